// voxelize_idx.hip -- DEVICE version of PG_OP.voxelize_idx (the loader step right before the
// forward; the reference runs it on the CPU with a hash map, libs/pointgroup_ops/src/voxelize/voxelize.cpp:58-152).
//
// Same results bit for bit (voxel ids in FIRST-ENCOUNTER order of the points, per-voxel point lists in ascending
// point index, coords of the voxel's first point), without a hash table or a sort of the points:
//   1. occupancy bitmask of the batch x S^3 grid + popcount prefix  -> a dense temporary id r per occupied voxel
//   2. firstpt[r] = min point index in the voxel (atomicMin)
//   3. bitmask over POINT indices of "is the first point of its voxel" + popcount prefix
//        -> voxel id = rank of its first point among all first points  (= first-encounter order)
//   4. counts (atomicAdd), maxActive (atomicMax); host reads {V, maxActive} once to size the outputs
//   5. unordered append of every point to its voxel's row, then a per-voxel insertion sort (rows are short)
#include "common.h"

void dcl_internal_zero_words(void *p, long long nwords, hipStream_t s);
int dcl_internal_scan_mask(const uint32_t *mask, int nwords, int32_t *wprefix, int32_t *scratch, hipStream_t s);

namespace {

__device__ __forceinline__ int lin_of(const int64_t *c, int S, int batch, bool &ok) {
  const long long b = c[0], x = c[1], y = c[2], z = c[3];
  ok = b >= 0 && b < batch && x >= 0 && x < S && y >= 0 && y < S && z >= 0 && z < S;
  return ok ? (int)(((b * S + x) * S + y) * S + z) : 0;
}
__device__ __forceinline__ int rank_of(const uint32_t *mask, const int32_t *wprefix, int lin) {
  const uint32_t m = mask[lin >> 5], bit = 1u << (lin & 31);
  return wprefix[lin >> 5] + __popc(m & (bit - 1));
}

__global__ void k_vi_mark(const int64_t *__restrict__ coords, int n, int S, int batch, uint32_t *__restrict__ vmask,
                          int32_t *__restrict__ err) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    bool ok;
    const int lin = lin_of(coords + (size_t)i * 4, S, batch, ok);
    if (!ok) { atomicExch(err, 1); continue; }
    atomicOr(&vmask[lin >> 5], 1u << (lin & 31));
  }
}
__global__ void k_vi_fill(int32_t *p, int n, int32_t v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_vi_first(const int64_t *__restrict__ coords, int n, int S, int batch, const uint32_t *__restrict__ vmask,
                           const int32_t *__restrict__ vprefix, int32_t *__restrict__ firstpt) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    bool ok;
    const int lin = lin_of(coords + (size_t)i * 4, S, batch, ok);
    if (ok) atomicMin(&firstpt[rank_of(vmask, vprefix, lin)], i);
  }
}
__global__ void k_vi_mark_first(const int64_t *__restrict__ coords, int n, int S, int batch,
                                const uint32_t *__restrict__ vmask, const int32_t *__restrict__ vprefix,
                                const int32_t *__restrict__ firstpt, uint32_t *__restrict__ pmask) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    bool ok;
    const int lin = lin_of(coords + (size_t)i * 4, S, batch, ok);
    if (ok && firstpt[rank_of(vmask, vprefix, lin)] == i) atomicOr(&pmask[i >> 5], 1u << (i & 31));
  }
}
// input_map[i] = voxel id; counts; maxActive; info = {V, maxActive}
__global__ void k_vi_ids(const int64_t *__restrict__ coords, int n, int S, int batch, const uint32_t *__restrict__ vmask,
                         const int32_t *__restrict__ vprefix, const int32_t *__restrict__ firstpt,
                         const uint32_t *__restrict__ pmask, const int32_t *__restrict__ pprefix, int npwords,
                         int32_t *__restrict__ input_map, int32_t *__restrict__ counts, int32_t *__restrict__ info) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    bool ok;
    const int lin = lin_of(coords + (size_t)i * 4, S, batch, ok);
    if (!ok) { input_map[i] = 0; continue; }
    const int f = firstpt[rank_of(vmask, vprefix, lin)];
    const int id = pprefix[f >> 5] + __popc(pmask[f >> 5] & ((1u << (f & 31)) - 1));
    input_map[i] = id;
    const int c = atomicAdd(&counts[id], 1) + 1;
    atomicMax(&info[1], c);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) info[0] = pprefix[npwords];
}
__global__ void k_vi_append(const int64_t *__restrict__ coords, int n, const int32_t *__restrict__ input_map,
                            const uint32_t *__restrict__ pmask, int max_active, int32_t *__restrict__ cursor,
                            int32_t *__restrict__ output_map, int64_t *__restrict__ output_coords) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int id = input_map[i];
    const int slot = atomicAdd(&cursor[id], 1);
    if (slot < max_active) output_map[(size_t)id * (max_active + 1) + 1 + slot] = i;
    if (pmask[i >> 5] & (1u << (i & 31))) {                       // first point of its voxel: voxelize.cpp:39-47
      const int64_t *c = coords + (size_t)i * 4;
      int64_t *o = output_coords + (size_t)id * 4;
      o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = c[3];
    }
  }
}
__global__ void k_vi_sort_rows(int V, int max_active, const int32_t *__restrict__ counts, int32_t *__restrict__ output_map) {
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x) {
    int32_t *row = output_map + (size_t)v * (max_active + 1);
    const int c = min(counts[v], max_active);
    row[0] = c;
    for (int a = 2; a <= c; ++a) {                                // insertion sort, ascending point index
      const int key = row[a];
      int j = a - 1;
      while (j >= 1 && row[j] > key) { row[j + 1] = row[j]; --j; }
      row[j + 1] = key;
    }
    for (int a = c + 1; a <= max_active; ++a) row[a] = 0;         // zero padding (voxelize.cpp:144-149)
  }
}

struct ViLayout { size_t vmask, vprefix, firstpt, pmask, pprefix, counts, cursor, scratch, total; int nvw, npw; };
bool vi_layout(int n, int batch, int S, ViLayout *L) {
  if (n < 0 || batch <= 0 || S <= 0 || (long long)batch * S * S * S > (1ll << 30)) return false;
  L->nvw = (int)(((long long)batch * S * S * S + 31) / 32);
  L->npw = (n + 31) / 32 > 0 ? (n + 31) / 32 : 1;
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off = (off + b + 255) / 256 * 256; return o; };
  const size_t nn = (size_t)(n > 0 ? n : 1);
  L->vmask = take(4 * (size_t)L->nvw); L->vprefix = take(4 * ((size_t)L->nvw + 1));
  L->firstpt = take(4 * nn); L->pmask = take(4 * (size_t)L->npw); L->pprefix = take(4 * ((size_t)L->npw + 1));
  L->counts = take(4 * nn); L->cursor = take(4 * nn);
  L->scratch = take(4 * ((size_t)L->nvw / 1024 + 4));
  L->total = off;
  return true;
}
template <typename T> T *at(void *b, size_t o) { return reinterpret_cast<T *>(reinterpret_cast<char *>(b) + o); }

}  // namespace

DCL_API int dcl_voxelize_idx_gpu_ws_bytes(int n, int batch, int S, int64_t *bytes_host) {
  ViLayout L;
  DCL_CHECK_ARG(bytes_host && vi_layout(n, batch, S, &L));
  *bytes_host = (int64_t)L.total;
  return 0;
}

// step 1: input_map (n), info_dev = {V, maxActive, error}; the caller reads info back to size the outputs.
DCL_API int dcl_voxelize_idx_gpu_count(const int64_t *coords, int n, int batch, int S, int mode, void *ws,
                                       int64_t ws_bytes, int32_t *input_map, int32_t *info_dev, dclStream_t stream) {
  ViLayout L;
  DCL_CHECK_ARG((mode == 3 || mode == 4) && ws && input_map && info_dev && vi_layout(n, batch, S, &L) &&
                ws_bytes >= (int64_t)L.total && (n == 0 || coords));
  hipStream_t s = (hipStream_t)stream;
  dcl_internal_zero_words(at<uint32_t>(ws, L.vmask), L.nvw, s);
  dcl_internal_zero_words(at<uint32_t>(ws, L.pmask), L.npw, s);
  dcl_internal_zero_words(at<int32_t>(ws, L.counts), n > 0 ? n : 1, s);
  dcl_internal_zero_words(at<int32_t>(ws, L.cursor), n > 0 ? n : 1, s);
  dcl_internal_zero_words(info_dev, 3, s);
  const int g = dcl_grid_1d(n > 0 ? n : 1, 256);
  hipLaunchKernelGGL(k_vi_fill, dim3(g), dim3(256), 0, s, at<int32_t>(ws, L.firstpt), n, 0x7fffffff);
  hipLaunchKernelGGL(k_vi_mark, dim3(g), dim3(256), 0, s, coords, n, S, batch, at<uint32_t>(ws, L.vmask), info_dev + 2);
  dcl_internal_scan_mask(at<uint32_t>(ws, L.vmask), L.nvw, at<int32_t>(ws, L.vprefix), at<int32_t>(ws, L.scratch), s);
  hipLaunchKernelGGL(k_vi_first, dim3(g), dim3(256), 0, s, coords, n, S, batch, at<uint32_t>(ws, L.vmask),
                     at<int32_t>(ws, L.vprefix), at<int32_t>(ws, L.firstpt));
  hipLaunchKernelGGL(k_vi_mark_first, dim3(g), dim3(256), 0, s, coords, n, S, batch, at<uint32_t>(ws, L.vmask),
                     at<int32_t>(ws, L.vprefix), at<int32_t>(ws, L.firstpt), at<uint32_t>(ws, L.pmask));
  dcl_internal_scan_mask(at<uint32_t>(ws, L.pmask), L.npw, at<int32_t>(ws, L.pprefix), at<int32_t>(ws, L.scratch), s);
  hipLaunchKernelGGL(k_vi_ids, dim3(g), dim3(256), 0, s, coords, n, S, batch, at<uint32_t>(ws, L.vmask),
                     at<int32_t>(ws, L.vprefix), at<int32_t>(ws, L.firstpt), at<uint32_t>(ws, L.pmask),
                     at<int32_t>(ws, L.pprefix), L.npw, input_map, at<int32_t>(ws, L.counts), info_dev);
  DCL_LAUNCH_CHECK();
  return 0;
}

// step 2: output_coords (V,4) i64, output_map (V, maxActive+1) i32 (fully written)
DCL_API int dcl_voxelize_idx_gpu_fill(const int64_t *coords, int n, int batch, int S, void *ws, const int32_t *input_map,
                                      int n_active, int max_active, int64_t *output_coords, int32_t *output_map,
                                      dclStream_t stream) {
  ViLayout L;
  DCL_CHECK_ARG(ws && input_map && vi_layout(n, batch, S, &L) && n_active >= 0 && max_active >= 1);
  if (n_active == 0) return 0;
  DCL_CHECK_ARG(coords && output_coords && output_map);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_vi_append, dim3(dcl_grid_1d(n, 256)), dim3(256), 0, s, coords, n, input_map,
                     at<uint32_t>(ws, L.pmask), max_active, at<int32_t>(ws, L.cursor), output_map, output_coords);
  hipLaunchKernelGGL(k_vi_sort_rows, dim3(dcl_grid_1d(n_active, 256)), dim3(256), 0, s, n_active, max_active,
                     at<int32_t>(ws, L.counts), output_map);
  DCL_LAUNCH_CHECK();
  return 0;
}

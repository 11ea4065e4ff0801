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

// ---- the crop builder's case in ONE launch: b crops of exactly n_per <= 1024 points each, batch-sorted, on 64^3 grids -------
// (YCBV/dataloader_test_YCBV.py:213-223 voxelises the b x 1024 sampled points of an image's crops; the general
// path above takes ~16 launches and a host read-back of {V, maxActive} for it.)  One workgroup of 1024 threads per crop, one
// point per thread, everything of the crop in LDS: occupancy bits of its 64^3 cells + popcount prefix -> a dense temporary
// id per occupied cell; the cell's first point by LDS atomicMin; first-encounter voxel ids = rank of that point among the
// crop's first points (ballots + a scan over the 16 waves); point counts by LDS atomics.  The crops' voxel counts meet
// through one word per crop in `comm` (crop c waits for the crops before it only; the words carry the call's generation
// number, so nothing has to be zeroed between calls).  Rows are CAPACITY-pitched: output_map has `pitch` ints per row
// (1 + the most points a voxel may hold), the rows behind the V live ones are zero, V / maxActive / an overflow flag land in info -- no host
// read-back sizes anything.  Same results as the general path: ids in first-encounter order, rows ascending, zero padded.
constexpr int kViCropThreads = 1024;
constexpr int kViCropS = 64;
constexpr int kViCropWords = kViCropS * kViCropS * kViCropS / 32;      // 8192

template <typename OccT>
__global__ __launch_bounds__(kViCropThreads) void k_vi_crops(const int64_t *__restrict__ coords, int n_per, int batch, int pitch,
                                                             int32_t *__restrict__ comm, int gen, int32_t *__restrict__ input_map,
                                                             OccT *__restrict__ output_coords, int32_t *__restrict__ output_map,
                                                             int32_t *__restrict__ info /* {V, maxActive, error} */) {
  __shared__ uint32_t s_mask[kViCropWords];
  __shared__ uint16_t s_prefix[kViCropWords];                         // <= 1024 occupied cells per crop
  __shared__ int32_t s_first[kViCropThreads], s_vid[kViCropThreads], s_cnt[kViCropThreads], s_cur[kViCropThreads];
  __shared__ int32_t s_wave[kViCropThreads / 64 + 1], s_red[4];
  const int c = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool live = t < n_per;
  const int gi = c * n_per + t;                                         // the point's global row
  for (int w = t; w < kViCropWords; w += kViCropThreads) s_mask[w] = 0u;
  s_first[t] = 0x7fffffff;
  s_cnt[t] = 0;
  s_cur[t] = 0;
  if (t < 4) s_red[t] = 0;
  __syncthreads();
  int lin = 0;
  bool ok = false;
  long long cb = 0, cx = 0, cy = 0, cz = 0;
  if (live) {
    const int64_t *p = coords + (size_t)gi * 4;
    cb = p[0]; cx = p[1]; cy = p[2]; cz = p[3];
    ok = cb == c && cx >= 0 && cx < kViCropS && cy >= 0 && cy < kViCropS && cz >= 0 && cz < kViCropS;
    lin = ok ? (int)((cx * kViCropS + cy) * kViCropS + cz) : 0;
    if (ok) atomicOr(&s_mask[lin >> 5], 1u << (lin & 31));
    else s_red[2] = 1;                                                   // a point outside its crop's grid: error
  }
  __syncthreads();
  // popcount prefix of the 8192 words: 8 words per thread, block scan of the thread sums
  {
    int sum = 0;
    uint32_t wds[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { wds[q] = s_mask[t * 8 + q]; sum += __popc(wds[q]); }
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (t == 0) { int run = 0; for (int w = 0; w < kViCropThreads / 64; ++w) { const int v = s_wave[w]; s_wave[w] = run; run += v; } }
    __syncthreads();
    int run = s_wave[wave] + inc - sum;
#pragma unroll
    for (int q = 0; q < 8; ++q) { s_prefix[t * 8 + q] = (uint16_t)run; run += __popc(wds[q]); }
  }
  __syncthreads();
  int r = 0;
  if (ok) {
    const uint32_t m = s_mask[lin >> 5], bit = 1u << (lin & 31);
    r = s_prefix[lin >> 5] + __popc(m & (bit - 1));
    atomicMin(&s_first[r], t);
  }
  __syncthreads();
  // first-encounter id = rank of the cell's first point among the crop's first points (in point order)
  const bool first = ok && s_first[r] == t;
  const unsigned long long bal = __ballot(first);
  if (lane == 0) s_wave[wave] = __popcll(bal);
  __syncthreads();
  if (t == 0) { int run = 0; for (int w = 0; w < kViCropThreads / 64; ++w) { const int v = s_wave[w]; s_wave[w] = run; run += v; } s_wave[kViCropThreads / 64] = run; }
  __syncthreads();
  const int V_c = s_wave[kViCropThreads / 64];
  if (first) s_vid[r] = s_wave[wave] + __popcll(bal & ((1ull << lane) - 1ull));
  __syncthreads();
  const int vid = ok ? s_vid[r] : 0;
  if (ok) atomicAdd(&s_cnt[vid], 1);
  __syncthreads();
  if (t < V_c) atomicMax(&s_red[1], s_cnt[t]);
  // the crops' voxel counts meet: mine is published, the ones before me are summed (bounded only by their running: a crop
  // waits for lower-numbered workgroups of a grid of <= 64 one-per-CU workgroups)
  if (t == 0) {
    __hip_atomic_store(comm + 2 * c + 1, V_c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(comm + 2 * c, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  int base = 0;
  if (t < c) {
    int spins = 0;                                                       // bounded all the same: ~2 s, then the error flag
    while (__hip_atomic_load(comm + 2 * t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != gen) {
      if (++spins > (1 << 22)) { s_red[2] = 1; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    base = __hip_atomic_load(comm + 2 * t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (t < 64) {                                                          // batch <= 64: one wave sums the bases
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) base += __shfl_xor(base, d, 64);
    if (t == 0) s_red[0] = base;
  }
  __syncthreads();
  base = s_red[0];
  const int ma_c = s_red[1];
  if (t == 0) {
    atomicMax(&info[1], ma_c);
    if (c == batch - 1) info[0] = base + V_c;
    if (s_red[2] || ma_c + 1 > pitch) atomicExch(&info[2], 1);           // a point outside the grid, or a voxel beyond the pitch
  }
  if (live) input_map[gi] = ok ? base + vid : 0;
  // rows: unordered append (LDS cursors), then every voxel's short row sorted ascending and zero padded
  if (ok) {
    const int slot = atomicAdd(&s_cur[vid], 1);
    if (slot + 1 < pitch) output_map[(size_t)(base + vid) * pitch + 1 + slot] = gi;
    if (first) {
      OccT *o = output_coords + (size_t)(base + vid) * 4;
      o[0] = (OccT)cb; o[1] = (OccT)cx; o[2] = (OccT)cy; o[3] = (OccT)cz;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t < V_c) {
    int32_t *row = output_map + (size_t)(base + t) * pitch;
    const int cnt = min(s_cnt[t], pitch - 1);
    row[0] = cnt;
    for (int a = 2; a <= cnt; ++a) {                                     // insertion sort, ascending point index
      const int key = row[a];
      int j = a - 1;
      while (j >= 1 && row[j] > key) { row[j + 1] = row[j]; --j; }
      row[j + 1] = key;
    }
    for (int a = cnt + 1; a < pitch; ++a) row[a] = 0;                    // zero padding (voxelize.cpp:144-149)
  }
  // the rows behind the live ones (capacity b * n_per): zeroed by the last crop's workgroup, which knows V -- a consumer that
  // walks all capacity rows (the whole-forward graph's voxel means) must find count 0 there, not stale memory
  if (c == batch - 1) {
    const size_t lo = (size_t)(base + V_c) * pitch, hi = (size_t)batch * n_per * pitch;
    for (size_t e = lo + t; e < hi; e += kViCropThreads) output_map[e] = 0;
    for (size_t e = (size_t)(base + V_c) * 4 + t; e < (size_t)batch * n_per * 4; e += kViCropThreads) output_coords[e] = (OccT)0;
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

// b crops of n_per points each (rows c*n_per .. (c+1)*n_per - 1 belong to crop c; n_per <= 1024, 64^3 grid, b <= 64), ONE
// launch, capacity-pitched outputs: output_map (b * n_per rows, `pitch` ints each: 1 + the most points per voxel it
// can hold; rows behind the V live ones zero), output_coords (b * n_per rows, 4) as int64 (occ32 = 0) or int32 (occ32 = 1: what the backbone runner takes), input_map
// (b*n_per), info = {V, maxActive, error}: error = a point outside its crop's grid or a voxel with more than pitch - 1 points
// (its row is then truncated).  comm: 2 * b ints that persist between calls (zero before the first), gen: a number that
// differs from the previous call's (the caller counts calls).  No host read-back is needed to size anything.
DCL_API int dcl_voxelize_idx_crops(const int64_t *coords, int batch, int n_per, int S, int mode, int pitch, int32_t *comm, int gen,
                                   int32_t *input_map, void *output_coords, int occ32, int32_t *output_map, int32_t *info_dev,
                                   dclStream_t stream) {
  DCL_CHECK_ARG((mode == 3 || mode == 4) && batch >= 1 && batch <= 64 && n_per >= 1 && n_per <= kViCropThreads && S == kViCropS &&
                pitch >= 2 && coords && comm && gen != 0 && input_map && output_coords && output_map && info_dev);
  hipStream_t s = (hipStream_t)stream;
  dcl_internal_zero_words(info_dev, 3, s);
  if (occ32)
    hipLaunchKernelGGL(k_vi_crops<int32_t>, dim3(batch), dim3(kViCropThreads), 0, s, coords, n_per, batch, pitch, comm, gen,
                       input_map, reinterpret_cast<int32_t *>(output_coords), output_map, info_dev);
  else
    hipLaunchKernelGGL(k_vi_crops<int64_t>, dim3(batch), dim3(kViCropThreads), 0, s, coords, n_per, batch, pitch, comm, gen,
                       input_map, reinterpret_cast<int64_t *>(output_coords), output_map, info_dev);
  DCL_LAUNCH_CHECK();
  return 0;
}

// row_order.hip -- device-side row ordering of a sparse-conv layer's output rows, for work dealt in USED chunks.
//
// indiceConv (libs/spconv/include/spconv/spconv_ops.h:284-344) walks the 27 kernel offsets of every output row; in
// gather form an output row of a dilating SparseConv3d has on average 12 of its 27 neighbours (rulebook density 0.45), a
// submanifold row 21 (0.8).  The implicit-GEMM kernel (sparse_conv.hip) skips an offset only when NO row of its 128-row
// tile (32-row wave slice) uses it, so in the natural row order -- ascending linear voxel index -- it issues 0.80-0.99 of
// the 27 x rows slots.  Rows with the same neighbourhood SHAPE skip together: this pass gives every row a 9-bit key
// ("is there a neighbour in plane x = -1 / 0 / +1, y = ..., z = ...") and orders the rows by it with a STABLE counting
// sort, so that a tile holds rows of one shape (issued work 0.52-0.56 of the slots on the dilating layers, 0.80-0.92 on the
// submanifold ones, measured on the backbone's active sets).  The order is internal to the conv launch -- tile slot i
// computes output row order[i] and stores it there -- so features stay in the reference's row order (spconv_ops.h:126)
// and results are unchanged up to the fp32 summation split points of the stream-K decomposition; because the sort is
// stable and every step is deterministic, a given input always produces the same bits.
//
// Per job (one conv layer) five steps, each ONE launch for all jobs of a backbone pass (blockIdx.y = job):
//   k_order_keys     27-bit neighbour mask per row (bit tests in the input set's occupancy words: 9 z-row fetches per
//                    row, no rank look-ups) + per-1024-row-block histogram of the keys
//   k_order_scan     exclusive scan of the (key-major, block-minor) histogram: one workgroup per job
//   k_order_scatter  order[offset(key, block) + stable rank inside the block] = row   (rank: 9 ballots per wave, the
//                    block's 16 waves in sequence)
//   k_order_tiles    per 128-row tile of the order: OR of its rows' masks -> step mask (the offsets in the kernel's
//                    visiting order) and its popcount
//   k_order_prefix   exclusive scan of the tiles' used-step counts (the unit prefix the conv kernel searches)
// All index / bit work: L2-bound, no MFMA.
#include "common.h"

int dcl_internal_order_rows(const DclOrderJobs &jobs, int njobs, dclStream_t stream);

namespace {

constexpr int kSortBlock = 1024;      // rows per counting-sort block (16 waves)
constexpr int kKeys = 512;            // 9-bit keys

__device__ __forceinline__ int live_rows(const DclOrderJob &j) {
  int n = j.n_dev ? *j.n_dev : j.n_host;
  return n < j.cap ? (n < 0 ? 0 : n) : j.cap;
}

// the S-bit occupancy row (b, x, y, *) of a grid with side S in {8, 16, 32, 64} as a 64-bit word (bit z = voxel z)
__device__ __forceinline__ unsigned long long z_row(const uint32_t *__restrict__ mask, int S, int b, int x, int y) {
  if ((unsigned)x >= (unsigned)S || (unsigned)y >= (unsigned)S) return 0ull;
  const long long lin0 = (((long long)b * S + x) * S + y) * S;
  const int w = (int)(lin0 >> 5);
  if (S == 64) return (unsigned long long)mask[w] | ((unsigned long long)mask[w + 1] << 32);
  const uint32_t v = mask[w] >> (int)(lin0 & 31);
  return S == 32 ? (unsigned long long)v : (unsigned long long)(v & ((1u << S) - 1u));
}

// offsets k = kz + 3 ky + 9 kx (k = x - o*s + p, geometry.h:61-70) of the present neighbours of output voxel q (k3, s1, p1)
__device__ __forceinline__ uint32_t neighbour_mask27(const uint32_t *__restrict__ mask, int S, int4 q) {
  uint32_t m = 0;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const unsigned long long row = z_row(mask, S, q.x, q.y - 1 + kx, q.z - 1 + ky);
      // bits z-1, z, z+1 of the row -> kz = 0, 1, 2 (z-1 < 0 and z+1 >= S fall off the row)
      const uint32_t three = q.w > 0 ? (uint32_t)((row >> (q.w - 1)) & 7ull) : (uint32_t)((row << 1) & 7ull);
      m |= three << (3 * ky + 9 * kx);
    }
  return m;
}

__device__ __forceinline__ int plane_key(uint32_t m) {
  constexpr uint32_t X0 = 0x1FFu, Y0 = 0x7u | (0x7u << 9) | (0x7u << 18), Z0 = 0x1249249u;
  int key = 0;
  key |= (m & X0) ? 1 : 0;          key |= (m & (X0 << 9)) ? 2 : 0;    key |= (m & (X0 << 18)) ? 4 : 0;
  key |= (m & Y0) ? 8 : 0;          key |= (m & (Y0 << 3)) ? 16 : 0;   key |= (m & (Y0 << 6)) ? 32 : 0;
  key |= (m & Z0) ? 64 : 0;         key |= (m & (Z0 << 1)) ? 128 : 0;  key |= (m & (Z0 << 2)) ? 256 : 0;
  return key;
}

__global__ __launch_bounds__(kSortBlock) void k_order_keys(const DclOrderJobs jobs) {
  const DclOrderJob &j = jobs.job[blockIdx.y];
  const int n = live_rows(j);
  const int nblk = (n + kSortBlock - 1) / kSortBlock;
  __shared__ int h[kKeys];
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    for (int i = threadIdx.x; i < kKeys; i += kSortBlock) h[i] = 0;
    __syncthreads();
    const int r = blk * kSortBlock + threadIdx.x;
    if (r < n) {
      const int4 q = reinterpret_cast<const int4 *>(j.out_indices)[r];
      const uint32_t m = neighbour_mask27(j.in_mask, j.S_in, q);
      j.rowmask[r] = m;
      atomicAdd(&h[plane_key(m)], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kKeys; i += kSortBlock) j.hist[(size_t)i * j.nblk_cap + blk] = h[i];
    __syncthreads();
  }
}

// workgroup of 1024 threads: exclusive scan of per-thread values
__device__ __forceinline__ int block_excl_scan_1024(int v, int *total) {
  __shared__ int wsum[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    base += i < wid ? wsum[i] : 0;
    tot += wsum[i];
  }
  *total = tot;
  __syncthreads();
  return base + inc - v;
}

__global__ __launch_bounds__(1024) void k_order_scan(const DclOrderJobs jobs) {
  const DclOrderJob &j = jobs.job[blockIdx.x];
  const int n = live_rows(j);
  const int nblk = (n + kSortBlock - 1) / kSortBlock;
  const int E = kKeys * nblk;                                   // entries in (key, block) order
  const int per = (E + 1023) / 1024;
  const int e0 = threadIdx.x * per, e1 = min(E, e0 + per);
  int s = 0;
  {
    int key = nblk > 0 ? e0 / nblk : 0, blk = nblk > 0 ? e0 - key * nblk : 0;      // one division per thread, then counters
    for (int e = e0; e < e1; ++e) {
      s += j.hist[(size_t)key * j.nblk_cap + blk];
      if (++blk == nblk) { blk = 0; ++key; }
    }
  }
  int total;
  int run = block_excl_scan_1024(s, &total);
  int key = nblk > 0 ? e0 / nblk : 0, blk = nblk > 0 ? e0 - key * nblk : 0;
  for (int e = e0; e < e1; ++e) {
    int32_t *p = j.hist + (size_t)key * j.nblk_cap + blk;
    const int c = *p;
    *p = run;
    run += c;
    if (++blk == nblk) { blk = 0; ++key; }
  }
}

__global__ __launch_bounds__(kSortBlock) void k_order_scatter(const DclOrderJobs jobs) {
  const DclOrderJob &j = jobs.job[blockIdx.y];
  const int n = live_rows(j);
  const int nblk = (n + kSortBlock - 1) / kSortBlock;
  __shared__ int cnt[kKeys];                                     // rows of the block's earlier waves per key
  __shared__ int off[kKeys];                                     // global offset of (key, this block)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    for (int i = threadIdx.x; i < kKeys; i += kSortBlock) {
      cnt[i] = 0;
      off[i] = j.hist[(size_t)i * j.nblk_cap + blk];
    }
    __syncthreads();
    const int r = blk * kSortBlock + threadIdx.x;
    const bool valid = r < n;
    const int key = valid ? plane_key(j.rowmask[r]) : 0;
    // lanes of this wave with the same key (9 ballots), in lane order = row order: the stable rank inside the wave
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 9; ++bit) {
      const unsigned long long bal = __ballot((key >> bit) & 1);
      peers &= ((key >> bit) & 1) ? bal : ~bal;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    const int mine = __popcll(peers);
    for (int w = 0; w < kSortBlock / 64; ++w) {                    // the block's waves in sequence: stable across waves
      if (wave == w && valid) {
        const int base = cnt[key];
        __builtin_amdgcn_wave_barrier();
        j.order[off[key] + base + rank] = r;
        if (rank == 0) cnt[key] = base + mine;
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ uint32_t step_mask_of(uint32_t m27, int subm) {
  if (!subm) return m27;
  // visiting order of a submanifold conv: the centre offset (k = 13) first, then k ascending (spconv_ops.h:289-299)
  return ((m27 >> 13) & 1u) | ((m27 & 0x1FFFu) << 1) | (m27 & ~0x3FFFu);
}

__global__ __launch_bounds__(256) void k_order_tiles(const DclOrderJobs jobs) {
  const DclOrderJob &j = jobs.job[blockIdx.y];
  const int n = live_rows(j);
  const int ntiles = (n + 127) / 128;
  const int lane = threadIdx.x & 63;
  for (int tile = blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += gridDim.x * 4) {
    const int i0 = tile * 128 + lane, i1 = i0 + 64;
    uint32_t m = 0;
    if (i0 < n) m |= j.rowmask[j.order[i0]];
    if (i1 < n) m |= j.rowmask[j.order[i1]];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d, 64);
    if (lane == 0) {
      const uint32_t sm = step_mask_of(m, j.subm);
      j.smask[tile] = sm;
      j.tile_cnt[tile] = __popc(sm);
    }
  }
}

__global__ __launch_bounds__(1024) void k_order_prefix(const DclOrderJobs jobs) {
  const DclOrderJob &j = jobs.job[blockIdx.x];
  const int n = live_rows(j);
  const int ntiles = (n + 127) / 128;
  const int per = (ntiles + 1023) / 1024;
  const int t0 = threadIdx.x * per, t1 = min(ntiles, t0 + per);
  int s = 0;
  for (int t = t0; t < t1; ++t) s += j.tile_cnt[t];
  int total;
  int run = block_excl_scan_1024(s, &total);
  for (int t = t0; t < t1; ++t) {
    j.bal[t] = run;
    run += j.tile_cnt[t];
  }
  if (threadIdx.x == 1023) j.bal[ntiles] = total;
}

}  // namespace

int dcl_internal_order_rows(const DclOrderJobs &jobs, int njobs, dclStream_t stream) {
  DCL_CHECK_ARG(njobs >= 1 && njobs <= DCL_ORDER_MAX_JOBS);
  long long most = 1;
  for (int i = 0; i < njobs; ++i) {
    const DclOrderJob &j = jobs.job[i];
    DCL_CHECK_ARG(j.out_indices && j.in_mask && j.rowmask && j.hist && j.order && j.tile_cnt && j.bal && j.smask && j.cap > 0 &&
                  (j.S_in == 8 || j.S_in == 16 || j.S_in == 32 || j.S_in == 64) &&
                  j.nblk_cap >= (j.cap + kSortBlock - 1) / kSortBlock && (j.n_dev || (j.n_host >= 0 && j.n_host <= j.cap)));
    const long long rows = j.n_dev ? j.cap : j.n_host;
    if (rows > most) most = rows;
  }
  hipStream_t s = (hipStream_t)stream;
  // workgroups loop over their job's sort blocks / tiles: the grids are sized for the live work of typical launches, not
  // for the row capacities of capacity mode
  const int gb = (int)(most / kSortBlock + 1 < 256 ? most / kSortBlock + 1 : 256);
  const int gt = (int)(most / 512 + 1 < 512 ? most / 512 + 1 : 512);
  hipLaunchKernelGGL(k_order_keys, dim3(gb, njobs), dim3(kSortBlock), 0, s, jobs);
  hipLaunchKernelGGL(k_order_scan, dim3(njobs), dim3(1024), 0, s, jobs);
  hipLaunchKernelGGL(k_order_scatter, dim3(gb, njobs), dim3(kSortBlock), 0, s, jobs);
  hipLaunchKernelGGL(k_order_tiles, dim3(gt, njobs), dim3(256), 0, s, jobs);
  hipLaunchKernelGGL(k_order_prefix, dim3(njobs), dim3(1024), 0, s, jobs);
  DCL_LAUNCH_CHECK();
  return 0;
}

// ---- op-level entry: ONE layer (tests, the spconv shim); workspace layout queried with dcl_order_rows_ws_bytes
namespace {
struct OrderLayout {
  size_t rowmask, hist, tile_cnt, total;
  int nblk_cap, tiles_cap;
};
bool order_layout(int cap, OrderLayout *L) {
  if (cap <= 0) return false;
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  L->nblk_cap = (cap + kSortBlock - 1) / kSortBlock;
  L->tiles_cap = (cap + 127) / 128;
  size_t off = 0;
  L->rowmask = off; off = up(off + sizeof(uint32_t) * (size_t)cap);
  L->hist = off;    off = up(off + sizeof(int32_t) * (size_t)kKeys * L->nblk_cap);
  L->tile_cnt = off; off = up(off + sizeof(int32_t) * (size_t)L->tiles_cap);
  L->total = off;
  return true;
}
}  // namespace

DCL_API int dcl_order_rows_ws_bytes(int cap, int64_t *bytes_host) {
  OrderLayout L;
  DCL_CHECK_ARG(bytes_host && order_layout(cap, &L));
  *bytes_host = (int64_t)L.total;
  return 0;
}

DCL_API int dcl_order_rows(const int32_t *out_indices, const int32_t *n_out_dev, int n_out_host, int cap,
                           const uint32_t *in_mask, int S_in, int subm, void *ws, int64_t ws_bytes, int32_t *order,
                           int32_t *bal, uint32_t *smask, dclStream_t stream) {
  OrderLayout L;
  DCL_CHECK_ARG(ws && order_layout(cap, &L) && ws_bytes >= (int64_t)L.total);
  DclOrderJobs jobs{};
  DclOrderJob &j = jobs.job[0];
  char *base = reinterpret_cast<char *>(ws);
  j.out_indices = out_indices; j.n_dev = n_out_dev; j.n_host = n_out_host; j.in_mask = in_mask; j.S_in = S_in; j.cap = cap;
  j.subm = subm; j.rowmask = reinterpret_cast<uint32_t *>(base + L.rowmask); j.hist = reinterpret_cast<int32_t *>(base + L.hist);
  j.order = order; j.tile_cnt = reinterpret_cast<int32_t *>(base + L.tile_cnt); j.bal = bal; j.smask = smask;
  j.nblk_cap = L.nblk_cap;
  return dcl_internal_order_rows(jobs, 1, stream);
}

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
// Per job (one conv layer) the rows are taken in WINDOWS of 8192 consecutive rows (64 tiles), each sorted on its own by one
// workgroup in LDS: measured on the backbone's sets a window sort reaches the issued work of a global sort within 2-3 %
// (rows that are close in linear index share their crop and their stretch of surface), and it needs no global histogram,
// scan or scatter.  Two launches for all jobs of a backbone pass (blockIdx.y = job):
//   k_order_masks    27-bit neighbour mask per row, one thread per row (bit tests in the input set's occupancy words: 9
//                    z-row fetches per row, no rank look-ups)
//   k_order_windows  mask -> 9-bit key -> three stable 3-bit radix passes over (key, local row) in LDS (per 64-row chunk the bucket
//   counts are ballots; one wave scans the 8 x 128 counters) -> order[] -> per 128-row tile the OR of its rows' masks =
//   step mask (offsets in the kernel's visiting order) and its popcount -> ticket; the job's last workgroup scans the
//   tiles' used-step counts into the unit prefix the conv kernel searches.
// All index / bit work: L2- and LDS-bound, no MFMA.
#include "common.h"

int dcl_internal_order_rows(const DclOrderJobs &jobs, int njobs, dclStream_t stream);

namespace {

constexpr int kWin = 8192;            // rows per sort window (64 tiles of 128 rows), one workgroup of 1024 threads each
constexpr int kChunks = kWin / 64;    // 64-row chunks of a window

__device__ __forceinline__ int live_rows(const DclOrderJob &j) {
  int n = j.n_dev ? *j.n_dev : j.n_host;
  return n < j.cap ? (n < 0 ? 0 : n) : j.cap;
}

// the S-bit occupancy row (b, x, y, *) of a grid with side S in {8, 16, 32, 64} as a 64-bit word (bit z = voxel z)
__device__ __forceinline__ unsigned long long z_row(const uint32_t *__restrict__ mask, int S, int b, int x, int y) {
  if ((unsigned)x >= (unsigned)S || (unsigned)y >= (unsigned)S) return 0ull;
  const int lin0 = ((b * S + x) * S + y) * S;                     // (grids are < 2^31 cells: checked where they are made)
  const int w = lin0 >> 5;
  if (S == 64) return (unsigned long long)mask[w] | ((unsigned long long)mask[w + 1] << 32);
  const uint32_t v = mask[w] >> (lin0 & 31);
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

__device__ __forceinline__ uint32_t step_mask_of(uint32_t m27, int subm) {
  if (!subm) return m27;
  // visiting order of a submanifold conv: the centre offset (k = 13) first, then k ascending (spconv_ops.h:289-299)
  return ((m27 >> 13) & 1u) | ((m27 & 0x1FFFu) << 1) | (m27 & ~0x3FFFu);
}


// one thread per row, all jobs of the pass: the 27-bit neighbour mask (wide launch: the 9 z-row fetches per row are
// latency, which a window's single workgroup cannot hide)
__global__ __launch_bounds__(256) void k_order_masks(const DclOrderJobs jobs) {
  const DclOrderJob &j = jobs.job[blockIdx.y];
  const int n = live_rows(j);
  for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
    const int4 q = reinterpret_cast<const int4 *>(j.out_indices)[r];
    j.rowmask[r] = neighbour_mask27(j.in_mask, j.S_in, q);
  }
}

#ifdef DCL_DIAG
__device__ unsigned long long g_order_stamps[16];     // diagnostic build: s_memrealtime (100 MHz) of workgroup (0,0)'s phases
#define ORDER_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_order_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ORDER_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(1024) void k_order_windows(const DclOrderJobs jobs) {
  const DclOrderJob &j = jobs.job[blockIdx.y];
  const int n = live_rows(j);
  const int nwin = (n + kWin - 1) / kWin;
  extern __shared__ uint32_t order_lds[];
  uint32_t *A = order_lds, *B = order_lds + kWin, *M = order_lds + 2 * kWin;
  int *cnt = reinterpret_cast<int *>(order_lds + 3 * kWin);          // [8 buckets][kChunks]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int win = blockIdx.x; win < nwin; win += gridDim.x) {
    const int base = win * kWin;
    const int live = n - base < kWin ? n - base : kWin;
    ORDER_STAMP(0);
    // 1. the rows' neighbour masks (k_order_masks) -> keys; item = key << 13 | local row
    uint32_t mm[kWin / 1024];
#pragma unroll
    for (int r = 0; r < kWin / 1024; ++r) {
      const int i = r * 1024 + tid;
      mm[r] = i < live ? j.rowmask[base + i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < kWin / 1024; ++r) {
      const int i = r * 1024 + tid;
      if (i < live) {
        M[i] = mm[r];
        A[i] = ((uint32_t)plane_key(mm[r]) << 13) | (uint32_t)i;
      }
    }
    __syncthreads();
    ORDER_STAMP(1);
    // 2. stable LSD radix sort by the 9-bit key: 3 passes of 3 bits.  A wave owns 8 consecutive chunks of 64 items.
    for (int pass = 0; pass < 3; ++pass) {
      ORDER_STAMP(2 + 3 * pass);
      uint32_t item[kChunks / 16];
      int dig[kChunks / 16], rank[kChunks / 16];
      cnt[tid] = 0;                                                  // 8 x kChunks = 1024 counters: buckets nobody is in stay 0
      __syncthreads();
#pragma unroll
      for (int r = 0; r < kChunks / 16; ++r) {
        const int c = wave * (kChunks / 16) + r, i = c * 64 + lane;
        const bool valid = i < live;
        item[r] = valid ? A[i] : 0u;
        dig[r] = valid ? (int)((item[r] >> (13 + 3 * pass)) & 7u) : 8;
        // the chunk's lanes with my digit, from three ballots (one per digit bit) instead of one per bucket
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 3; ++bit) {
          const unsigned long long bal = __ballot((dig[r] >> bit) & 1);
          peers &= ((dig[r] >> bit) & 1) ? bal : ~bal;
        }
        rank[r] = __builtin_amdgcn_mbcnt_hi((unsigned)(peers >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)peers, 0u));
        if (valid && rank[r] == 0) cnt[dig[r] * kChunks + c] = __popcll(peers);   // the bucket's first lane publishes its size
      }
      __syncthreads();
      ORDER_STAMP(3 + 3 * pass);
      if (wave == 0) {                                               // exclusive scan of the 8 x kChunks counters (bucket-major)
        constexpr int PER = 8 * kChunks / 64;
        int v[PER], s = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) { v[q] = cnt[lane * PER + q]; s += v[q]; }
        int inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int t = __shfl_up(inc, d, 64);
          if (lane >= d) inc += t;
        }
        int run = inc - s;
#pragma unroll
        for (int q = 0; q < PER; ++q) { cnt[lane * PER + q] = run; run += v[q]; }
      }
      __syncthreads();
      ORDER_STAMP(4 + 3 * pass);
      int dst[kChunks / 16];
#pragma unroll
      for (int r = 0; r < kChunks / 16; ++r)                         // all offsets first (independent LDS reads), then the writes
        dst[r] = dig[r] < 8 ? cnt[dig[r] * kChunks + wave * (kChunks / 16) + r] + rank[r] : -1;
#pragma unroll
      for (int r = 0; r < kChunks / 16; ++r)
        if (dst[r] >= 0) B[dst[r]] = item[r];
      __syncthreads();
      uint32_t *t = A; A = B; B = t;
    }
    ORDER_STAMP(11);
    // 3. the order, and per tile its step mask / used-step count
#pragma unroll 2
    for (int r = 0; r < kWin / 1024; ++r) {
      const int i = r * 1024 + tid;
      if (i < live) j.order[base + i] = base + (int)(A[i] & (kWin - 1));
    }
    const int tiles = (live + 127) / 128;
    for (int t = wave; t < tiles; t += 16) {
      const int i0 = t * 128 + lane, i1 = i0 + 64;
      uint32_t m = 0;
      if (i0 < live) m |= M[A[i0] & (kWin - 1)];
      if (i1 < live) m |= M[A[i1] & (kWin - 1)];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d, 64);
      if (lane == 0) {
        const uint32_t sm = step_mask_of(m, j.subm);
        j.smask[win * (kWin / 128) + t] = sm;
        // write-through (sc1): the payload of the in-launch hand-off below needs no release fence then
        int32_t *p = j.tile_cnt + win * (kWin / 128) + t;
        const int c = __popc(sm);
        asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(c) : "memory");
      }
    }
    __syncthreads();
  }
  // 4. the job's last workgroup turns the tiles' counts into the unit prefix.  In-launch hand-off by the guide's counter
  // recipe (Guideline 16, R1): sc1 stores above, every storing wave drains them, workgroup barrier, ONE relaxed agent-scope
  // ticket; the last arriver acquires (agent scope) and reads with plain loads.  The ticket is left at zero.
  ORDER_STAMP(12);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ORDER_STAMP(13);
  if (tid == 0) {
    const int old = __hip_atomic_fetch_add(j.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == (int)gridDim.x - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(j.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    cnt[0] = last;
  }
  __syncthreads();
  ORDER_STAMP(14);
  if (!cnt[0]) return;
  __syncthreads();
  const int ntiles = (n + 127) / 128;
  const int per = (ntiles + 1023) / 1024;
  const int t0 = tid * per, t1 = min(ntiles, t0 + per);
  int s = 0;
  for (int t = t0; t < t1; ++t) s += j.tile_cnt[t];
  // workgroup exclusive scan over 1024 threads
  int *wsum = cnt;
  int inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    before += i < wave ? wsum[i] : 0;
    total += wsum[i];
  }
  int run = before + inc - s;
  for (int t = t0; t < t1; ++t) {
    j.bal[t] = run;
    run += j.tile_cnt[t];
  }
  if (tid == 1023) j.bal[ntiles] = total;
  ORDER_STAMP(15);
}

}  // namespace

int dcl_internal_order_rows(const DclOrderJobs &jobs, int njobs, dclStream_t stream) {
  DCL_CHECK_ARG(njobs >= 1 && njobs <= DCL_ORDER_MAX_JOBS);
  long long most = 1;
  for (int i = 0; i < njobs; ++i) {
    const DclOrderJob &j = jobs.job[i];
    DCL_CHECK_ARG(j.out_indices && j.in_mask && j.rowmask && j.order && j.tile_cnt && j.bal && j.smask && j.ticket && j.cap > 0 &&
                  (j.S_in == 8 || j.S_in == 16 || j.S_in == 32 || j.S_in == 64) &&
                  (j.n_dev || (j.n_host >= 0 && j.n_host <= j.cap)));
    const long long rows = j.n_dev ? j.cap : j.n_host;
    if (rows > most) most = rows;
  }
  // workgroups loop over their job's windows: the grid is sized for the live work of typical launches (<= 32 windows =
  // 262 144 rows per job in flight at once), not for the row capacities of capacity mode
  const long long wins = (most + kWin - 1) / kWin;
  const int gx = (int)(wins < 32 ? wins : 32);
  const size_t lds = (size_t)(3 * kWin + 8 * kChunks) * sizeof(uint32_t);
  (void)hipFuncSetAttribute((const void *)k_order_windows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_order_masks, dim3(dcl_grid_1d(most, 256, 1024), njobs), dim3(256), 0, (hipStream_t)stream, jobs);
  hipLaunchKernelGGL(k_order_windows, dim3(gx, njobs), dim3(1024), lds, (hipStream_t)stream, jobs);
  DCL_LAUNCH_CHECK();
  return 0;
}

#ifdef DCL_DIAG
DCL_API int dcl_debug_order_stamps(unsigned long long *host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_order_stamps), sizeof(unsigned long long) * 16);
}
#endif

// ---- op-level entry: ONE layer (tests, the spconv shim); workspace layout queried with dcl_order_rows_ws_bytes
namespace {
struct OrderLayout {
  size_t rowmask, tile_cnt, ticket, total;
};
bool order_layout(int cap, OrderLayout *L) {
  if (cap <= 0) return false;
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  size_t off = 0;
  L->rowmask = off;  off = up(off + sizeof(uint32_t) * (size_t)cap);
  L->tile_cnt = off; off = up(off + sizeof(int32_t) * ((size_t)(cap + 127) / 128 + 1));
  L->ticket = off;   off = up(off + 64);
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
  j.subm = subm; j.order = order; j.tile_cnt = reinterpret_cast<int32_t *>(base + L.tile_cnt); j.bal = bal; j.smask = smask;
  j.ticket = reinterpret_cast<int32_t *>(base + L.ticket);
  j.rowmask = reinterpret_cast<uint32_t *>(base + L.rowmask);
  dcl_internal_zero_words(j.ticket, 1, (hipStream_t)stream);       // (the backbone runner's tickets sit in a region it zeroes anyway)
  return dcl_internal_order_rows(jobs, 1, stream);
}

// rulebook.hip -- device-side sparse-conv rulebooks without sort, dense int grid or host sync.
//
// Replaces getIndicePair<3> (libs/spconv/include/spconv/spconv_ops.h:27-137) and its kernels
// (indice.cu.h:24-208): prepareIndicePairsKernel + torch::_unique + assignGridAndIndiceOut +
// assignIndicePairs for conv/pool, prepareSubMGrid + getSubMIndicePairs for submanifold conv.
//
// MI355X design: an active set lives in an occupancy BITMASK over the batch x S^3 grid
// (64^3 bits = 32 KiB per crop, L2-resident) plus an exclusive popcount prefix per 32-bit word.
// The rank of a voxel in ascending linear-index order -- exactly the output numbering the
// reference obtains from sort+unique (spconv_ops.h:126) -- is then
//     wprefix[word] + popc(mask[word] & lowbits)
// so output ids need no sort, no 4 B/cell dense grid (1 MiB per crop in the reference) and no
// count read-back.  Rulebooks are emitted in GATHER form (nbr[k][o] = input row or -1), which
// lets the conv/pool kernels accumulate offsets k = 0..26 in the reference's order in registers,
// without scatter atomics.  All work is integer/bit traffic: HBM/L2-bound, no MFMA.
#include "common.h"

namespace {

constexpr int kScanWords = 1024;  // mask words per scan block (256 threads x 4)

__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// block of 256 threads: returns exclusive prefix of v, *total = block sum
__device__ __forceinline__ int block_excl_scan_256(int v, int *total) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int inc = wave_incl_scan(v);
  if (lane == 63) wsum[wid] = inc;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) base += (i < wid) ? wsum[i] : 0;
  *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return base + inc - v;
}

__global__ void k_block_popc(const uint32_t *__restrict__ mask, int nwords, int32_t *__restrict__ block_sums) {
  const int w0 = blockIdx.x * kScanWords + threadIdx.x * 4;
  int s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += (w0 + j < nwords) ? __popc(mask[w0 + j]) : 0;
  int total;
  block_excl_scan_256(s, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ void k_scan_words(const uint32_t *__restrict__ mask, int nwords,
                             const int32_t *__restrict__ block_sums, int32_t *__restrict__ wprefix) {
  int part = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) part += block_sums[i];
  int base;
  block_excl_scan_256(part, &base);
  const int w0 = blockIdx.x * kScanWords + threadIdx.x * 4;
  int c[4], s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    c[j] = (w0 + j < nwords) ? __popc(mask[w0 + j]) : 0;
    s += c[j];
  }
  int total;
  int ex = block_excl_scan_256(s, &total) + base;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (w0 + j < nwords) wprefix[w0 + j] = ex;
    ex += c[j];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) wprefix[nwords] = base + total;
}

// small masks (<= 32 scan blocks): one launch -- every block counts the bits in front of it itself instead of reading
// block sums written by a separate k_block_popc launch (the whole mask is at most 128 KiB and L2-resident)
__global__ void k_scan_words_fused(const uint32_t *__restrict__ mask, int nwords, int32_t *__restrict__ wprefix) {
  int part = 0;
  const int before = (int)blockIdx.x * kScanWords;              // multiple of 4 words, mask is 16-B aligned
  const uint4 *m4 = reinterpret_cast<const uint4 *>(mask);
  for (int i = threadIdx.x; i < (before >> 2); i += 256) {
    const uint4 v = m4[i];
    part += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  }
  int base;
  block_excl_scan_256(part, &base);
  const int w0 = blockIdx.x * kScanWords + threadIdx.x * 4;
  int c[4], s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    c[j] = (w0 + j < nwords) ? __popc(mask[w0 + j]) : 0;
    s += c[j];
  }
  int total;
  int ex = block_excl_scan_256(s, &total) + base;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (w0 + j < nwords) wprefix[w0 + j] = ex;
    ex += c[j];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) wprefix[nwords] = base + total;
}

__device__ __forceinline__ int grid_lookup(const uint32_t *__restrict__ mask, const int32_t *__restrict__ wprefix,
                                           const int32_t *__restrict__ perm, int lin) {
  const int w = lin >> 5;
  const uint32_t m = mask[w];
  const uint32_t bit = 1u << (lin & 31);
  if (!(m & bit)) return -1;
  const int r = wprefix[w] + __popc(m & (bit - 1));
  return perm ? perm[r] : r;
}

// rows whose batch index lies in [b_lo, b_lo + nb) enter the grid, re-based to batch 0 (a batch window lets several
// backbone passes share one occupied-voxel array)
__global__ void k_mark_rows(const int32_t *__restrict__ indices, const int32_t *__restrict__ n_dev, int n_host, int S,
                            int b_lo, int nb, uint32_t *__restrict__ mask) {
  const int n = n_dev ? min(*n_dev, n_host) : n_host;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int4 p = reinterpret_cast<const int4 *>(indices)[i];
    const int bb = p.x - b_lo;
    if ((unsigned)bb >= (unsigned)nb) continue;
    const int lin = ((bb * S + p.y) * S + p.z) * S + p.w;
    atomicOr(&mask[lin >> 5], 1u << (lin & 31));
  }
}

__global__ void k_fill_perm(const int32_t *__restrict__ indices, const int32_t *__restrict__ n_dev, int n_host, int S,
                            int b_lo, int nb, const uint32_t *__restrict__ mask, const int32_t *__restrict__ wprefix,
                            int32_t *__restrict__ perm) {
  const int n = n_dev ? min(*n_dev, n_host) : n_host;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int4 p = reinterpret_cast<const int4 *>(indices)[i];
    const int bb = p.x - b_lo;
    if ((unsigned)bb >= (unsigned)nb) continue;
    const int lin = ((bb * S + p.y) * S + p.z) * S + p.w;
    perm[grid_lookup(mask, wprefix, nullptr, lin)] = i;
  }
}

// every input voxel marks the outputs it reaches: o = (p + pad - k) / stride when divisible
// (the set getValidOutPos enumerates, geometry.h:23-85, for dilation 1).
__global__ void k_mark_conv_outputs(const int32_t *__restrict__ in_indices, const int32_t *__restrict__ n_dev,
                                    int n_host, int S_out, int ks, int stride, int pad,
                                    uint32_t *__restrict__ out_mask) {
  const int n = n_dev ? *n_dev : n_host;
  const int kvol = ks * ks * ks;
  const long long total = (long long)n * kvol;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / kvol);
    const int k = (int)(t - (long long)i * kvol);
    const int kx = k / (ks * ks), ky = (k / ks) % ks, kz = k % ks;
    const int4 p = reinterpret_cast<const int4 *>(in_indices)[i];
    const int tx = p.y + pad - kx, ty = p.z + pad - ky, tz = p.w + pad - kz;
    if (tx < 0 || ty < 0 || tz < 0) continue;
    if (tx % stride || ty % stride || tz % stride) continue;
    const int ox = tx / stride, oy = ty / stride, oz = tz / stride;
    if (ox >= S_out || oy >= S_out || oz >= S_out) continue;
    const int lin = ((p.x * S_out + ox) * S_out + oy) * S_out + oz;
    atomicOr(&out_mask[lin >> 5], 1u << (lin & 31));
  }
}

// ---- mask-driven output sets (no atomics, no input row list) ------------------------------------------------------
// A z-row of the grid is S <= 64 bits; conv(k3,s1,p1) dilates by one cell on every axis, pool(k3,s2,p1) maps
// inputs {2o-1, 2o, 2o+1} to output o.  Both are pure bit operations on rows: one thread per OUTPUT word.
__device__ __forceinline__ unsigned long long load_zrow(const uint32_t *__restrict__ mask, int b, int x, int y, int S) {
  if ((unsigned)x >= (unsigned)S || (unsigned)y >= (unsigned)S) return 0ull;
  const long long off = (((long long)b * S + x) * S + y) * S;          // bit offset, a multiple of S
  const int w = (int)(off >> 5);
  if (S == 64) return (unsigned long long)mask[w] | ((unsigned long long)mask[w + 1] << 32);
  return ((unsigned long long)mask[w] >> (off & 31)) & ((1ull << S) - 1ull);
}

__device__ __forceinline__ unsigned long long compress_even_bits(unsigned long long x) {
  x &= 0x5555555555555555ull;
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
  x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
  x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
  x = (x | (x >> 16)) & 0x00000000ffffffffull;
  return x;
}

template <int STRIDE>   // 1: conv k3 s1 p1 (S_out = S_in);  2: pool k3 s2 p1 (S_out = S_in / 2)
__device__ __forceinline__ uint32_t out_word_k3(const uint32_t *__restrict__ in_mask, int batch, int S_in, int S_out, int w) {
  const int rows_per_word = S_out >= 32 ? 1 : 32 / S_out;
  const unsigned long long in_rowmask = S_in == 64 ? ~0ull : ((1ull << S_in) - 1ull);
  uint32_t word = 0;
  for (int rr = 0; rr < rows_per_word; ++rr) {
    const long long bit0 = (long long)w * 32 + (long long)rr * S_out;   // first bit of this output row (part)
    const long long row = bit0 / S_out;                                   // = (b*S_out + ox)*S_out + oy
    const int oy = (int)(row % S_out);
    const int ox = (int)((row / S_out) % S_out);
    const int b = (int)(row / ((long long)S_out * S_out));
    if (b >= batch) break;
    unsigned long long u = 0ull;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) u |= load_zrow(in_mask, b, ox * STRIDE + dx, oy * STRIDE + dy, S_in);
    unsigned long long t = (u | (u << 1) | (u >> 1)) & in_rowmask;
    if (STRIDE == 2) t = compress_even_bits(t);
    if (S_out == 64) word = (uint32_t)(t >> ((w & 1) * 32));
    else word |= (uint32_t)t << (rr * S_out);
  }
  return word;
}

template <int STRIDE>
__global__ void k_out_mask_k3(const uint32_t *__restrict__ in_mask, int batch, int S_in, int S_out, int nwords_out,
                              uint32_t *__restrict__ out_mask) {
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < nwords_out; w += gridDim.x * blockDim.x)
    out_mask[w] = out_word_k3<STRIDE>(in_mask, batch, S_in, S_out, w);
}

// The whole mask chain of a backbone pass (conv set, pool set) x 4 levels in ONE launch for 64^3 grids: a crop's occupancy
// is 32 KiB of bits, so one workgroup per crop keeps the current and the next mask in LDS and walks the 8 stages with a
// barrier between them (the row arithmetic of k_out_mask_k3 on crop-local rows; words never straddle crops because every
// level's S^3 is a multiple of 32).  Replaces 8 dependent launches at the head of every pass.  All index arithmetic is
// 32-bit shifts (S is a power of two): with one workgroup per crop the 64-bit divisions of the general kernel would be the
// whole run time.
constexpr int kChainS = 64, kChainWords = kChainS * kChainS * kChainS / 32, kChainThreads = 1024;
// (branch-free: an out-of-range neighbour row reads row (0, 0) and is masked out, so that the nine rows' LDS reads of an
// output row are independent loads in flight together instead of nine guarded round trips)
template <bool IN64>   // IN64: the input grid is 64 wide (two words per row), known to the compiler -- a uniform branch on S
                       // around the LDS read made every one of the nine reads of an output row wait for the one before it
__device__ __forceinline__ unsigned long long local_zrow(const uint32_t *m, int x, int y, int S, int lg) {
  const bool ok = (unsigned)x < (unsigned)S && (unsigned)y < (unsigned)S;
  const int off = (((ok ? x : 0) << lg) + (ok ? y : 0)) << lg;          // bit offset of the z-row inside the crop
  const int w = off >> 5;
  unsigned long long v;
  if constexpr (IN64) v = (unsigned long long)m[w] | ((unsigned long long)m[w + 1] << 32);
  else {
    const uint32_t r = m[w] >> (off & 31);
    v = (unsigned long long)(r & (S == 32 ? ~0u : (1u << S) - 1u));
  }
  return ok ? v : 0ull;
}
template <int STRIDE, bool IN64>
__device__ __forceinline__ unsigned long long local_out_row(const uint32_t *src, int row, int S_in, int lg_in, int lg_out) {
  const int ox = row >> lg_out, oy = row & ((1 << lg_out) - 1);
  unsigned long long u = 0ull;
#pragma unroll
  for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) u |= local_zrow<IN64>(src, ox * STRIDE + dx, oy * STRIDE + dy, S_in, lg_in);
  unsigned long long t = (u | (u << 1) | (u >> 1)) & (IN64 ? ~0ull : ((1ull << S_in) - 1ull));
  if (STRIDE == 2) t = compress_even_bits(t);
  return t;
}
template <int STRIDE, bool IN64>
__device__ __forceinline__ void chain_stage(const uint32_t *src, uint32_t *dst, uint32_t *__restrict__ gout, int S_in,
                                            int S_out, int tid) {
  const int lg_in = 31 - __clz(S_in), lg_out = 31 - __clz(S_out);
  const int rows = S_out << lg_out;
  if (S_out == 64) {                                                    // one thread per row = two words
    for (int row = tid; row < rows; row += kChainThreads) {
      const unsigned long long t = local_out_row<STRIDE, IN64>(src, row, S_in, lg_in, lg_out);
      const uint2 v = make_uint2((uint32_t)t, (uint32_t)(t >> 32));
      reinterpret_cast<uint2 *>(dst)[row] = v;
      reinterpret_cast<uint2 *>(gout)[row] = v;
    }
  } else {
    // S_out <= 32: at most 1024 rows = one per thread; the 32 / S_out rows of a word sit in consecutive lanes and are
    // OR-ed together by shuffles (every lane of the workgroup takes part; a thread per WORD would leave the coarse
    // levels to a handful of threads walking 4-8 rows each -- they were the longest stages of the chain)
    const int lg_rpw = 5 - lg_out, sub = tid & ((1 << lg_rpw) - 1);
    uint32_t bits = 0;
    if (tid < rows) bits = (uint32_t)local_out_row<STRIDE, IN64>(src, tid, S_in, lg_in, lg_out) << (sub << lg_out);
    for (int d = 1; d < (1 << lg_rpw); d <<= 1) bits |= __shfl_xor(bits, d, 64);
    if (tid < rows && sub == 0) {
      dst[tid >> lg_rpw] = bits;
      gout[tid >> lg_rpw] = bits;
    }
  }
}
// (barriers: common.h dcl_lds_barrier -- a stage must not wait for its own global stores to land before the next reads LDS)
__device__ __forceinline__ unsigned long long dilate_z64(unsigned long long u) { return u | (u << 1) | (u >> 1); }
// Stage 0 (conv set of level 0, 64^3 -> 64^3) for 1024 threads: thread = (y, 4 consecutive x); the 3 x 3 neighbourhoods of
// its four rows are 6 x-columns of 3 rows, OR-ed over y first: 18 row reads for 4 outputs instead of 36 guarded ones.  Lanes
// walk y: every 64-bit LDS read is conflict-free.
__device__ __forceinline__ void chain_stage0_64(const uint32_t *src32, uint32_t *dst32, uint32_t *__restrict__ gout32, int tid) {
  const unsigned long long *src = reinterpret_cast<const unsigned long long *>(src32);
  unsigned long long *dst = reinterpret_cast<unsigned long long *>(dst32);
  uint2 *gout = reinterpret_cast<uint2 *>(gout32);
  const int y = tid & 63, x0 = (tid >> 6) << 2;
  const int ym = y > 0 ? y - 1 : y, yp = y < 63 ? y + 1 : y;            // (a clamped row ORs in the row itself: harmless)
  unsigned long long u[6];
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    const int x = x0 - 1 + d;
    u[d] = 0ull;
    if ((unsigned)x < 64u) u[d] = src[(x << 6) + ym] | src[(x << 6) + y] | src[(x << 6) + yp];        // (wave-uniform test)
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned long long t = dilate_z64(u[k] | u[k + 1] | u[k + 2]);
    const int row = ((x0 + k) << 6) + y;
    dst[row] = t;
    gout[row] = make_uint2((uint32_t)t, (uint32_t)(t >> 32));
  }
}
__global__ void __launch_bounds__(kChainThreads) k_mask_chain64(const uint32_t *__restrict__ mask0, const DclGeoSets g) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[2][kChainWords];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < kChainWords / 4; i += kChainThreads)
    reinterpret_cast<uint4 *>(buf[0])[i] = reinterpret_cast<const uint4 *>(mask0 + (size_t)b * kChainWords)[i];
  dcl_lds_barrier();
  chain_stage0_64(buf[0], buf[1], const_cast<uint32_t *>(g.mask[0]) + (size_t)b * kChainWords, tid);
  dcl_lds_barrier();
  chain_stage<2, true>(buf[1], buf[0], const_cast<uint32_t *>(g.mask[1]) + (size_t)b * 1024, kChainS, kChainS / 2, tid);
  dcl_lds_barrier();
  int S_in = kChainS / 2, cur = 0;
#pragma unroll 1
  for (int i = 2; i < 8; ++i) {
    const int S_out = g.S[i];
    uint32_t *gout = const_cast<uint32_t *>(g.mask[i]) + (size_t)b * ((S_out * S_out * S_out) >> 5);
    if (i & 1) chain_stage<2, false>(buf[cur], buf[cur ^ 1], gout, S_in, S_out, tid);
    else chain_stage<1, false>(buf[cur], buf[cur ^ 1], gout, S_in, S_out, tid);
    dcl_lds_barrier();
    cur ^= 1;
    S_in = S_out;
  }
}

// ---- the WHOLE geometry stage of a pass of a handful of crops in one launch ------------------------------------------------
// (64^3 grids, at most kGeoSmallBatch crops: one-image calls and small batches, where the stage's 8 launches -- zero, mark, scan, fill_perm,
// mask chain, block counts, word prefixes, enumerate -- are a third of the sparse half's critical path.)  One workgroup
// per crop: marks the crop's voxels in an LDS bitmask, walks the 8-stage mask chain in LDS (as k_mask_chain64), counts
// every set, exchanges the nine counts with the other crops' workgroups (a flag per crop in `comm`, zeroed by the caller
// before the launch; crop c waits for crops < c only, and workgroups are dispatched in index order, so the wait cannot
// deadlock), then writes word prefixes, decoded rows and the level-0 permutation of its own crop at the ranks
// base(crops before it) + local rank -- the same ascending linear-index numbering as the separate launches, bit for bit.
#ifdef DCL_DIAG
// tools/geo_stamps.py: s_memrealtime (100 MHz) at the phase boundaries of workgroup 0, thread 0 -- diagnostic library only
__device__ unsigned long long g_geo_stamps[32];
#define GEO_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_geo_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define GEO_STAMP_END(i) do { __syncthreads(); GEO_STAMP(i); } while (0)
#else
#define GEO_STAMP(i) do { } while (0)
#define GEO_STAMP_END(i) do { } while (0)
#endif
DCL_HOOK_INT(kGeoSmallBatch, 16);     // most crops of a pass that takes the one-launch geometry stage (comm holds 64 crops).
// (Stage time by HIP events, one launch vs the separate launches, 1024-point crops: 8 crops 22 us either way (the former limit),
//  12: 26 vs 36, 20: 55 vs 63, 32: 62 vs 71, 40: 65 vs 74 -- but the whole forward of 32 crops does not move (3.884 vs 3.887 ms:
//  the stage hides behind the point branch there) and 12 crops gain 0.6 %; every workgroup walks the whole voxel list, so at
//  32 x 12288-point crops (41k rows) the one launch is the slower one, 75 vs 72 us: hence also a bound on the rows.)
constexpr int kGeoSmallRows = 32768;
constexpr int kGeoSmallMax = 64, kGeoCommStride = 16;
static_assert(kGeoSmallMax * kGeoCommStride == kChainThreads, "k_geometry_small: one thread per exchange word");
struct GeoSmallArgs {
  const int32_t *occ, *n_dev;
  int n_host, batch_lo, batch;
  uint32_t *mask0;
  int32_t *wprefix0, *perm0, *comm;
  // optional rider: PG_OP.voxelize_fp of the pass's points (voxelize.hip: voxelize_fp_kernel, same arithmetic) by the
  // workgroups behind the crops' -- it depends on the input only, and the launch otherwise leaves all but `batch` CUs idle
  const float *vx_feats;
  const int32_t *vx_rules;
  float *vx_out;
  int vx_rows, vx_ma, vx_planes, vx_avg;
};
// ---- k_geometry_small, second form.  What the first form (one thread = 8 consecutive words, nine rolled 1024-thread scans,
// per-thread row decoding) spent its 34 us on, by stamps (tools/geo_stamps.py, one crop): 12 us in the mask chain -- 4 of
// them in the first stage (36 guarded LDS row reads per thread) and ~1 us per later stage, each of which waited at its
// barrier for its own global stores to LAND (__syncthreads waits for vmcnt(0)); 5.5 us in the scans (every thread summing
// 9 x 16 wave totals); 12 us decoding rows, because a crop's voxels sit in a few hundred neighbouring words and the threads
// that owned them walked 100+ bits each while the rest of the workgroup waited.  This form:
//   * barriers that wait for LDS only (the stage's global stores stay in flight); all nine sets stay in LDS (sets 2..8 in
//     their own 9 KiB), so nothing is read back from memory;
//   * the first stage reads 18 rows per thread for 4 outputs (a thread owns 4 consecutive x at one y);
//   * ownership interleaved inside a wave (lane L owns words 64 j + L of the wave's 512), the counts of set 0, set 1 and
//     of set 1's non-empty words packed into ONE DPP scan per round, one more packed scan for sets 2 + 3, one for 4..8;
//     wave totals are scanned by ONE wave per set;
//   * set 1 (5/8 of all rows) is decoded from a list of its non-empty words dealt round-robin over the 1024 threads.
// Same ranks, same rows, bit for bit (the ranks are prefix sums of the same popcounts in the same word order).
constexpr int kSmallOff2 = 0, kSmallOff3 = 1024, kSmallOff4 = 2048, kSmallWords = 2048 + 128 + 128 + 16 + 16 + 2;
// inclusive scan over the 64 lanes by DPP (row_shr 1/2/4/8 inside rows of 16, then row_bcast 15 / 31); fields packed into
// v scan independently as long as none overflows into the next
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
  return v;
}

__global__ void __launch_bounds__(kChainThreads) k_geometry_small(const GeoSmallArgs a, const DclGeoSets g) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[2][kChainWords];     // sets 0 and 1 (64^3), kept to the end
  __shared__ __attribute__((aligned(16))) uint32_t small[kSmallWords + 6];  // sets 2..8 back to back
  __shared__ uint32_t s_list[kChainWords];              // non-empty words of set 1: word | rank inside the crop << 13
  __shared__ int32_t s_wp0[kChainWords];                // set 0's word prefixes (the level-0 permutation reads them back)
  __shared__ int s_w[10][16], s_wbase[10][16], s_tot[10], s_base[9];     // [9] = set 1's non-empty words
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (c >= a.batch) {                                   // rider workgroups: voxelize_fp (see GeoSmallArgs)
    const long long total = (long long)a.vx_rows * a.vx_planes;
    for (long long t = (long long)(c - a.batch) * kChainThreads + tid; t < total; t += (long long)(gridDim.x - a.batch) * kChainThreads) {
      const int row = (int)(t / a.vx_planes);
      const int plane = (int)(t - (long long)row * a.vx_planes);
      const int32_t *r = a.vx_rules + (size_t)row * (a.vx_ma + 1);
      const int n_active = r[0];
      const float mult = (a.vx_avg && n_active > 0) ? 1.0f / (float)n_active : 1.0f;
      float acc = 0.0f;
      for (int i = 1; i <= n_active; ++i) acc = acc + mult * a.vx_feats[(size_t)r[i] * a.vx_planes + plane];
      a.vx_out[t] = acc;
    }
    return;
  }
  GEO_STAMP(0);
  if (c == 0 && tid < 16 && g.zero_words) g.zero_words[tid] = 0;            // tickets of later launches of the pass
  // 1. this crop's occupancy (rows of other crops are skipped; rows are re-based by batch_lo)
  const int n = a.n_dev ? min(*a.n_dev, a.n_host) : a.n_host;        // (asked for before the zeroing: two dependent round trips)
  for (int i = tid; i < kChainWords / 4; i += kChainThreads) reinterpret_cast<uint4 *>(buf[0])[i] = make_uint4(0u, 0u, 0u, 0u);
  if (tid < 160) s_w[tid >> 4][tid & 15] = 0;
  if (tid < 9) s_base[tid] = 0;
  dcl_lds_barrier();
  // (every crop's workgroup walks the whole voxel list: the crop ids of four rows first -- independent 4-byte loads -- then
  //  the rows that are this crop's)
  for (int i0 = tid; i0 < n; i0 += 4 * kChainThreads) {
    int bx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bx[u] = i0 + u * kChainThreads < n ? a.occ[4 * (size_t)(i0 + u * kChainThreads)] : a.batch_lo - 1;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (bx[u] - a.batch_lo != c) continue;
      const int4 p = reinterpret_cast<const int4 *>(a.occ)[i0 + u * kChainThreads];
      const int lin = ((((p.y << 6) + p.z) << 6)) + p.w;
      atomicOr(&buf[0][lin >> 5], 1u << (lin & 31));
    }
  }
  dcl_lds_barrier();
  GEO_STAMP(1);
  {
    uint4 *m0 = reinterpret_cast<uint4 *>(a.mask0 + (size_t)c * kChainWords);
    m0[2 * tid] = reinterpret_cast<const uint4 *>(buf[0])[2 * tid];
    m0[2 * tid + 1] = reinterpret_cast<const uint4 *>(buf[0])[2 * tid + 1];
  }
  GEO_STAMP(2);
  // 2. the mask chain: stage 0 (chain_stage0_64), stage 1 out of the 64-wide grid, then the six stages of the narrow grids
  chain_stage0_64(buf[0], buf[1], const_cast<uint32_t *>(g.mask[0]) + (size_t)c * kChainWords, tid);
  dcl_lds_barrier();
  GEO_STAMP(16);
  chain_stage<2, true>(buf[1], small, const_cast<uint32_t *>(g.mask[1]) + (size_t)c * 1024, kChainS, kChainS / 2, tid);
  dcl_lds_barrier();
  GEO_STAMP(17);
  {
    const uint32_t *src = small;
    int S_in = kChainS / 2, off = 1024;
#pragma unroll 1
    for (int i = 2; i < 8; ++i) {
      const int S_out = g.S[i], nw = (S_out * S_out * S_out) >> 5;
      uint32_t *gout = const_cast<uint32_t *>(g.mask[i]) + (size_t)c * nw;
      uint32_t *dst = small + off;
      if (i & 1) chain_stage<2, false>(src, dst, gout, S_in, S_out, tid);
      else chain_stage<1, false>(src, dst, gout, S_in, S_out, tid);
      dcl_lds_barrier();
      GEO_STAMP(16 + i);
      src = dst;
      off += nw;
      S_in = S_out;
    }
  }
  GEO_STAMP(3);
  // 3. counts.  Sets 0 / 1: lane L of wave W owns words 512 W + 64 j + L, j = 0..7 -- neighbouring words, where a crop's
  // voxels cluster, go to different lanes.  Per round ONE scan of  popc(set 0) | popc(set 1) << 12 | (set 1 word != 0) << 24
  // (<= 2048 | 2048 | 64 per round: no field overflows); e01[j] / en[j] = this word's exclusive ranks inside the wave.
  uint32_t e01[8], en[8];
  {
    uint32_t run0 = 0, run1 = 0, runn = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int w = (wid << 9) + (j << 6) + lane;
      const uint32_t m1 = buf[1][w];
      const uint32_t p = (uint32_t)__popc(buf[0][w]) | ((uint32_t)__popc(m1) << 12) | ((m1 ? 1u : 0u) << 24);
      const uint32_t inc = wave_incl_scan_dpp(p);
      const uint32_t ex = inc - p, tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      e01[j] = (run0 + (ex & 0xfffu)) | ((run1 + ((ex >> 12) & 0xfffu)) << 16);          // <= 16384 each
      en[j] = runn + (ex >> 24);
      run0 += tot & 0xfffu; run1 += (tot >> 12) & 0xfffu; runn += tot >> 24;
    }
    if (lane == 0) { s_w[0][wid] = (int)run0; s_w[1][wid] = (int)run1; s_w[9][wid] = (int)runn; }
  }
  // sets 2 / 3 (32^3: 1024 words, word t to thread t), one packed scan
  uint32_t e23;
  {
    const uint32_t p = (uint32_t)__popc(small[kSmallOff2 + tid]) | ((uint32_t)__popc(small[kSmallOff3 + tid]) << 16);
    const uint32_t inc = wave_incl_scan_dpp(p);
    e23 = inc - p;
    if (lane == 63) { s_w[2][wid] = (int)(inc & 0xffffu); s_w[3][wid] = (int)(inc >> 16); }
  }
  // sets 4..8 (128 + 128 + 16 + 16 + 2 words, back to back in `small`): item v = tid < 290.  Set 4 = waves 0, 1; set 5 = waves
  // 2, 3; sets 6 / 7 / 8 = lanes 0..15 / 16..31 / 32, 33 of wave 4: one plain scan per wave, set boundaries by readlane.
  int sv = -1, wv = 0, e48 = 0;                         // this thread's item: set, word inside the set, exclusive rank
  if (wid < 5) {                                        // (wave-uniform)
    const bool has = tid < 290;
    const uint32_t p = has ? (uint32_t)__popc(small[kSmallOff4 + tid]) : 0u;
    const uint32_t inc = wave_incl_scan_dpp(p);
    const int ex = (int)(inc - p);
    if (wid < 4) {
      sv = 4 + (wid >> 1); wv = tid & 127; e48 = ex;
      if (lane == 63) s_w[sv][wid] = (int)inc;
    } else {
      const int at16 = __builtin_amdgcn_readlane(ex, 16), at32 = __builtin_amdgcn_readlane(ex, 32), end = __builtin_amdgcn_readlane((int)inc, 33);
      if (lane < 16) { sv = 6; wv = lane; e48 = ex; }
      else if (lane < 32) { sv = 7; wv = lane - 16; e48 = ex - at16; }
      else if (lane < 34) { sv = 8; wv = lane - 32; e48 = ex - at32; }
      if (lane == 0) { s_w[6][4] = at16; s_w[7][4] = at32 - at16; s_w[8][4] = end - at32; }
    }
  }
  GEO_STAMP(4);
  dcl_lds_barrier();
  if (wid < 10) {                                       // wave s: exclusive scan of set s's 16 wave totals
    const int v = lane < 16 ? s_w[wid][lane] : 0;
    const int inc = (int)wave_incl_scan_dpp((uint32_t)v);
    if (lane < 16) s_wbase[wid][lane] = inc - v;
    if (lane == 15) s_tot[wid] = inc;
  }
  dcl_lds_barrier();
  GEO_STAMP(5);
  // 4. counts out, bases in (a pass of ONE crop has nobody to tell).  Thread (c2, s) fetches set s's count of crop c2 < c: all
  // crops' flags and counts in flight together (nine lanes walking the crops one after the other were 2 round trips per crop)
  if (a.batch > 1) {
    int32_t *mine = a.comm + c * kGeoCommStride;
    if (tid == 0) {
      for (int s = 0; s < 9; ++s) __hip_atomic_store(mine + s, s_tot[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 15, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int c2 = tid >> 4, s = tid & 15;              // (kGeoSmallMax crops x kGeoCommStride words = the 1024 threads)
    if (c2 < c && s < 9) {
      const int32_t *theirs = a.comm + c2 * kGeoCommStride;
      while (__hip_atomic_load(theirs + 15, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
      atomicAdd(&s_base[s], __hip_atomic_load(theirs + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
  }
  dcl_lds_barrier();
  GEO_STAMP(6);
  // 5. word prefixes (every set) and decoded rows (sets 1..8) at base(crops before this one) + rank inside the crop
  auto decode = [&](int set, int S, int w, uint32_t m, int r) __attribute__((always_inline)) {
    const int lg = 31 - __clz(S);
    int4 *rows = reinterpret_cast<int4 *>(g.indices[set - 1]);
    const int cap = g.cap[set - 1];
    while (m) {
      const int bit = __ffs(m) - 1;
      m &= m - 1;
      const int lin = (w << 5) + bit;
      if (r < cap) rows[r] = make_int4(c, lin >> (2 * lg), (lin >> lg) & (S - 1), lin & (S - 1));
      ++r;
    }
  };
  {
    const int b0 = s_base[0] + s_wbase[0][wid], b1 = s_wbase[1][wid], bn = s_wbase[9][wid], g1 = s_base[1];
    int32_t *wp0 = a.wprefix0 + (size_t)c * kChainWords, *wp1 = g.wprefix[0] + (size_t)c * kChainWords;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int w = (wid << 9) + (j << 6) + lane;
      const int r0 = b0 + (int)(e01[j] & 0xffffu), r1 = b1 + (int)(e01[j] >> 16);       // r1: inside the crop
      wp0[w] = r0;
      s_wp0[w] = r0;
      wp1[w] = g1 + r1;
      if (buf[1][w]) s_list[bn + (int)en[j]] = (uint32_t)w | ((uint32_t)r1 << 13);
    }
  }
  {
    const int r2 = s_base[2] + s_wbase[2][wid] + (int)(e23 & 0xffffu), r3 = s_base[3] + s_wbase[3][wid] + (int)(e23 >> 16);
    g.wprefix[1][(size_t)c * 1024 + tid] = r2;
    g.wprefix[2][(size_t)c * 1024 + tid] = r3;
    decode(2, 32, tid, small[kSmallOff2 + tid], r2);
    decode(3, 32, tid, small[kSmallOff3 + tid], r3);
  }
  if (sv >= 0) {
    const int S = g.S[sv - 1], nw = (S * S * S) >> 5;
    const int r = s_base[sv] + s_wbase[sv][wid] + e48;
    g.wprefix[sv - 1][(size_t)c * nw + wv] = r;
    decode(sv, S, wv, small[kSmallOff4 + tid], r);
  }
  dcl_lds_barrier();                                        // set 1's list and set 0's prefixes are in LDS
  {
    const int ne = s_tot[9], g1 = s_base[1];
#pragma unroll 1
    for (int e = tid; e < ne; e += kChainThreads) {
      const uint32_t ent = s_list[e];
      const int w = (int)(ent & 0x1fffu);
      // (a word of a 64-wide grid is half a z-row: x, y and the z half are the word's, only the bit moves)
      uint32_t m = buf[1][w];
      int r = g1 + (int)(ent >> 13);
      int4 row = make_int4(c, w >> 7, (w >> 1) & 63, (w & 1) << 5);
      int4 *rows = reinterpret_cast<int4 *>(g.indices[0]) + r;
      int left = g.cap[0] - r;
      while (m) {
        const int bit = __ffs(m) - 1;
        m &= m - 1;
        if (left > 0) *rows = make_int4(row.x, row.y, row.z, row.w + bit);
        ++rows;
        --left;
      }
    }
  }
  GEO_STAMP(7);
  if (c == a.batch - 1 && tid < 9) {                    // totals: the entry past the last crop's words, and the live counts
    const int s = tid, S = s == 0 ? kChainS : g.S[s - 1], nw = (S * S * S) >> 5;
    const int total = s_base[s] + s_tot[s];
    (s == 0 ? a.wprefix0 : g.wprefix[s - 1])[(size_t)a.batch * nw] = total;
    if (s > 0 && g.n_out[s - 1]) *g.n_out[s - 1] = total;
  }
  GEO_STAMP(8);
  // 6. level-0 permutation: rank -> row of the caller's voxel list (mask and prefixes of set 0 are still in LDS)
  if (a.perm0) {
    for (int i0 = tid; i0 < n; i0 += 4 * kChainThreads) {
      int bx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) bx[u] = i0 + u * kChainThreads < n ? a.occ[4 * (size_t)(i0 + u * kChainThreads)] : a.batch_lo - 1;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (bx[u] - a.batch_lo != c) continue;
        const int i = i0 + u * kChainThreads;
        const int4 p = reinterpret_cast<const int4 *>(a.occ)[i];
        const int lin = ((((p.y << 6) + p.z) << 6)) + p.w;
        a.perm0[s_wp0[lin >> 5] + __popc(buf[0][lin >> 5] & ((1u << (lin & 31)) - 1u))] = i;
      }
    }
  }
  GEO_STAMP_END(9);
}

// decode every set bit into its (b,x,y,z) row at its rank (assignGridAndIndiceOutKernel,
// indice.cu.h:112-128, without the sort that precedes it).
__global__ void k_enumerate(const uint32_t *__restrict__ mask, const int32_t *__restrict__ wprefix, int nwords,
                            int S, int32_t *__restrict__ out_indices, int cap, int32_t *__restrict__ n_out_dev) {
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += gridDim.x * blockDim.x) {
    uint32_t m = mask[w];
    int r = wprefix[w];
    while (m) {
      const int bit = __ffs(m) - 1;
      m &= m - 1;
      int lin = (w << 5) + bit;
      int4 o;
      o.w = lin % S; lin /= S;
      o.z = lin % S; lin /= S;
      o.y = lin % S; lin /= S;
      o.x = lin;
      if (r < cap) reinterpret_cast<int4 *>(out_indices)[r] = o;
      ++r;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && n_out_dev) *n_out_dev = wprefix[nwords];
}

// ---- the same scan + enumerate for several sets in one launch each (blockIdx.y = set): the masks of a backbone pass
// depend only on each other, so the geometry stage first chains the 8 mask kernels and then ranks / decodes all sets at once
__global__ void k_block_popc_sets(const DclGeoSets g) {
  const int set = blockIdx.y, nwords = g.nwords[set];
  if (g.zero_words && blockIdx.x == 0 && set == 0 && threadIdx.x < 16) g.zero_words[threadIdx.x] = 0;   // tickets of later launches of the pass
  if ((int)blockIdx.x * kScanWords >= nwords) return;
  const uint32_t *__restrict__ mask = g.mask[set];
  const int w0 = blockIdx.x * kScanWords + threadIdx.x * 4;
  int s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += (w0 + j < nwords) ? __popc(mask[w0 + j]) : 0;
  int total;
  block_excl_scan_256(s, &total);
  if (threadIdx.x == 0) g.block_sums[set][blockIdx.x] = total;
}

__global__ void k_scan_words_sets(const DclGeoSets g) {
  const int set = blockIdx.y, nwords = g.nwords[set];
  if ((int)blockIdx.x * kScanWords >= nwords) return;
  const uint32_t *__restrict__ mask = g.mask[set];
  const int32_t *__restrict__ block_sums = g.block_sums[set];
  int32_t *__restrict__ wprefix = g.wprefix[set];
  int part = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) part += block_sums[i];
  int base;
  block_excl_scan_256(part, &base);
  const int w0 = blockIdx.x * kScanWords + threadIdx.x * 4;
  int c[4], s = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    c[j] = (w0 + j < nwords) ? __popc(mask[w0 + j]) : 0;
    s += c[j];
  }
  int total;
  int ex = block_excl_scan_256(s, &total) + base;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (w0 + j < nwords) wprefix[w0 + j] = ex;
    ex += c[j];
  }
  if (((int)blockIdx.x + 1) * kScanWords >= nwords && threadIdx.x == 0) {      // last block of this set
    wprefix[nwords] = base + total;
    if (g.n_out[set]) *g.n_out[set] = base + total;
  }
}

__global__ void k_enumerate_sets(const DclGeoSets g) {
  const int set = blockIdx.y, nwords = g.nwords[set], S = g.S[set], cap = g.cap[set];
  const uint32_t *__restrict__ mask = g.mask[set];
  const int32_t *__restrict__ wprefix = g.wprefix[set];
  int4 *__restrict__ out = reinterpret_cast<int4 *>(g.indices[set]);
  for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += gridDim.x * blockDim.x) {
    uint32_t m = mask[w];
    int r = wprefix[w];
    while (m) {
      const int bit = __ffs(m) - 1;
      m &= m - 1;
      int lin = (w << 5) + bit;
      int4 o;
      o.w = lin % S; lin /= S;
      o.z = lin % S; lin /= S;
      o.y = lin % S; lin /= S;
      o.x = lin;
      if (r < cap) out[r] = o;
      ++r;
    }
  }
}

// gather-form rulebook: for output o and offset k the feeding input sits at p = o*stride - pad + k.
__global__ void k_build_nbr(const int32_t *__restrict__ out_indices, const int32_t *__restrict__ n_out_dev,
                            int n_out_host, const uint32_t *__restrict__ in_mask,
                            const int32_t *__restrict__ in_wprefix, const int32_t *__restrict__ in_perm,
                            int S_in, int ks, int stride, int pad, int32_t *__restrict__ nbr, int cap) {
  int n = n_out_dev ? *n_out_dev : n_out_host;
  n = n < cap ? n : cap;
  const int kvol = ks * ks * ks;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x) {
    const int4 q = reinterpret_cast<const int4 *>(out_indices)[o];
    const int bx = q.y * stride - pad, by = q.z * stride - pad, bz = q.w * stride - pad;
    int k = 0;
    for (int kx = 0; kx < ks; ++kx) {
      const int px = bx + kx;
      for (int ky = 0; ky < ks; ++ky) {
        const int py = by + ky;
        for (int kz = 0; kz < ks; ++kz, ++k) {
          const int pz = bz + kz;
          int v = -1;
          if ((unsigned)px < (unsigned)S_in && (unsigned)py < (unsigned)S_in && (unsigned)pz < (unsigned)S_in)
            v = grid_lookup(in_mask, in_wprefix, in_perm, ((q.x * S_in + px) * S_in + py) * S_in + pz);
          nbr[(size_t)k * cap + o] = v;
        }
      }
    }
    (void)kvol;
  }
}

__global__ void k_pairs_init(int32_t *__restrict__ pairs, long long n, int32_t *__restrict__ indice_num, int kvol) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
    pairs[t] = -1;
  if (blockIdx.x == 0 && (int)threadIdx.x < kvol) indice_num[threadIdx.x] = 0;
}

__global__ void k_pairs_export(const int32_t *__restrict__ nbr, int cap, const int32_t *__restrict__ n_out_dev,
                               int n_out_host, int kvol, int32_t *__restrict__ pairs, int n_in_cap,
                               int32_t *__restrict__ indice_num) {
  int n = n_out_dev ? *n_out_dev : n_out_host;
  n = n < cap ? n : cap;
  const long long total = (long long)n * kvol;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(t / n);
    const int o = (int)(t - (long long)k * n);
    const int v = nbr[(size_t)k * cap + o];
    if (v < 0) continue;
    const int c = atomicAdd(&indice_num[k], 1);
    if (c < n_in_cap) {
      pairs[((size_t)k * 2 + 0) * n_in_cap + c] = v;
      pairs[((size_t)k * 2 + 1) * n_in_cap + c] = o;
    }
  }
}

// reference pair lists -> gather table.  An output row occurs at most once per kernel offset (spconv_ops.h:296-344 relies
// on the same fact for its non-atomic scatter-add), so the writes of one offset never collide.
__global__ void k_pairs_import(const int32_t *__restrict__ pairs, int pair_stride, const int32_t *__restrict__ indice_num,
                               int kvol, int n_in, int n_out, int32_t *__restrict__ nbr, int cap,
                               int32_t *__restrict__ bad) {
  const long long total = (long long)kvol * pair_stride;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(t / pair_stride);
    const int j = (int)(t - (long long)k * pair_stride);
    if (j >= indice_num[k]) continue;
    const int i = pairs[((size_t)k * 2 + 0) * pair_stride + j];
    const int o = pairs[((size_t)k * 2 + 1) * pair_stride + j];
    if ((unsigned)i >= (unsigned)n_in || (unsigned)o >= (unsigned)n_out) {
      if (bad) atomicAdd(bad, 1);
      continue;
    }
    nbr[(size_t)k * cap + o] = i;
  }
}

// indiceSummaryRF on the pair format (summaryRF.cu:26-41): rf[out] += 1 per pair.  rf must be zeroed by the caller's launch
// order (k_fill_i32 below).
__global__ void k_pairs_summary_rf(const int32_t *__restrict__ pairs, int pair_stride, const int32_t *__restrict__ indice_num,
                                   int kvol, int n_out, int32_t *__restrict__ rf) {
  const long long total = (long long)kvol * pair_stride;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(t / pair_stride);
    const int j = (int)(t - (long long)k * pair_stride);
    if (j >= indice_num[k]) continue;
    const int o = pairs[((size_t)k * 2 + 1) * pair_stride + j];
    if ((unsigned)o < (unsigned)n_out) atomicAdd(&rf[o], 1);
  }
}

__global__ void k_fill_i32(int32_t *__restrict__ p, long long n, int32_t v, int32_t *__restrict__ zero_word) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
    p[t] = v;
  if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;
}

int scan_mask(const uint32_t *mask, int nwords, int32_t *wprefix, int32_t *scratch, hipStream_t s) {
  const int nblocks = dcl_div_up(nwords, kScanWords);
  if (nblocks <= 32 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0) {
    hipLaunchKernelGGL(k_scan_words_fused, dim3(nblocks), dim3(256), 0, s, mask, nwords, wprefix);
    return 0;
  }
  hipLaunchKernelGGL(k_block_popc, dim3(nblocks), dim3(256), 0, s, mask, nwords, scratch);
  hipLaunchKernelGGL(k_scan_words, dim3(nblocks), dim3(256), 0, s, mask, nwords, scratch, wprefix);
  return 0;
}

inline long long grid_words(int batch, int S) { return ((long long)batch * S * S * S + 31) / 32; }

}  // namespace

// library-internal: zero `nwords` 32-bit words with a kernel (a plain kernel node when the stream is being captured;
// memset nodes are avoided in the whole-forward graph)
__global__ void k_zero_words(uint32_t *__restrict__ p, long long nwords) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  uint4 *p4 = reinterpret_cast<uint4 *>(p);
  const long long n4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? nwords >> 2 : 0;
  for (long long j = i; j < n4; j += stride) p4[j] = make_uint4(0u, 0u, 0u, 0u);
  for (long long j = 4 * n4 + i; j < nwords; j += stride) p[j] = 0u;
}
#ifdef DCL_DIAG
// diagnostic: one lane writes the 100 MHz wall clock into *slot -- a time stamp INSIDE a stream / a captured graph, to see
// what really overlaps in an unprofiled replay (tools/graph_timeline.py)
__global__ void k_debug_stamp(unsigned long long *slot) { *slot = __builtin_amdgcn_s_memrealtime(); }
DCL_API int dcl_debug_stamp(unsigned long long *slot_dev, dclStream_t stream) {
  DCL_CHECK_ARG(slot_dev);
  hipLaunchKernelGGL(k_debug_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, slot_dev);
  DCL_LAUNCH_CHECK();
  return 0;
}
#endif
void dcl_internal_zero_words(void *p, long long nwords, hipStream_t s) {
  if (nwords <= 0) return;
  hipLaunchKernelGGL(k_zero_words, dim3(dcl_grid_1d((nwords + 3) / 4, 256, 2048)), dim3(256), 0, s, (uint32_t *)p, nwords);
}

// library-internal (hidden visibility): exclusive popcount prefix of a bitmask, wprefix[nwords] = total
int dcl_internal_scan_mask(const uint32_t *mask, int nwords, int32_t *wprefix, int32_t *scratch, hipStream_t s) {
  return scan_mask(mask, nwords, wprefix, scratch, s);
}

// n_rows_dev (optional): the live row count on the device; n_rows then only bounds it (capacity mode, graph capture)
int dcl_internal_grid_from_indices(const int32_t *indices, const int32_t *n_rows_dev, int n_rows, int batch_lo, int batch,
                                   int S, uint32_t *mask, int32_t *wprefix, int32_t *perm, int32_t *scratch,
                                   dclStream_t stream);

DCL_API int dcl_grid_from_indices(const int32_t *indices, int n_rows, int batch, int S, uint32_t *mask,
                                  int32_t *wprefix, int32_t *perm, int32_t *scratch, dclStream_t stream) {
  return dcl_internal_grid_from_indices(indices, nullptr, n_rows, 0, batch, S, mask, wprefix, perm, scratch, stream);
}

int dcl_internal_grid_from_indices(const int32_t *indices, const int32_t *n_rows_dev, int n_rows, int batch_lo, int batch,
                                   int S, uint32_t *mask, int32_t *wprefix, int32_t *perm, int32_t *scratch,
                                   dclStream_t stream) {
  DCL_CHECK_ARG(batch > 0 && S > 0 && n_rows >= 0 && mask && wprefix && scratch);
  DCL_CHECK_ARG(grid_words(batch, S) < (1ll << 26));
  hipStream_t s = (hipStream_t)stream;
  const int nwords = (int)grid_words(batch, S);
  dcl_internal_zero_words(mask, nwords, s);
  if (n_rows > 0) {
    DCL_CHECK_ARG(indices);
    hipLaunchKernelGGL(k_mark_rows, dim3(dcl_grid_1d(n_rows, 256)), dim3(256), 0, s, indices, n_rows_dev, n_rows, S,
                       batch_lo, batch, mask);
  }
  scan_mask(mask, nwords, wprefix, scratch, s);
  if (perm && n_rows > 0)
    hipLaunchKernelGGL(k_fill_perm, dim3(dcl_grid_1d(n_rows, 256)), dim3(256), 0, s, indices, n_rows_dev, n_rows, S,
                       batch_lo, batch, mask, wprefix, perm);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_conv_out_grid(const int32_t *in_indices, const int32_t *n_in_dev, int n_in_host,
                              const uint32_t *in_mask, int batch,
                              int S_in, int ksize, int stride, int padding, uint32_t *out_mask,
                              int32_t *out_wprefix, int32_t *out_indices, int32_t *n_out_dev, int cap_out,
                              int32_t *scratch, dclStream_t stream) {
  DCL_CHECK_ARG(batch > 0 && S_in > 0 && ksize >= 1 && ksize <= 3 && stride >= 1 && padding >= 0);
  DCL_CHECK_ARG(in_indices && out_mask && out_wprefix && out_indices && scratch && cap_out > 0 && n_in_host >= 0);
  const int S_out = (S_in + 2 * padding - (ksize - 1) - 1) / stride + 1;   // spconv/ops.py:19-30
  DCL_CHECK_ARG(S_out > 0 && grid_words(batch, S_out) < (1ll << 26));
  hipStream_t s = (hipStream_t)stream;
  const int nwords = (int)grid_words(batch, S_out);
  const bool pow2 = (S_in & (S_in - 1)) == 0 && S_in >= 4 && S_in <= 64;
  if (in_mask && pow2 && ksize == 3 && padding == 1 && (stride == 1 || stride == 2)) {
    // bit-parallel path: the output mask is a function of the input MASK alone
    if (stride == 1)
      hipLaunchKernelGGL((k_out_mask_k3<1>), dim3(dcl_grid_1d(nwords, 256)), dim3(256), 0, s, in_mask, batch, S_in,
                         S_out, nwords, out_mask);
    else
      hipLaunchKernelGGL((k_out_mask_k3<2>), dim3(dcl_grid_1d(nwords, 256)), dim3(256), 0, s, in_mask, batch, S_in,
                         S_out, nwords, out_mask);
  } else {
    dcl_internal_zero_words(out_mask, nwords, s);
    const int kvol = ksize * ksize * ksize;
    const long long in_work = (long long)n_in_host * kvol;                  // n_in_host bounds *n_in_dev
    hipLaunchKernelGGL(k_mark_conv_outputs, dim3(dcl_grid_1d(in_work > 0 ? in_work : 1, 256)), dim3(256), 0, s,
                       in_indices, n_in_dev, n_in_host, S_out, ksize, stride, padding, out_mask);
  }
  scan_mask(out_mask, nwords, out_wprefix, scratch, s);
  hipLaunchKernelGGL(k_enumerate, dim3(dcl_grid_1d(nwords, 256)), dim3(256), 0, s, out_mask, out_wprefix, nwords,
                     S_out, out_indices, cap_out, n_out_dev);
  DCL_LAUNCH_CHECK();
  return 0;
}

// internal (backbone.hip): bit-parallel output mask of a k3 p1 conv (stride 1) / pool (stride 2) from the input mask
int dcl_internal_out_mask_k3(const uint32_t *in_mask, int batch, int S_in, int stride, uint32_t *out_mask,
                             dclStream_t stream) {
  DCL_CHECK_ARG(in_mask && out_mask && batch > 0 && (S_in & (S_in - 1)) == 0 && S_in >= 4 && S_in <= 64 &&
                (stride == 1 || stride == 2));
  const int S_out = S_in / stride;
  const int nwords = (int)grid_words(batch, S_out);
  if (stride == 1)
    hipLaunchKernelGGL((k_out_mask_k3<1>), dim3(dcl_grid_1d(nwords, 256)), dim3(256), 0, (hipStream_t)stream, in_mask,
                       batch, S_in, S_out, nwords, out_mask);
  else
    hipLaunchKernelGGL((k_out_mask_k3<2>), dim3(dcl_grid_1d(nwords, 256)), dim3(256), 0, (hipStream_t)stream, in_mask,
                       batch, S_in, S_out, nwords, out_mask);
  DCL_LAUNCH_CHECK();
  return 0;
}

// internal (backbone.hip): word prefixes, row counts and (b,x,y,z) rows of `nsets` masks in three launches
// all 8 masks of a pass from the level-0 mask in one launch (64^3 grids only; the caller falls back to the chained
// dcl_internal_out_mask_k3 launches otherwise).  g.mask[i] / g.S[i] must be filled for i = 0..7.
bool dcl_internal_mask_chain_ok(int S) { return S == kChainS; }
int dcl_internal_mask_chain(const uint32_t *mask0, int batch, const DclGeoSets &g, dclStream_t stream) {
  if (batch <= 0) return 0;
  for (int i = 0; i < 8; ++i) DCL_CHECK_ARG(g.mask[i] && g.S[i] == kChainS >> ((i + 1) >> 1));      // 64, 32, 32, 16, 16, 8, 8, 4
  hipLaunchKernelGGL(k_mask_chain64, dim3(batch), dim3(kChainThreads), 0, (hipStream_t)stream, mask0, g);
  DCL_LAUNCH_CHECK();
  return 0;
}

// the geometry stage of a pass of a handful of crops in one launch (+ the zeroing of its exchange words); see
// k_geometry_small.  comm: kGeoSmallMax * 16 ints of scratch.
bool dcl_internal_geometry_small_ok(int batch, int S, int rows) {
  return S == kChainS && batch >= 1 && batch <= kGeoSmallBatch && batch <= kGeoSmallMax && (batch <= 8 || rows <= kGeoSmallRows);
}
#ifdef DCL_DIAG
DCL_API void dcl_debug_geometry_small_batch(int n) { kGeoSmallBatch = n; }
DCL_API int dcl_debug_geometry_small_stamps(unsigned long long *host32) {
  return hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_geo_stamps), sizeof(g_geo_stamps)) == hipSuccess ? 0 : DCL_EINVAL;
}
#endif
int dcl_internal_geometry_small(const int32_t *occ, const int32_t *n_dev, int n_host, int batch_lo, int batch, uint32_t *mask0,
                                int32_t *wprefix0, int32_t *perm0, int32_t *comm, const DclGeoSets &g, const DclVoxelizeRider *vx,
                                dclStream_t stream) {
  DCL_CHECK_ARG(dcl_internal_geometry_small_ok(batch, kChainS, n_host) && mask0 && wprefix0 && comm && (n_host == 0 || occ));
  for (int i = 0; i < 8; ++i) DCL_CHECK_ARG(g.mask[i] && g.wprefix[i] && g.indices[i] && g.S[i] == kChainS >> ((i + 1) >> 1));      // 64, 32, 32, 16, 16, 8, 8, 4
  hipStream_t s = (hipStream_t)stream;
  if (batch > 1) dcl_internal_zero_words(comm, (long long)kGeoSmallMax * kGeoCommStride, s);      // (one crop exchanges nothing)
  GeoSmallArgs a{occ, n_dev, n_host, batch_lo, batch, mask0, wprefix0, perm0, comm, nullptr, nullptr, nullptr, 0, 0, 1, 0};
  int riders = 0;
  if (vx && vx->rows > 0) {
    a.vx_feats = vx->feats; a.vx_rules = vx->rules; a.vx_out = vx->out;
    a.vx_rows = vx->rows; a.vx_ma = vx->max_active; a.vx_planes = vx->planes; a.vx_avg = vx->average;
    const long long total = (long long)vx->rows * vx->planes;
    riders = (int)((total + kChainThreads - 1) / kChainThreads);
    if (riders > 240) riders = 240;                    // (a workgroup of this kernel takes a CU to itself)
  }
  hipLaunchKernelGGL(k_geometry_small, dim3(batch + riders), dim3(kChainThreads), 0, s, a, g);
  DCL_LAUNCH_CHECK();
  return 0;
}

int dcl_internal_scan_enumerate_sets(const DclGeoSets &g, int nsets, dclStream_t stream) {
  DCL_CHECK_ARG(nsets >= 1 && nsets <= 8);
  int max_words = 0;
  for (int i = 0; i < nsets; ++i) {
    DCL_CHECK_ARG(g.mask[i] && g.wprefix[i] && g.indices[i] && g.block_sums[i] && g.nwords[i] > 0 && g.S[i] > 0);
    if (g.nwords[i] > max_words) max_words = g.nwords[i];
  }
  hipStream_t s = (hipStream_t)stream;
  const int nblk = dcl_div_up(max_words, kScanWords);
  hipLaunchKernelGGL(k_block_popc_sets, dim3(nblk, nsets), dim3(256), 0, s, g);
  hipLaunchKernelGGL(k_scan_words_sets, dim3(nblk, nsets), dim3(256), 0, s, g);
  hipLaunchKernelGGL(k_enumerate_sets, dim3(dcl_grid_1d(max_words, 256), nsets), dim3(256), 0, s, g);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_rulebook_gather(const int32_t *out_indices, const int32_t *n_out_dev, int n_out_host,
                                const uint32_t *in_mask, const int32_t *in_wprefix, const int32_t *in_perm,
                                int batch, int S_in, int ksize, int stride, int padding, int32_t *nbr, int cap,
                                dclStream_t stream) {
  DCL_CHECK_ARG(batch > 0 && S_in > 0 && ksize >= 1 && ksize <= 3 && stride >= 1 && padding >= 0 && cap > 0);
  DCL_CHECK_ARG(out_indices && in_mask && in_wprefix && nbr && n_out_host >= 0 && n_out_host <= cap);
  const int rows = n_out_dev ? cap : n_out_host;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(k_build_nbr, dim3(dcl_grid_1d(rows, 256)), dim3(256), 0, (hipStream_t)stream, out_indices,
                     n_out_dev, n_out_host, in_mask, in_wprefix, in_perm, S_in, ksize, stride, padding, nbr, cap);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_rulebook_conv(const int32_t *in_indices, const int32_t *n_in_dev, int n_in_host,
                              const uint32_t *in_mask, const int32_t *in_wprefix, const int32_t *in_perm,
                              int batch, int S_in, int ksize, int stride, int padding,
                              uint32_t *out_mask, int32_t *out_wprefix, int32_t *out_indices,
                              int32_t *n_out_dev, int32_t *nbr, int cap_out, int32_t *scratch,
                              dclStream_t stream) {
  DCL_CHECK_ARG(in_mask && in_wprefix && nbr);
  int rc = dcl_conv_out_grid(in_indices, n_in_dev, n_in_host, in_mask, batch, S_in, ksize, stride, padding, out_mask,
                             out_wprefix, out_indices, n_out_dev, cap_out, scratch, stream);
  if (rc) return rc;
  const int S_out = (S_in + 2 * padding - (ksize - 1) - 1) / stride + 1;
  const int nwords = (int)grid_words(batch, S_out);
  return dcl_rulebook_gather(out_indices, out_wprefix + nwords, 0, in_mask, in_wprefix, in_perm, batch, S_in, ksize,
                             stride, padding, nbr, cap_out, stream);
}

DCL_API int dcl_rulebook_subm(const int32_t *indices, const int32_t *n_dev, int n_host,
                              const uint32_t *mask, const int32_t *wprefix, const int32_t *perm,
                              int batch, int S, int ksize, int32_t *nbr, int cap, dclStream_t stream) {
  DCL_CHECK_ARG(ksize == 1 || ksize == 3);
  return dcl_rulebook_gather(indices, n_dev, n_host, mask, wprefix, perm, batch, S, ksize, 1, ksize / 2, nbr, cap,
                             stream);                                       // spconv_ops.h:76-79
}

DCL_API int dcl_rulebook_to_pairs(const int32_t *nbr, int cap, const int32_t *n_out_dev, int n_out_host,
                                  int kvol, int32_t *indice_pairs, int n_in_cap, int32_t *indice_num,
                                  dclStream_t stream) {
  DCL_CHECK_ARG(nbr && indice_pairs && indice_num && cap > 0 && kvol > 0 && kvol <= 27 && n_in_cap >= 0);
  hipStream_t s = (hipStream_t)stream;
  const long long np = (long long)kvol * 2 * n_in_cap;
  hipLaunchKernelGGL(k_pairs_init, dim3(dcl_grid_1d(np > 0 ? np : 1, 256)), dim3(256), 0, s, indice_pairs, np,
                     indice_num, kvol);
  hipLaunchKernelGGL(k_pairs_export, dim3(dcl_grid_1d((long long)cap * kvol, 256)), dim3(256), 0, s, nbr, cap,
                     n_out_dev, n_out_host, kvol, indice_pairs, n_in_cap, indice_num);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_rulebook_from_pairs(const int32_t *indice_pairs, int pair_stride, const int32_t *indice_num_dev, int kvol,
                                    int n_in, int n_out, int32_t *nbr, int cap, int32_t *bad_pairs_dev,
                                    dclStream_t stream) {
  DCL_CHECK_ARG(indice_pairs && indice_num_dev && nbr && kvol > 0 && kvol <= 27 && pair_stride >= 0 && n_in >= 0 &&
                n_out >= 0 && cap > 0 && n_out <= cap);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fill_i32, dim3(dcl_grid_1d((long long)kvol * cap, 256)), dim3(256), 0, s, nbr, (long long)kvol * cap,
                     -1, bad_pairs_dev);
  if (pair_stride > 0 && n_out > 0)
    hipLaunchKernelGGL(k_pairs_import, dim3(dcl_grid_1d((long long)kvol * pair_stride, 256)), dim3(256), 0, s, indice_pairs,
                       pair_stride, indice_num_dev, kvol, n_in, n_out, nbr, cap, bad_pairs_dev);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_indice_summary_rf(const int32_t *indice_pairs, int pair_stride, const int32_t *indice_num_dev, int kvol,
                                  int n_out, int32_t *rf, dclStream_t stream) {
  DCL_CHECK_ARG(indice_pairs && indice_num_dev && kvol > 0 && kvol <= 27 && pair_stride >= 0 && n_out >= 0 &&
                (rf || n_out == 0));
  if (n_out == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_fill_i32, dim3(dcl_grid_1d(n_out, 256)), dim3(256), 0, s, rf, (long long)n_out, 0, nullptr);
  if (pair_stride > 0)
    hipLaunchKernelGGL(k_pairs_summary_rf, dim3(dcl_grid_1d((long long)kvol * pair_stride, 256)), dim3(256), 0, s,
                       indice_pairs, pair_stride, indice_num_dev, kvol, n_out, rf);
  DCL_LAUNCH_CHECK();
  return 0;
}

// sparse_conv.hip -- sparse 3-D convolution and sparse average pooling over gather-form rulebooks.
//
// Replaces indiceConv<float> (libs/spconv/include/spconv/spconv_ops.h:253-349: per kernel offset a
// gather kernel, a cuBLAS SGEMM and a scatter-add kernel, i.e. ~81 launches + one device->host
// sync per layer) and indiceSummaryRF + indiceAvgPool (pool_ops.h:141-208; summaryRF.cu:26-41;
// avgpool.cu:96-176: 54 launches + 2 syncs per pool) by ONE launch each.
//
// Output-stationary: a wavefront owns 32 consecutive output voxels and all Cout channels, walks the
// kernel offsets in the reference's order (k ascending; the centre offset first for submanifold
// conv, spconv_ops.h:289-299) and accumulates gathered-row x W[k] products in fp32 MFMA
// accumulators (v_mfma_f32_32x32x2_f32, exact fp32: an fmaf chain), so no scatter, no atomics and
// a fixed summation order.  Offsets none of the wave's 32 rows use are skipped (wave-uniform).
// The BatchNorm1d(eval)+ReLU that follows every conv in the backbone (models/Modules.py:36-40) is
// the epilogue.  Bound: MFMA fp32 (2*pairs*Cin*Cout flop); features and weights are L2-resident.
#include "common.h"
#include "conv_body.h"
#include <atomic>
#include <mutex>
#include <utility>
#include <array>
#include <vector>

int dcl_internal_sparse_conv_fwd(const float *feat, const DclNbrSrc &nbr, int cap, const int32_t *n_out_dev,
                                 int n_out_host, const float *W, int cin, int cout, int kvol, int subm, const float *scale,
                                 const float *shift, int relu, float *out, float *scratch, int64_t scratch_floats,
                                 dclStream_t stream, int counters_ready = 0, const DclRowOrder *ord = nullptr);
int dcl_internal_sparse_avgpool_fwd(const float *feat, const DclNbrSrc &nbr, int cap, const int32_t *n_out_dev,
                                    int n_out_host, int c, int kvol, float *out, int32_t *rf, dclStream_t stream);
int dcl_internal_sparse_conv_fwd_sides(const DclConvSides &sides, int nsides, int cin, int cout, int kvol, int subm, int relu,
                                       float *scratch, int64_t scratch_floats, dclStream_t stream, int counters_ready = 0,
                                       int *counters_state = nullptr);
int dcl_internal_sparse_avgpool_fwd_sides(const DclConvSides &sides, int nsides, int c, int kvol, int32_t *rf,
                                          const int32_t *rf_in, dclStream_t stream);

namespace {


// ---- generic VALU kernel: any Cin/Cout (used for the 7->16 stem and as an A/B check) -------------
__global__ void k_sparse_conv_valu(const float *__restrict__ feat, const DclNbrSrc src, int cap,
                                   const int32_t *__restrict__ n_out_dev, int n_out_host,
                                   const float *__restrict__ W, int cin, int cout, int kvol, int subm,
                                   const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                                   float *__restrict__ out) {
  int n = n_out_dev ? *n_out_dev : n_out_host;
  n = n < cap ? n : cap;
  const long long total = (long long)n * cout;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(t / cout);
    const int co = (int)(t - (long long)row * cout);
    float acc = 0.0f;
    for (int s = 0; s < kvol; ++s) {
      const int k = offset_at(s, kvol, subm);
      const int v = dcl_nbr_at(src, cap, k, row);
      if (v < 0) continue;
      const float *f = feat + (size_t)v * cin;
      const float *w = W + (size_t)k * cin * cout + co;
      float part = 0.0f;
      for (int ci = 0; ci < cin; ++ci) part = __fmaf_rn(f[ci], w[(size_t)ci * cout], part);
      acc = acc + part;                       // per-offset GEMM result added to out (spconv_ops.h:326-344)
    }
    if (scale) acc = acc * scale[co] + shift[co];
    if (relu) acc = fmaxf(acc, 0.0f);
    out[t] = acc;
  }
}

// ---- stem kernel (conv_body.h: conv_stem_body) ------------------------------------------------------------------
template <int CIN, int COUT, int LPR>
__global__ __launch_bounds__(256) void k_sparse_conv_stem(const DclConvSides sides, int nsides, int kvol, int subm, int relu) {
  __shared__ __attribute__((aligned(16))) float Ws[2 * 27 * (CIN * COUT + 1)];   // both sides' filters, pitched per offset
  __shared__ int32_t s_nb[256 * 27];                                             // every lane's 27 neighbour rows
  conv_stem_body<CIN, COUT, 256, LPR>(sides, nsides, kvol, subm, relu, Ws, s_nb, blockIdx.x, gridDim.x);
}


// ---- MFMA kernel: Cin % 8 == 0, Cout % (32*NT) == 0 ---------------------------------------------
// wave = 32 output rows x (32*NT) output channels; grid.y walks the channel tiles so that small
// (deep) layers still put >= 2-3 waves on every SIMD.
template <int NT>
__global__ __launch_bounds__(256) void k_sparse_conv_mfma(
    const float *__restrict__ feat, const DclNbrSrc src, int cap, const int32_t *__restrict__ n_out_dev,
    int n_out_host, const float *__restrict__ W, int cin, int cout, int kvol, int subm,
    const float *__restrict__ scale, const float *__restrict__ shift, int relu, float *__restrict__ out) {
  int n = n_out_dev ? *n_out_dev : n_out_host;
  n = n < cap ? n : cap;
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  const int waves_per_block = blockDim.x >> 6;
  const int ntiles = (n + 31) >> 5;
  const int col0 = blockIdx.y * (32 * NT);
  for (int tile = blockIdx.x * waves_per_block + (threadIdx.x >> 6); tile < ntiles;
       tile += gridDim.x * waves_per_block) {
    const int row = tile * 32 + r;
    const bool valid = row < n;
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    for (int s = 0; s < kvol; ++s) {
      const int k = offset_at(s, kvol, subm);
      const int v = valid ? dcl_nbr_at(src, cap, k, row) : -1;
      if (__ballot(v >= 0) == 0ull) continue;                       // nobody in this tile uses offset k
      const float *fp = feat + (size_t)(v >= 0 ? v : 0) * cin + h * 4;
      const float *wp = W + (size_t)k * cin * cout + col0 + r;
      for (int c8 = 0; c8 < cin; c8 += 8) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v >= 0) a = *reinterpret_cast<const float4 *>(fp + c8);
        const float av[4] = {a.x, a.y, a.z, a.w};
        // MFMA step t contracts channels {c8+t, c8+4+t}: lane half h supplies channel c8+4h+t.
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float *wrow = wp + (size_t)(c8 + h * 4 + t) * cout;
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], wrow[j * 32], acc[j], 0, 0, 0);
        }
      }
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // MFMA -> VALU read of the accumulator (hipcc 7.2 omitted the wait states when the accumulator is re-read behind a barrier: stale acc[15])
    // C/D layout: col = lane&31 (cout within tile), row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = col0 + j * 32 + r;
      const float sc = scale ? scale[co] : 1.0f;
      const float sh = scale ? shift[co] : 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int orow = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (orow < n) {
          float x = acc[j][e];
          if (scale) x = x * sc + sh;
          if (relu) x = fmaxf(x, 0.0f);
          out[(size_t)orow * cout + co] = x;
        }
      }
    }
  }
}

// ---- MFMA kernel with the WHOLE filter resident in LDS (conv_body.h: conv_wlds_body) --------------------------------
template <int CIN, int COUT_T, bool SUBM>
__global__ __launch_bounds__(CIN == 16 ? 1024 : 512) void k_sparse_conv_wlds(const DclConvSides sides, int nsides, int relu) {
  extern __shared__ __attribute__((aligned(16))) float wl_lds[];             // [27][CIN][32]
  conv_wlds_body<CIN, COUT_T, SUBM, (CIN == 16 ? 1024 : 512)>(sides, nsides, relu, wl_lds, blockIdx.x, gridDim.x, blockIdx.y);
}


#ifdef DCL_DIAG   // superseded by k_sparse_conv_dma for every shape it takes: kept in the diagnostic library as an A/B reference
// ---- implicit-GEMM MFMA kernel: gathered A tile AND weight tile through LDS ------------------------------------
// The 27 offsets x Cin input channels form one long contraction axis of "virtual channels"; it is walked in chunks of
// KC virtual channels (= G = KC/Cin whole offsets, visited in the reference's order).  Per chunk the workgroup stages
//   As[BM rows][KC]  gathered input rows -- each row segment is read by Cin/4 consecutive lanes (full 16-B-per-lane
//                    coalescing; the per-lane row gather of k_sparse_conv_lds used 32 B of every 128-B line it pulled)
//   Bs[KC][BN]       the matching W rows
// through registers (global loads for chunk j+1 are in flight during chunk j's MFMAs), then every wave feeds its 32x32
// output tile from LDS (A: ds_read_b128, B: ds_read_b32, both conflict-free).  Chunks whose offsets no row of the
// workgroup uses are skipped.  WC = waves along the channel axis: BM = 32*(4/WC) rows, BN = 32*WC channels.
template <int CIN, int WC, int KC>
__global__ __launch_bounds__(256, 2) void k_sparse_conv_tile(
    const float *__restrict__ feat, const DclNbrSrc src, int cap, const int32_t *__restrict__ n_out_dev,
    int n_out_host, const float *__restrict__ W, int cout, int kvol, int subm, const float *__restrict__ scale,
    const float *__restrict__ shift, int relu, float *__restrict__ out, float *__restrict__ partial, int nsplit) {
  constexpr int WR = 4 / WC;
  constexpr int BM = 32 * WR, BN = 32 * WC;
  constexpr int G = KC / CIN;                      // whole offsets per chunk
  constexpr int AP = KC + 4;                       // A row pitch (floats): conflict-free ds_read_b128 across rows
  constexpr int NA = BM * KC / 4 / 256;            // float4 of A per thread per chunk
  constexpr int NB = KC * BN / 4 / 256;            // float4 of B per thread per chunk
  static_assert(KC % CIN == 0 && NA >= 1 && NB >= 1, "tile shape");
  extern __shared__ __attribute__((aligned(16))) float conv_lds[];
  float *As = conv_lds;                            // [BM][AP]
  float *Bs = conv_lds + BM * AP;                  // [KC][BN]
  int32_t *Ns = reinterpret_cast<int32_t *>(Bs + KC * BN);   // [27][BM] neighbour rows of this row block
  __shared__ unsigned s_kmask;

  int n = n_out_dev ? *n_out_dev : n_out_host;
  n = n < cap ? n : cap;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;
  const int col0 = blockIdx.y * BN;
  const int nblk = (n + BM - 1) / BM;
  // split-K (gridDim.z = nsplit > 1, launches with few row tiles): this workgroup owns the chunks [j_lo, nchunks) of the
  // walk and leaves raw partial sums for k_conv_split_reduce
  const int nchunks_all = (kvol + G - 1) / G;
  const int j_lo = (int)((long long)blockIdx.z * nchunks_all / nsplit);
  const int nchunks = (int)((long long)(blockIdx.z + 1) * nchunks_all / nsplit);

  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int row0 = blk * BM;
    if (tid == 0) s_kmask = 0;
    __syncthreads();
    // neighbour table of the row block + which offsets it uses at all
    unsigned mymask = 0;
    const int s_lo = j_lo * G, s_hi = min(kvol, nchunks * G);          // this split's offsets (walk order)
    for (int e = tid; e < (s_hi - s_lo) * BM; e += 256) {
      const int si = e / BM, rr = e - si * BM;
      const int k = offset_at(s_lo + si, kvol, subm);
      const int v = (row0 + rr < n) ? dcl_nbr_at(src, cap, k, row0 + rr) : -1;
      Ns[k * BM + rr] = v;
      mymask |= (v >= 0 ? 1u : 0u) << k;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mymask |= __shfl_xor(mymask, d, 64);
    if (lane == 0 && mymask) atomicOr(&s_kmask, mymask);
    __syncthreads();
    const unsigned kmask = s_kmask;
    auto chunk_used = [&](int j) {
      unsigned m = 0;
      for (int g = 0; g < G; ++g) {
        const int s = j * G + g;
        if (s < kvol) m |= (kmask >> offset_at(s, kvol, subm)) & 1u;
      }
      return m != 0;
    };
    auto next_used = [&](int from) { int q = from; while (q < nchunks && !chunk_used(q)) ++q; return q; };

    float4 areg[NA], breg[NB];
    auto fetch = [&](int j) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int f = tid + i * 256;
        const int rr = f / (KC / 4), vc = (f - rr * (KC / 4)) * 4;
        const int g = vc / CIN, ch = vc - g * CIN;
        const int s = j * G + g;
        int v = -1;
        if (s < kvol) v = Ns[offset_at(s, kvol, subm) * BM + rr];
        areg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v >= 0) areg[i] = *reinterpret_cast<const float4 *>(feat + (size_t)v * CIN + ch);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int f = tid + i * 256;
        const int vc = f / (BN / 4), c4 = (f - vc * (BN / 4)) * 4;
        const int g = vc / CIN, ch = vc - g * CIN;
        const int s = j * G + g;
        breg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < kvol)
          breg[i] = *reinterpret_cast<const float4 *>(W + ((size_t)offset_at(s, kvol, subm) * CIN + ch) * cout + col0 + c4);
      }
    };
    auto stash = [&]() {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int f = tid + i * 256;
        const int rr = f / (KC / 4), vc = (f - rr * (KC / 4)) * 4;
        *reinterpret_cast<float4 *>(As + rr * AP + vc) = areg[i];
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) *reinterpret_cast<float4 *>(Bs + (tid + i * 256) * 4) = breg[i];
    };

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    int j = next_used(j_lo);
    if (j < nchunks) fetch(j);
    while (j < nchunks) {
      __syncthreads();                                   // everyone is done reading the previous chunk's tiles
      stash();
      __syncthreads();
      const int jn = next_used(j + 1);
      if (jn < nchunks) fetch(jn);                       // in flight during the MFMAs below
      // wave-level skip: none of this wave's 32 rows has a neighbour under any offset of the chunk
      bool mine = false;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int sg = j * G + g;
        if (sg < kvol) mine |= Ns[offset_at(sg, kvol, subm) * BM + wr * 32 + r] >= 0;
      }
      const float *ap = As + (wr * 32 + r) * AP + 4 * h;
      const float *bp = Bs + (4 * h) * BN + wc * 32 + r;
      if (__ballot(mine) != 0ull) {
        // operands of step i+1 are read from LDS while step i's four MFMAs run (two register sets, order pinned with
        // sched_group_barrier: 5 DS reads, then 4 MFMAs) -- otherwise every group of MFMAs starts with the LDS latency
        float4 a_cur = *reinterpret_cast<const float4 *>(ap);
        float b_cur[4] = {bp[0], bp[BN], bp[2 * BN], bp[3 * BN]};
#pragma unroll
        for (int i = 0; i < KC / 8; ++i) {
          float4 a_nxt = a_cur;
          float b_nxt[4] = {b_cur[0], b_cur[1], b_cur[2], b_cur[3]};
          if (i + 1 < KC / 8) {
            a_nxt = *reinterpret_cast<const float4 *>(ap + 8 * (i + 1));
#pragma unroll
            for (int t = 0; t < 4; ++t) b_nxt[t] = bp[(8 * (i + 1) + t) * BN];
            __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur[0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur[1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur[2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur[3], acc, 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          a_cur = a_nxt;
#pragma unroll
          for (int t = 0; t < 4; ++t) b_cur[t] = b_nxt[t];
        }
      }
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // MFMA -> VALU hazard pad (hipcc 7.2 omitted the wait states when the accumulator is re-read behind a barrier: stale acc[15])
      j = jn;
    }
    const int co = col0 + wc * 32 + r;
    const float sc = scale ? scale[co] : 1.0f;
    const float sh = scale ? shift[co] : 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int orow = row0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (orow < n) {
        float x = acc[e];
        if (nsplit > 1) {
          partial[((size_t)blockIdx.z * cap + orow) * cout + co] = x;
        } else {
          if (scale) x = x * sc + sh;
          if (relu) x = fmaxf(x, 0.0f);
          out[(size_t)orow * cout + co] = x;
        }
      }
    }
    __syncthreads();
  }
}

#endif  // DCL_DIAG

// ---- implicit-GEMM MFMA kernel fed by LDS-DMA (conv_body.h: conv_dma_body) ------------------------------------------
template <int CIN, int WR, int WCW, int NT, bool ORD>
__global__ __launch_bounds__(64 * WR * WCW, WR * WCW / 2) void k_sparse_conv_dma(   // 2 workgroups per CU (LDS allows 2)
    const DclConvSides sides, int nsides, int cout, int kvol, int subm, int relu, float *__restrict__ partial, int stream_k,
    int aligned_ns, int xcd_remap, int32_t *__restrict__ tile_counters, int use_bal_arg) {
  extern __shared__ __attribute__((aligned(16))) float conv_lds[];   // [stage 0: A|B][stage 1: A|B][Ns 27*BM][kmask (4)][rows BM]
  conv_dma_body<CIN, WR, WCW, NT, ORD>(sides, nsides, cout, kvol, subm, relu, partial, stream_k, aligned_ns, xcd_remap, tile_counters,
                                       use_bal_arg, conv_lds, blockIdx.x, gridDim.x);
}


#ifdef DCL_DIAG
// out = act(scale * (P_0 + P_1 + ... ) + shift), partial sums added in split order; thread = 4 channels of a row
__global__ void k_conv_split_reduce(const float *__restrict__ partial, int nsplit, int cap, const int32_t *__restrict__ n_out_dev,
                                    int n_out_host, int cout, const float *__restrict__ scale,
                                    const float *__restrict__ shift, int relu, float *__restrict__ out) {
  int n = n_out_dev ? *n_out_dev : n_out_host;
  n = n < cap ? n : cap;
  const int c4 = cout >> 2;
  const long long total = (long long)n * c4;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(t % c4);
    float4 a = reinterpret_cast<const float4 *>(partial)[t];
    for (int z = 1; z < nsplit; ++z) {
      const float4 b = reinterpret_cast<const float4 *>(partial + (size_t)z * cap * cout)[t];
      a.x = a.x + b.x; a.y = a.y + b.y; a.z = a.z + b.z; a.w = a.w + b.w;
    }
    if (scale) {
      const float4 sc = reinterpret_cast<const float4 *>(scale)[q], sh = reinterpret_cast<const float4 *>(shift)[q];
      a.x = a.x * sc.x + sh.x; a.y = a.y * sc.y + sh.y; a.z = a.z * sc.z + sh.z; a.w = a.w * sc.w + sh.w;
    }
    if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    reinterpret_cast<float4 *>(out)[t] = a;
  }
}

#endif  // DCL_DIAG

// Deferred combine of a stream-K launch (conv_body.h: conv_frag_reduce_body).  grid = (tiles_cap, NW*NT*4*64/256), 256 threads.
template <int WR, int WCW, int NT>
__global__ __launch_bounds__(256) void k_conv_frag_reduce(const float *__restrict__ partial, const DclConvSides sides, int nsides,
                                                          int cout, int C, int G, int min_u, int relu) {
  conv_frag_reduce_body<WR, WCW, NT>(partial, sides, nsides, cout, C, G, min_u, relu, blockIdx.x, blockIdx.y * 256);
}


constexpr int kConvMaxSplit = 8;       // K-splits of a launch with many row tiles
constexpr int kConvFewRows = 4096;     // at most this many output rows (capacity): up to one split per kernel offset
static int conv_split_cap(long long rows) { return rows <= kConvFewRows ? 27 : kConvMaxSplit; }
// launches that are latency-bound on the contraction walk: few output rows -- or, in capacity mode (whole-forward
// hipGraph, a handful of crops), a row capacity of at most two crops' worth of cells, whatever the live count turns out to be
DCL_HOOK_INT(kConvFewRowsCap, 65536);
static bool conv_few_rows(int rows, bool capacity_mode) { return capacity_mode ? rows <= kConvFewRowsCap : rows <= kConvFewRows; }
// ... unless the caller says how many rows it EXPECTS (DclConvSide::n_host beside n_dev: the backbone runner knows the batch
// size and the level): a capacity says little -- level 4 of 32 crops has a capacity of 16384 rows and fills 3/4 of it, level 3
// of 6 crops has one of 24576 and fills a fifth
// Up to 24576 expected rows a launch is better off as a few-row launch (short runs of chunks that never straddle two tiles,
// combine as a launch of its own): whole forward, graph replay, 2 / 4 / 6 / 8 / 12 crops 0.616 -> 0.583 / 0.870 -> 0.790 /
// 1.118 -> 1.060 / 1.325 -> 1.27 / 1.79 -> 1.76 ms against the former 6144 (tools/ab_hook.py dcl_debug_conv_few_hint).  A
// launch that carries a row order keeps the 6144: the few-row form takes no order, and at 16 / 32 crops the deep levels lose
// more by that (2.27 -> 2.31, 4.03 -> 4.21 ms) than the short runs win.
DCL_HOOK_INT(kConvFewRowsHint, 24576);
constexpr int kConvFewRowsOrdered = 6144;
static bool conv_launch_is_few(const DclConvSides &sides, int nsides) {
  const bool capacity_mode = sides.s[0].n_dev != nullptr;
  int rows = 0;
  bool hinted = capacity_mode, ordered = false;
  for (int i = 0; i < nsides; ++i) {
    const DclConvSide &S = sides.s[i];
    const int r_i = S.n_dev ? (S.n_host > 0 ? S.n_host : S.cap) : S.n_host;
    hinted = hinted && S.n_host > 0;
    ordered = ordered || S.ord.order != nullptr;
    rows = r_i > rows ? r_i : rows;
  }
  const int limit = ordered && kConvFewRowsHint > kConvFewRowsOrdered ? kConvFewRowsOrdered : (int)kConvFewRowsHint;
  if (hinted || !capacity_mode) return rows <= limit;         // expected (capacity mode with a hint) or exact row counts
  return conv_few_rows(rows, capacity_mode);
}
// A/B and tuning switches: process-wide atomics set through dcl_debug_* in the DIAGNOSTIC library (-DDCL_DIAG, tests/_diag/),
// compile-time constants in the product library -- the product has no hooks, no superseded kernels and no getenv
DCL_HOOK_INT(g_conv_xcd_remap, 1);   // 0 = plain blockIdx order
#ifdef DCL_CONV_STAMPS
static std::atomic<int> g_stamp_select{-1}, g_stamp_count{0};
#endif
DCL_HOOK_INT(g_conv_slots, 512);     // workgroups a launch is dealt over (2 x 256 resident slots)
DCL_HOOK_INT(g_conv_wlds, 1);        // 1 = Cin 16 / 32 -> 32 layers with many rows on the filter-resident kernel, 0 = LDS-DMA kernel
DCL_HOOK_INT(g_conv_few_tiles, 1);   // 1 = few-row launches on 64-row tiles, 0 = 128-row tiles for every launch
DCL_HOOK_INT(g_conv_few_chunks, 4);  // chunks per workgroup (at least) of a few-row launch
DCL_HOOK_INT(g_conv_split, 0);       // 0 = automatic, n = force n-way split-K when scratch allows, -1 = never more than kConvMaxSplit, -2 = never split, -3 = few-row combine inside the launch
#ifdef DCL_DIAG
static std::atomic<int> g_conv_order_mode{0};   // A/B: 0 = as given, 1 = ignore the row order (natural rows), 2 = order but nominal chunk units
#endif

// How one LDS-DMA conv launch is decomposed.
static DclConvPlan plan_conv_dma(const DclConvSides &sides, int nsides, int CIN, int BM, int BN, int cout, int kvol, bool have_scratch,
                                 long long scratch_floats, int kSlots) {
  constexpr int KC = 32;
  DclConvPlan P{};
  const bool capacity_mode = sides.s[0].n_dev != nullptr;
  int tiles = 0;
  for (int i = 0; i < nsides; ++i) {
    const int r_i = sides.s[i].n_dev ? sides.s[i].cap : sides.s[i].n_host;
    tiles += dcl_div_up(r_i, BM) * (cout / BN);
  }
  // With scratch (partial-tile slots + tickets) a launch can be decomposed three ways; the cheapest by a small cost model
  // (chunk units on the critical path of a CU slot, fixed cost f per segment: neighbour table, first operand fetch,
  // publish) is taken:
  //   whole tiles      rounds(tiles) * (C + f)                    -- enough tiles, or just under a multiple of the slots
  //   aligned split-K  rounds(tiles * ns) * (C / ns + f + 1)      -- tiles * ns just fills the 2 x 256 resident slots
  //   stream-K         ceil(units / 512) + 2 f + 1                -- everything else (no partial rounds, no idle slots)
  // A handful of crops (one-image calls) is latency-bound on the chunk loop: stream-K with kFewChunks chunks per workgroup.
  const int nchunks = dcl_div_up(kvol * CIN, KC);
  constexpr int kFix = 4;
  // chunks per workgroup of a few-row launch: about four -- and a divisor of the tile's chunk count (27 -> 3, 54 -> 3, 14 -> 2,
  // 108 -> 4) where there is one, so that no workgroup's run of chunks straddles two tiles: a straddling workgroup runs TWO
  // segments -- two neighbour tables, two first fetches, two publishes, 16 us instead of 8 at one crop -- and the launch
  // lasts as long as its longest workgroup
  int kFewChunks = g_conv_few_chunks;
  if (kFewChunks == 4) {
    int best = 0;
    for (int d = 2; d <= 6; ++d)
      if (nchunks % d == 0 && (best == 0 || (d > 4 ? d - 4 : 4 - d) < (best > 4 ? best - 4 : 4 - best))) best = d;
    if (best) kFewChunks = best;
  }
  const long long units = (long long)tiles * nchunks;
  int stream_k = 0, aligned_ns = 0, G = tiles < 65535 * 16 ? tiles : 65535 * 16;
  bool deferred = false, counters = false, split = false;
  const bool never = g_conv_split == -2;
  if (have_scratch && scratch_floats > kConvCounterWords && tiles <= kConvCounterWords && !never) {
    const long long slots_fit = (scratch_floats - kConvCounterWords) / ((long long)2 * BM * BN);    // 2 slots per workgroup
    const bool few = conv_launch_is_few(sides, nsides);
    long long g_stream = units / kFewChunks < kSlots ? units / kFewChunks : kSlots;
    if (g_stream > slots_fit) g_stream = slots_fit;
    long long best = dcl_div_up(tiles, kSlots) * (long long)(nchunks + kFix) * 8;                   // whole tiles
    int mode = 0, best_ns = 1;
    if (g_conv_split > 0) {                                        // test / tuning hook: force an aligned split
      mode = 1;
      best_ns = (int)g_conv_split < nchunks ? (int)g_conv_split : nchunks;
    } else if (few || capacity_mode) {
      mode = 2;                                                    // live sizes unknown or tiny: even shares, >= kFewChunks
    } else {
      for (int ns = 2; ns <= kConvMaxSplit && ns * 8 <= nchunks; ++ns) {
        const long long c = dcl_div_up((long long)tiles * ns, kSlots) * (long long)(dcl_div_up(nchunks, ns) + kFix + 1) * 8 + ns;
        if ((long long)tiles * ns <= slots_fit && c < best) { best = c; mode = 1; best_ns = ns; }
      }
      if (g_stream >= 1) {
        const long long c = (dcl_div_up(units, g_stream) + 2 * kFix + 1) * 8 + 4;
        if (c < best) { best = c; mode = 2; }
      }
    }
    if (mode == 1 && best_ns >= 2 && (long long)tiles * best_ns <= slots_fit) {
      aligned_ns = best_ns;
      G = tiles * best_ns;
    } else if (mode == 2 && g_stream >= 1) {
      stream_k = few ? kFewChunks : 1;
      G = (int)g_stream;
    }
    if (aligned_ns || stream_k) {
      split = true;
      deferred = few && stream_k == kFewChunks && g_conv_split != -3;     // few rows: many segments per tile, combine = own launch (-3: A/B, in the launch)
      counters = !deferred;
    }
  }
  // Row order (row_order.hip).  The order itself works with every decomposition whose combine runs inside the launch (the
  // deferred combine of few-row launches maps tile slots to rows on its own: such launches take no order).  Used-chunk
  // dealing replaces the nominal units whenever the launch would split tiles anyway (aligned split-K / stream-K);
  // launches whose tiles fit one round of whole tiles keep them (measured: forced stream-K only adds the combine there).
  bool have_order = true, have_bal = true;
  for (int i = 0; i < nsides; ++i) {
    have_order = have_order && sides.s[i].ord.order != nullptr;
    have_bal = have_bal && sides.s[i].ord.bal != nullptr && sides.s[i].ord.smask != nullptr;
  }
  int use_bal = 0;
  bool keep_order = false;
  if (have_order && !deferred && BM == 128) {
    keep_order = true;
    // (round 5: also launches whose tiles would fit ONE round of whole tiles.  An ordered tile's cost is its USED chunks --
    // 13 to 54 of 54 on the 64 -> 64 layer of 32 crops -- and one round of whole tiles lasted as long as its most expensive
    // tile: workgroup life 76 us on average, 150 us for the last one; dealt in used chunks every workgroup gets the same share)
    bool one_round_too = tiles * 2 > kSlots;
#ifdef DCL_DIAG
    if (g_conv_order_mode == 3) one_round_too = false;                                         // A/B: the former rule
#endif
    if (have_bal && CIN >= 32 && (aligned_ns || stream_k || one_round_too) && have_scratch && scratch_floats > kConvCounterWords &&
        tiles <= kConvCounterWords && !never) {
      use_bal = 1;
      stream_k = 1;
      aligned_ns = 0;
      const long long slots_fit = (scratch_floats - kConvCounterWords) / ((long long)2 * BM * BN);
      G = (int)(kSlots < slots_fit ? kSlots : slots_fit);
      split = true;
      counters = true;                                            // (already zeroed above: aligned / stream-K launches own them)
    }
  }
#ifdef DCL_DIAG
  if (g_conv_order_mode == 1) { keep_order = false; use_bal = 0; }                             // A/B: natural row order
  if (g_conv_order_mode == 2) use_bal = 0;                                                     // A/B: order, nominal units
#endif
  P.stream_k = stream_k; P.aligned_ns = aligned_ns; P.G = G; P.use_bal = use_bal; P.deferred = deferred ? 1 : 0;
  P.split = split ? 1 : 0; P.counters = counters ? 1 : 0; P.keep_order = keep_order ? 1 : 0; P.tiles = tiles; P.nchunks = nchunks;
  return P;
}

template <int CIN, int WR, int WCW, int NT>
static void launch_conv_dma(const DclConvSides &sides, int nsides, int cout, int kvol, int subm, int relu, float *scratch,
                            long long scratch_floats, int counters_ready, int *counters_state, hipStream_t s) {
  constexpr int BM = 32 * WR, BN = 32 * NT * WCW, KC = 32;
  const size_t lds = (size_t)(2 * (BM * KC + KC * BN) + 27 * BM + 8 + BM) * sizeof(float);
  (void)hipFuncSetAttribute((const void *)k_sparse_conv_dma<CIN, WR, WCW, NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  (void)hipFuncSetAttribute((const void *)k_sparse_conv_dma<CIN, WR, WCW, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
#ifdef DCL_CONV_STAMPS
  {
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_sparse_conv_dma<CIN, WR, WCW, NT, true>, 64 * WR * WCW, lds);
    fprintf(stderr, "k_sparse_conv_dma<%d,%d,%d,%d>: dynamic LDS %zu B, occupancy API says %d workgroups per CU\n", CIN, WR, WCW, NT, lds, occ);
  }
#endif
  const DclConvPlan P = plan_conv_dma(sides, nsides, CIN, BM, BN, cout, kvol, scratch != nullptr, scratch_floats, g_conv_slots);
#ifdef DCL_CONV_STAMPS
  const int stamp_sel = g_stamp_select.load(), stamp_no = g_stamp_count.fetch_add(1);
  const int stamp_bit = (stamp_sel < 0 || stamp_sel == stamp_no) ? 16 : 0;
  if (stamp_bit)
    fprintf(stderr, "stamped launch %d: Cin %d tile %dx%d sides %d  G %d stream_k %d aligned_ns %d use_bal %d deferred %d tiles %d chunks/tile %d\n",
            stamp_no, CIN, BM, BN, nsides, P.G, P.stream_k, P.aligned_ns, P.use_bal, P.deferred, P.tiles, P.nchunks);
#else
  constexpr int stamp_bit = 0;
#endif
  float *partial = scratch;
  int32_t *counters = nullptr;
  if (P.split) {
    partial = scratch + kConvCounterWords;
    if (P.counters) {
      counters = reinterpret_cast<int32_t *>(scratch);
      // (counters_state: the caller's "the tickets are zero" flag of a pass -- the first launch that needs them zeroes them, a
      // pass of few-row launches, whose combine is a launch of its own, never does)
      if (!counters_ready && !(counters_state && *counters_state)) dcl_internal_zero_words(counters, kConvCounterWords, s);
      if (counters_state) *counters_state = 1;
    }
  }
  DclConvSides sd = sides;
  if (!P.keep_order)
    for (int i = 0; i < nsides; ++i) sd.s[i].ord = DclRowOrder{nullptr, nullptr, nullptr};
  bool any_order = false;
  for (int i = 0; i < nsides; ++i) any_order = any_order || sd.s[i].ord.order != nullptr;
  if (any_order)
    hipLaunchKernelGGL((k_sparse_conv_dma<CIN, WR, WCW, NT, true>), dim3(P.G), dim3(64 * WR * WCW), lds, s, sd, nsides, cout, kvol,
                       subm, relu, partial, P.stream_k, P.aligned_ns, (int)g_conv_xcd_remap | stamp_bit, counters, P.use_bal);
  else
    hipLaunchKernelGGL((k_sparse_conv_dma<CIN, WR, WCW, NT, false>), dim3(P.G), dim3(64 * WR * WCW), lds, s, sd, nsides, cout, kvol,
                       subm, relu, partial, P.stream_k, P.aligned_ns, (int)g_conv_xcd_remap | stamp_bit, counters, 0);
  if (P.deferred)
    hipLaunchKernelGGL((k_conv_frag_reduce<WR, WCW, NT>), dim3(P.tiles, WR * WCW * NT), dim3(256), 0, s, partial, sd, nsides, cout,
                       P.nchunks, P.G, P.stream_k, relu);
}

#ifdef DCL_DIAG
template <int CIN, int WC, int KC>
static void launch_conv_tile(int rows, const float *feat, const DclNbrSrc &nbr, int cap, const int32_t *n_out_dev,
                             int n_out_host, const float *W, int cout, int kvol, int subm, const float *scale,
                             const float *shift, int relu, float *out, float *scratch, long long scratch_floats,
                             int counters_ready, hipStream_t s) {
  (void)counters_ready;
  if (scratch && scratch_floats > kConvCounterWords) { scratch += kConvCounterWords; scratch_floats -= kConvCounterWords; }
  else scratch = nullptr;
  constexpr int WR = 4 / WC, BM = 32 * WR, BN = 32 * WC;
  const size_t lds = (size_t)(BM * (KC + 4) + KC * BN + 27 * BM) * sizeof(float);
  (void)hipFuncSetAttribute((const void *)k_sparse_conv_tile<CIN, WC, KC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  const int nblk = dcl_div_up(rows, BM);
  // one-image calls: a handful of row tiles walking all 27 offsets is pure latency -- split the walk (>= 1 chunk each)
  const int nchunks = dcl_div_up(kvol, KC / CIN);
  int nsplit = 1;
  if (scratch && g_conv_split >= 0 && conv_few_rows(rows, n_out_dev != nullptr)) {
    nsplit = g_conv_split > 0 ? (int)g_conv_split : kConvMaxSplit;
    if (nsplit > nchunks) nsplit = nchunks;
    while (nsplit > 1 && (long long)nsplit * cap * cout > scratch_floats) --nsplit;
    if (nsplit < 1) nsplit = 1;
  }
  hipLaunchKernelGGL((k_sparse_conv_tile<CIN, WC, KC>), dim3(nblk < 65535 ? nblk : 65535, cout / BN, nsplit), dim3(256), lds,
                     s, feat, nbr, cap, n_out_dev, n_out_host, W, cout, kvol, subm, scale, shift, relu, out, scratch, nsplit);
  if (nsplit > 1)
    hipLaunchKernelGGL(k_conv_split_reduce, dim3(dcl_grid_1d((long long)rows * (cout / 4), 256)), dim3(256), 0, s, scratch,
                       nsplit, cap, n_out_dev, n_out_host, cout, scale, shift, relu, out);
}

#endif  // DCL_DIAG

// ---- sparse average pool (conv_body.h: avgpool_body) ----------------------------------------------------------------
template <int PF, bool K27>
__global__ __launch_bounds__(256) void k_sparse_avgpool(const DclConvSides sides, int nsides, int c, int kvol,
                                                        int32_t *__restrict__ rf_out, const int32_t *__restrict__ rf_in) {
  __shared__ int32_t s_v[64 * 27];
  avgpool_body<256, PF, K27>(sides, nsides, c, kvol, rf_out, rf_in, s_v, blockIdx.x, gridDim.x);
}


__global__ void k_sparse_avgpool_scalar(const float *__restrict__ feat, const DclNbrSrc src, int cap,
                                        const int32_t *__restrict__ n_out_dev, int n_out_host, int c, int kvol,
                                        float *__restrict__ out, int32_t *__restrict__ rf_out,
                                        const int32_t *__restrict__ rf_in) {
  int n = n_out_dev ? *n_out_dev : n_out_host;
  n = n < cap ? n : cap;
  const long long total = (long long)n * c;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(t / c);
    const int ch = (int)(t - (long long)row * c);
    int rf = 0;
    for (int k = 0; k < kvol; ++k) rf += dcl_nbr_at(src, cap, k, row) >= 0;
    if (rf_in) rf = rf_in[row];
    const float d = (float)rf;
    float acc = 0.f;
    for (int k = 0; k < kvol; ++k) {
      const int v = dcl_nbr_at(src, cap, k, row);
      if (v >= 0) acc = acc + feat[(size_t)v * c + ch] / d;
    }
    out[t] = acc;
    if (rf_out && ch == 0) rf_out[row] = rf;
  }
}

}  // namespace

DCL_HOOK_INT(g_force_valu, 0);   // 1 = plain VALU kernel for every conv, 2 = MFMA kernel without LDS staging (the general Cin % 8 fallback), 4 = register-staged tile kernel instead of the LDS-DMA one, 5 = 4-wave 128x64 tiles for Cout = 64 (the former default)
#ifdef DCL_DIAG
DCL_API void dcl_debug_force_valu_conv(int on) { g_force_valu = on; }
DCL_API void dcl_debug_conv_split(int n) { g_conv_split = n; }
DCL_API void dcl_debug_conv_few_chunks(int n) { g_conv_few_chunks = n >= 1 ? n : 4; }
DCL_API void dcl_debug_conv_few_tiles(int on) { g_conv_few_tiles = on; }
DCL_API void dcl_debug_conv_wlds(int on) { g_conv_wlds = on; }
DCL_API void dcl_debug_conv_few_cap(int rows) { kConvFewRowsCap = rows; }
DCL_API void dcl_debug_conv_few_hint(int rows) { kConvFewRowsHint = rows; }
DCL_API void dcl_debug_conv_order_mode(int mode) { g_conv_order_mode = mode; }
DCL_API void dcl_debug_conv_slots(int n) { g_conv_slots = (n >= 64 && n <= 512) ? n : 512; }
DCL_API void dcl_debug_conv_xcd_remap(int on) { g_conv_xcd_remap = on; }
#endif
#ifdef DCL_CONV_STAMPS
// host: kStampWgs * kStampSegs * 8 uint64 (the stamps; phase = 1: wave 0's per-phase cycle sums); clear = 1 zeroes both tables
extern "C" __attribute__((visibility("default"))) int dcl_debug_conv_stamps(unsigned long long *host, int phase, int clear) {
  constexpr size_t bytes = sizeof(unsigned long long) * kStampWgs * kStampSegs * 16;
  if (clear) {
    static unsigned long long zeros[kStampWgs * kStampSegs * 16];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_phase), zeros, bytes);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), zeros, bytes);
  }
  if (phase) return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_phase), bytes);
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_stamps), bytes);
}
// stamp only the n-th LDS-DMA conv launch from now on (n < 0: every launch, each overwriting the last)
extern "C" __attribute__((visibility("default"))) void dcl_debug_conv_stamps_select(int n) {
  g_stamp_select = n;
  g_stamp_count = 0;
}
#endif
// most K-splits a conv launch over `rows` output rows may use (sizes the partial-sum scratch; backbone.hip)
int dcl_internal_conv_split_cap(long long rows) { return conv_split_cap(rows); }

DCL_API int dcl_sparse_conv_scratch_floats(int rows_cap, int cout, int64_t *floats_host) {
  DCL_CHECK_ARG(rows_cap >= 0 && cout > 0 && floats_host);
  // stream-K: two partial-tile slots (128 x min(Cout,128) floats) per workgroup of the 512-slot grid + the tile tickets;
  // the register-staged tile kernel (A/B variant) keeps split-major row partials
  const int64_t stream = (int64_t)2 * 512 * 128 * (cout < 128 ? cout : 128);
  const int64_t rowsplit = (int64_t)conv_split_cap(rows_cap) * (((int64_t)rows_cap + 127) / 128 * 128) * cout;
  *floats_host = (stream > rowsplit ? stream : rowsplit) + kConvCounterWords;
  return 0;
}

DCL_API int dcl_sparse_conv_fwd(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev,
                                int n_out_host, const float *W, int cin, int cout, int kvol, int subm,
                                const float *scale, const float *shift, int relu, float *out,
                                dclStream_t stream) {
  return dcl_sparse_conv_fwd_ws(feat, nbr, cap, n_out_dev, n_out_host, W, cin, cout, kvol, subm, scale, shift, relu, out,
                                nullptr, 0, stream);
}

DCL_API int dcl_sparse_conv_fwd_ws(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev,
                                   int n_out_host, const float *W, int cin, int cout, int kvol, int subm,
                                   const float *scale, const float *shift, int relu, float *out, float *scratch,
                                   int64_t scratch_floats, dclStream_t stream) {
  DCL_CHECK_ARG(nbr);
  const DclNbrSrc src = {nbr, nullptr, nullptr, nullptr, nullptr, 0, 1, 0};
  return dcl_internal_sparse_conv_fwd(feat, src, cap, n_out_dev, n_out_host, W, cin, cout, kvol, subm, scale, shift, relu,
                                      out, scratch, scratch_floats, stream);
}

// Same with a row order from dcl_order_rows (row_order.hip): tile slots compute rows in that order, work is dealt in used
// chunks where the launch splits tiles.  Results equal dcl_sparse_conv_fwd_ws up to the fp32 summation split points.
DCL_API int dcl_sparse_conv_fwd_ordered(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev,
                                        int n_out_host, const float *W, int cin, int cout, int kvol, int subm,
                                        const float *scale, const float *shift, int relu, float *out, float *scratch,
                                        int64_t scratch_floats, const int32_t *order, const int32_t *bal,
                                        const uint32_t *smask, dclStream_t stream) {
  DCL_CHECK_ARG(nbr && order && (bal == nullptr) == (smask == nullptr));
  const DclNbrSrc src = {nbr, nullptr, nullptr, nullptr, nullptr, 0, 1, 0};
  const DclRowOrder ord{order, bal, smask};
  return dcl_internal_sparse_conv_fwd(feat, src, cap, n_out_dev, n_out_host, W, cin, cout, kvol, subm, scale, shift, relu,
                                      out, scratch, scratch_floats, stream, 0, &ord);
}

// ---- measurement facility (bench.py's `roofline_sparse_conv`): while enabled, every sparse-conv call is bracketed by
// HIP events on the stream it is launched on; dcl_profile_conv_end() waits for them and returns the summed device time.
// Mutex-protected; not for use under stream capture (events would become graph nodes).
namespace {
struct ConvProfile {
  std::mutex mu;
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  std::vector<std::array<int32_t, 4>> what;      // per call: cin, cout, subm, sides
};
ConvProfile g_conv_prof;
}  // namespace

DCL_API int dcl_profile_conv_begin(void) {
  std::lock_guard<std::mutex> lock(g_conv_prof.mu);
  for (auto &e : g_conv_prof.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  g_conv_prof.ev.clear();
  g_conv_prof.what.clear();
  g_conv_prof.on = true;
  return 0;
}

DCL_API int dcl_profile_conv_end_calls(double *ms_total_host, int32_t *calls_host, float *ms_per_call_host,
                                       int32_t *what_per_call_host, int32_t cap) {
  std::lock_guard<std::mutex> lock(g_conv_prof.mu);
  g_conv_prof.on = false;
  double total = 0.0;
  int32_t i = 0;
  for (auto &e : g_conv_prof.ev) {
    float ms = 0.f;
    if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) total += ms;
    if (i < cap) {
      if (ms_per_call_host) ms_per_call_host[i] = ms;
      if (what_per_call_host)
        for (int q = 0; q < 4; ++q) what_per_call_host[4 * i + q] = g_conv_prof.what[i][q];
    }
    ++i;
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  if (ms_total_host) *ms_total_host = total;
  if (calls_host) *calls_host = (int32_t)g_conv_prof.ev.size();
  g_conv_prof.ev.clear();
  g_conv_prof.what.clear();
  return 0;
}

DCL_API int dcl_profile_conv_end(double *ms_total_host, int32_t *calls_host) {
  return dcl_profile_conv_end_calls(ms_total_host, calls_host, nullptr, nullptr, 0);
}

static int conv_dispatch(const DclConvSides &sides, int nsides, int cin, int cout, int kvol, int subm, int relu, float *scratch,
                         int64_t scratch_floats, int counters_ready, int *counters_state, dclStream_t stream);

// library-internal: one layer of up to two problems ("sides") in one launch; `src` may be an implicit rulebook (native
// backbone runner).  Timed as ONE conv call by the measurement facility above.
int dcl_internal_sparse_conv_fwd_sides(const DclConvSides &sides, int nsides, int cin, int cout, int kvol, int subm, int relu,
                                       float *scratch, int64_t scratch_floats, dclStream_t stream, int counters_ready,
                                       int *counters_state) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool timed = false;
  {
    std::lock_guard<std::mutex> lock(g_conv_prof.mu);
    timed = g_conv_prof.on;
  }
  if (timed) {
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, (hipStream_t)stream);
  }
  const int rc = conv_dispatch(sides, nsides, cin, cout, kvol, subm, relu, scratch, scratch_floats, counters_ready, counters_state,
                               stream);
  if (timed) {
    (void)hipEventRecord(e1, (hipStream_t)stream);
    std::lock_guard<std::mutex> lock(g_conv_prof.mu);
    g_conv_prof.ev.emplace_back(e0, e1);
    g_conv_prof.what.push_back({cin, cout, subm, nsides});
  }
  return rc;
}

int dcl_internal_sparse_conv_fwd(const float *feat, const DclNbrSrc &nbr, int cap, const int32_t *n_out_dev,
                                 int n_out_host, const float *W, int cin, int cout, int kvol, int subm, const float *scale,
                                 const float *shift, int relu, float *out, float *scratch, int64_t scratch_floats,
                                 dclStream_t stream, int counters_ready, const DclRowOrder *ord) {
  DclConvSides sides{};
  DclConvSide &S = sides.s[0];
  S.feat = feat; S.src = nbr; S.n_dev = n_out_dev; S.n_host = n_out_dev ? 0 : n_out_host; S.cap = cap;
  S.W = W; S.scale = scale; S.shift = shift; S.out = out;
  S.ord = ord ? *ord : DclRowOrder{nullptr, nullptr, nullptr};
  return dcl_internal_sparse_conv_fwd_sides(sides, 1, cin, cout, kvol, subm, relu, scratch, scratch_floats, stream,
                                            counters_ready);
}

// Which kernel family / tile shape a layer takes.
static DclConvChoice conv_choose(const DclConvSides &sides, int nsides, int cin, int cout, int kvol, bool have_scratch) {
  DclConvChoice c{DCL_CONV_GENERIC, 0, 0, 0};
  const bool mfma_ok = g_force_valu != 1 && (cin % 8 == 0) && (cout % 32 == 0);
  const bool lds_ok = mfma_ok && g_force_valu != 2 && (cin == 16 || cin == 32 || cin == 64 || cin == 128);
#ifdef DCL_DIAG
  const bool diag_tile = lds_ok && g_force_valu == 4;
#else
  const bool diag_tile = false;
#endif
  if (lds_ok && !diag_tile) {
    // LDS-DMA implicit-GEMM kernel; the tile shape follows Cout.  Few-row launches (a handful of crops; latency-bound) take
    // 64-row tiles: twice the workgroups, half the MFMA time per chunk and half the neighbour table per workgroup.
    const bool is_few = conv_launch_is_few(sides, nsides);
    const bool few_tiles = g_conv_few_tiles != 0 && have_scratch && cout % 64 == 0 && is_few;
    // wide, shallow layers (Cin 16 / 32 -> 32 channels, many rows): the filter-resident kernel, no staging, no barriers
    // (Measured and dropped: the middle layers (Cin 32 / 64, filter too big for LDS) with the rows in registers and one
    // offset's filter slice staged per step, workgroups of 2 / 4 waves in lock step over the 27 offsets, no split-K -- 99 /
    // 105 / 161 us for the 32->64 / 64->64 / 64->128 layers against 87 / 79 / 142 of the LDS-DMA kernel: a barrier and a
    // drained vmcnt per offset cost more than the row staging they replace.)
    // (the 32 -> 64 layer as two 32-column halves -- instantiated, measured, not used: 99 us against the DMA kernel's 86)
    // Two sides in one call (one-stream schedule): the 16-channel layer goes out as a launch per side (44 us each against 105
    // for the grouped LDS-DMA launch), the 32-channel one keeps the grouped LDS-DMA launch (72 us against 2 x 48).
    const bool wlds32 = ((cin == 32 && cout == 32 && (nsides == 1 || g_conv_wlds == 3)) || (g_conv_wlds == 2 && cout == 64 && cin == 32));
    if (g_conv_wlds != 0 && ((cout == 32 && cin == 16) || wlds32) && kvol == 27 && !is_few) {
      c.family = DCL_CONV_WLDS;
      return c;
    }
    c.family = DCL_CONV_DMA;
#ifdef DCL_DIAG
    if (g_force_valu == 5 && cout % 128 != 0 && cout % 64 == 0) { c.WR = 4; c.WCW = 1; c.NT = 2; return c; }   // A/B: the former 4 waves of 32x64
#endif
    if (few_tiles) {                                   // few rows: 64-row tiles, 64x128 (4 waves of 32x64) or 64x64 (4 waves of 32x32)
      c.WR = 2; c.WCW = 2; c.NT = cout % 128 == 0 ? 2 : 1;
    } else if (cout % 64 != 0) {                       // Cout = 32: 128x32 tiles, 4 waves
      c.WR = 4; c.WCW = 1; c.NT = 1;
    } else if (cout % 128 == 0) {                      // 128x128 tiles, 8 waves
      c.WR = 4; c.WCW = 2; c.NT = 2;
    } else {
      // Cout % 64 == 0: 128x64 tiles, EIGHT waves of 32x32 (each wave issues 3 DMA pieces per chunk instead of 6 and has 16
      // MFMAs instead of 32 behind them: 84 -> 80 us on the 32->64 layer, 80 -> 70 on the 64->64 one; four waves of 32x64
      // were the form until the DMA pieces of a wave went out as grouped statements)
      c.WR = 4; c.WCW = 2; c.NT = 1;
    }
    return c;
  }
  if (cin == 7 && cout == 16 && kvol <= 27 && g_force_valu != 1 && !mfma_ok) c.family = DCL_CONV_STEM;
  return c;
}

static int conv_dispatch(const DclConvSides &sides_in, int nsides_in, int cin, int cout, int kvol, int subm, int relu,
                         float *scratch, int64_t scratch_floats, int counters_ready, int *counters_state, dclStream_t stream) {
  DCL_CHECK_ARG(nsides_in >= 1 && nsides_in <= 2 && cin > 0 && cout > 0 && kvol > 0 && kvol <= 27);
  // sides without rows drop out (an empty level of one backbone)
  DclConvSides sides{};
  int nsides = 0;
  for (int i = 0; i < nsides_in; ++i) {
    const DclConvSide &S = sides_in.s[i];
    DCL_CHECK_ARG(S.feat && (S.src.nbr || (S.src.out_indices && S.src.in_mask && S.src.in_wprefix && kvol == 27)) && S.W && S.out &&
                  S.cap > 0);
    DCL_CHECK_ARG((S.scale == nullptr) == (S.shift == nullptr));
    DCL_CHECK_ARG(S.n_dev || (S.n_host >= 0 && S.n_host <= S.cap));
    DCL_CHECK_ARG((S.n_dev != nullptr) == (sides_in.s[0].n_dev != nullptr));           // capacity mode: all sides or none
    if (S.n_dev || S.n_host > 0) sides.s[nsides++] = S;
  }
  if (nsides == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const DclConvChoice ch = conv_choose(sides, nsides, cin, cout, kvol, scratch != nullptr);
  if (ch.family == DCL_CONV_WLDS) {
    // Cin = 16: both sides' filters fit the LDS together (2 x 54 KiB) -> ONE launch over both sides' tiles; Cin = 32: a launch per
    // side (one workgroup per CU is all a filter leaves room for)
    const int per_launch = cin == 16 ? nsides : 1;
    const size_t lds = (size_t)per_launch * 27 * cin * 32 * sizeof(float);
    const dim3 grid(256 / (cout / 32), cout / 32), block(cin == 16 ? 1024 : 512);
#define WLDS_LAUNCH(CI, CO, SB)                                                                                              \
    do {                                                                                                                   \
      (void)hipFuncSetAttribute((const void *)k_sparse_conv_wlds<CI, CO, SB>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                (int)lds);                                                                                 \
      hipLaunchKernelGGL((k_sparse_conv_wlds<CI, CO, SB>), grid, block, lds, s, one, per_launch, relu);                    \
    } while (0)
    for (int side_i = 0; side_i < nsides; side_i += per_launch) {
      DclConvSides one{};
      for (int q = 0; q < per_launch; ++q) one.s[q] = sides.s[side_i + q];
      if (cin == 16) { if (subm) WLDS_LAUNCH(16, 32, true); else WLDS_LAUNCH(16, 32, false); }
      else if (cout == 32) { if (subm) WLDS_LAUNCH(32, 32, true); else WLDS_LAUNCH(32, 32, false); }
      else { if (subm) WLDS_LAUNCH(32, 64, true); else WLDS_LAUNCH(32, 64, false); }
    }
#undef WLDS_LAUNCH
    DCL_LAUNCH_CHECK();
    return 0;
  }
  if (ch.family == DCL_CONV_DMA) {
#define DMA_ARGS sides, nsides, cout, kvol, subm, relu, scratch, (long long)scratch_floats, counters_ready, counters_state, s
#define DMA_CASE(WR_, WCW_, NT_)                                          \
    switch (cin) {                                                      \
      case 16: launch_conv_dma<16, WR_, WCW_, NT_>(DMA_ARGS); break;    \
      case 32: launch_conv_dma<32, WR_, WCW_, NT_>(DMA_ARGS); break;    \
      case 64: launch_conv_dma<64, WR_, WCW_, NT_>(DMA_ARGS); break;    \
      default: launch_conv_dma<128, WR_, WCW_, NT_>(DMA_ARGS); break;   \
    }
    if (ch.WR == 2 && ch.NT == 2) { DMA_CASE(2, 2, 2) }
    else if (ch.WR == 2) { DMA_CASE(2, 2, 1) }
    else if (ch.WCW == 1 && ch.NT == 1) { DMA_CASE(4, 1, 1) }
#ifdef DCL_DIAG
    else if (ch.WCW == 1 && ch.NT == 2) { DMA_CASE(4, 1, 2) }
#endif
    else if (ch.NT == 2) { DMA_CASE(4, 2, 2) }
    else { DMA_CASE(4, 2, 1) }
#undef DMA_CASE
#undef DMA_ARGS
    DCL_LAUNCH_CHECK();
    return 0;
  }
  if (ch.family == DCL_CONV_STEM) {
    int rows = 0, expect = 0;                            // capacity mode: the grid by capacity, the kernel form by the expected rows
    for (int i = 0; i < nsides; ++i) {
      rows += sides.s[i].n_dev ? sides.s[i].cap : sides.s[i].n_host;
      expect += sides.s[i].form_rows > 0 ? sides.s[i].form_rows
                                         : (sides.s[i].n_dev ? (sides.s[i].n_host > 0 ? sides.s[i].n_host : sides.s[i].cap) : sides.s[i].n_host);
    }
    // many rows: a lane per row; a handful of crops: four lanes per row (conv_body.h: conv_stem_body).  The two forms add a row's
    // neighbours in different orders, so inside the backbone runner the choice goes by form_rows -- crops x 3600, the same
    // number launch by launch and under graph capture (round-5 review: the capacity hint of one and the exact count of the other
    // could put the same batch on either side of the switch)
    if (expect > 49152)
      hipLaunchKernelGGL((k_sparse_conv_stem<7, 16, 1>), dim3(dcl_grid_1d(rows, 256, 768)), dim3(256), 0, s, sides, nsides, kvol, subm,
                         relu);
    else
      hipLaunchKernelGGL((k_sparse_conv_stem<7, 16, 4>), dim3(dcl_grid_1d(rows, 64, 768)), dim3(256), 0, s, sides, nsides, kvol, subm,
                         relu);
    DCL_LAUNCH_CHECK();
    return 0;
  }
  {
    const bool mfma_ok = g_force_valu != 1 && (cin % 8 == 0) && (cout % 32 == 0);
#ifdef DCL_DIAG
    const bool diag_tile = mfma_ok && g_force_valu == 4 && (cin == 16 || cin == 32 || cin == 64 || cin == 128);
#endif
    // the general kernels take one problem per launch
    for (int i = 0; i < nsides; ++i) {
      const DclConvSide &S = sides.s[i];
      const int rows = S.n_dev ? S.cap : S.n_host;
#ifdef DCL_DIAG
      if (diag_tile) {                                                       // A/B: register-staged tile kernel
        float *scr = i == 0 ? scratch : nullptr;                             // (one scratch: the second side goes unsplit)
#define TILE_ARGS rows, S.feat, S.src, S.cap, S.n_dev, S.n_host, S.W, cout, kvol, subm, S.scale, S.shift, relu, S.out, scr, \
                  (long long)scratch_floats, counters_ready, s
        if (cout % 64 == 0) {
          switch (cin) {
            case 16: launch_conv_tile<16, 2, 128>(TILE_ARGS); break;
            case 32: launch_conv_tile<32, 2, 128>(TILE_ARGS); break;
            case 64: launch_conv_tile<64, 2, 128>(TILE_ARGS); break;
            default: launch_conv_tile<128, 2, 128>(TILE_ARGS); break;
          }
        } else {
          switch (cin) {
            case 16: launch_conv_tile<16, 1, 64>(TILE_ARGS); break;
            case 32: launch_conv_tile<32, 1, 64>(TILE_ARGS); break;
            case 64: launch_conv_tile<64, 1, 64>(TILE_ARGS); break;
            default: launch_conv_tile<128, 1, 128>(TILE_ARGS); break;
          }
        }
#undef TILE_ARGS
        continue;
      }
#endif
      if (mfma_ok) {
        const int ntiles = dcl_div_up(rows, 32);
        const int nt = (cout % 64 == 0 && (long long)ntiles * (cout / 64) >= 4096) ? 2 : 1;
        const int ytiles = cout / (32 * nt);
        const int blocks = dcl_grid_1d(ntiles, 4, 256 * 8);
        if (nt == 2)
          hipLaunchKernelGGL((k_sparse_conv_mfma<2>), dim3(blocks, ytiles), dim3(256), 0, s, S.feat, S.src, S.cap, S.n_dev,
                             S.n_host, S.W, cin, cout, kvol, subm, S.scale, S.shift, relu, S.out);
        else
          hipLaunchKernelGGL((k_sparse_conv_mfma<1>), dim3(blocks, ytiles), dim3(256), 0, s, S.feat, S.src, S.cap, S.n_dev,
                             S.n_host, S.W, cin, cout, kvol, subm, S.scale, S.shift, relu, S.out);
      } else {
        hipLaunchKernelGGL(k_sparse_conv_valu, dim3(dcl_grid_1d((long long)rows * cout, 256)), dim3(256), 0, s, S.feat, S.src,
                           S.cap, S.n_dev, S.n_host, S.W, cin, cout, kvol, subm, S.scale, S.shift, relu, S.out);
      }
    }
  }
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_sparse_avgpool_fwd(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev,
                                   int n_out_host, int c, int kvol, float *out, int32_t *rf,
                                   dclStream_t stream) {
  DCL_CHECK_ARG(nbr);
  const DclNbrSrc src = {nbr, nullptr, nullptr, nullptr, nullptr, 0, 1, 0};
  return dcl_internal_sparse_avgpool_fwd(feat, src, cap, n_out_dev, n_out_host, c, kvol, out, rf, stream);
}

static int avgpool_launch(const float *feat, const DclNbrSrc &nbr, int cap, const int32_t *n_out_dev, int n_out_host, int c,
                          int kvol, float *out, int32_t *rf, const int32_t *rf_in, dclStream_t stream);

DCL_API int dcl_sparse_avgpool_fwd_rf(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev,
                                      int n_out_host, int c, int kvol, const int32_t *summaryrf, float *out,
                                      dclStream_t stream) {
  DCL_CHECK_ARG(nbr && summaryrf);
  const DclNbrSrc src = {nbr, nullptr, nullptr, nullptr, nullptr, 0, 1, 0};
  return avgpool_launch(feat, src, cap, n_out_dev, n_out_host, c, kvol, out, nullptr, summaryrf, stream);
}

int dcl_internal_sparse_avgpool_fwd(const float *feat, const DclNbrSrc &nbr, int cap, const int32_t *n_out_dev,
                                    int n_out_host, int c, int kvol, float *out, int32_t *rf, dclStream_t stream) {
  return avgpool_launch(feat, nbr, cap, n_out_dev, n_out_host, c, kvol, out, rf, nullptr, stream);
}

static int avgpool_launch(const float *feat, const DclNbrSrc &nbr, int cap, const int32_t *n_out_dev, int n_out_host, int c,
                          int kvol, float *out, int32_t *rf, const int32_t *rf_in, dclStream_t stream) {
  DclConvSides sides{};
  DclConvSide &S = sides.s[0];
  S.feat = feat; S.src = nbr; S.cap = cap; S.n_dev = n_out_dev; S.n_host = n_out_dev ? 0 : n_out_host; S.out = out;
  return dcl_internal_sparse_avgpool_fwd_sides(sides, 1, c, kvol, rf, rf_in, stream);
}

// up to two pools (the two backbones' pools of a level) in one launch; rf / rf_in only with one side
int dcl_internal_sparse_avgpool_fwd_sides(const DclConvSides &sides_in, int nsides_in, int c, int kvol, int32_t *rf,
                                          const int32_t *rf_in, dclStream_t stream) {
  DCL_CHECK_ARG(nsides_in >= 1 && nsides_in <= 2 && c > 0 && kvol > 0 && kvol <= 27 && (nsides_in == 1 || (!rf && !rf_in)));
  DclConvSides sides{};
  int nsides = 0;
  long long rows = 0, expect = 0;                        // capacity mode: the grid by capacity, the kernel form by the expected rows
  for (int i = 0; i < nsides_in; ++i) {
    const DclConvSide &S = sides_in.s[i];
    DCL_CHECK_ARG(S.feat && (S.src.nbr || (S.src.out_indices && S.src.in_mask && S.src.in_wprefix && kvol == 27)) && S.out && S.cap > 0);
    DCL_CHECK_ARG(S.n_dev || (S.n_host >= 0 && S.n_host <= S.cap));
    if (S.n_dev || S.n_host > 0) {
      sides.s[nsides++] = S;
      rows += S.n_dev ? S.cap : S.n_host;
      expect += S.n_dev ? (S.n_host > 0 ? S.n_host : S.cap) : S.n_host;
    }
  }
  if (nsides == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int c4 = c / 4;
  if (c % 4 == 0 && c4 >= 4 && c4 <= 64 && 256 % c4 == 0) {
    const dim3 grid(dcl_grid_1d(rows * c4, 256, 2048));
    if (kvol != 27)                                        // (op-level calls with another window: the network's pools are 3^3)
      hipLaunchKernelGGL((k_sparse_avgpool<14, false>), grid, dim3(256), 0, s, sides, nsides, c, kvol, rf, rf_in);
    else if (expect * c4 <= 256 * 512)                     // at most two workgroups per CU: one round of 27 gathers
      hipLaunchKernelGGL((k_sparse_avgpool<27, true>), grid, dim3(256), 0, s, sides, nsides, c, kvol, rf, rf_in);
    else
      hipLaunchKernelGGL((k_sparse_avgpool<14, true>), grid, dim3(256), 0, s, sides, nsides, c, kvol, rf, rf_in);
  } else {
    for (int i = 0; i < nsides; ++i) {
      const DclConvSide &S = sides.s[i];
      hipLaunchKernelGGL(k_sparse_avgpool_scalar, dim3(dcl_grid_1d((long long)(S.n_dev ? S.cap : S.n_host) * c, 256)), dim3(256), 0,
                         s, S.feat, S.src, S.cap, S.n_dev, S.n_host, c, kvol, S.out, rf, rf_in);
    }
  }
  DCL_LAUNCH_CHECK();
  return 0;
}

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
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

int dcl_internal_sparse_conv_fwd(const float *feat, const DclNbrSrc &nbr, int cap, const int32_t *n_out_dev,
                                 int n_out_host, const float *W, int cin, int cout, int kvol, int subm, const float *scale,
                                 const float *shift, int relu, float *out, float *scratch, int64_t scratch_floats,
                                 dclStream_t stream, int counters_ready = 0, const DclRowOrder *ord = nullptr);
int dcl_internal_sparse_avgpool_fwd(const float *feat, const DclNbrSrc &nbr, int cap, const int32_t *n_out_dev,
                                    int n_out_host, int c, int kvol, float *out, int32_t *rf, dclStream_t stream);
int dcl_internal_sparse_conv_fwd_sides(const DclConvSides &sides, int nsides, int cin, int cout, int kvol, int subm, int relu,
                                       float *scratch, int64_t scratch_floats, dclStream_t stream, int counters_ready = 0);
int dcl_internal_sparse_avgpool_fwd_sides(const DclConvSides &sides, int nsides, int c, int kvol, int32_t *rf,
                                          const int32_t *rf_in, dclStream_t stream);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// offset visiting order: reference adds the centre GEMM first for subm, then k ascending.
__device__ __forceinline__ int offset_at(int step, int kvol, int subm) {
  if (!subm) return step;
  const int centre = kvol / 2;
  if (step == 0) return centre;
  return step <= centre ? step - 1 : step;
}

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

// ---- stem kernel: small Cin/Cout known at compile time (DCL-Net: 7 -> 16) ------------------------------------
// one thread per output row, all COUT channels in registers; W (kvol x CIN x COUT) lives in LDS and is read as
// wave-uniform broadcasts.  Same summation order as the generic kernel (per offset an ascending-ci fmaf chain, then one add).
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_sparse_conv_stem(const DclConvSides sides, int nsides, int kvol, int subm, int relu) {
  // FOUR lanes per output row: lane g of a row's quad walks the kernel offsets s = g, g + 4, ... (each offset: neighbour
  // look-up, CIN loads, a CIN x COUT fmaf block -> one partial row added to the lane's sum in ascending s), then the four
  // lane sums are added as (l0 + l1) + (l2 + l3) by two butterfly rounds.  One thread per row made every row a serial chain
  // of 27 dependent look-ups -- 28 us for the 3 200 rows of a one-crop call, 23 us at 108 000 rows.
  __shared__ __attribute__((aligned(16))) float Ws[2 * 27 * CIN * COUT];         // both sides' filters
  static_assert(COUT % 16 == 0, "four lanes write COUT / 4 channels each as float4s");
  int n0 = sides.s[0].n_dev ? *sides.s[0].n_dev : sides.s[0].n_host;
  n0 = n0 < sides.s[0].cap ? n0 : sides.s[0].cap;
  int n1 = 0;
  if (nsides > 1) {
    n1 = sides.s[1].n_dev ? *sides.s[1].n_dev : sides.s[1].n_host;
    n1 = n1 < sides.s[1].cap ? n1 : sides.s[1].cap;
  }
  for (int i = threadIdx.x; i < kvol * CIN * COUT; i += 256) {
    Ws[i] = sides.s[0].W[i];
    if (nsides > 1) Ws[27 * CIN * COUT + i] = sides.s[1].W[i];
  }
  __syncthreads();
  const int g = threadIdx.x & 3;
  for (int q = blockIdx.x * 64 + (threadIdx.x >> 2); q < ((n0 + n1 + 63) & ~63); q += gridDim.x * 64) {   // whole quads stay together
    const bool live = q < n0 + n1;
    const int second = (live && q >= n0) ? 1 : 0;
    const DclConvSide &S = sides.s[second];
    const int row = live ? q - (second ? n0 : 0) : 0;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
    for (int s = g; s < kvol; s += 4) {
      const int k = offset_at(s, kvol, subm);
      const int v = live ? dcl_nbr_at(S.src, S.cap, k, row) : -1;
      if (v < 0) continue;
      float f[CIN];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) f[ci] = S.feat[(size_t)v * CIN + ci];
      const float *w = Ws + second * 27 * CIN * COUT + k * CIN * COUT;
      float part[COUT];
#pragma unroll
      for (int co = 0; co < COUT; ++co) part[co] = 0.0f;
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int co = 0; co < COUT; ++co) part[co] = __fmaf_rn(f[ci], w[ci * COUT + co], part[co]);
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] = acc[co] + part[co];
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {                  // (l0 + l1) + (l2 + l3): commutative adds, the same bits in all four lanes
      acc[co] = acc[co] + __shfl_xor(acc[co], 1, 64);
      acc[co] = acc[co] + __shfl_xor(acc[co], 2, 64);
    }
    if (!live) continue;
    constexpr int PER = COUT / 4;                        // channels written by each of the four lanes
    float o[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      float x = 0.f;
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) x = g == gg ? acc[gg * PER + j] : x;       // static register indices
      const int co = g * PER + j;
      if (S.scale) x = x * S.scale[co] + S.shift[co];
      if (relu) x = fmaxf(x, 0.0f);
      o[j] = x;
    }
    float4 *dst = reinterpret_cast<float4 *>(S.out + (size_t)row * COUT + g * PER);
#pragma unroll
    for (int j = 0; j < PER / 4; ++j) dst[j] = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
  }
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

// ---- MFMA kernel with the WHOLE filter resident in LDS: the wide, shallow layers (Cin 16 / 32 -> Cout 32) -----------------
// The first two MFMA layers of a backbone have the most rows (10^5 at 32 crops) and the fewest channels: 27 * Cin * 32 floats
// of weights are 54 / 108 KiB -- they fit the CU's LDS whole.  In the LDS-DMA implicit GEMM these layers were bound by the
// per-chunk hand-shake (a 128 x 32 tile is 16 MFMAs per wave between two barriers: MFMA busy 0.27-0.35); here there is no
// staging of the gathered rows and NO barrier after the filter load: a wave owns 32 output rows, looks up their 27
// neighbour rows once, and per kernel offset every lane loads ITS row's channels straight into the registers that are the
// MFMA's A operand (lane (r, h) holds channels 8h .. 8h+7 of every 16-channel group, so MFMA step i contracts the channel
// pair {i, 8 + i}; the B operand W[k][8h + i][col r] comes from LDS), the loads of the next offset in flight under the
// MFMAs of this one.  16 waves per CU hide the rest.  Offsets none of the wave's rows has are skipped.  Summation: per
// output the offsets in the reference's visiting order, inside an offset the MFMA's pair order -- a different fp32
// association than the DMA kernel's (both within the tolerance of the parity tests).
// Up to two problems per launch: the workgroups are split between the sides in proportion to their row tiles.
// (Cin = 16: 16 waves per workgroup at 128 registers; Cin = 32 has half the row tiles and twice the registers per offset in
// flight: 8 waves at 256 registers, five offsets ahead)
// Cout = 64 (the 32 -> 64 layer): blockIdx.y picks one of the two 32-column halves of the filter (108 KiB each); both halves
// gather the same rows (L2 traffic, not HBM).
template <int CIN, int COUT_T, bool SUBM>
__global__ __launch_bounds__(CIN == 16 ? 1024 : 512) void k_sparse_conv_wlds(const DclConvSides sides, int nsides, int relu) {
  constexpr int COUT = 32, KV = 27, GRP = CIN / 16;                          // 32 columns per workgroup; 16-channel groups per row
  constexpr int NTHR = CIN == 16 ? 1024 : 512, NWAVE = NTHR / 64;
  extern __shared__ __attribute__((aligned(16))) float wl_lds[];             // [27][CIN][32]
  const int col0 = blockIdx.y * COUT;
  int n0 = sides.s[0].n_dev ? *sides.s[0].n_dev : sides.s[0].n_host;
  n0 = n0 < sides.s[0].cap ? n0 : sides.s[0].cap;
  int n1 = 0;
  if (nsides > 1) {
    n1 = sides.s[1].n_dev ? *sides.s[1].n_dev : sides.s[1].n_host;
    n1 = n1 < sides.s[1].cap ? n1 : sides.s[1].cap;
  }
  const int t0 = (n0 + 31) >> 5, t1 = (n1 + 31) >> 5;
  const int G = gridDim.x;
  int g0 = G;                                                                // workgroups of side 0
  if (t1 > 0) {
    g0 = (int)(((long long)G * t0 + (t0 + t1) / 2) / (t0 + t1));
    g0 = g0 < 1 ? 1 : (g0 > G - 1 ? G - 1 : g0);
    if (t0 == 0) g0 = 0;
  }
  const int second = (int)blockIdx.x >= g0 ? 1 : 0;
  const DclConvSide &S = sides.s[second];
  const int n = second ? n1 : n0, ntiles = second ? t1 : t0;
  const int wg = blockIdx.x - (second ? g0 : 0), nwg = second ? G - g0 : g0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const float *__restrict__ feat = S.feat;
  {
    const float4 *Wg = reinterpret_cast<const float4 *>(S.W + col0);
    float4 *Wl = reinterpret_cast<float4 *>(wl_lds);
    for (int i = threadIdx.x; i < KV * CIN * COUT / 4; i += NTHR) Wl[i] = Wg[(i >> 3) * (COUT_T / 4) + (i & 7)];
  }
  __syncthreads();
  // (running the first tile's look-ups under the filter load was measured: the 27 live row numbers across the load push
  // the 16-wave variant over its 128 registers -- 43 -> 52 us)
  for (int tile = wg * NWAVE + wave; tile < ntiles; tile += nwg * NWAVE) {
    const int row = tile * 32 + r;
    const bool valid = row < n;
    int v[KV];                                                               // the 27 neighbour rows of this lane's output row
#pragma unroll
    for (int st = 0; st < KV; ++st) v[st] = valid ? dcl_nbr_at(S.src, S.cap, offset_at(st, KV, SUBM ? 1 : 0), row) : -1;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    constexpr int PF = CIN == 16 ? 4 : 5;              // offsets whose rows are in flight ahead of the MFMAs (register ring)
    float4 ring[PF + 1][2 * GRP];
    auto fetch = [&](int vv, float4 (&a)[2 * GRP]) {                          // a missing neighbour reads row 0 and is zeroed below
      const float4 *fp = reinterpret_cast<const float4 *>(feat + (size_t)(vv >= 0 ? vv : 0) * CIN + 8 * h);
#pragma unroll
      for (int g = 0; g < GRP; ++g) { a[2 * g] = fp[4 * g]; a[2 * g + 1] = fp[4 * g + 1]; }
    };
#pragma unroll
    for (int st = 0; st < PF; ++st) fetch(v[st], ring[st]);
#pragma unroll
    for (int st = 0; st < KV; ++st) {
      if (st + PF < KV) fetch(v[st + PF], ring[(st + PF) % (PF + 1)]);
      if (__ballot(v[st] >= 0) != 0ull) {
        const int k = offset_at(st, KV, SUBM ? 1 : 0);
        const bool have = v[st] >= 0;
        const float *wk = wl_lds + (k * CIN + 8 * h) * COUT + r;
        const float4 (&cur)[2 * GRP] = ring[st % (PF + 1)];
#pragma unroll
        for (int g = 0; g < GRP; ++g) {
          const float av[8] = {cur[2 * g].x, cur[2 * g].y, cur[2 * g].z, cur[2 * g].w,
                               cur[2 * g + 1].x, cur[2 * g + 1].y, cur[2 * g + 1].z, cur[2 * g + 1].w};
#pragma unroll
          for (int i = 0; i < 8; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(have ? av[i] : 0.0f, wk[(16 * g + i) * COUT], acc, 0, 0, 0);
        }
      }
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // MFMA -> VALU read of the accumulator
    const float sc = S.scale ? S.scale[col0 + r] : 1.0f, sh = S.scale ? S.shift[col0 + r] : 0.0f;
    float *__restrict__ out = S.out;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int orow = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (orow < n) {
        float x = acc[e];
        if (S.scale) x = x * sc + sh;
        if (relu) x = fmaxf(x, 0.0f);
        out[(size_t)orow * COUT_T + col0 + r] = x;
      }
    }
  }
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

// ---- implicit-GEMM MFMA kernel fed by LDS-DMA (Cout % 64 == 0): same 64x64 output tile and virtual-channel walk as
// k_sparse_conv_tile, but both operand tiles of a 64-channel chunk go global -> LDS with global_load_lds_dwordx4 (no
// staging registers, no ds_write), double-buffered: the DMA of the next used chunk is issued right after the ONE barrier
// per chunk and has the whole chunk of MFMAs to land.  A rows are gathered by the DMA itself (per-lane global address
// = the neighbour's feature row, or a zero line for a missing neighbour); since a DMA instruction fills 1 KiB of
// contiguous LDS, bank conflicts are avoided by swizzling instead of padding: the 16-B column c of row r is stored at
// column c ^ (r & 15) (A), and W rows with bit 2 of their index set swap their 32-column halves (B).
__device__ float4 g_conv_zero_line = {0.f, 0.f, 0.f, 0.f};
#ifdef DCL_CONV_STAMPS
// diagnostic build only (tools/conv_stamps.py): s_memrealtime (100 MHz) of workgroup phases, 8 stamps per workgroup
constexpr int kStampWgs = 16384;
__device__ unsigned long long g_conv_stamps[kStampWgs * 8];
__device__ unsigned long long g_conv_phase[kStampWgs * 8];     // wave 0's shader cycles in: DMA wait, barrier, issue, MFMA block; chunks
#define CONV_STAMP(i)                                                                                          \
  do {                                                                                                         \
    const int wg__ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                           \
    if (threadIdx.x == 0 && wg__ < kStampWgs) g_conv_stamps[wg__ * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define CONV_STAMP(i) do { } while (0)
#endif
typedef __attribute__((address_space(3))) void conv_lds_void_t;
__device__ __forceinline__ void conv_glds16(const void *gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_byte_addr)
               : "memory");
}
// N pieces whose LDS destinations are 1 KiB apart, ONE M0 value: the destination of piece i is the instruction's offset
// field (i * 1024), which the hardware adds to the GLOBAL address as well -- the caller's source pointer of piece i is
// pre-decremented by i * 1024 bytes.  One wave pays ~45 cycles per piece for this form against ~58 for an M0 write per
// piece (tools/ubench_glds.hip; four waves issuing at once: 83 against 119).
template <int N>
__device__ __forceinline__ void conv_glds16_group(const float *const *gsrc, unsigned lds_byte_addr) {
  static_assert(N == 1 || N == 2 || N == 4, "pieces per group");
  unsigned keep;
  if constexpr (N == 1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc[0]), "s"(lds_byte_addr) : "memory");
  else if constexpr (N == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc[0]), "v"(gsrc[1]), "s"(lds_byte_addr) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %2, off offset:1024\n\tglobal_load_lds_dwordx4 %3, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %4, off offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc[0]), "v"(gsrc[1]), "v"(gsrc[2]), "v"(gsrc[3]), "s"(lds_byte_addr) : "memory");
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// 16-B write-through (sc1) store: the payload of an in-launch hand-off (cdna_hip_programming.md Guideline 16, R1)
__device__ __forceinline__ void conv_store16_wt(f32x4 *p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned conv_lds_addr(const float *p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(conv_lds_void_t *)p);
}

// Tile shape: WR x WCW waves, each 32 rows x (32*NT) channels => BM = 32*WR rows, BN = 32*NT*WCW channels per workgroup,
// KC = 32 virtual channels per chunk.  What bounds this kernel is the L2 -> LDS operand traffic (2*BM*BN*KC flop per
// (BM+BN)*KC*4 bytes), not LDS or the MFMA pipe: 64x64 tiles (16 flop/B) saturated at ~45 % of the MFMA peak, hence
// 128x128 (Cout % 128 == 0, 8 waves, 32 flop/B) and 128x64 (4 waves) here.
// ORD: the launch carries a row order (sides.s[].ord; the two deep levels of a batch of 14 crops or more) -- a
// compile-time switch, because the natural-order instantiation must not pay registers for the order's bookkeeping (the
// 8-wave variants sit at the 128-VGPR limit of two workgroups per CU)
template <int CIN, int WR, int WCW, int NT, bool ORD>
__global__ __launch_bounds__(64 * WR * WCW, WR * WCW / 2) void k_sparse_conv_dma(   // 2 workgroups per CU (LDS allows 2)
    const DclConvSides sides, int nsides, int cout, int kvol, int subm, int relu, float *__restrict__ partial, int stream_k,
    int aligned_ns, int xcd_remap, int32_t *__restrict__ tile_counters, int use_bal_arg) {
  const int use_bal = ORD ? use_bal_arg : 0;
  constexpr int NW = WR * WCW, NTHR = 64 * NW;
  constexpr int BM = 32 * WR, BN = 32 * NT * WCW, KC = 32;
  constexpr int AT = BM * KC, BT = KC * BN, ST = AT + BT;      // floats per stage
  constexpr int A_INSTR = BM / 8;                              // 1-KiB DMA instructions per A tile (8 rows of 128 B each)
  constexpr int B_ROWS_PER = 256 / BN;                         // W rows per 1-KiB DMA instruction
  constexpr int B_INSTR = KC / B_ROWS_PER;
  static_assert(A_INSTR % NW == 0 && B_INSTR % NW == 0 && (BN == 32 || BN == 64 || BN == 128), "tile shape");
  constexpr int BSWZ = BN >= 64 ? 1 : 0;                       // W-row half swap (rows 32 floats wide have no halves to swap)
  extern __shared__ __attribute__((aligned(16))) float conv_lds[];   // [stage 0: A|B][stage 1: A|B][Ns 27*BM][kmask (4)][rows BM]
  int32_t *Ns = reinterpret_cast<int32_t *>(conv_lds + 2 * ST);
  unsigned *s_kmask = reinterpret_cast<unsigned *>(Ns + 27 * BM);
  int32_t *s_rows = reinterpret_cast<int32_t *>(s_kmask + 4);         // output row of every tile slot (-1 = none): the row order

  // the launch's problems ("sides": the observed / template backbone of the same layer; one for a plain call): their
  // live row counts and tile counts -- every workgroup needs both to find its place in the common unit sequence
  int n0 = sides.s[0].n_dev ? *sides.s[0].n_dev : sides.s[0].n_host;
  n0 = n0 < sides.s[0].cap ? n0 : sides.s[0].cap;
  int n1 = 0;
  if (nsides > 1) {
    n1 = sides.s[1].n_dev ? *sides.s[1].n_dev : sides.s[1].n_host;
    n1 = n1 < sides.s[1].cap ? n1 : sides.s[1].cap;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave / WCW, wc = wave % WCW;
  // Work decomposition ("stream-K").  The launch's work is the sequence of chunk units (tile 0: chunks 0..C-1, tile 1:
  // ..., tiles numbered column-tile-fastest); workgroup w owns the contiguous units [w*U, (w+1)*U) with U = ceil(total / G)
  // for the G workgroups of the launch, so every workgroup does the same amount of MFMA work whatever the number of
  // tiles (no partial last round on the 2 x 256 resident slots, no idle CUs when a deep layer has fewer tiles than
  // slots).  A workgroup's range covers at most two partial tiles (its first and its last segment) and whole tiles in
  // between; a partial segment is published to the workgroup's slot 0 / 1 and the tile's ticket is drawn -- the last
  // arriver adds the tile's segments in ascending chunk order (deterministic) and runs the epilogue.  stream_k == 0:
  // one whole tile per workgroup (grid-stride), no partials -- or, aligned_ns >= 2, classic split-K: workgroup w owns
  // segment w % ns of tile w / ns (chunks [seg*C/ns, (seg+1)*C/ns)), the choice when tiles * ns just fills the slots.
  // XCD-aware placement (speed only): workgroup ids are dealt round-robin over the 8 XCDs; renumber so that the
  // workgroups sharing an XCD (= one L2) hold neighbouring unit ranges (column tiles of a row tile, neighbouring row
  // tiles, whose gathered input rows overlap) -- not in capacity mode, where the live work occupies the low ids only
  int wid = blockIdx.x;
  const int G = gridDim.x;
  if (xcd_remap && sides.s[0].n_dev == nullptr) {
    const int xq = G >> 3, xr = G & 7, xcd = wid & 7;
    wid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (wid >> 3);
  }
  const int nblk0 = (n0 + BM - 1) / BM, nblk1 = (n1 + BM - 1) / BM, ncol = cout / BN;
  const int C = (kvol * CIN + KC - 1) / KC;                    // chunks per tile
  // (32-bit unit arithmetic: tiles * C < 2^31 for every launch the host code makes -- 64-bit divisions would cost
  // dozens of VGPRs in a kernel that sits at the 128-register limit)
  // Row order (row_order.hip): tile slot i computes output row ord.order[i] -- rows sorted by the shape of their
  // neighbourhood, so that a tile's rows use the same few kernel offsets.  use_bal (stream-K only, CIN >= 32): the units are
  // USED chunks -- ord.bal[0..nblk] is the prefix of the row tiles' used-step counts, ord.smask[] their step masks -- so a
  // row tile with few used offsets costs its workgroups proportionally less.  The unit sequence of a grouped launch is side
  // 0's units followed by side 1's (tiles never straddle the sides).
  constexpr int CPKH = CIN >= KC ? CIN / KC : 1;
  const int units0 = use_bal ? sides.s[0].ord.bal[nblk0] * ncol * CPKH : nblk0 * ncol * C;
  const int units1 = nsides > 1 ? (use_bal ? sides.s[1].ord.bal[nblk1] * ncol * CPKH : nblk1 * ncol * C) : 0;
  const int total = units0 + units1;
  const int nblk = nblk0 + nblk1;                              // row tiles of the launch (aligned split-K numbers them through)
  int U = C, u = wid * C, u_end = total;                       // stream_k == 0: tile wid, wid + G, ...
  if (aligned_ns) {
    const int tl = wid / aligned_ns, seg = wid - tl * aligned_ns;
    u = tl * C + seg * C / aligned_ns;
    u_end = tl < nblk * ncol ? tl * C + (seg + 1) * C / aligned_ns : u;
  } else if (stream_k) {
    U = (total + G - 1) / G;
    if (U < stream_k) U = stream_k;                            // few rows: at least this many chunks per workgroup
    u = wid * U;
    u_end = u + U < total ? u + U : total;
  }
  if (u > total) u = total;
  const float *zero = reinterpret_cast<const float *>(&g_conv_zero_line);

  while (u < u_end) {
    int tile, j_begin, nchunks, tile_lo, tile_hi, blk, by;
    bool whole;
    // which side this segment belongs to, and that side's problem (uniform: scalar loads of ONE side's descriptor)
    const int second = u >= units0 ? 1 : 0;
    const DclConvSide &S = sides.s[second];
    const float *__restrict__ feat = S.feat;
    const DclNbrSrc src = S.src;
    const int cap = S.cap;
    const float *__restrict__ W = S.W;
    const DclRowOrder ord = ORD ? S.ord : DclRowOrder{nullptr, nullptr, nullptr};
    const int32_t *__restrict__ bal = use_bal ? ord.bal : nullptr;
    const int n = second ? n1 : n0;
    const int ubase = second ? units0 : 0, tbase = second ? nblk0 * ncol : 0;       // first unit / tile of the side
    const int ul = u - ubase;                                                        // the side's own unit index
    const int ul_end = (u_end < ubase + (second ? units1 : units0) ? u_end : ubase + (second ? units1 : units0)) - ubase;
    if (bal) {
      const int nblk_s = second ? nblk1 : nblk0;
      int lo = 0, hi = nblk_s;                                 // largest row tile whose first unit is <= ul
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (bal[mid] * ncol * CPKH <= ul) lo = mid; else hi = mid;
      }
      blk = lo;
      const int cnt = (bal[blk + 1] - bal[blk]) * CPKH, base = bal[blk] * ncol * CPKH;
      by = (ul - base) / cnt;
      tile_lo = base + by * cnt;
      tile_hi = tile_lo + cnt;
      const int v0 = ul - tile_lo, v1 = cnt < v0 + (ul_end - ul) ? cnt : v0 + (ul_end - ul);
      const unsigned smk = ord.smask[blk];
      auto nominal = [&](int v) -> int {                       // used chunk v of the tile -> its nominal chunk index
        unsigned m = smk;
        for (int q = v / CPKH; q > 0; --q) m &= m - 1u;
        return __builtin_ctz(m) * CPKH + v % CPKH;
      };
      j_begin = nominal(v0);
      nchunks = nominal(v1 - 1) + 1;
      whole = v0 == 0 && v1 == cnt;
      tile = blk * ncol + by;
      u += v1 - v0;
    } else {
      tile = ul / C;
      j_begin = ul - tile * C;
      nchunks = (stream_k || aligned_ns) ? (C < j_begin + (ul_end - ul) ? C : j_begin + (ul_end - ul)) : C;   // end chunk of the segment
      whole = j_begin == 0 && nchunks == C;
      blk = tile / ncol;
      by = tile - blk * ncol;
      tile_lo = tile * C;
      tile_hi = tile_lo + C;
      u += (stream_k || aligned_ns) ? nchunks - j_begin : G * C;
    }
    tile_lo += ubase;                                          // slots / tickets are numbered over the whole launch
    tile_hi += ubase;
    tile += tbase;
    const int row0 = blk * BM, col0 = by * BN;
    CONV_STAMP(0);
#ifdef DCL_CONV_STAMPS
    {
      const int wg__ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      if (threadIdx.x == 0 && wg__ < kStampWgs)
        g_conv_stamps[wg__ * 8 + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |
                                      (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // XCC_ID, HW_ID
    }
#endif
    if (tid == 0) *s_kmask = 0;
    constexpr bool ordered = ORD;                          // (natural order: the slot -> row map is arithmetic, no LDS round trip)
    if (ordered)
      for (int rr = tid; rr < BM; rr += NTHR) s_rows[rr] = row0 + rr < n ? ord.order[row0 + rr] : -1;
    __syncthreads();
    // neighbour rows of the offsets this workgroup's chunk range touches (all 27 without split-K)
    unsigned mymask = 0;
    const int sx_lo = (j_begin * KC) / CIN;
    const int sx_hi = min(kvol - 1, (nchunks * KC - 1) / CIN);
#pragma unroll NW == 4 ? 8 : 4                             // several rounds of lookups in flight (2-3 dependent loads each)
    for (int e = tid; e < (sx_hi - sx_lo + 1) * BM; e += NTHR) {
      const int si = e / BM, rr = e - si * BM;
      const int k = offset_at(sx_lo + si, kvol, subm);
      const int orow = ordered ? s_rows[rr] : (row0 + rr < n ? row0 + rr : -1);
      const int v = orow >= 0 ? dcl_nbr_at(src, cap, k, orow) : -1;
      Ns[k * BM + rr] = v;
      mymask |= (v >= 0 ? 1u : 0u) << k;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mymask |= __shfl_xor(mymask, d, 64);
    if (lane == 0 && mymask) atomicOr(s_kmask, mymask);
    __syncthreads();
    const unsigned kmask = *s_kmask;
    // ---- chunk control, scalar-light.  Steps (= kernel offsets in visiting order, offset_at) own CPK = CIN/KC chunks
    // each, or a chunk spans SPC = KC/CIN steps (CIN = 16).  `smask` marks the steps whose offset has a neighbour in this
    // tile, `wsmask` those with one among this wave's 32 rows; the next used chunk is a find-first-set away.
    constexpr int CPK = CIN >= KC ? CIN / KC : 1, SPC = CIN >= KC ? 1 : KC / CIN;
    unsigned smask = 0;
    for (int sx = 0; sx < kvol; ++sx) smask |= ((kmask >> offset_at(sx, kvol, subm)) & 1u) << sx;
    auto next_used = [&](int from) -> int {            // smallest used chunk >= from (or nchunks)
      if (from >= nchunks) return nchunks;
      if constexpr (SPC == 1) {
        const int sx = from / CPK;
        if ((smask >> sx) & 1u) return from;
        const unsigned rest = sx + 1 < 32 ? (smask >> (sx + 1)) << (sx + 1) : 0u;
        if (!rest) return nchunks;
        const int q = __builtin_ctz(rest) * CPK;
        return q < nchunks ? q : nchunks;
      } else {
        for (int q = from; q < nchunks; ++q)
          if ((smask >> (q * SPC)) & ((1u << SPC) - 1u)) return q;
        return nchunks;
      }
    };

    // A row = 32 floats = 8 blocks of 16 B, block c of row q stored at c ^ ((q >> 1) & 7): 16 consecutive rows read with
    // ds_read_b128 then cover all 64 banks once.  W row kk with bit 2 set swaps its 32-float halves (the two lane
    // halves of an MFMA read rows 4 apart).
    // The issue of a chunk's operand DMAs sits between the barrier and the MFMA block of every wave, so it is kept
    // short: what depends only on the lane (tile row, swizzled 16-B piece, W row / column piece) is computed once per
    // tile; per chunk the neighbour rows of ALL of the wave's A pieces are read from LDS first (independent reads, one
    // wait) and only then the DMAs go out -- an asm statement with a memory clobber between two LDS reads would serialise
    // read -> wait -> DMA per piece.
    constexpr int A_PER = A_INSTR / NW, B_PER = B_INSTR / NW, LPR = BN / 4;
    // (the lane constants are recomputed per chunk -- a handful of VALU ops -- rather than kept in registers across the
    // MFMA block: the 8-wave 128x128 variant sits at the 128-VGPR limit of two workgroups per CU)
    auto lane_consts = [&](int (&a_row)[A_PER], int (&a_chs)[A_PER], int (&b_kk)[B_PER], int (&b_off)[B_PER]) {
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int g = wave * A_PER + i;
        a_row[i] = g * 8 + (lane >> 3);
        a_chs[i] = ((lane & 7) ^ ((a_row[i] >> 1) & 7)) << 2;
      }
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int g = wave * B_PER + i;
        b_kk[i] = g * B_ROWS_PER + lane / LPR;
        const int pcol = lane % LPR;
        const int bcol = col0 + ((BSWZ ? pcol ^ (((b_kk[i] >> 2) & 1) << 3) : pcol) << 2);
        b_off[i] = (CIN >= KC ? b_kk[i] : (b_kk[i] % CIN)) * cout + bcol;      // W row inside the chunk's offset, column piece
      }
    };
    // prep(j): the global source of every DMA piece of chunk j (registers), piece i of a group of GRP pieces pre-decremented
    // by i KiB (conv_glds16_group); fire_all(stage): the chunk's pieces go out, W first (their sources are ready first).
    constexpr int NPIECE = A_PER + B_PER;
    constexpr int AGRP = A_PER >= 4 ? 4 : A_PER, BGRP = B_PER >= 4 ? 4 : B_PER;
    static_assert(A_PER % AGRP == 0 && B_PER % BGRP == 0 && (AGRP == 1 || AGRP == 2 || AGRP == 4) && (BGRP == 1 || BGRP == 2 || BGRP == 4), "DMA groups");
    const float *psrc[NPIECE];
    auto prep = [&](int j) {
      const float **asrc = psrc, **bsrc = psrc + A_PER;
      int a_row[A_PER], a_chs[A_PER], b_kk[B_PER], b_off[B_PER];
      lane_consts(a_row, a_chs, b_kk, b_off);
      if constexpr (SPC == 1) {                          // the chunk lies inside ONE kernel offset: uniform k and channel base
        const int sx = j / CPK, chb = (j - sx * CPK) * KC;
        const int k = offset_at(sx, kvol, subm);
        const int32_t *nrow = Ns + k * BM;
        int v[A_PER];
#pragma unroll
        for (int i = 0; i < A_PER; ++i) v[i] = nrow[a_row[i]];
        const float *wbase = W + ((size_t)k * CIN + chb) * cout;
#pragma unroll
        for (int i = 0; i < B_PER; ++i) bsrc[i] = wbase + b_off[i] - (i % BGRP) * 256;
        const float *fbase = feat + chb;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) asrc[i] = (v[i] >= 0 ? fbase + ((size_t)v[i] * CIN + a_chs[i]) : zero) - (i % AGRP) * 256;
      } else {                                           // CIN = 16: the chunk spans two offsets, the piece decides which
        const int sx0 = j * SPC;
        const int k0 = offset_at(sx0, kvol, subm);
        const bool has1 = sx0 + 1 < kvol;
        const int k1 = has1 ? offset_at(sx0 + 1, kvol, subm) : k0;
        int v[A_PER];
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
          const bool second = a_chs[i] >= CIN;
          v[i] = Ns[(second ? k1 : k0) * BM + a_row[i]];
          if (second && !has1) v[i] = -1;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
          const bool second = b_kk[i] >= CIN;
          bsrc[i] = ((second && !has1) ? zero : W + (size_t)(second ? k1 : k0) * CIN * cout + b_off[i]) - (i % BGRP) * 256;
        }
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
          asrc[i] = (v[i] >= 0 ? feat + ((size_t)v[i] * CIN + (a_chs[i] & (CIN - 1))) : zero) - (i % AGRP) * 256;
      }
    };
    auto fire_all = [&](int stage) {
      float *As = conv_lds + stage * ST, *Bs = As + AT;
#pragma unroll
      for (int g = 0; g < B_PER; g += BGRP) conv_glds16_group<BGRP>(psrc + A_PER + g, conv_lds_addr(Bs + (wave * B_PER + g) * 256));
#pragma unroll
      for (int g = 0; g < A_PER; g += AGRP) conv_glds16_group<AGRP>(psrc + g, conv_lds_addr(As + (wave * A_PER + g) * 256));
    };
    // steps under which at least one of THIS WAVE's 32 rows has a neighbour (wave-level skip of a chunk's MFMA block)
    unsigned wsmask = 0;
    {
      // (the lane's row is made opaque here so that its LDS address is formed per segment: hoisted out of the tile loop
      // it was the one value the 8-wave variant spilled -- and a kernel with a scratch segment does not get its second
      // workgroup per CU at dispatch time)
      int wrow = wr * 32 + r;
      asm volatile("" : "+v"(wrow));
      for (int sx = 0; sx < kvol; ++sx)
        if ((smask >> sx) & 1u)
          wsmask |= (__ballot(Ns[offset_at(sx, kvol, subm) * BM + wrow] >= 0) != 0ull ? 1u : 0u) << sx;
    }
    wsmask = __builtin_amdgcn_readfirstlane(wsmask);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    int j = next_used(j_begin), cur = 0;
    CONV_STAMP(1);
    // Four-wave tiles have registers to spare (2 workgroups x 4 waves per CU = 2 waves per SIMD): the sources of the chunk
    // AFTER the next are prepared under the MFMA block (LDS reads of the neighbour table + address arithmetic), so that only
    // the DMA instructions themselves stand between the barrier and the MFMAs.  Eight-wave tiles prepare right before they
    // fire (128 x 128: at their 128 registers; 128 x 64: measured both ways, 80 / 70 us against 82 / 71 with the early prep).
    constexpr bool PIPE = NW == 4;
    if (j < nchunks) {
      prep(j);
      fire_all(0);
    }
    int jn = next_used(j + 1);
    if (PIPE && jn < nchunks) prep(jn);
    bool first_chunk = true;
#ifdef DCL_CONV_STAMPS
    unsigned long long ph_wait = 0, ph_bar = 0, ph_issue = 0, ph_mfma = 0, ph_t = __builtin_amdgcn_s_memtime(), ph_n = 0;
#define PH(acc) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); acc += t__ - ph_t; ph_t = t__; } while (0)
#else
#define PH(acc) do { } while (0)
#endif
    while (j < nchunks) {
      __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's DMA pieces of chunk j have landed
      PH(ph_wait);
      __syncthreads();                                   // ... everyone's have, and stage cur^1 has no reader left
      PH(ph_bar);
      if (first_chunk) { CONV_STAMP(2); first_chunk = false; }
      if (jn < nchunks) {                                // (spreading the pieces over the MFMA groups was measured: the wave
        if (!PIPE) prep(jn);                             //  pays the same per piece there, nothing is hidden)
        fire_all(cur ^ 1);
      }
      const int jn2 = next_used(jn + 1);
      if (PIPE && jn2 < nchunks) prep(jn2);
      PH(ph_issue);
      // wave-level skip: none of this wave's 32 rows has a neighbour under any offset of the chunk
      const bool mine = SPC == 1 ? ((wsmask >> (j / CPK)) & 1u) != 0 : ((wsmask >> (j * SPC)) & ((1u << SPC) - 1u)) != 0;
      if (mine) {
        const float *arow = conv_lds + cur * ST + (wr * 32 + r) * KC;
        const float *bcol = nullptr;
        const int sw = (r >> 1) & 7;
        // B operand of column tile t: logical column wc*32*NT + 32*t + r; the XOR with 32*h commutes with + 32*t only
        // through the XOR itself, so the tile offset is applied as an XOR too (32*t has no bits below 32)
#pragma unroll
        for (int i = 0; i < KC / 8; ++i) {
          const float4 a = *reinterpret_cast<const float4 *>(arow + (((2 * i + h) ^ sw) << 2));
          const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const float bv = conv_lds[cur * ST + AT + (8 * i + 4 * h + q) * BN + ((wc * 32 * NT + 32 * t + r) ^ (BSWZ * 32 * h))];
              acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv, acc[t], 0, 0, 0);
            }
        }
        (void)bcol;
      }
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");   // MFMA -> VALU hazard pad (hipcc 7.2 omitted the wait states when the accumulator is re-read behind a barrier: stale acc[15])
      PH(ph_mfma);
#ifdef DCL_CONV_STAMPS
      ++ph_n;
#endif
      j = jn;
      jn = jn2;
      cur ^= 1;
    }
#ifdef DCL_CONV_STAMPS
    {
      const int wg__ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      if (threadIdx.x == 0 && wg__ < kStampWgs) {
        g_conv_phase[wg__ * 8 + 0] = ph_wait; g_conv_phase[wg__ * 8 + 1] = ph_bar; g_conv_phase[wg__ * 8 + 2] = ph_issue;
        g_conv_phase[wg__ * 8 + 3] = ph_mfma; g_conv_phase[wg__ * 8 + 4] = ph_n;
      }
    }
#endif
    CONV_STAMP(3);
    if (!whole) {
      // ---- in-launch combine (last arriver).  Publish: write-through (sc1) stores of this split's partial tile, every
      // storing wave drains them, workgroup barrier, ONE lane takes the tile's ticket.  The workgroup that draws the last
      // ticket acquires (agent scope), re-reads ALL partials with plain loads and adds them in split order -- the same
      // order, hence the same bits, as k_conv_split_reduce -- then the epilogue.  The counter is left at zero.
      // partial tiles live in FRAGMENT order -- [split][tile][wave][t][e/4][lane] float4, i.e. every lane stores and later
      // re-reads its own accumulator registers as 16-B pieces, 1 KiB contiguous per wave instruction -- whole rows of the
      // tile, padding rows included (the scratch is sized for row tiles, not rows)
      const size_t tile_f4 = (size_t)NW * NT * 4 * 64;
      // slot of a workgroup's segment of `tile`: 2*w if the tile holds w's first unit, else 2*w + 1
      // (w * U >= tile * C  <=>  the tile holds w's first unit, for the workgroups w that touch the tile at all)
      auto slot_of = [&](int w) -> size_t { return (size_t)(2 * w + ((aligned_ns || w * U >= tile_lo) ? 0 : 1)); };
      f32x4 *mine = reinterpret_cast<f32x4 *>(partial) + slot_of(wid) * tile_f4 + (size_t)wave * NT * 4 * 64 + lane;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
          v.x = acc[t][4 * q]; v.y = acc[t][4 * q + 1]; v.z = acc[t][4 * q + 2]; v.w = acc[t][4 * q + 3];
          if (tile_counters) conv_store16_wt(mine + (t * 4 + q) * 64, v);
          else mine[(t * 4 + q) * 64] = v;                  // deferred combine (k_conv_frag_reduce, next launch): plain stores
        }
      if (tile_counters == nullptr) {                      // few-row launches: the combine is a launch of its own
        __syncthreads();
        continue;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      CONV_STAMP(4);
      const int w_first = aligned_ns ? tile * aligned_ns : tile_lo / U;
      const int w_last = aligned_ns ? w_first + aligned_ns - 1 : (tile_hi - 1) / U;
      if (tid == 0) {
        int32_t *ctr = tile_counters + tile;
        const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = old == w_last - w_first ? 1u : 0u;
        if (last) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_kmask[1] = last;
      }
      __syncthreads();
      CONV_STAMP(5);
      if (s_kmask[1]) {
        // split-major: the NT*4 pieces of one split are independent loads in flight together; per element the sum is
        // P_0 + P_1 + ... in split order
        const f32x4 *base = reinterpret_cast<const f32x4 *>(partial) + (size_t)wave * NT * 4 * 64 + lane;
        const int nseg = w_last - w_first + 1;
        // ZU segments' pieces of ONE column tile are in flight together (few-row launches have up to 27 segments per tile:
        // one segment per load latency would make the combine the longest phase of the launch; both column tiles at once
        // would not fit the 128 registers of two workgroups per CU); the adds stay in segment order
        constexpr int ZU = NT == 2 ? 3 : 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll 1
          for (int z0 = 0; z0 < nseg; z0 += ZU) {
            asm volatile("" ::: "memory");                                   // keep the next block's loads behind this point
            f32x4 v[ZU][4];
#pragma unroll
            for (int uu = 0; uu < ZU; ++uu) {
              const int zc = z0 + uu < nseg ? z0 + uu : nseg - 1;            // clamped: loaded, not added
              const f32x4 *pz = base + slot_of(w_first + zc) * tile_f4 + t * 4 * 64;
#pragma unroll
              for (int q = 0; q < 4; ++q) v[uu][q] = pz[q * 64];
            }
#pragma unroll
            for (int uu = 0; uu < ZU; ++uu) {
              if (z0 + uu >= nseg) break;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 w = v[uu][q];
                if (z0 + uu == 0) {
                  acc[t][4 * q] = w.x; acc[t][4 * q + 1] = w.y; acc[t][4 * q + 2] = w.z; acc[t][4 * q + 3] = w.w;
                } else {
                  acc[t][4 * q] = acc[t][4 * q] + w.x; acc[t][4 * q + 1] = acc[t][4 * q + 1] + w.y;
                  acc[t][4 * q + 2] = acc[t][4 * q + 2] + w.z; acc[t][4 * q + 3] = acc[t][4 * q + 3] + w.w;
                }
              }
            }
          }
        }
        // (the epilogue's pointers are fetched from the side's descriptor HERE, through an opaque index, so that they are
        // not kept in scalar registers across the chunk loop: the kernel runs at the SGPR limit)
        const int sec_e = __builtin_amdgcn_readfirstlane(second);
        const float *__restrict__ scale = sides.s[sec_e].scale;
        const float *__restrict__ shift = sides.s[sec_e].shift;
        float *__restrict__ out = sides.s[sec_e].out;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int co = col0 + wc * 32 * NT + 32 * t + r;
          const float sc = scale ? scale[co] : 1.0f;
          const float sh = scale ? shift[co] : 0.0f;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int slot = wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int orow = ordered ? s_rows[slot] : (row0 + slot < n ? row0 + slot : -1);
            if (orow >= 0) {
              float x = acc[t][e];
              if (scale) x = x * sc + sh;
              if (relu) x = fmaxf(x, 0.0f);
              out[(size_t)orow * cout + co] = x;
            }
          }
        }
      }
      __syncthreads();
      CONV_STAMP(6);
      continue;
    }
    const int sec_e = __builtin_amdgcn_readfirstlane(second);
    const float *__restrict__ scale = sides.s[sec_e].scale;
    const float *__restrict__ shift = sides.s[sec_e].shift;
    float *__restrict__ out = sides.s[sec_e].out;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int co = col0 + wc * 32 * NT + 32 * t + r;
      const float sc = scale ? scale[co] : 1.0f;
      const float sh = scale ? shift[co] : 0.0f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int slot = wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int orow = ordered ? s_rows[slot] : (row0 + slot < n ? row0 + slot : -1);
        if (orow >= 0) {
          float x = acc[t][e];
          if (scale) x = x * sc + sh;
          if (relu) x = fmaxf(x, 0.0f);
          out[(size_t)orow * cout + co] = x;
        }
      }
    }
    __syncthreads();
  }
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

// Deferred combine of a stream-K launch (few-row launches: a tile has up to 27 segments, which the last arriver would have
// to add in as many dependent rounds of loads -- here every thread owns one 16-B piece of a tile and has all of its
// segments' loads in flight at once).  Same unit arithmetic as k_sparse_conv_dma; tiles owned by ONE workgroup were
// written by it directly and are skipped.  grid = (tiles_cap, NW*NT*4*64/256), 256 threads.
template <int WR, int WCW, int NT>
__global__ __launch_bounds__(256) void k_conv_frag_reduce(const float *__restrict__ partial, const DclConvSides sides, int nsides,
                                                          int cout, int C, int G, int min_u, int relu) {
  constexpr int NW = WR * WCW, BM = 32 * WR, BN = 32 * NT * WCW;
  int n0 = sides.s[0].n_dev ? *sides.s[0].n_dev : sides.s[0].n_host;
  n0 = n0 < sides.s[0].cap ? n0 : sides.s[0].cap;
  int n1 = 0;
  if (nsides > 1) {
    n1 = sides.s[1].n_dev ? *sides.s[1].n_dev : sides.s[1].n_host;
    n1 = n1 < sides.s[1].cap ? n1 : sides.s[1].cap;
  }
  const int ncol = cout / BN;
  const int tiles0 = (n0 + BM - 1) / BM * ncol, tiles1 = (n1 + BM - 1) / BM * ncol;
  const int tile = blockIdx.x;                                               // numbered over the whole launch: side 0's, then side 1's
  if (tile >= tiles0 + tiles1) return;
  const int total = (tiles0 + tiles1) * C;
  int U = (total + G - 1) / G;
  if (U < min_u) U = min_u;
  const int w_first = (tile * C) / U, w_last = ((tile + 1) * C - 1) / U;
  if (w_first == w_last) return;                                             // one owner: written by the conv kernel
  const int piece = blockIdx.y * 256 + threadIdx.x;                          // [wave][t][q][lane] inside the tile
  const int lane = piece & 63, q = (piece >> 6) & 3, t = (piece >> 8) % NT, wave = piece / (256 * NT);
  const size_t tile_f4 = (size_t)NW * NT * 4 * 64;
  const f32x4 *base = reinterpret_cast<const f32x4 *>(partial) + piece;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  constexpr int ZR = 9;                                                      // segments in flight per round (27 = 3 rounds)
#pragma unroll 1
  for (int w0 = w_first; w0 <= w_last; w0 += ZR) {
    f32x4 v[ZR];
#pragma unroll
    for (int i = 0; i < ZR; ++i) {
      const int w = w0 + i <= w_last ? w0 + i : w_last;                      // clamped: loaded, not added
      v[i] = base[(size_t)(2 * w + (w * U >= tile * C ? 0 : 1)) * tile_f4];
    }
#pragma unroll
    for (int i = 0; i < ZR; ++i) {
      const bool first = w0 + i == w_first, live = w0 + i <= w_last;
      const f32x4 sum = {a.x + v[i].x, a.y + v[i].y, a.z + v[i].z, a.w + v[i].w};
      a = first ? v[i] : (live ? sum : a);
    }
  }
  const int second = tile >= tiles0 ? 1 : 0;
  const DclConvSide &S = sides.s[second];
  const int n = second ? n1 : n0, lt = tile - (second ? tiles0 : 0);
  const int blk = lt / ncol, by = lt - blk * ncol;
  const int r = lane & 31, h = lane >> 5, wr = wave / WCW, wc = wave % WCW;
  const int co = by * BN + wc * 32 * NT + 32 * t + r;
  const float sc = S.scale ? S.scale[co] : 1.0f, sh = S.scale ? S.shift[co] : 0.0f;
  const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int orow = blk * BM + wr * 32 + c + 8 * q + 4 * h;                 // accumulator element e = 4 q + c
    if (orow < n) {
      float x = av[c];
      if (S.scale) x = x * sc + sh;
      if (relu) x = fmaxf(x, 0.0f);
      S.out[(size_t)orow * cout + co] = x;
    }
  }
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
DCL_HOOK_INT(kConvFewRowsHint, 6144);
static bool conv_launch_is_few(const DclConvSides &sides, int nsides) {
  const bool capacity_mode = sides.s[0].n_dev != nullptr;
  int rows = 0;
  bool hinted = capacity_mode;
  for (int i = 0; i < nsides; ++i) {
    const DclConvSide &S = sides.s[i];
    const int r_i = S.n_dev ? (S.n_host > 0 ? S.n_host : S.cap) : S.n_host;
    hinted = hinted && S.n_host > 0;
    rows = r_i > rows ? r_i : rows;
  }
  return hinted ? rows <= kConvFewRowsHint : conv_few_rows(rows, capacity_mode);
}
// A/B and tuning switches: process-wide atomics set through dcl_debug_* in the DIAGNOSTIC library (-DDCL_DIAG, tests/_diag/),
// compile-time constants in the product library -- the product has no hooks, no superseded kernels and no getenv
DCL_HOOK_INT(g_conv_xcd_remap, 1);   // 0 = plain blockIdx order
DCL_HOOK_INT(g_conv_slots, 512);     // workgroups a launch is dealt over (2 x 256 resident slots)
DCL_HOOK_INT(g_conv_wlds, 1);        // 1 = Cin 16 / 32 -> 32 layers with many rows on the filter-resident kernel, 0 = LDS-DMA kernel
DCL_HOOK_INT(g_conv_few_tiles, 1);   // 1 = few-row launches on 64-row tiles, 0 = 128-row tiles for every launch
DCL_HOOK_INT(g_conv_few_chunks, 4);  // chunks per workgroup (at least) of a few-row launch
DCL_HOOK_INT(g_conv_split, 0);       // 0 = automatic, n = force n-way split-K when scratch allows, -1 = never more than kConvMaxSplit, -2 = never split, -3 = few-row combine inside the launch
#ifdef DCL_DIAG
static std::atomic<int> g_conv_order_mode{0};   // A/B: 0 = as given, 1 = ignore the row order (natural rows), 2 = order but nominal chunk units
#endif

template <int CIN, int WR, int WCW, int NT>
static void launch_conv_dma(const DclConvSides &sides, int nsides, int cout, int kvol, int subm, int relu, float *scratch,
                            long long scratch_floats, int counters_ready, hipStream_t s) {
  constexpr int BM = 32 * WR, BN = 32 * NT * WCW, KC = 32;
  const size_t lds = (size_t)(2 * (BM * KC + KC * BN) + 27 * BM + 4 + BM) * sizeof(float);
  (void)hipFuncSetAttribute((const void *)k_sparse_conv_dma<CIN, WR, WCW, NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  (void)hipFuncSetAttribute((const void *)k_sparse_conv_dma<CIN, WR, WCW, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  const bool capacity_mode = sides.s[0].n_dev != nullptr;
  int tiles = 0;
  for (int i = 0; i < nsides; ++i) {
    const int r_i = sides.s[i].n_dev ? sides.s[i].cap : sides.s[i].n_host;
    tiles += dcl_div_up(r_i, BM) * (cout / BN);
  }
#ifdef DCL_CONV_STAMPS
  {
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_sparse_conv_dma<CIN, WR, WCW, NT, true>, 64 * WR * WCW, lds);
    fprintf(stderr, "k_sparse_conv_dma<%d,%d,%d,%d>: dynamic LDS %zu B, occupancy API says %d workgroups per CU\n", CIN, WR, WCW, NT, lds, occ);
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, (const void *)k_sparse_conv_dma<CIN, WR, WCW, NT, true>) == hipSuccess)
      fprintf(stderr, "  numRegs %d sharedSizeBytes %zu maxThreadsPerBlock %d maxDynamicSharedSizeBytes %d\n", fa.numRegs,
              fa.sharedSizeBytes, fa.maxThreadsPerBlock, fa.maxDynamicSharedSizeBytes);
    for (size_t l : {(size_t)16384, (size_t)32768, (size_t)49152, (size_t)65536, (size_t)73728, (size_t)77824, (size_t)79376, (size_t)81920}) {
      for (int thr : {256, 512}) {
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_sparse_conv_dma<CIN, WR, WCW, NT, true>, thr, l);
        fprintf(stderr, "  lds %zu threads %d -> %d;", l, thr, occ);
      }
      fprintf(stderr, "\n");
    }
  }
#endif
  // With scratch (partial-tile slots + tickets) a launch can be decomposed three ways; the cheapest by a small cost model
  // (chunk units on the critical path of a CU slot, fixed cost f per segment: neighbour table, first operand fetch,
  // publish) is taken:
  //   whole tiles      rounds(tiles) * (C + f)                    -- enough tiles, or just under a multiple of the slots
  //   aligned split-K  rounds(tiles * ns) * (C / ns + f + 1)      -- tiles * ns just fills the 2 x 256 resident slots
  //   stream-K         ceil(units / 512) + 2 f + 1                -- everything else (no partial rounds, no idle slots)
  // A handful of crops (one-image calls) is latency-bound on the chunk loop: stream-K with kFewChunks chunks per workgroup.
  const int nchunks = dcl_div_up(kvol * CIN, KC);
  constexpr int kFix = 4;
  const int kFewChunks = g_conv_few_chunks;
  const int kSlots = g_conv_slots;
  const long long units = (long long)tiles * nchunks;
  int stream_k = 0, aligned_ns = 0, G = tiles < 65535 * 16 ? tiles : 65535 * 16;
  bool deferred = false;
  int32_t *counters = nullptr;
  float *partial = scratch;
  const bool never = g_conv_split == -2;
  if (scratch && scratch_floats > kConvCounterWords && tiles <= kConvCounterWords && !never) {
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
      partial = scratch + kConvCounterWords;
      deferred = few && stream_k == kFewChunks && g_conv_split != -3;     // few rows: many segments per tile, combine = own launch (-3: A/B, in the launch)
      if (!deferred) {
        counters = reinterpret_cast<int32_t *>(scratch);
        if (!counters_ready) dcl_internal_zero_words(counters, kConvCounterWords, s);
      }
    }
  }
  // Row order (row_order.hip).  The order itself works with every decomposition whose combine runs inside the launch (the
  // deferred combine of few-row launches maps tile slots to rows on its own: such launches take no order).  Used-chunk
  // dealing replaces the nominal units whenever the launch would split tiles anyway (aligned split-K / stream-K);
  // launches whose tiles fit one round of whole tiles keep them (measured: forced stream-K only adds the combine there).
  DclConvSides sd = sides;
  bool have_order = true, have_bal = true;
  for (int i = 0; i < nsides; ++i) {
    have_order = have_order && sd.s[i].ord.order != nullptr;
    have_bal = have_bal && sd.s[i].ord.bal != nullptr && sd.s[i].ord.smask != nullptr;
  }
  int use_bal = 0;
  if (have_order && !deferred && BM == 128) {
    if (have_bal && CIN >= 32 && (aligned_ns || stream_k) && scratch && scratch_floats > kConvCounterWords) {
      use_bal = 1;
      stream_k = 1;
      aligned_ns = 0;
      const long long slots_fit = (scratch_floats - kConvCounterWords) / ((long long)2 * BM * BN);
      G = (int)(kSlots < slots_fit ? kSlots : slots_fit);
      partial = scratch + kConvCounterWords;
      counters = reinterpret_cast<int32_t *>(scratch);            // (already zeroed above: aligned / stream-K launches own them)
    }
  } else {
    for (int i = 0; i < nsides; ++i) sd.s[i].ord = DclRowOrder{nullptr, nullptr, nullptr};
  }
#ifdef DCL_DIAG
  if (g_conv_order_mode == 1) {                                                                // A/B: natural row order
    for (int i = 0; i < nsides; ++i) sd.s[i].ord = DclRowOrder{nullptr, nullptr, nullptr};
    use_bal = 0;
  }
  if (g_conv_order_mode == 2) use_bal = 0;                                                     // A/B: order, nominal units
#endif
  bool any_order = false;
  for (int i = 0; i < nsides; ++i) any_order = any_order || sd.s[i].ord.order != nullptr;
  if (any_order)
    hipLaunchKernelGGL((k_sparse_conv_dma<CIN, WR, WCW, NT, true>), dim3(G), dim3(64 * WR * WCW), lds, s, sd, nsides, cout, kvol,
                       subm, relu, partial, stream_k, aligned_ns, (int)g_conv_xcd_remap, counters, use_bal);
  else
    hipLaunchKernelGGL((k_sparse_conv_dma<CIN, WR, WCW, NT, false>), dim3(G), dim3(64 * WR * WCW), lds, s, sd, nsides, cout, kvol,
                       subm, relu, partial, stream_k, aligned_ns, (int)g_conv_xcd_remap, counters, 0);
  if (deferred)
    hipLaunchKernelGGL((k_conv_frag_reduce<WR, WCW, NT>), dim3(tiles, WR * WCW * NT), dim3(256), 0, s, partial, sd, nsides, cout,
                       nchunks, G, stream_k, relu);
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

// ---- sparse average pool ------------------------------------------------------------------------
// thread = (output row, 4 channels): rf = #valid offsets (summaryRF.cu:39), then
// out = ((0 + f_k0/rf) + f_k1/rf) + ... in ascending offset order (avgpool.cu:130).
__global__ __launch_bounds__(256) void k_sparse_avgpool(const DclConvSides sides, int nsides, int c, int kvol,
                                                        int32_t *__restrict__ rf_out, const int32_t *__restrict__ rf_in) {
  // the c/4 threads of an output row share its 27 neighbour rows through LDS (one lookup per (row, offset) per block).
  // Up to two problems per launch (the two backbones' pools of a level): the row blocks of side 0, then those of side 1;
  // rf_out / rf_in (the op-level API) belong to side 0 of a one-sided launch.
  __shared__ int32_t s_v[64 * 27];
  int n0 = sides.s[0].n_dev ? *sides.s[0].n_dev : sides.s[0].n_host;
  n0 = n0 < sides.s[0].cap ? n0 : sides.s[0].cap;
  int n1 = 0;
  if (nsides > 1) {
    n1 = sides.s[1].n_dev ? *sides.s[1].n_dev : sides.s[1].n_host;
    n1 = n1 < sides.s[1].cap ? n1 : sides.s[1].cap;
  }
  const int c4 = c >> 2;                                   // 4..64 and a divisor of 256 (checked by the launcher)
  const int rpb = 256 / c4;                                // output rows per block step
  const int tid = threadIdx.x;
  const int rr = tid / c4, q = tid - rr * c4;
  const int nb0 = (n0 + rpb - 1) / rpb, nb1 = (n1 + rpb - 1) / rpb;
  for (int bi = blockIdx.x; bi < nb0 + nb1; bi += gridDim.x) {
    const int second = bi >= nb0 ? 1 : 0;
    const DclConvSide &S = sides.s[second];
    const float *__restrict__ feat = S.feat;
    float *__restrict__ out = S.out;
    const int n = second ? n1 : n0, cap = S.cap;
    const int row0 = (bi - (second ? nb0 : 0)) * rpb;
    __syncthreads();
#pragma unroll 4                                           // the lookups of up to 4 rounds in flight together (2 dependent loads each)
    for (int e = tid; e < rpb * kvol; e += 256) {
      const int r2 = e / kvol, k = e - r2 * kvol;
      s_v[r2 * 27 + k] = row0 + r2 < n ? dcl_nbr_at(S.src, cap, k, row0 + r2) : -1;
    }
    __syncthreads();
    const int row = row0 + rr;
    if (row >= n) continue;
    const int32_t *v = s_v + rr * 27;
    int rf = 0;
    for (int k = 0; k < kvol; ++k) rf += v[k] >= 0;
    if (rf_in) rf = rf_in[row];                            // caller's summaryrf (indice_avgpool_fp32's 5th argument)
    const float d = (float)rf;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // 14 neighbour rows in flight per round (a missing neighbour loads row 0 and is not added): the loads of the whole
    // window used to be 27 dependent steps -- a branch on the LDS value in front of each -- which is what a pool of a few
    // hundred rows (one-image calls) spent its time on.  The terms are still added in ascending offset order.  (Rounds of 9
    // and of 27 were measured too: 24 / 21 / 20 / 12 us for the four pools of 32 crops either way -- 27 in flight cost the
    // occupancy that 9 lacked in depth -- against 19 / 17 / 16 / 12 with two rounds of 14.  The kernel is bound by its chain of
    // dependent loads, not by the 108 divisions per thread: a three-instruction exact division changed nothing.)
    constexpr int PF = 14;
    for (int k0 = 0; k0 < kvol; k0 += PF) {
      float4 f[PF];
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const int vk = k0 + j < kvol ? v[k0 + j] : -1;
        f[j] = reinterpret_cast<const float4 *>(feat + (size_t)(vk < 0 ? 0 : vk) * c)[q];
      }
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const bool ok = k0 + j < kvol && v[k0 + j] >= 0;
        acc.x = ok ? acc.x + f[j].x / d : acc.x; acc.y = ok ? acc.y + f[j].y / d : acc.y;
        acc.z = ok ? acc.z + f[j].z / d : acc.z; acc.w = ok ? acc.w + f[j].w / d : acc.w;
      }
    }
    reinterpret_cast<float4 *>(out + (size_t)row * c)[q] = acc;
    if (rf_out && q == 0) rf_out[row] = rf;
  }
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
extern "C" __attribute__((visibility("default"))) int dcl_debug_conv_stamps(unsigned long long *host, int n_wg, int clear) {
  if (clear) {
    static unsigned long long zeros[kStampWgs * 8];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), zeros, sizeof(zeros));
  }
  if (n_wg > kStampWgs) n_wg = kStampWgs;
  if (clear == 0 && n_wg < 0)
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_phase), sizeof(unsigned long long) * 8 * (-n_wg));
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_conv_stamps), sizeof(unsigned long long) * 8 * n_wg);
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
};
ConvProfile g_conv_prof;
}  // namespace

DCL_API int dcl_profile_conv_begin(void) {
  std::lock_guard<std::mutex> lock(g_conv_prof.mu);
  for (auto &e : g_conv_prof.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  g_conv_prof.ev.clear();
  g_conv_prof.on = true;
  return 0;
}

DCL_API int dcl_profile_conv_end(double *ms_total_host, int32_t *calls_host) {
  std::lock_guard<std::mutex> lock(g_conv_prof.mu);
  g_conv_prof.on = false;
  double total = 0.0;
  for (auto &e : g_conv_prof.ev) {
    float ms = 0.f;
    if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) total += ms;
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  if (ms_total_host) *ms_total_host = total;
  if (calls_host) *calls_host = (int32_t)g_conv_prof.ev.size();
  g_conv_prof.ev.clear();
  return 0;
}

static int conv_dispatch(const DclConvSides &sides, int nsides, int cin, int cout, int kvol, int subm, int relu, float *scratch,
                         int64_t scratch_floats, int counters_ready, dclStream_t stream);

// library-internal: one layer of up to two problems ("sides") in one launch; `src` may be an implicit rulebook (native
// backbone runner).  Timed as ONE conv call by the measurement facility above.
int dcl_internal_sparse_conv_fwd_sides(const DclConvSides &sides, int nsides, int cin, int cout, int kvol, int subm, int relu,
                                       float *scratch, int64_t scratch_floats, dclStream_t stream, int counters_ready) {
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
  const int rc = conv_dispatch(sides, nsides, cin, cout, kvol, subm, relu, scratch, scratch_floats, counters_ready, stream);
  if (timed) {
    (void)hipEventRecord(e1, (hipStream_t)stream);
    std::lock_guard<std::mutex> lock(g_conv_prof.mu);
    g_conv_prof.ev.emplace_back(e0, e1);
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

static int conv_dispatch(const DclConvSides &sides_in, int nsides_in, int cin, int cout, int kvol, int subm, int relu,
                         float *scratch, int64_t scratch_floats, int counters_ready, dclStream_t stream) {
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
    const bool few_tiles = g_conv_few_tiles != 0 && scratch && cout % 64 == 0 && is_few;
    // wide, shallow layers (Cin 16 / 32 -> 32 channels, many rows): the filter-resident kernel, no staging, no barriers
    // (Measured and dropped: the middle layers (Cin 32 / 64, filter too big for LDS) with the rows in registers and one
    // offset's filter slice staged per step, workgroups of 2 / 4 waves in lock step over the 27 offsets, no split-K -- 99 /
    // 105 / 161 us for the 32->64 / 64->64 / 64->128 layers against 87 / 79 / 142 of the LDS-DMA kernel: a barrier and a
    // drained vmcnt per offset cost more than the row staging they replace.)
    // (the 32 -> 64 layer as two 32-column halves -- instantiated, measured, not used: 99 us against the DMA kernel's 86)
    // Two sides in one call (one-stream schedule): the 16-channel layer goes out as a launch per side (44 us each against 105
    // for the grouped LDS-DMA launch), the 32-channel one keeps the grouped LDS-DMA launch (72 us against 2 x 48).
    if (g_conv_wlds != 0 && ((cout == 32 && (cin == 16 || (cin == 32 && nsides == 1))) || (g_conv_wlds == 2 && cout == 64 && cin == 32)) &&
        kvol == 27 && !is_few) {
      const size_t lds = (size_t)27 * cin * 32 * sizeof(float);
      const dim3 grid(256 / (cout / 32), cout / 32), block(cin == 16 ? 1024 : 512);
#define WLDS_LAUNCH(CI, CO, SB)                                                                                              \
      do {                                                                                                                   \
        (void)hipFuncSetAttribute((const void *)k_sparse_conv_wlds<CI, CO, SB>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                  (int)lds);                                                                                 \
        hipLaunchKernelGGL((k_sparse_conv_wlds<CI, CO, SB>), grid, block, lds, s, sides, nsides, relu);                      \
      } while (0)
      // (a launch per side: one workgroup per CU is all the filter leaves room for, so two sides in one launch only halve
      // each side's CUs -- measured 101 us for both against 2 x 44)
      const DclConvSides both = sides;
      const int nboth = nsides;
      for (int side_i = 0; side_i < nboth; ++side_i) {
        DclConvSides sides{};
        sides.s[0] = both.s[side_i];
        const int nsides = 1;
        if (cin == 16) { if (subm) WLDS_LAUNCH(16, 32, true); else WLDS_LAUNCH(16, 32, false); }
        else if (cout == 32) { if (subm) WLDS_LAUNCH(32, 32, true); else WLDS_LAUNCH(32, 32, false); }
        else { if (subm) WLDS_LAUNCH(32, 64, true); else WLDS_LAUNCH(32, 64, false); }
      }
#undef WLDS_LAUNCH
      DCL_LAUNCH_CHECK();
      return 0;
    }
#define DMA_ARGS sides, nsides, cout, kvol, subm, relu, scratch, (long long)scratch_floats, counters_ready, s
#ifdef DCL_DIAG
    if (g_force_valu == 5 && cout % 128 != 0 && cout % 64 == 0) {          // A/B: the former 4 waves of 32x64 on 128x64 tiles
      switch (cin) {
        case 16: launch_conv_dma<16, 4, 1, 2>(DMA_ARGS); break;
        case 32: launch_conv_dma<32, 4, 1, 2>(DMA_ARGS); break;
        case 64: launch_conv_dma<64, 4, 1, 2>(DMA_ARGS); break;
        default: launch_conv_dma<128, 4, 1, 2>(DMA_ARGS); break;
      }
    } else
#endif
    if (few_tiles) {                                                        // few rows: 64-row tiles (see above)
      if (cout % 128 == 0) {
        switch (cin) {
          case 16: launch_conv_dma<16, 2, 2, 2>(DMA_ARGS); break;
          case 32: launch_conv_dma<32, 2, 2, 2>(DMA_ARGS); break;
          case 64: launch_conv_dma<64, 2, 2, 2>(DMA_ARGS); break;
          default: launch_conv_dma<128, 2, 2, 2>(DMA_ARGS); break;
        }
      } else {                                                              // 64x64 tiles, 4 waves of 32x32
        switch (cin) {
          case 16: launch_conv_dma<16, 2, 2, 1>(DMA_ARGS); break;
          case 32: launch_conv_dma<32, 2, 2, 1>(DMA_ARGS); break;
          case 64: launch_conv_dma<64, 2, 2, 1>(DMA_ARGS); break;
          default: launch_conv_dma<128, 2, 2, 1>(DMA_ARGS); break;
        }
      }
    } else if (cout % 64 != 0) {                                            // Cout = 32: LDS-DMA kernel on 128x32 tiles
      switch (cin) {
        case 16: launch_conv_dma<16, 4, 1, 1>(DMA_ARGS); break;
        case 32: launch_conv_dma<32, 4, 1, 1>(DMA_ARGS); break;
        case 64: launch_conv_dma<64, 4, 1, 1>(DMA_ARGS); break;
        default: launch_conv_dma<128, 4, 1, 1>(DMA_ARGS); break;
      }
    } else if (cout % 128 == 0) {                                           // 128x128 tiles, 8 waves
      switch (cin) {
        case 16: launch_conv_dma<16, 4, 2, 2>(DMA_ARGS); break;
        case 32: launch_conv_dma<32, 4, 2, 2>(DMA_ARGS); break;
        case 64: launch_conv_dma<64, 4, 2, 2>(DMA_ARGS); break;
        default: launch_conv_dma<128, 4, 2, 2>(DMA_ARGS); break;
      }
    } else {
      // Cout % 64 == 0: 128x64 tiles, EIGHT waves of 32x32 (each wave issues 3 DMA pieces per chunk instead of 6 and has 16
      // MFMAs instead of 32 behind them: 84 -> 80 us on the 32->64 layer, 80 -> 70 on the 64->64 one; four waves of 32x64
      // were the form until the DMA pieces of a wave went out as grouped statements)
      switch (cin) {
        case 16: launch_conv_dma<16, 4, 2, 1>(DMA_ARGS); break;
        case 32: launch_conv_dma<32, 4, 2, 1>(DMA_ARGS); break;
        case 64: launch_conv_dma<64, 4, 2, 1>(DMA_ARGS); break;
        default: launch_conv_dma<128, 4, 2, 1>(DMA_ARGS); break;
      }
    }
#undef DMA_ARGS
  } else if (cin == 7 && cout == 16 && kvol <= 27 && g_force_valu != 1 && !mfma_ok) {
    int rows = 0;
    for (int i = 0; i < nsides; ++i) rows += sides.s[i].n_dev ? sides.s[i].cap : sides.s[i].n_host;
    hipLaunchKernelGGL((k_sparse_conv_stem<7, 16>), dim3(dcl_grid_1d(rows, 64)), dim3(256), 0, s, sides, nsides, kvol, subm,
                       relu);
  } else {
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
  long long rows = 0;
  for (int i = 0; i < nsides_in; ++i) {
    const DclConvSide &S = sides_in.s[i];
    DCL_CHECK_ARG(S.feat && (S.src.nbr || (S.src.out_indices && S.src.in_mask && S.src.in_wprefix && kvol == 27)) && S.out && S.cap > 0);
    DCL_CHECK_ARG(S.n_dev || (S.n_host >= 0 && S.n_host <= S.cap));
    if (S.n_dev || S.n_host > 0) {
      sides.s[nsides++] = S;
      rows += S.n_dev ? S.cap : S.n_host;
    }
  }
  if (nsides == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int c4 = c / 4;
  if (c % 4 == 0 && c4 >= 4 && c4 <= 64 && 256 % c4 == 0) {
    hipLaunchKernelGGL(k_sparse_avgpool, dim3(dcl_grid_1d(rows * c4, 256, 2048)), dim3(256), 0, s, sides, nsides, c, kvol, rf, rf_in);
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

// linear_dma.hip -- the per-point linear layers of DCL-Net's dense half as an OWN fp32 MFMA GEMM core (no vendor library):
//     y[M x N] = act(x[M x K] Wt[K x N] + bias[N]),   row-major, every matrix with its own row pitch,
// i.e. the Conv1d(k=1) / 1x1x1 Conv3d + folded BatchNorm + ReLU stacks of the reference (models/Modules.py:58-97, 173-201:
// BasicBlock_3DCONV / Head_MultiLayerPerceptron; call sites models/DCL_Net.py:188-235, models/refiner.py:78-95), which the
// reference runs as cuDNN pointwise convolutions / cuBLAS SGEMMs.  Until round 5 these were hipBLASLt calls (40 % of a forward).
//
// Kernel k_linear_dma<BM, BN, WGR, EPI>: a 256-thread workgroup (4 waves, WGR x WGC) owns one BM x BN tile of y; two
// workgroups share a CU (2 x 64 KiB of LDS, one wave of each per SIMD), so one's prologue / epilogue / barrier waits run
// under the other's MFMAs.  K goes in chunks of 32: the x tile [BM][32] and the Wt tile [32][BN] are fetched global -> LDS by
// LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, four pieces under one M0 value), double-buffered, one raw
// barrier per chunk (lgkmcnt only: the DMAs of the next chunk stay in flight across it), x rows XOR-swizzled on the SOURCE
// side so that the b128 fragment reads are conflict-free.  A wave computes (BM/WGR) x (BN/WGC) as 32 x 32 blocks of
// v_mfma_f32_32x32x2_f32 -- exact fp32, one rounding per product, the sum of an output element a single fmaf chain over k
// (chunk by chunk; inside a chunk k runs 0,4,1,5,2,6,3,7,8,12,...) -- 64 x 64 per wave in the main shape: four MFMAs per pair
// of operand registers.  Workgroup ids are renumbered XCD-aware: the workgroups that share an L2 walk consecutive tiles of
// one row block, so an x tile comes from HBM once and Wt (<= 2 MiB) stays resident in every L2.
// Bound: fp32 MFMA (157.3 TFLOP/s); algorithmic work 2 M N K flop; HBM bytes 4 (M K + K N + M N).
//
// EPI = 1 (dcl_linear_pool_fwd): the confidence-weighted pooling of models/DCL_Net.py:223-228 as the GEMM's epilogue --
//     part[row tile][c] = sum over the tile's rows j of  w[j] * relu(x_j . Wt[:, c] + bias[c])
// so the (b*N) x 1024 activation of the last fuser layer is never stored (1.9 GB written + read at the stress shape) and the
// weighted column-sum kernel disappears; dcl_pool_finish adds a crop's tile partials in a fixed order (deterministic).
#include <hip/hip_runtime.h>
#include "common.h"

namespace {

typedef float ld_f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void ld_lds_void_t;

constexpr int kLdKC = 32;                                            // k per chunk

__device__ __forceinline__ unsigned ld_lds_addr(const float *p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(ld_lds_void_t *)p);
}
// One LDS-DMA piece: 64 lanes x 16 B from (wave-uniform 64-bit base in an SGPR pair) + (per-lane 32-bit byte offset) to LDS at
// (wave-uniform) lds_byte_addr + lane * 16.  Inline asm on purpose (cdna guide 5.7; dense.hip: glds16): issued through the
// builtin hipcc drains the DMA before the next ds_read of the array; an asm load is not in the compiler's counters, the
// kernel waits for it itself.  M0 is saved / restored inside the statement.
__device__ __forceinline__ void ld_glds16_s(unsigned voff, const float *sbase, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
struct LinDmaArgs {
  const float *x, *Wt, *bias;
  float *y;
  long long ldx, ldw, ldy;
  int M, N, K, relu;
  const float *roww;     // EPI 1: row j (crop j / rows_per_crop, point j % rows_per_crop) weighs roww[crop * w_stride + point]
  int rows_per_crop;
  long long w_stride;
  float *part;           // EPI 1: [row tiles][ldp] partial weighted column sums;  EPI 2: out[M] (roww = w3 with stride w_stride)
  const float *b3;       // EPI 2: the last layer's bias (one float)
  long long ldp;
  int xcd_remap;
};

__device__ __forceinline__ int ld_rowmap(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// KS = 2 (launches of few tiles: a handful of crops): EIGHT waves per tile, two groups of four that take the even / the odd
// 32-deep chunks of K -- a tile's chain of chunks is half as long, which is what a launch that cannot fill the chip is bound
// by -- each group with its own half of every ring stage; at the end group 1 hands its accumulators over through LDS and
// group 0 adds them (acc0 + acc1: one fixed order) and runs the epilogue.
template <int BM, int BN, int WGR, int EPI, int KS>
__global__ __launch_bounds__(256 * KS, KS == 4 ? 4 : 2) void k_linear_dma(const LinDmaArgs a) {
  constexpr int WGC = 4 / WGR;
  constexpr int WM = BM / WGR, WN = BN / WGC;            // a wave's tile
  constexpr int MB = WM / 32, NB = WN / 32;              // ... in 32 x 32 MFMA blocks
  constexpr int AT = BM * kLdKC, BT = kLdKC * BN, ST = AT + BT;     // floats per stage and wave group
  constexpr int RING = 2 * KS * ST;                      // the whole ring: 2 stages x KS groups
  static_assert(KS == 1 || ((KS == 2 || KS == 4) && EPI == 0), "the K split comes with the plain epilogue");
  static_assert(EPI != 2 || WGR * WGC == 4, "row-dot epilogue");
  constexpr int APW = BM / 32, BPW = BN / 32;            // 1-KiB DMA pieces per wave and chunk (A: 8 rows each; B: 1 KiB of k-rows)
  static_assert(APW == 2 || APW == 4, "x pieces per wave");
  static_assert(BPW == 2 || BPW == 4, "Wt pieces per wave");
  constexpr int BLPR = BN / 4;                           // lanes per k-row of the Wt tile
  constexpr int BKPP = 64 / BLPR;                        // k-rows per Wt piece
  extern __shared__ __attribute__((aligned(16))) float ld_lds[];    // 2 stages x [A BM x 32 | B 32 x BN] (+ EPI 1: weights, sums)

  // ---- which tiles: the workgroup is PERSISTENT (2 per CU resident: gridDim = 512, fewer for small problems) and walks its
  // tiles in a loop, so that the first chunk of tile t + 1 is fetched under the last chunk of tile t and a tile costs no
  // dispatch, no address set-up round trip and no exposed first fetch.  Workgroup ids are dealt round-robin over the 8 XCDs:
  // XCD x owns a contiguous range of tiles in (row block, column block) order and its workgroups (slot = id / 8) walk it
  // with stride gridDim / 8 -- the workgroups that share an L2 work on neighbouring tiles of the same row blocks.
  const int ntn = (a.N + BN - 1) / BN, ntm = (a.M + BM - 1) / BM, tiles = ntm * ntn;
  int t_lo, t_hi, t_stride;                                // this workgroup's tiles: t_lo, t_lo + t_stride, ... < t_hi
  {
    const int id = blockIdx.x, nwg = gridDim.x;
    if (a.xcd_remap && (nwg & 7) == 0) {
      const int xq = tiles >> 3, xr = tiles & 7, xcd = id & 7;
      const int x_lo = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
      t_lo = x_lo + (id >> 3);
      t_hi = x_lo + xq + (xcd < xr ? 1 : 0);
      t_stride = nwg >> 3;
    } else {
      t_lo = id; t_hi = tiles; t_stride = nwg;
    }
  }
  if (t_lo >= t_hi) return;
  const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = KS == 1 ? 0 : wave8 >> 2, wave = KS == 1 ? wave8 : (wave8 & 3);      // K group, wave inside it
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave / WGC, wc = wave - wr * WGC;
  const int kchunks = a.K / kLdKC;                         // 32-deep chunks of K
  const int nchunks = (kchunks + KS - 1) / KS;             // steps of the chunk loop: group g takes chunk step * KS + g
  const long long bchunk = (long long)kLdKC * a.ldw;
  const unsigned lds0 = ld_lds_addr(ld_lds);
  const int n4 = (a.N + 3) & ~3;

  // ---- the producer: which chunk of which tile is fetched next.  A wave's DMA pieces per chunk: x pieces APW*wave .. (8 rows of
  // 128 B each), Wt pieces BPW*wave .. (BKPP k-rows each); a piece's source = a wave-uniform base (SGPR pair, advanced per chunk
  // by scalar adds) + a per-lane byte offset that only changes with the tile (one VGPR per piece).  Rows >= M / columns >= N are
  // fetched from the last valid row / column group instead of a zero line: they only feed accumulators that are never stored
  // (EPI 1: that get weight 0).
  unsigned aoff[APW], boff[BPW];
  const float *abase, *bbase;
  int p_tile = t_lo, p_chunk = 0;
  auto producer_tile = [&]() {                             // offsets and bases of tile p_tile, chunk 0
    const int ptm = p_tile / ntn, ptn = p_tile - ptm * ntn;
    const int prow0 = ptm * BM, pcol0 = ptn * BN;
#pragma unroll
    for (int i = 0; i < APW; ++i) {
      const int arow = (APW * wave + i) * 8 + (lane >> 3);                      // tile row
      const int achs = ((lane & 7) ^ ((arow >> 1) & 7)) << 2;                   // source column inside the chunk (swizzle)
      const int srow = min(prow0 + arow, a.M - 1) - prow0;
      aoff[i] = (unsigned)(((long long)srow * a.ldx + achs) * 4);
    }
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      const int bkk = (BPW * wave + i) * BKPP + lane / BLPR;                    // k-row inside the chunk
      const int bcol = min(pcol0 + ((lane % BLPR) << 2), n4 - 4) - pcol0;       // (a row of Wt holds N rounded up to 4 floats)
      boff[i] = (unsigned)(((long long)bkk * a.ldw + bcol) * 4);
    }
    abase = a.x + (size_t)prow0 * a.ldx + grp * kLdKC;
    bbase = a.Wt + pcol0 + (size_t)grp * bchunk;
  };
  auto producer_advance = [&]() {                          // behind the last piece of a chunk
    if (++p_chunk < nchunks) {
      abase += KS * kLdKC;
      bbase += KS * bchunk;
    } else {
      p_chunk = 0;
      p_tile += t_stride;
      if (p_tile < t_hi) producer_tile();
    }
  };
  // (a group whose chunk of the producer's step lies past K -- the odd group in the last step of an odd chunk count -- fetches
  //  nothing and, below, multiplies nothing)
  auto issue_a = [&](int i, int stage) {
    if (KS == 1 || p_chunk * KS + grp < kchunks)
      ld_glds16_s(aoff[i], abase, lds0 + (unsigned)(((stage * KS + grp) * ST + (APW * wave + i) * 256) * 4));
  };
  auto issue_b = [&](int i, int stage) {
    if (KS == 1 || p_chunk * KS + grp < kchunks)
      ld_glds16_s(boff[i], bbase, lds0 + (unsigned)(((stage * KS + grp) * ST + AT + (BPW * wave + i) * 256) * 4));
  };

  const int sw = (r >> 1) & 7;
  const float *arow0 = ld_lds + grp * ST + (wr * WM + r) * kLdKC;
  const float *bcol0 = ld_lds + grp * ST + AT + wc * WN + r + 4 * h * BN;
  // operand fragments of one 8-deep k step: x[row][8i + 4h .. +3] per 32-row block (one b128), Wt[8i + 4h + q][col] per column block
  struct Frag { float4 a[MB]; float b[NB][4]; };
  auto load_a = [&](Frag &f, int stage, int i) {
    const float *arow = arow0 + stage * KS * ST;
#pragma unroll
    for (int m = 0; m < MB; ++m) f.a[m] = *reinterpret_cast<const float4 *>(arow + m * 32 * kLdKC + (((2 * i + h) ^ sw) << 2));
  };
  auto load_b = [&](Frag &f, int stage, int i, int n) {
    const float *bcol = bcol0 + stage * KS * ST;
#pragma unroll
    for (int q = 0; q < 4; ++q) f.b[n][q] = bcol[(8 * i + q) * BN + n * 32];
  };
  ld_f32x16 acc[MB][NB];
  auto mfma_group = [&](const Frag &f, int q) {
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const float av = q == 0 ? f.a[m].x : q == 1 ? f.a[m].y : q == 2 ? f.a[m].z : f.a[m].w;
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, f.b[n][q], acc[m][n], 0, 0, 0);
    }
  };

  producer_tile();
#pragma unroll
  for (int i = 0; i < BPW; ++i) issue_b(i, 0);
#pragma unroll
  for (int i = 0; i < APW; ++i) issue_a(i, 0);
  producer_advance();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's pieces of the first chunk have landed
  int cc = 0;                                              // chunks consumed so far: chunk cc sits in stage cc & 1
  for (int tile = t_lo; tile < t_hi; tile += t_stride) {
    const int tm = tile / ntn, tn = tile - tm * ntn;
    const int row0 = tm * BM, col0 = tn * BN;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;
    if constexpr (EPI == 1) {                              // the tile's row weights -> LDS behind the ring (read by the epilogue;
      if (tid < BM) {                                      //  the previous tile's epilogue ended with a barrier)
        const int row = row0 + tid, crop = row / a.rows_per_crop;
        ld_lds[RING + tid] = row < a.M ? a.roww[(size_t)crop * a.w_stride + (row - crop * a.rows_per_crop)] : 0.0f;
      }
    }
    for (int c = 0; c < nchunks; ++c, ++cc) {
      const int st = cc & 1;
      const bool more = p_tile < t_hi;                     // the producer still has a chunk to fetch (this tile's next, or the
                                                           // NEXT tile's first: it lands under this tile's last chunk + epilogue)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (every wave has waited for its own pieces of this chunk: end of the
      __builtin_amdgcn_s_barrier();                        //  previous one) everyone's are in; nobody reads the other stage any more ...
      // ... which the producer's chunk lands in, under this chunk's MFMAs.  One instruction stream that keeps the matrix pipe
      // fed by itself: the chunk is 4 k steps x 4 groups of MB x NB MFMAs; behind every group goes ONE piece of other work -- a
      // slice of the NEXT step's fragment reads, or one DMA piece (steps 0 and 1: they need the rest of the chunk to land) --
      // so that nothing but the first fragment of a chunk is ever waited for.  The scheduler is pinned group by group (left
      // alone it reads each step's operands right before using them and issues the DMAs back to back).
      const bool mine = KS == 1 || c * KS + grp < kchunks;  // (wave-uniform) this group has a chunk in this step
      Frag f[2];
      if (mine) {
        load_a(f[0], st, 0);
#pragma unroll
        for (int n = 0; n < NB; ++n) load_b(f[0], st, 0, n);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (mine) {
            mfma_group(f[i & 1], q);
            if (i < 3) {
              if (q == 0) load_a(f[(i + 1) & 1], st, i + 1);
              else if (q - 1 < NB) load_b(f[(i + 1) & 1], st, i + 1, q - 1);
            }
          }
          if (more && i == 0 && q < BPW) issue_b(q, st ^ 1);
          if (more && i == 1 && q < APW) issue_a(q, st ^ 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (more) producer_advance();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the next chunk have landed (and, once per tile,
                                                           // the previous tile's stores: long done)
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // MFMA -> VALU read of the accumulators

    if constexpr (KS > 1) {
      // groups 1 .. KS-1 hand their partial sums over through LDS -- the ring itself: a K-split launch is one tile per workgroup
      // (launch_linear_dma), nothing is being fetched any more -- and group 0 adds them in group order (one fixed order)
      dcl_lds_barrier();                                   // every wave is past its last fragment read
      float *xch = ld_lds + ((wave * MB * NB) * 16) * 64 + lane;
      constexpr int XG = 4 * MB * NB * 16 * 64;            // floats of one group's accumulators
      if (grp > 0) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) xch[(grp - 1) * XG + ((m * NB + n) * 16 + e) * 64] = acc[m][n][e];
      }
      dcl_lds_barrier();
      if (grp == 0) {
#pragma unroll
        for (int gsrc = 0; gsrc < KS - 1; ++gsrc)
#pragma unroll
          for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
              for (int e = 0; e < 16; ++e) acc[m][n][e] = acc[m][n][e] + xch[gsrc * XG + ((m * NB + n) * 16 + e) * 64];
      }
    }
    if constexpr (EPI == 0) {
      if (KS > 1 && grp > 0) continue;                     // group 0 holds the sums
      float *__restrict__ y = a.y;
      const bool whole = row0 + BM <= a.M && col0 + BN <= a.N;     // (workgroup-uniform) interior tile: no per-element checks
      float bias[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const int co = col0 + wc * WN + n * 32 + r;
        bias[n] = (a.bias && co < a.N) ? a.bias[co] : 0.0f;
      }
      if (whole) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            float *yp = y + (size_t)(row0 + wr * WM + m * 32 + 4 * h) * a.ldy + (col0 + wc * WN + n * 32 + r);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              float v = acc[m][n][e] + bias[n];
              if (a.relu) v = fmaxf(v, 0.0f);
              yp[(size_t)((e & 3) + 8 * (e >> 2)) * a.ldy] = v;
            }
          }
      } else {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n) {
            const int co = col0 + wc * WN + n * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int orow = row0 + wr * WM + m * 32 + ld_rowmap(e, h);
              float v = acc[m][n][e] + bias[n];
              if (a.relu) v = fmaxf(v, 0.0f);
              if (orow < a.M && co < a.N) y[(size_t)orow * a.ldy + co] = v;
            }
          }
      }
    } else if constexpr (EPI == 2) {
      // row-dot epilogue (the confidence regressor's last two layers, models/DCL_Net.py:115-126: ... -> 128 -> 1): the tile
      // spans all N <= BN columns;  out[row] = sum_c relu(acc[row][c] + bias[c]) * w3[c] + b3  -- per lane over its column
      // blocks, over the 32 lanes of a half wave by a butterfly, over the WGC waves of a row through LDS in wave order.
      float w3c[NB], bias[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const int co = col0 + wc * WN + n * 32 + r;
        bias[n] = (a.bias && co < a.N) ? a.bias[co] : 0.0f;
        w3c[n] = co < a.N ? a.roww[(size_t)co * a.w_stride] : 0.0f;
      }
      float *red = ld_lds + RING;                          // [WGC][BM]
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float sdot = 0.0f;
#pragma unroll
          for (int n = 0; n < NB; ++n) sdot = __fmaf_rn(fmaxf(acc[m][n][e] + bias[n], 0.0f), w3c[n], sdot);
#pragma unroll
          for (int d = 16; d >= 1; d >>= 1) sdot += __shfl_xor(sdot, d, 64);
          if (r == 0) red[wc * BM + wr * WM + m * 32 + ld_rowmap(e, h)] = sdot;
        }
      dcl_lds_barrier();
      if (tid < BM && row0 + tid < a.M) {
        float sdot = red[tid];
#pragma unroll
        for (int w = 1; w < WGC; ++w) sdot += red[w * BM + tid];
        a.part[row0 + tid] = sdot + a.b3[0];
      }
      dcl_lds_barrier();                                   // (the next tile rewrites the sums)
    } else {
      // weighted column sums of the tile's rows: per wave over its WM rows (registers, then the two lane halves), then the WGR
      // waves of a column through LDS in wave order -- a fixed order, the same bits every run.  (The tile's row weights sit
      // in LDS behind the ring since the tile's first barrier; the sums go behind them: the ring itself may be receiving the
      // next tile's first chunk.)
      const float *wl = ld_lds + RING + wr * WM + 4 * h;
      float *red = ld_lds + RING + BM;                     // [WGR][BN]
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        const int cl = wc * WN + n * 32 + r, co = col0 + cl;
        const float bias = (a.bias && co < a.N) ? a.bias[co] : 0.0f;
        float s = 0.0f;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float4 w4 = *reinterpret_cast<const float4 *>(wl + m * 32 + 8 * g4);    // rows 8 g4 + 4 h + 0..3 = e 4 g4 .. + 3
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = acc[m][n][4 * g4 + j] + bias;
              if (a.relu) v = fmaxf(v, 0.0f);
              s = __fmaf_rn(v, wv[j], s);
            }
          }
        s += __shfl_xor(s, 32, 64);
        if (h == 0) red[wr * BN + cl] = s;
      }
      dcl_lds_barrier();
      if (tid < BN && col0 + tid < a.N) {
        float s = red[tid];
#pragma unroll
        for (int w = 1; w < WGR; ++w) s += red[w * BN + tid];
        a.part[(size_t)tm * a.ldp + col0 + tid] = s;
      }
      dcl_lds_barrier();                                   // (the next tile rewrites the weights / sums)
    }
  }
}

DCL_HOOK_INT(g_lin_tile, 0);          // diagnostic: 0 = automatic tile shape, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 / 5 = 64x64 with K split over 8 / 16 waves
DCL_HOOK_INT(g_lin_xcd, 1);
DCL_HOOK_INT(g_lin_persist, 1 << 20); // rounds of resident workgroups from which a launch is PERSISTENT (1 << 20: by K, see launch_linear_dma)

int lin_cu_count() {                    // CUs of the current device (cached per device)
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

template <int BM, int BN, int WGR, int EPI, int KS = 1>
int launch_linear_dma(const LinDmaArgs &a, hipStream_t stream) {
  const long long tiles = (long long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  if (tiles > 0x7fffffffll) {
    dcl_set_error("dcl_linear_fwd: too many tiles");
    return DCL_EINVAL;
  }
  // One workgroup per tile.  (The kernel's tile loop also runs PERSISTENT -- gridDim = the resident slots, every workgroup walking
  // its share of the tiles with the next tile's first chunk fetched under the current tile's last one -- and alone on the GPU
  // that is worth 5 % on the K = 256 layers (8 chunks per tile) and nothing at K = 480 / 512.  Inside a forward it measured
  // nothing either way, and persistent workgroups hold their CU slots for the whole launch: whatever runs beside them -- the
  // other direction's launches, the next call's sparse stage on its high-priority streams -- only gets in at their end.
  // Same-job A/B of the stress step: 23.60 ms persistent, 23.52 one workgroup per tile, 23.46 with the side streams at high
  // priority.  So only the short-K launches walk (below); the diagnostic library has the switch: dcl_debug_linear_persist.)
  constexpr size_t lds = (size_t)2 * KS * (BM + BN) * kLdKC * sizeof(float) + (EPI == 1 ? (BM + WGR * BN) * sizeof(float) : 0) +
                         (EPI == 2 ? (size_t)(4 / WGR) * BM * sizeof(float) : 0) +
                         0;
  static_assert(KS == 1 || (size_t)(KS - 1) * BM * BN <= (size_t)2 * KS * (BM + BN) * kLdKC, "the K groups' exchange fits the ring");
  constexpr int per_cu = (160 * 1024) / (int)lds > 4 ? 4 : (160 * 1024) / (int)lds;
  const long long slots = (long long)per_cu * lin_cu_count();
  // (short K -- the K = 128 / 256 layers, 4-8 chunks per tile -- is where the walk pays: a tile's first fetch is a quarter of
  //  its life; those launches are persistent from four rounds of slots on)
  const long long rounds = (int)g_lin_persist != (1 << 20) ? (long long)g_lin_persist : (a.K <= 256 ? 4 : (1 << 20));
  const unsigned grid = (unsigned)(KS == 1 && tiles >= rounds * slots ? slots : tiles);      // (K-split launches: one tile per workgroup)
  static bool attr_set = false;                            // (idempotent; a race sets it twice)
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_linear_dma<BM, BN, WGR, EPI, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_linear_dma<BM, BN, WGR, EPI, KS>), dim3(grid), dim3(256 * KS), lds, stream, a);
  return 0;
}

}  // namespace

#ifdef DCL_DIAG
DCL_API void dcl_debug_linear_tile(int t) { g_lin_tile = t; }
DCL_API void dcl_debug_linear_xcd_remap(int on) { g_lin_xcd = on; }
DCL_API void dcl_debug_linear_persist(int rounds) { g_lin_persist = rounds; }
#endif

// can this layer run on the own core?  (16-byte DMA pieces: aligned bases and pitches; K in whole chunks)
static bool lin_dma_ok(const float *x, int64_t ldx, const float *Wt, int64_t ldw, int N, int K) {
  return K >= kLdKC && K % kLdKC == 0 && ldx % 4 == 0 && ldw % 4 == 0 && ((size_t)x & 15) == 0 && ((size_t)Wt & 15) == 0 &&
         ldw >= (N + 3) / 4 * 4;
}

// Tile shape by a cost model: a CU works its tiles off at a fixed MFMA rate however many workgroups share it, so a launch
// lasts  ceil(tiles / CUs) * BM * BN / eff  (eff: smaller wave tiles re-read more operands per MFMA).  1 = 128 x 128,
// 2 = 128 x 64, 3 = 64 x 64.
static int lin_pick_tile(int M, int N) {
  const int forced = (int)g_lin_tile;
  if (forced >= 1 && forced <= 5) return forced;
  const int cus = lin_cu_count();
  static const int bm[3] = {128, 128, 64}, bn[3] = {128, 64, 64};
  static const double eff[3] = {1.0, 0.96, 0.93};
  int best = 3;
  double best_cost = 0.0;
  for (int i = 0; i < 3; ++i) {
    if (N <= 64 && bn[i] > 64) continue;
    const long long t = (long long)((M + bm[i] - 1) / bm[i]) * ((N + bn[i] - 1) / bn[i]);
    const double cost = (double)((t + cus - 1) / cus) * bm[i] * bn[i] / eff[i];
    if (best_cost == 0.0 || cost < best_cost) { best = i + 1; best_cost = cost; }
  }
  return best;
}

DCL_API int dcl_linear_dma_fwd(const float *x, int64_t ldx, const float *Wt, int64_t ldw, const float *bias, float *y, int64_t ldy,
                               int M, int N, int K, int relu, dclStream_t stream) {
  DCL_CHECK_ARG(M >= 0 && N > 0 && K > 0 && x && Wt && y && ldx >= K && ldw >= N && ldy >= N);
  DCL_CHECK_ARG(lin_dma_ok(x, ldx, Wt, ldw, N, K));
  if (M == 0) return 0;
  LinDmaArgs a{x, Wt, bias, y, ldx, ldw, ldy, M, N, K, relu, nullptr, 1, 0, nullptr, nullptr, 0, (int)g_lin_xcd};
  int rc, tile = lin_pick_tile(M, N);
  // a launch of 64 x 64 tiles that cannot even give every CU two of them is bound by ONE tile's chain of chunks: eight waves per
  // tile then, the two halves of K side by side (k_linear_dma<.., KS = 2>)
  // (and a launch that cannot give every CU even one: sixteen waves, four quarters of K)
  const long long t64 = (long long)((M + 63) / 64) * ((N + 63) / 64);
  if (tile == 3 && (int)g_lin_tile == 0 && t64 <= 2ll * lin_cu_count() && K >= 4 * kLdKC) tile = t64 <= lin_cu_count() && K >= 8 * kLdKC ? 5 : 4;
  switch (tile) {
    case 1: rc = launch_linear_dma<128, 128, 2, 0>(a, (hipStream_t)stream); break;
    case 2: rc = launch_linear_dma<128, 64, 2, 0>(a, (hipStream_t)stream); break;
    case 4: rc = launch_linear_dma<64, 64, 2, 0, 2>(a, (hipStream_t)stream); break;
    case 5: rc = launch_linear_dma<64, 64, 2, 0, 4>(a, (hipStream_t)stream); break;
    default: rc = launch_linear_dma<64, 64, 2, 0>(a, (hipStream_t)stream); break;
  }
  if (rc) return rc;
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_linear_pool_fwd(const float *x, int64_t ldx, const float *Wt, int64_t ldw, const float *bias, const float *roww,
                                int rows_per_crop, int64_t w_stride, float *part, int64_t ldp, int M, int N, int K, int relu,
                                dclStream_t stream) {
  DCL_CHECK_ARG(M > 0 && N > 0 && K > 0 && x && Wt && roww && part && ldx >= K && ldw >= N && ldp >= N && rows_per_crop >= 1 &&
                w_stride >= 0);
  DCL_CHECK_ARG(lin_dma_ok(x, ldx, Wt, ldw, N, K));
  LinDmaArgs a{x, Wt, bias, nullptr, ldx, ldw, 0, M, N, K, relu, roww, rows_per_crop, w_stride, part, nullptr, ldp, (int)g_lin_xcd};
  int rc = launch_linear_dma<128, 128, 2, 1>(a, (hipStream_t)stream);
  if (rc) return rc;
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_linear_rowdot_fwd(const float *x, int64_t ldx, const float *Wt, int64_t ldw, const float *bias, const float *w3,
                                  int64_t ldw3, const float *b3, float *out, int M, int N, int K, dclStream_t stream) {
  DCL_CHECK_ARG(M >= 0 && N > 0 && N <= 128 && K > 0 && x && Wt && w3 && b3 && out && ldx >= K && ldw >= N && ldw3 >= 1);
  DCL_CHECK_ARG(lin_dma_ok(x, ldx, Wt, ldw, N, K));
  if (M == 0) return 0;
  LinDmaArgs a{x, Wt, bias, nullptr, ldx, ldw, 0, M, N, K, 1, w3, 1, ldw3, out, b3, 0, (int)g_lin_xcd};
  int rc = launch_linear_dma<128, 128, 2, 2>(a, (hipStream_t)stream);
  if (rc) return rc;
  DCL_LAUNCH_CHECK();
  return 0;
}

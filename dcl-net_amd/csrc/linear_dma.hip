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
__device__ __attribute__((aligned(256))) float g_ld_zero[1024];      // the zero line (static storage: all zero); 4 KiB so that
                                                                     // the pre-decremented piece pointers below stay inside it

__device__ __forceinline__ unsigned ld_lds_addr(const float *p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(ld_lds_void_t *)p);
}
// N pieces of one LDS-DMA group: 64 lanes x 16 B each from per-lane global addresses to LDS at (wave-uniform) base + piece *
// 1 KiB + lane * 16, all under ONE M0 value.  The instruction's offset field moves the global address too, so source pointer
// i arrives pre-decremented by i KiB.  Inline asm on purpose (cdna guide 5.7; dense.hip: glds16): issued through the builtin
// hipcc drains the DMA before the next ds_read of the array; an asm load is not in the compiler's counters, the kernel
// waits for it itself.
__device__ __forceinline__ void ld_glds16_x4(const float *g0, const float *g1, const float *g2, const float *g3, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\t"
               "global_load_lds_dwordx4 %2, off offset:1024\n\t"
               "global_load_lds_dwordx4 %3, off offset:2048\n\t"
               "global_load_lds_dwordx4 %4, off offset:3072\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g0), "v"(g1), "v"(g2), "v"(g3), "s"(lds_byte_addr) : "memory");
}
// one piece: per-lane 32-bit byte offset + wave-uniform 64-bit base (SGPR pair)
__device__ __forceinline__ void ld_glds16_s(unsigned voff, const float *sbase, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void ld_glds16_x2(const float *g0, const float *g1, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\t"
               "global_load_lds_dwordx4 %2, off offset:1024\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g0), "v"(g1), "s"(lds_byte_addr) : "memory");
}

struct LinDmaArgs {
  const float *x, *Wt, *bias;
  float *y;
  long long ldx, ldw, ldy;
  int M, N, K, relu;
  const float *roww;     // EPI 1: weight of every row of x (M floats)
  float *part;           // EPI 1: [row tiles][ldp] partial weighted column sums
  long long ldp;
  int xcd_remap, stagger;
};

__device__ __forceinline__ int ld_rowmap(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

template <int BM, int BN, int WGR, int EPI>
__global__ __launch_bounds__(256, 2) void k_linear_dma(const LinDmaArgs a) {
  constexpr int WGC = 4 / WGR;
  constexpr int WM = BM / WGR, WN = BN / WGC;            // a wave's tile
  constexpr int MB = WM / 32, NB = WN / 32;              // ... in 32 x 32 MFMA blocks
  constexpr int AT = BM * kLdKC, BT = kLdKC * BN, ST = AT + BT;     // floats per stage
  constexpr int APW = BM / 32, BPW = BN / 32;            // 1-KiB DMA pieces per wave and chunk (A: 8 rows each; B: 1 KiB of k-rows)
  static_assert(APW == 2 || APW == 4, "x pieces per wave");
  static_assert(BPW == 2 || BPW == 4, "Wt pieces per wave");
  constexpr int BLPR = BN / 4;                           // lanes per k-row of the Wt tile
  constexpr int BKPP = 64 / BLPR;                        // k-rows per Wt piece
  extern __shared__ __attribute__((aligned(16))) float ld_lds[];    // 2 stages x [A BM x 32 | B 32 x BN]

  const int ntn = (a.N + BN - 1) / BN;
  // XCD-aware renumbering (speed only): workgroup ids are dealt round-robin over the 8 XCDs; give each XCD a contiguous range
  // of tiles in (row block, column block) order
  int tile;
  {
    const int nwg = gridDim.x, id = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    tile = !a.xcd_remap ? id : (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
  }
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int row0 = tm * BM, col0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave / WGC, wc = wave - wr * WGC;

  // ---- this wave's DMA pieces: x pieces APW*wave .. (8 rows of 128 B each), Wt pieces BPW*wave .. (BKPP k-rows each).  A piece's
  // source = a wave-uniform base (SGPR pair, advanced per chunk by scalar adds) + a per-lane byte offset that never changes
  // (one VGPR per piece).  Rows >= M / columns >= N are fetched from the last valid row / column group instead of a zero
  // line: they only feed accumulators that are never stored (EPI 1: that get weight 0).
  unsigned aoff[APW], boff[BPW];
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    const int arow = (APW * wave + i) * 8 + (lane >> 3);                        // tile row
    const int achs = ((lane & 7) ^ ((arow >> 1) & 7)) << 2;                     // source column inside the chunk (swizzle)
    const int srow = min(row0 + arow, a.M - 1) - row0;
    aoff[i] = (unsigned)(((long long)srow * a.ldx + achs) * 4);
  }
  const int n4 = (a.N + 3) & ~3;
#pragma unroll
  for (int i = 0; i < BPW; ++i) {
    const int bkk = (BPW * wave + i) * BKPP + lane / BLPR;                      // k-row inside the chunk
    const int bcol = min(col0 + ((lane % BLPR) << 2), n4 - 4) - col0;           // (a row of Wt holds N rounded up to 4 floats)
    boff[i] = (unsigned)(((long long)bkk * a.ldw + bcol) * 4);
  }
  const float *abase = a.x + (size_t)row0 * a.ldx;         // chunk 0 of the tile's rows / columns (wave-uniform)
  const float *bbase = a.Wt + col0;
  const long long bchunk = (long long)kLdKC * a.ldw;
  const unsigned lds0 = ld_lds_addr(ld_lds);

  ld_f32x16 acc[MB][NB];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;

  if (a.stagger > 0) {                                     // (diagnostic) the workgroup in the odd wave slots starts late
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4);             // HW_REG_HW_ID[3:0]: wave slot on the SIMD
    if (slot & 1)
      for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }

  const int nchunks = a.K / kLdKC;
  const int sw = (r >> 1) & 7;
  const float *arow0 = ld_lds + (wr * WM + r) * kLdKC;
  const float *bcol0 = ld_lds + AT + wc * WN + r + 4 * h * BN;
  // operand fragments of one 8-deep k step: x[row][8i + 4h .. +3] per 32-row block (one b128), Wt[8i + 4h + q][col] per column block
  struct Frag { float4 a[MB]; float b[NB][4]; };
  auto load_a = [&](Frag &f, int stage, int i) {
    const float *arow = arow0 + stage * ST;
#pragma unroll
    for (int m = 0; m < MB; ++m) f.a[m] = *reinterpret_cast<const float4 *>(arow + m * 32 * kLdKC + (((2 * i + h) ^ sw) << 2));
  };
  auto load_b = [&](Frag &f, int stage, int i, int n) {
    const float *bcol = bcol0 + stage * ST;
#pragma unroll
    for (int q = 0; q < 4; ++q) f.b[n][q] = bcol[(8 * i + q) * BN + n * 32];
  };
  auto mfma_group = [&](const Frag &f, int q) {
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      const float av = q == 0 ? f.a[m].x : q == 1 ? f.a[m].y : q == 2 ? f.a[m].z : f.a[m].w;
#pragma unroll
      for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, f.b[n][q], acc[m][n], 0, 0, 0);
    }
  };
  auto issue_a = [&](int i, int stage) { ld_glds16_s(aoff[i], abase, lds0 + (unsigned)((stage * ST + (APW * wave + i) * 256) * 4)); };
  auto issue_b = [&](int i, int stage) { ld_glds16_s(boff[i], bbase, lds0 + (unsigned)((stage * ST + AT + (BPW * wave + i) * 256) * 4)); };
#pragma unroll
  for (int i = 0; i < BPW; ++i) issue_b(i, 0);
#pragma unroll
  for (int i = 0; i < APW; ++i) issue_a(i, 0);
  for (int c = 0; c < nchunks; ++c) {
    const int st = c & 1;
    const bool more = c + 1 < nchunks;
    abase += kLdKC;                                        // -> chunk c + 1
    bbase += bchunk;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of chunk c have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // everyone's have; nobody reads chunk c - 1's stage any more ...
    // ... which chunk c + 1 lands in, under this chunk's MFMAs.  One instruction stream that keeps the matrix pipe fed by
    // itself: the chunk is 4 k steps x 4 groups of MB x NB MFMAs; behind every group goes ONE piece of other work -- a slice of
    // the NEXT step's fragment reads, or one DMA piece of the next chunk (steps 0 and 1: they need the rest of the chunk to
    // land) -- so that nothing but the first fragment of a chunk is ever waited for.  The scheduler is pinned group by group
    // (left alone it reads each step's operands right before using them and issues the DMAs back to back).
    Frag f[2];
    load_a(f[0], st, 0);
#pragma unroll
    for (int n = 0; n < NB; ++n) load_b(f[0], st, 0, n);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mfma_group(f[i & 1], q);
        if (i < 3) {
          if (q == 0) load_a(f[(i + 1) & 1], st, i + 1);
          else if (q - 1 < NB) load_b(f[(i + 1) & 1], st, i + 1, q - 1);
        }
        if (more && i == 0 && q < BPW) issue_b(q, st ^ 1);
        if (more && i == 1 && q < APW) issue_a(q, st ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");        // MFMA -> VALU read of the accumulators

  if constexpr (EPI == 0) {
    float *__restrict__ y = a.y;
    const bool whole = row0 + BM <= a.M && col0 + BN <= a.N;       // (workgroup-uniform) interior tile: no per-element checks
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
  } else {
    // weighted column sums of the tile's rows: per wave over its WM rows (registers, then the two lane halves), then the WGR
    // waves of a column through LDS in wave order -- a fixed order, the same bits every run
    float wrow[MB][16];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int orow = row0 + wr * WM + m * 32 + ld_rowmap(e, h);
        wrow[m][e] = orow < a.M ? a.roww[orow] : 0.0f;
      }
    __syncthreads();                                       // (all waves are past their last LDS reads: reuse it)
    float *red = ld_lds;                                   // [WGR][BN]
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int cl = wc * WN + n * 32 + r, co = col0 + cl;
      const float bias = (a.bias && co < a.N) ? a.bias[co] : 0.0f;
      float s = 0.0f;
#pragma unroll
      for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[m][n][e] + bias;
          if (a.relu) v = fmaxf(v, 0.0f);
          s = __fmaf_rn(v, wrow[m][e], s);
        }
      s += __shfl_xor(s, 32, 64);
      if (h == 0) red[wr * BN + cl] = s;
    }
    __syncthreads();
    if (tid < BN && col0 + tid < a.N) {
      float s = red[tid];
#pragma unroll
      for (int w = 1; w < WGR; ++w) s += red[w * BN + tid];
      a.part[(size_t)tm * a.ldp + col0 + tid] = s;
    }
  }
}

template <int BM, int BN, int WGR, int EPI>
int launch_linear_dma(const LinDmaArgs &a, hipStream_t stream) {
  const long long tiles = (long long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  if (tiles > 0x7fffffffll) {
    dcl_set_error("dcl_linear_fwd: too many tiles");
    return DCL_EINVAL;
  }
  constexpr size_t lds = (size_t)2 * (BM + BN) * kLdKC * sizeof(float);
  static bool attr_set = false;                            // (idempotent; a race sets it twice)
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_linear_dma<BM, BN, WGR, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_linear_dma<BM, BN, WGR, EPI>), dim3((unsigned)tiles), dim3(256), lds, stream, a);
  return 0;
}

DCL_HOOK_INT(g_lin_tile, 0);          // diagnostic: 0 = automatic tile shape, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 64x128
DCL_HOOK_INT(g_lin_xcd, 1);
DCL_HOOK_INT(g_lin_stagger, 0);

}  // namespace

#ifdef DCL_DIAG
DCL_API void dcl_debug_linear_tile(int t) { g_lin_tile = t; }
DCL_API void dcl_debug_linear_xcd_remap(int on) { g_lin_xcd = on; }
DCL_API void dcl_debug_linear_stagger(int n) { g_lin_stagger = n; }
#endif

// can this layer run on the own core?  (16-byte DMA pieces: aligned bases and pitches; K in whole chunks)
static bool lin_dma_ok(const float *x, int64_t ldx, const float *Wt, int64_t ldw, int N, int K) {
  return K >= kLdKC && K % kLdKC == 0 && ldx % 4 == 0 && ldw % 4 == 0 && ((size_t)x & 15) == 0 && ((size_t)Wt & 15) == 0 &&
         ldw >= (N + 3) / 4 * 4;
}

static int lin_pick_tile(int M, int N) {
  const int forced = (int)g_lin_tile;
  if (forced) return forced;
  if (N <= 64) return M >= 128 * 512 ? 2 : 3;                     // 128x64 while that still fills the 512 slots, else 64x64
  const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
  if (t128 >= 384) return 1;
  return (N % 128) > 64 || (N % 128) == 0 ? 3 : 3;
}

DCL_API int dcl_linear_dma_fwd(const float *x, int64_t ldx, const float *Wt, int64_t ldw, const float *bias, float *y, int64_t ldy,
                               int M, int N, int K, int relu, dclStream_t stream) {
  DCL_CHECK_ARG(M >= 0 && N > 0 && K > 0 && x && Wt && y && ldx >= K && ldw >= N && ldy >= N);
  DCL_CHECK_ARG(lin_dma_ok(x, ldx, Wt, ldw, N, K));
  if (M == 0) return 0;
  LinDmaArgs a{x, Wt, bias, y, ldx, ldw, ldy, M, N, K, relu, nullptr, nullptr, 0, (int)g_lin_xcd, (int)g_lin_stagger};
  int rc;
  switch (lin_pick_tile(M, N)) {
    case 1: rc = launch_linear_dma<128, 128, 2, 0>(a, (hipStream_t)stream); break;
    case 2: rc = launch_linear_dma<128, 64, 2, 0>(a, (hipStream_t)stream); break;
    case 4: rc = launch_linear_dma<64, 128, 2, 0>(a, (hipStream_t)stream); break;
    default: rc = launch_linear_dma<64, 64, 2, 0>(a, (hipStream_t)stream); break;
  }
  if (rc) return rc;
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_linear_pool_fwd(const float *x, int64_t ldx, const float *Wt, int64_t ldw, const float *bias, const float *roww,
                                float *part, int64_t ldp, int M, int N, int K, int relu, dclStream_t stream) {
  DCL_CHECK_ARG(M > 0 && N > 0 && K > 0 && x && Wt && roww && part && ldx >= K && ldw >= N && ldp >= N);
  DCL_CHECK_ARG(lin_dma_ok(x, ldx, Wt, ldw, N, K));
  LinDmaArgs a{x, Wt, bias, nullptr, ldx, ldw, 0, M, N, K, relu, roww, part, ldp, (int)g_lin_xcd, (int)g_lin_stagger};
  int rc = launch_linear_dma<128, 128, 2, 1>(a, (hipStream_t)stream);
  if (rc) return rc;
  DCL_LAUNCH_CHECK();
  return 0;
}

// linear_split.hip -- the big per-point linear layers on the bf16 matrix pipe WITHOUT giving up fp32 results:
//     y[M x N] = act(x[M x K] Wt[K x N] + bias[N])        (the layers of linear_dma.hip: models/Modules.py:58-97, 173-201;
//                                                          call sites models/DCL_Net.py:188-235, models/refiner.py:78-95)
// Every fp32 operand is written as the EXACT sum of three bf16 pieces, v = h + m + l (h = RN_bf16(v), m = RN_bf16(v - h),
// l = RN_bf16(v - h - m): 24 significant bits = 8 + 8 + 8, the two residuals are exact in fp32, the third piece is exact), and a
// product x w is accumulated as the six piece products of weight >= 2^-16,
//     x w  ~=  xl wh + xh wl + xm wm + xm wh + xh wm + xh wh        (dropped: xm wl + xl wm + xl wl, |.| <= 2^-25 |x w|),
// each an EXACT product of two 8-bit significands, summed in the fp32 accumulators of v_mfma_f32_32x32x16_bf16.  The dropped terms
// are below the rounding an fp32 FMA chain commits per step (2^-24 of the running sum); tests/test_gpu_ops.py compares this core
// and the fp32-MFMA core with float64 side by side (same error size).  Why: on gfx950 the fp32 MFMA runs at 1/16 of the bf16 one,
// so six bf16 products per fp32 product are 2.7x the fp32 pipe's rate on paper; what the chip sustains under load (it lowers its
// clock to ~1.75 GHz in bf16 MFMA loops on random data) is 290-300 TFLOP/s fp32-equivalent for the bare six-product body and
// 254 with the operand reads and the split beside it (tools/ubench_mfma_split.hip), against 155 for a bare fp32 MFMA loop.
//
// Kernel k_linear_split<EPI>: a 256-thread workgroup owns a 256 x 128 tile of y, wave w the rows 64 w .. 64 w + 63 of all 128
// columns (2 x 4 blocks of 32 x 32: 128 accumulator registers; an x value is split ONCE per workgroup, by the only wave that
// uses it).  K goes in 16-deep chunks (one MFMA k step): the x tile [256][16] fp32 and the three weight-piece tiles [128][16]
// bf16 are fetched global -> LDS by LDS-DMA (global_load_lds_dwordx4), two stages, one barrier per chunk, two workgroups per CU.
// The weight pieces are prepared once per layer (dcl_linear_split_weight: a layer's weights never change), already in tile order
// and already bank-swizzled, so a chunk's 12 KiB of them are one linear copy.  x rows are XOR-swizzled on the source side.
// Bound: bf16 MFMA at six products per fp32 product; algorithmic work 2 M N K flop (12 M N K bf16 flop executed).
#include <hip/hip_runtime.h>
#include "common.h"

namespace {

typedef float sp_f32x16 __attribute__((ext_vector_type(16)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned sp_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void sp_lds_void_t;

constexpr int kSpKC = 16;                                  // k per chunk = one MFMA k step
constexpr int kSpBM = 256, kSpBN = 128;                    // workgroup tile
constexpr int kSpA = kSpBM * kSpKC * 4;                    // bytes of an x tile          (16 KiB)
constexpr int kSpP = kSpBN * kSpKC * 2;                    // bytes of one weight-piece tile (4 KiB)
constexpr int kSpB = 3 * kSpP;                             // ... of the three             (12 KiB)
constexpr int kSpST = kSpA + kSpB;                         // a stage                      (28 KiB)

__device__ __forceinline__ unsigned sp_lds_addr(const void *p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(sp_lds_void_t *)p);
}
// one LDS-DMA piece: 64 lanes x 16 B from (wave-uniform base, SGPR pair) + (per-lane byte offset) to LDS at lds_byte_addr + lane * 16
// (inline asm on purpose, see linear_dma.hip: ld_glds16_s)
__device__ __forceinline__ void sp_glds16(unsigned voff, const void *sbase, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
// two floats -> two bf16 (round to nearest even: v_cvt_pk_bf16_f32), low half = a
__device__ __forceinline__ unsigned sp_cvt2(float a, float b) {
  const sp_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, sp_bf16x2));
}
// (x0, x1) -> packed pieces h, m, l with x = h + m + l exactly
__device__ __forceinline__ void sp_split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  // (the empty asm statements keep the two subtractions of a pair from being fused into one v_pk_add_f32: packed fp32 VALU beside
  //  MFMAs costs more than the two scalar instructions it replaces -- MI355X_MICROARCH.md, cycle constants)
  h = sp_cvt2(x0, x1);
  float r0 = x0 - __uint_as_float(h << 16);
  asm volatile("" : "+v"(r0));
  float r1 = x1 - __uint_as_float(h & 0xffff0000u);
  asm volatile("" : "+v"(r1));
  m = sp_cvt2(r0, r1);
  float s0 = r0 - __uint_as_float(m << 16);
  asm volatile("" : "+v"(s0));
  float s1 = r1 - __uint_as_float(m & 0xffff0000u);
  asm volatile("" : "+v"(s1));
  l = sp_cvt2(s0, s1);
}
__device__ __forceinline__ sp_bf16x8 sp_bf(sp_u32x4 v) { return __builtin_bit_cast(sp_bf16x8, v); }

struct LinSplitArgs {
  const float *x;
  const unsigned char *planes;                             // dcl_linear_split_weight's output
  const float *bias;
  float *y;
  long long ldx, ldy;
  int M, N, K, relu;
  const float *roww;     // EPI 1: row weights (see linear_dma.hip);  EPI 2: the last layer's weights
  int rows_per_crop;
  long long w_stride;
  float *part;           // EPI 1: [128-row tiles][ldp];  EPI 2: out[M]
  const float *b3;
  long long ldp;
  int xcd_remap;
};

__device__ __forceinline__ int sp_rowmap(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// weight pieces in tile order: [K/16 chunks][ceil(N/128) column tiles][3 pieces][128 columns][2 halves of the 16 k][8 bf16], the
// half index XORed with bit 3 of the column (the fragment reads of sixteen consecutive columns then cover all LDS banks)
__global__ __launch_bounds__(256) void k_split_weight(const float *__restrict__ Wt, long long ldw, int K, int N, unsigned *__restrict__ planes) {
  const int ntn = (N + kSpBN - 1) / kSpBN;
  const long long pieces = (long long)(K / kSpKC) * ntn * kSpBN * 2;       // 16-byte pieces per plane
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < pieces; i += (long long)gridDim.x * 256) {
    const int hs = (int)(i & 1), c = (int)((i >> 1) & (kSpBN - 1));
    const long long ct = i >> 8;                           // chunk * ntn + column tile
    const int kc = (int)(ct / ntn), tn = (int)(ct - (long long)kc * ntn);
    const int hh = hs ^ ((c >> 3) & 1), col = tn * kSpBN + c, k0 = kc * kSpKC + 8 * hh;
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float w0 = col < N ? Wt[(size_t)(k0 + 2 * e) * ldw + col] : 0.0f, w1 = col < N ? Wt[(size_t)(k0 + 2 * e + 1) * ldw + col] : 0.0f;
      sp_split2(w0, w1, h[e], m[e], l[e]);
    }
    unsigned *dst = planes + (size_t)ct * (kSpB / 4) + (size_t)(c * 2 + hs) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { dst[e] = h[e]; dst[kSpP / 4 + e] = m[e]; dst[2 * (kSpP / 4) + e] = l[e]; }
  }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void k_linear_split(const LinSplitArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];   // 2 stages x [x tile | 3 weight-piece tiles] (+ epilogue scratch)
  const int ntn = (a.N + kSpBN - 1) / kSpBN, ntm = (a.M + kSpBM - 1) / kSpBM, tiles = ntm * ntn;
  // workgroup ids are dealt round-robin over the 8 XCDs: XCD x owns a contiguous range of tiles in (row block, column block) order,
  // so the workgroups that share an L2 work on the column tiles of the same row blocks (linear_dma.hip)
  int t_lo, t_hi, t_stride;
  {
    const int id = blockIdx.x, nwg = gridDim.x;
    if ((a.xcd_remap & 1) && (nwg & 7) == 0) {
      const int xq = tiles >> 3, xr = tiles & 7, xcd = id & 7;
      const int x_lo = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;
      t_lo = x_lo + (id >> 3);
      t_hi = x_lo + xq + (xcd < xr ? 1 : 0);
      t_stride = nwg >> 3;
    } else {
      t_lo = id; t_hi = tiles; t_stride = nwg;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nchunks = a.K / kSpKC;
  const int whatif = a.xcd_remap >> 1;                     // (diagnostic library: timing-only what-if runs; 0 in the product)
  const unsigned lds0 = sp_lds_addr(sp_lds);
  const size_t bstep = (size_t)ntn * kSpB;                 // bytes from a column tile's chunk to its next chunk

  // fragment addresses inside a stage (bytes): x[row][8 h .. 8 h + 7] of row block i = pieces 2 h, 2 h + 1 of the row, swizzled by
  // the row; weight piece p of column block j = piece (h ^ bit 3 of the column) of column 32 j + r
  const int asw = (r >> 2) & 3;
  const unsigned char *afrag = sp_lds + (wave * 64 + r) * 64;
  const unsigned char *bfrag = sp_lds + kSpA + r * 32 + ((h ^ ((r >> 3) & 1)) << 4);

  for (int tile = t_lo; tile < t_hi; tile += t_stride) {
    const int tm = tile / ntn, tn = tile - tm * ntn;
    const int row0 = tm * kSpBM, col0 = tn * kSpBN;
    // this wave's DMA pieces per chunk: its own 64 x rows (4 pieces of 16 rows x 64 B), and 3 of the 12 KiB of weight pieces.  Rows
    // >= M are fetched from the last valid row: they only feed accumulators that are never stored (EPI 1: that get weight 0).
    unsigned aoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 64 + i * 16 + (lane >> 2);
      const int p = (lane & 3) ^ ((row >> 2) & 3);
      const int srow = min(row0 + row, a.M - 1) - row0;
      aoff[i] = (unsigned)(((long long)srow * a.ldx + 4 * p) * 4);
    }
    const unsigned boff = (unsigned)tid * 16u;
    const float *abase = a.x + (size_t)row0 * a.ldx;
    const unsigned char *bbase = a.planes + (size_t)tn * kSpB;
    // the 7 DMA pieces of a chunk: 3 of the weight pieces, 4 x-row groups; bases advance behind the last one
    auto issue_piece = [&](int k, int stage) {
      const unsigned s0 = lds0 + (unsigned)(stage * kSpST);
      if (k < 3) sp_glds16(boff + (unsigned)(k * 4096), bbase, s0 + (unsigned)(kSpA + k * 4096 + wave * 1024));
      else sp_glds16(aoff[k - 3], abase, s0 + (unsigned)((wave * 64 + (k - 3) * 16) * 64));
      if (k == 6) { abase += kSpKC; bbase += bstep; }
    };
    auto issue = [&](int stage) {
#pragma unroll
      for (int k = 0; k < 7; ++k) issue_piece(k, stage);
    };
    sp_f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    dcl_lds_barrier();                                     // (a previous tile's last reads / epilogue scratch)
    if constexpr (EPI == 1) {                              // the tile's row weights -> LDS behind the ring
      float *wl = reinterpret_cast<float *>(sp_lds + 2 * kSpST);
      const int row = row0 + tid, crop = row / a.rows_per_crop;
      wl[tid] = row < a.M ? a.roww[(size_t)crop * a.w_stride + (row - crop * a.rows_per_crop)] : 0.0f;
    }
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int c = 0; c < nchunks; ++c) {
      const int st = c & 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // everyone's pieces of chunk c are in (each wave waited for its own at the
      __builtin_amdgcn_s_barrier();                        //  end of the previous chunk); nobody reads the other stage any more ...
      // ... which chunk c + 1 lands in, under this chunk's MFMAs.  One instruction stream, pinned group by group (left alone the
      // compiler splits all of x first, reads every weight fragment next and -- worst -- hoists the wait for the NEXT chunk's DMA in
      // front of this chunk's MFMAs): first the fragment reads of row block 0 and column block 0 with the DMA issue under their
      // latency, the split of row block 0, then eight blocks of six MFMAs, the first four each with a quarter of row block 1's
      // split and the next column block's reads beside them.
      const bool more = c + 1 < nchunks;
      const unsigned char *sa = afrag + st * kSpST, *sb = bfrag + st * kSpST;
      float4 av[2][2];
      sp_u32x4 ap[2][3], bp[4][3];
      auto read_a = [&](int i) {
#pragma unroll
        for (int q = 0; q < 2; ++q) av[i][q] = *reinterpret_cast<const float4 *>(sa + i * 32 * 64 + (((2 * h + q) ^ asw) << 4));
      };
      auto read_b = [&](int j) {
#pragma unroll
        for (int p = 0; p < 3; ++p) bp[j][p] = *reinterpret_cast<const sp_u32x4 *>(sb + p * kSpP + j * 32 * 32);
      };
      auto split_a = [&](int i, int part) {                // values 2 part, 2 part + 1 of the row block's eight
        unsigned ph, pm, pl;
        const float4 v = av[i][part >> 1];
        if (part & 1) sp_split2(v.z, v.w, ph, pm, pl); else sp_split2(v.x, v.y, ph, pm, pl);
        ap[i][0][part] = ph; ap[i][1][part] = pm; ap[i][2][part] = pl;
      };
      read_a(0);
      read_b(0);
      read_a(1);
      if (more && !(whatif & 1)) issue(st ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!(whatif & 2)) {
#pragma unroll
        for (int part = 0; part < 4; ++part) split_a(0, part);
      } else {
#pragma unroll
        for (int p = 0; p < 3; ++p) ap[0][p] = sp_u32x4{__float_as_uint(av[0][0].x), __float_as_uint(av[0][0].y), __float_as_uint(av[0][1].x), __float_as_uint(av[0][1].y)};
      }
      read_b(1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int blk = 0; blk < 8; ++blk) {
        const int i = blk >> 2, j = blk & 3;               // small terms first
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(ap[i][2]), sp_bf(bp[j][0]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(ap[i][0]), sp_bf(bp[j][2]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(ap[i][1]), sp_bf(bp[j][1]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(ap[i][1]), sp_bf(bp[j][0]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(ap[i][0]), sp_bf(bp[j][1]), acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sp_bf(ap[i][0]), sp_bf(bp[j][0]), acc[i][j], 0, 0, 0);
        if (blk < 4) {
          if (!(whatif & 2)) split_a(1, blk);
          else { ap[1][0][blk] = __float_as_uint(av[1][blk >> 1].x); ap[1][1][blk] = __float_as_uint(av[1][blk >> 1].y); ap[1][2][blk] = __float_as_uint(av[1][blk >> 1].z); }
        }
        if (blk < 2) read_b(blk + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the next chunk have landed
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // MFMA -> VALU read of the accumulators

    if constexpr (EPI == 0) {
      if (whatif & 4) continue;
      float *__restrict__ y = a.y;
      const bool whole = row0 + kSpBM <= a.M && col0 + kSpBN <= a.N;
      float bias[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = col0 + j * 32 + r;
        bias[j] = (a.bias && co < a.N) ? a.bias[co] : 0.0f;
      }
      if (whole) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float *yp = y + (size_t)(row0 + wave * 64 + i * 32 + 4 * h) * a.ldy + (col0 + j * 32 + r);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              float v = acc[i][j][e] + bias[j];
              if (a.relu) v = fmaxf(v, 0.0f);
              __builtin_nontemporal_store(v, yp + (size_t)((e & 3) + 8 * (e >> 2)) * a.ldy);   // (an activation of hundreds of MB: read
            }                                                                                  //  once by the next layer, from HBM anyway)
          }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int co = col0 + j * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int orow = row0 + wave * 64 + i * 32 + sp_rowmap(e, h);
              float v = acc[i][j][e] + bias[j];
              if (a.relu) v = fmaxf(v, 0.0f);
              if (orow < a.M && co < a.N) y[(size_t)orow * a.ldy + co] = v;
            }
          }
      }
    } else if constexpr (EPI == 3) {
      // attention-V epilogue: the activation leaves the kernel AS the three bf16 pieces the attention's P.V product reads
      // (csrc/dense.hip: k_cross_attn_split; layout of k_attn_split_v, which this makes unnecessary for these channels) and is
      // never stored as fp32.  The accumulators already have that layout's shape: lane (r, h) of column block j holds, in elements
      // 0..7 / 8..15 of row block i, the keys (e & 3) + 8 (e >> 2) + 4 h of two consecutive 16-key half tiles for channel
      // col0 + 32 j + r -- one fragment of 8 keys each.  Rows = keys of crop row / rows_per_crop (a tile never straddles crops:
      // rows_per_crop % 256 == 0, checked by the entry point); ldp = half tiles per crop.
      unsigned char *vp = reinterpret_cast<unsigned char *>(a.part);
      const int crop = row0 / a.rows_per_crop, key0 = row0 - crop * a.rows_per_crop;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = col0 + j * 32 + r;
        const float bias = (a.bias && c < a.N) ? a.bias[c] : 0.0f;
        const int hs = h ^ ((c >> 3) & 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const long long ht = (long long)crop * a.ldp + ((key0 + wave * 64 + i * 32) >> 4) + hf;
            sp_u32x4 ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v0 = acc[i][j][8 * hf + 2 * e] + bias, v1 = acc[i][j][8 * hf + 2 * e + 1] + bias;
              if (a.relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
              unsigned qh, qm, ql;
              sp_split2(v0, v1, qh, qm, ql);
              ph[e] = qh; pm[e] = qm; pl[e] = ql;
            }
            unsigned char *dst = vp + (size_t)ht * (3 * 320 * 32) + (size_t)(c * 2 + hs) * 16;
            if (c < a.N) {
              *reinterpret_cast<sp_u32x4 *>(dst) = ph;
              *reinterpret_cast<sp_u32x4 *>(dst + 320 * 32) = pm;
              *reinterpret_cast<sp_u32x4 *>(dst + 2 * 320 * 32) = pl;
            }
          }
      }
    } else if constexpr (EPI == 2) {
      // row-dot epilogue (the confidence regressor's last two layers, models/DCL_Net.py:115-126: ... -> 128 -> 1): the tile spans all
      // N <= 128 columns and a wave all of them:  out[row] = sum_c relu(acc[row][c] + bias[c]) * w3[c] + b3 -- per lane over its four
      // column blocks, then a butterfly over the 32 lanes of a half wave
      float w3c[4], bias[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = col0 + j * 32 + r;
        bias[j] = (a.bias && co < a.N) ? a.bias[co] : 0.0f;
        w3c[j] = co < a.N ? a.roww[(size_t)co * a.w_stride] : 0.0f;
      }
      const float b3 = a.b3[0];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float sdot = 0.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j) sdot = __fmaf_rn(fmaxf(acc[i][j][e] + bias[j], 0.0f), w3c[j], sdot);
#pragma unroll
          for (int d = 16; d >= 1; d >>= 1) sdot += __shfl_xor(sdot, d, 64);
          const int orow = row0 + wave * 64 + i * 32 + sp_rowmap(e, h);
          if (r == 0 && orow < a.M) a.part[orow] = sdot + b3;
        }
    } else {
      // weighted column sums per 128-row pooling tile (waves 0, 1 / 2, 3): per wave over its 64 rows (registers, then the two
      // lane halves), then the two waves of a pooling tile through LDS in wave order -- a fixed order, the same bits every run
      const float *wl = reinterpret_cast<const float *>(sp_lds + 2 * kSpST) + wave * 64 + 4 * h;
      float *red = reinterpret_cast<float *>(sp_lds + 2 * kSpST) + kSpBM;          // [4 waves][128]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cl = j * 32 + r, co = col0 + cl;
        const float bias = (a.bias && co < a.N) ? a.bias[co] : 0.0f;
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float4 w4 = *reinterpret_cast<const float4 *>(wl + i * 32 + 8 * g4);
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float v = acc[i][j][4 * g4 + q] + bias;
              if (a.relu) v = fmaxf(v, 0.0f);
              s = __fmaf_rn(v, wv[q], s);
            }
          }
        s += __shfl_xor(s, 32, 64);
        if (h == 0) red[wave * kSpBN + cl] = s;
      }
      dcl_lds_barrier();
      {
        const int pt = tid >> 7, cl = tid & 127;           // pooling tile inside the workgroup tile, column
        if (col0 + cl < a.N && row0 + pt * 128 < a.M)
          a.part[(size_t)(2 * tm + pt) * a.ldp + col0 + cl] = red[(2 * pt) * kSpBN + cl] + red[(2 * pt + 1) * kSpBN + cl];
      }
    }
  }
}

DCL_HOOK_INT(g_sp_xcd, 1);            // bit 0: XCD-aware tile numbering; bits 1..: what-if runs (diagnostic library: dcl_debug_linear_split_whatif)

template <int EPI>
int launch_linear_split(const LinSplitArgs &a, hipStream_t stream) {
  const long long tiles = (long long)((a.M + kSpBM - 1) / kSpBM) * ((a.N + kSpBN - 1) / kSpBN);
  if (tiles > 0x7fffffffll) {
    dcl_set_error("dcl_linear_split_fwd: too many tiles");
    return DCL_EINVAL;
  }
  constexpr size_t lds = (size_t)2 * kSpST + (EPI == 1 ? (kSpBM + 4 * kSpBN) * sizeof(float) : 0);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)k_linear_split<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((k_linear_split<EPI>), dim3((unsigned)tiles), dim3(256), lds, stream, a);
  return 0;
}

bool lin_split_ok(const float *x, int64_t ldx, const void *planes, int K) {
  return K >= kSpKC && K % kSpKC == 0 && ldx % 4 == 0 && ((size_t)x & 15) == 0 && ((size_t)planes & 15) == 0;
}

}  // namespace

#ifdef DCL_DIAG
DCL_API void dcl_debug_linear_split_xcd_remap(int on) { g_sp_xcd = ((int)g_sp_xcd & ~1) | (on & 1); }
DCL_API void dcl_debug_linear_split_whatif(int bits) { g_sp_xcd = ((int)g_sp_xcd & 1) | (bits << 1); }
#endif

DCL_API int64_t dcl_linear_split_weight_bytes(int K, int N) {
  if (K <= 0 || N <= 0 || K % kSpKC) return 0;
  return (int64_t)(K / kSpKC) * ((N + kSpBN - 1) / kSpBN) * kSpB;
}

DCL_API int dcl_linear_split_weight(const float *Wt, int64_t ldw, int K, int N, void *planes, dclStream_t stream) {
  DCL_CHECK_ARG(Wt && planes && K > 0 && N > 0 && K % kSpKC == 0 && ldw >= N && ((size_t)planes & 15) == 0);
  const long long pieces = (long long)(K / kSpKC) * ((N + kSpBN - 1) / kSpBN) * kSpBN * 2;
  const unsigned grid = (unsigned)((pieces + 255) / 256 > 4096 ? 4096 : (pieces + 255) / 256);
  hipLaunchKernelGGL(k_split_weight, dim3(grid), dim3(256), 0, (hipStream_t)stream, Wt, (long long)ldw, K, N, (unsigned *)planes);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_linear_split_fwd(const float *x, int64_t ldx, const void *planes, const float *bias, float *y, int64_t ldy, int M, int N,
                                 int K, int relu, dclStream_t stream) {
  DCL_CHECK_ARG(M >= 0 && N > 0 && K > 0 && x && planes && y && ldx >= K && ldy >= N);
  DCL_CHECK_ARG(lin_split_ok(x, ldx, planes, K));
  if (M == 0) return 0;
  LinSplitArgs a{x, (const unsigned char *)planes, bias, y, ldx, ldy, M, N, K, relu, nullptr, 1, 0, nullptr, nullptr, 0, (int)g_sp_xcd};
  int rc = launch_linear_split<0>(a, (hipStream_t)stream);
  if (rc) return rc;
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_linear_split_pool_fwd(const float *x, int64_t ldx, const void *planes, const float *bias, const float *roww,
                                      int rows_per_crop, int64_t w_stride, float *part, int64_t ldp, int M, int N, int K, int relu,
                                      dclStream_t stream) {
  DCL_CHECK_ARG(M > 0 && N > 0 && K > 0 && x && planes && roww && part && ldx >= K && ldp >= N && rows_per_crop >= 1 && w_stride >= 0);
  DCL_CHECK_ARG(lin_split_ok(x, ldx, planes, K));
  LinSplitArgs a{x, (const unsigned char *)planes, bias, nullptr, ldx, 0, M, N, K, relu, roww, rows_per_crop, w_stride, part, nullptr, ldp,
                 (int)g_sp_xcd};
  int rc = launch_linear_split<1>(a, (hipStream_t)stream);
  if (rc) return rc;
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_linear_split_vpieces_fwd(const float *x, int64_t ldx, const void *planes, const float *bias, void *vplanes,
                                         int rows_per_crop, int M, int N, int K, int relu, dclStream_t stream) {
  DCL_CHECK_ARG(M > 0 && N > 0 && N <= 320 && N % 32 == 0 && K > 0 && x && planes && vplanes && ldx >= K && rows_per_crop > 0);
  DCL_CHECK_ARG(rows_per_crop % kSpBM == 0 && M % rows_per_crop == 0 && (((size_t)vplanes) & 15) == 0);
  DCL_CHECK_ARG(lin_split_ok(x, ldx, planes, K));
  LinSplitArgs a{x, (const unsigned char *)planes, bias, nullptr, ldx, 0, M, N, K, relu, nullptr, rows_per_crop, 0, (float *)vplanes, nullptr,
                 (long long)(rows_per_crop / 16), (int)g_sp_xcd};
  int rc = launch_linear_split<3>(a, (hipStream_t)stream);
  if (rc) return rc;
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_linear_split_rowdot_fwd(const float *x, int64_t ldx, const void *planes, const float *bias, const float *w3, int64_t ldw3,
                                        const float *b3, float *out, int M, int N, int K, dclStream_t stream) {
  DCL_CHECK_ARG(M >= 0 && N > 0 && N <= kSpBN && K > 0 && x && planes && w3 && b3 && out && ldx >= K && ldw3 >= 1);
  DCL_CHECK_ARG(lin_split_ok(x, ldx, planes, K));
  if (M == 0) return 0;
  LinSplitArgs a{x, (const unsigned char *)planes, bias, nullptr, ldx, 0, M, N, K, 1, w3, 1, ldw3, out, b3, 0, (int)g_sp_xcd};
  int rc = launch_linear_split<2>(a, (hipStream_t)stream);
  if (rc) return rc;
  DCL_LAUNCH_CHECK();
  return 0;
}

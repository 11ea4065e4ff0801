// conv_body.h -- device bodies of the sparse feature stage (conv / combine / avg-pool work items) behind the per-layer kernels
// of sparse_conv.hip (one launch per layer: item = blockIdx.x).  Reference: indiceConv<float>
// (libs/spconv/include/spconv/spconv_ops.h:253-349), indiceAvgPool (pool_ops.h:141-208), Backbone_SPCONV.forward
// (models/Modules.py:153-159).
#pragma once
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// offset visiting order: reference adds the centre GEMM first for subm, then k ascending.
__device__ __forceinline__ int offset_at(int step, int kvol, int subm) {
  if (!subm) return step;
  const int centre = kvol / 2;
  if (step == 0) return centre;
  return step <= centre ? step - 1 : step;
}
// ---- stem kernel: small Cin/Cout known at compile time (DCL-Net: 7 -> 16) ------------------------------------
// ONE lane per output row, all COUT channels in registers, W (kvol x CIN x COUT) in LDS.  The stem's rulebook is thin -- the
// dilated output set of level 0 has 3-4 of its 27 neighbours on average -- so a lane first looks its row's 27 neighbours up
// (nine columns of three, common.h: dcl_nbr_col; kept in a lane-private strip of LDS), then walks ONLY the present ones, in
// the reference's visiting order (spconv_ops.h:284-344: ascending k, the centre first for subm): per neighbour CIN loads and
// a CIN x COUT fmaf block with an ascending-ci chain per output, added to the row's sum -- the reference's own summation
// order.  (Until round 4 four lanes shared a row and every lane walked all of its seven offsets, present or not, because some
// lane of the wave always had one: 85 us for the 280 000 rows of 32 crops against 1.5 us of arithmetic.)  The filter rows are
// pitched CIN * COUT + 1 floats: lanes read different offsets' rows at the same (ci, co), and 112 floats apart they would
// share four banks.  No barrier after the filter load: a lane only reads LDS it has written itself.
// LPR = lanes per output row: 1 for launches of many rows; 4 for a handful of crops, where a row is a chain of dependent
// loads per present neighbour and there are lanes to spare: lane g of the row's quad takes the present neighbours number g,
// g + 4, ... (visiting order) and the four lane sums are added as (l0 + l1) + (l2 + l3).
// NTB = threads of the workgroup, NTB / LPR output rows per step; item = blockIdx.x of nitems = gridDim.x, row blocks round-robin.
template <int CIN, int COUT, int NTB, int LPR>
__device__ __forceinline__ void conv_stem_body(const DclConvSides &sides, int nsides, int kvol, int subm, int relu, float *Ws,
                                               int32_t *s_nb /* LDS: NTB * 27 ints */, int item, int nitems) {
  const int tix = (int)threadIdx.x;
  static_assert(COUT % 4 == 0, "rows are written as float4s");
  constexpr int WP = CIN * COUT + 1;                                              // filter pitch per offset (floats)
  const DclConvSide &S0 = sides.s[0];
  int n0 = S0.n_dev ? *S0.n_dev : S0.n_host;
  n0 = n0 < S0.cap ? n0 : S0.cap;
  int n1 = 0;
  if (nsides > 1) {
    n1 = sides.s[1].n_dev ? *sides.s[1].n_dev : sides.s[1].n_host;
    n1 = n1 < sides.s[1].cap ? n1 : sides.s[1].cap;
  }
  for (int i = tix; i < kvol * CIN * COUT; i += NTB) {
    const int k = i / (CIN * COUT), j = i - k * (CIN * COUT);
    Ws[k * WP + j] = S0.W[i];
    if (nsides > 1) Ws[27 * WP + k * WP + j] = sides.s[1].W[i];
  }
  __syncthreads();
  static_assert(LPR == 1 || LPR == 4, "lanes per row");
  constexpr int RB = NTB / LPR;                                                   // output rows per step
  const int g = tix & (LPR - 1);
  int32_t *nb = s_nb + (tix / LPR) * 27;                                          // the row's strip (stride 27: conflict-free)
  const int nblocks = (n0 + n1 + RB - 1) / RB;
  for (int blk = item; blk < nblocks; blk += nitems) {
    const int q = blk * RB + tix / LPR;
    const bool live = q < n0 + n1;
    if (LPR == 1 && !live) continue;                                              // (LPR = 4: whole quads stay for the butterfly)
    const int second = (live && q >= n0) ? 1 : 0;
    const DclConvSide &S = sides.s[second];
    const int row = live ? q - (second ? n0 : 0) : 0;
    unsigned present = 0;
    if (kvol == 27) {
      // LPR = 4: lane g looks up columns g, g + 4, g + 8; the quad shares the strip and ORs its masks
      const int4 q4 = live ? dcl_nbr_row(S.src, row) : make_int4(0, 0, 0, 0);
#pragma unroll
      for (int c0 = 0; c0 < 9; c0 += LPR) {
        const int col = c0 + g;
        if (col < 9 && live) {
          int t3[3];
          dcl_nbr_col(S.src, S.cap, col, row, q4, t3);
#pragma unroll
          for (int kz = 0; kz < 3; ++kz) {
            nb[3 * col + kz] = t3[kz];
            present |= (t3[kz] >= 0 ? 1u : 0u) << (3 * col + kz);
          }
        }
      }
    } else {
      for (int k = g; k < kvol; k += LPR) {
        const int v = live ? dcl_nbr_at(S.src, S.cap, k, row) : -1;
        nb[k] = v;
        present |= (v >= 0 ? 1u : 0u) << k;
      }
    }
    if (LPR == 4) {                                      // (the quad's lanes are neighbours in one wave: LDS writes above, reads below, in program order)
      present |= __shfl_xor(present, 1, 64);
      present |= __shfl_xor(present, 2, 64);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.0f;
    const float *wside = Ws + second * 27 * WP;
    const float *__restrict__ feat = S.feat;
    const int centre = kvol / 2;
    bool centre_first = subm && ((present >> centre) & 1u);
    if constexpr (LPR == 1) {
      // many rows: every present neighbour in visiting order; other waves hide the latency of the row loads
      while (present) {
        int k;
        if (centre_first) { k = centre; centre_first = false; } else { k = __builtin_ctz(present); }
        present &= ~(1u << k);
        const int v = nb[k];
        float f[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) f[ci] = feat[(size_t)v * CIN + ci];
        const float *w = wside + k * WP;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          float part = 0.0f;
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) part = __fmaf_rn(f[ci], w[ci * COUT + co], part);
          acc[co] = acc[co] + part;
        }
      }
    } else {
      // a handful of crops: this lane takes the present neighbours number g, g + 4, ... (visiting order), two rows in flight per
      // round (a chain of latencies otherwise), added in visiting order
      unsigned mine = 0;
      {
        unsigned rest = present;
        int turn = 0;
        if (centre_first) { if (g == 0) mine |= 1u << centre; rest &= ~(1u << centre); turn = 1; }
        while (rest) {
          const int k = __builtin_ctz(rest);
          rest &= rest - 1u;
          if (((turn++) & (LPR - 1)) == g) mine |= 1u << k;
        }
        centre_first = centre_first && g == 0;
      }
      auto next_k = [&]() -> int {
        int k;
        if (centre_first) { k = centre; centre_first = false; } else { k = __builtin_ctz(mine); }
        mine &= ~(1u << k);
        return k;
      };
      auto add_offset = [&](const float (&f)[CIN], int k) {
        const float *w = wside + k * WP;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          float part = 0.0f;
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) part = __fmaf_rn(f[ci], w[ci * COUT + co], part);
          acc[co] = acc[co] + part;
        }
      };
      while (mine) {
        const int ka = next_k();
        const int kb = mine ? next_k() : -1;
        const int va = nb[ka], vb = nb[kb >= 0 ? kb : ka];
        float fa[CIN], fb[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) { fa[ci] = feat[(size_t)va * CIN + ci]; fb[ci] = feat[(size_t)vb * CIN + ci]; }
        add_offset(fa, ka);
        if (kb >= 0) add_offset(fb, kb);
      }
    }
    if (LPR == 4) {
#pragma unroll
      for (int co = 0; co < COUT; ++co) {                // (l0 + l1) + (l2 + l3): commutative adds, the same bits in all four lanes
        acc[co] = acc[co] + __shfl_xor(acc[co], 1, 64);
        acc[co] = acc[co] + __shfl_xor(acc[co], 2, 64);
      }
      if (!live) continue;
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float x = acc[co];
      if (S.scale) x = x * S.scale[co] + S.shift[co];
      if (relu) x = fmaxf(x, 0.0f);
      acc[co] = x;
    }
    float4 *dst = reinterpret_cast<float4 *>(S.out + (size_t)row * COUT);
    if (LPR == 1) {
#pragma unroll
      for (int j = 0; j < COUT / 4; ++j) dst[j] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
    } else {                                             // lane g writes quarter g of the row
      static_assert(COUT % 16 == 0, "four lanes write COUT / 4 channels each as float4s");
      constexpr int PER = COUT / 4;
#pragma unroll
      for (int j = 0; j < PER / 4; ++j) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = 0.f;
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) x = g == gg ? acc[gg * PER + 4 * j + e] : x;   // static register indices
          o[e] = x;
        }
        dst[g * (PER / 4) + j] = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}
#ifdef DCL_CONV_STAMPS
// diagnostic build only (tools/conv_stamps.py): s_memrealtime (100 MHz) of workgroup phases, 8 stamps per segment of a
// workgroup (its first kStampSegs segments), in the launches the host marks (bit 4 of xcd_remap)
constexpr int kStampWgs = 1024, kStampSegs = 4;
__device__ unsigned long long g_conv_stamps[kStampWgs * kStampSegs * 16];
__device__ unsigned long long g_conv_phase[kStampWgs * kStampSegs * 16];     // wave 0's shader cycles in: DMA wait, barrier, issue, MFMA block; chunks
#define CONV_STAMP_SLOT (((int)blockIdx.x * kStampSegs + seg__) * 16)
#define CONV_STAMP_ON ((xcd_remap & 16) && threadIdx.x == 0 && (int)blockIdx.x < kStampWgs && seg__ < kStampSegs)
#define CONV_STAMP(i)                                                                                          \
  do {                                                                                                         \
    if (CONV_STAMP_ON) g_conv_stamps[CONV_STAMP_SLOT + (i)] = __builtin_amdgcn_s_memrealtime();                \
  } while (0)
#else
#define CONV_STAMP(i) do { } while (0)
#endif
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
// NTHR = threads of the workgroup; item = blockIdx.x of G = gridDim.x, ycol = blockIdx.y.
template <int CIN, int COUT_T, bool SUBM, int NTHR>
__device__ __forceinline__ void conv_wlds_body(const DclConvSides &sides, int nsides, int relu, float *wl_lds /* [27][CIN][32] */,
                                               int item, int G, int ycol) {
  const int tix = (int)threadIdx.x;
  constexpr int COUT = 32, KV = 27, GRP = CIN / 16;                          // 32 columns per workgroup; 16-channel groups per row
  constexpr int NWAVE = NTHR / 64;
  const int col0 = ycol * COUT;
  int n0 = sides.s[0].n_dev ? *sides.s[0].n_dev : sides.s[0].n_host;
  n0 = n0 < sides.s[0].cap ? n0 : sides.s[0].cap;
  int n1 = 0;
  if (nsides > 1) {
    n1 = sides.s[1].n_dev ? *sides.s[1].n_dev : sides.s[1].n_host;
    n1 = n1 < sides.s[1].cap ? n1 : sides.s[1].cap;
  }
  const int t0 = (n0 + 31) >> 5, t1 = (n1 + 31) >> 5;
  const int lane = tix & 63, wave = tix >> 6;
  const int r = lane & 31, h = lane >> 5;
  // EVERY side's filter is resident (two sides: Cin = 16 only, 2 x 54 KiB), and the wave tiles of all sides are dealt
  // round-robin over all waves of the launch: one launch for both backbones, and no side whose tiles come to "one and a bit"
  // per wave (32 crops, a launch per side: 1.07 and 1.7 tiles per wave -- two rounds each, the second nearly empty)
  for (int sd = 0; sd < nsides; ++sd) {
    const float4 *Wg = reinterpret_cast<const float4 *>(sides.s[sd].W + col0);
    float4 *Wl = reinterpret_cast<float4 *>(wl_lds) + sd * (KV * CIN * COUT / 4);
    for (int i = tix; i < KV * CIN * COUT / 4; i += NTHR) Wl[i] = Wg[(i >> 3) * (COUT_T / 4) + (i & 7)];
  }
  __syncthreads();
  // (running the first tile's look-ups under the filter load was measured: the 27 live row numbers across the load push
  // the 16-wave variant over its 128 registers -- 43 -> 52 us.  Round 5, measured and dropped: the filter in MFMA-operand
  // order (two ds_read_b128 instead of eight ds_read_b32 per group) with a zero line for missing neighbours -- no gain, 133
  // -> 140 us; a software pipeline over a wave's tiles with eight 256-register waves -- 153 us.  Probes of this kernel at
  // 32 crops: the gathers cost nothing measurable, the MFMAs 84 of 145 us, the look-ups 20, filter load + epilogue + launch 40.)
  // (slot = wave * G + item, not item * NWAVE + wave: the LAST, partly filled round of tiles then lands on every CU's first
  // waves -- one per SIMD, which runs its MFMAs unshared -- instead of filling all sixteen waves of the first few CUs: at 2.14
  // tiles per wave the third round of 578 tiles was a full round on 36 CUs while 220 waited)
  for (int gt = wave * G + item; gt < t0 + t1; gt += G * NWAVE) {
    const int second = gt >= t0 ? 1 : 0;
    const DclConvSide &S = sides.s[second];
    const float *__restrict__ feat = S.feat;
    const float *wl_side = wl_lds + second * (KV * CIN * COUT);
    const int n = second ? n1 : n0, tile = gt - (second ? t0 : 0);
    const int row = tile * 32 + r;
    const bool valid = row < n;
    int v[KV];                                                               // the 27 neighbour rows of this lane's output row
    {
      int vk[KV];                                                            // by offset k: nine columns of three z-neighbours
      const int4 q = valid ? dcl_nbr_row(S.src, row) : make_int4(0, 0, 0, 0);
#pragma unroll
      for (int col = 0; col < 9; ++col) {
        int t3[3] = {-1, -1, -1};
        if (valid) dcl_nbr_col(S.src, S.cap, col, row, q, t3);
        vk[3 * col] = t3[0]; vk[3 * col + 1] = t3[1]; vk[3 * col + 2] = t3[2];
      }
#pragma unroll
      for (int st = 0; st < KV; ++st) v[st] = vk[offset_at(st, KV, SUBM ? 1 : 0)];   // visiting order (a compile-time permutation)
    }
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
        const float *wk = wl_side + (k * CIN + 8 * h) * COUT + r;
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
    const DclConvSide &Se = S;
    const float sc = Se.scale ? Se.scale[col0 + r] : 1.0f, sh = Se.scale ? Se.shift[col0 + r] : 0.0f;
    float *__restrict__ out = Se.out;
    const bool have_scale = Se.scale != nullptr;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int orow = tile * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (orow < n) {
        float x = acc[e];
        if (have_scale) x = x * sc + sh;
        if (relu) x = fmaxf(x, 0.0f);
        out[(size_t)orow * COUT_T + col0 + r] = x;
      }
    }
  }
}
// ---- implicit-GEMM MFMA kernel fed by LDS-DMA (Cout % 64 == 0): same 64x64 output tile and virtual-channel walk as
// k_sparse_conv_tile, but both operand tiles of a 64-channel chunk go global -> LDS with global_load_lds_dwordx4 (no
// staging registers, no ds_write), double-buffered: the DMA of the next used chunk is issued right after the ONE barrier
// per chunk and has the whole chunk of MFMAs to land.  A rows are gathered by the DMA itself (per-lane global address
// = the neighbour's feature row, or a zero line for a missing neighbour); since a DMA instruction fills 1 KiB of
// contiguous LDS, bank conflicts are avoided by swizzling instead of padding: the 16-B column c of row r is stored at
// column c ^ (r & 15) (A), and W rows with bit 2 of their index set swap their 32-column halves (B).
__device__ float4 g_conv_zero_line = {0.f, 0.f, 0.f, 0.f};
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
// wid_in = blockIdx.x of G = gridDim.x.
template <int CIN, int WR, int WCW, int NT, bool ORD>
__device__ __forceinline__ void conv_dma_body(
    const DclConvSides &sides, int nsides, int cout, int kvol, int subm, int relu, float *__restrict__ partial, int stream_k,
    int aligned_ns, int xcd_remap, int32_t *__restrict__ tile_counters, int use_bal_arg, float *conv_lds, int wid_in, int G) {
  const int use_bal = ORD ? use_bal_arg : 0;
  constexpr int NW = WR * WCW, NTHR = 64 * NW;
  constexpr int BM = 32 * WR, BN = 32 * NT * WCW, KC = 32;
  constexpr int AT = BM * KC, BT = KC * BN, ST = AT + BT;      // floats per stage
  constexpr int A_INSTR = BM / 8;                              // 1-KiB DMA instructions per A tile (8 rows of 128 B each)
  constexpr int B_ROWS_PER = 256 / BN;                         // W rows per 1-KiB DMA instruction
  constexpr int B_INSTR = KC / B_ROWS_PER;
  static_assert(A_INSTR % NW == 0 && B_INSTR % NW == 0 && (BN == 32 || BN == 64 || BN == 128), "tile shape");
  constexpr int BSWZ = BN >= 64 ? 1 : 0;                       // W-row half swap (rows 32 floats wide have no halves to swap)
  // conv_lds: [stage 0: A|B][stage 1: A|B][Ns 27*BM][kmask (8)][rows BM]
  int32_t *Ns = reinterpret_cast<int32_t *>(conv_lds + 2 * ST);
  unsigned *s_kmask = reinterpret_cast<unsigned *>(Ns + 27 * BM);     // [0..WR-1]: offsets with a neighbour among the 32 rows of row group g; [4]: last-arriver flag
  int32_t *s_rows = reinterpret_cast<int32_t *>(s_kmask + 8);         // output row of every tile slot (-1 = none): the row order
  static_assert(NTHR % BM == 0 && WR <= 4, "a thread looks up ONE tile row (tid % BM) under every offset: its mask bits are that row's");

  // the launch's problems ("sides": the observed / template backbone of the same layer; one for a plain call): their
  // live row counts and tile counts -- every workgroup needs both to find its place in the common unit sequence
  int n0, n1 = 0;
  n0 = sides.s[0].n_dev ? *sides.s[0].n_dev : sides.s[0].n_host;
  n0 = n0 < sides.s[0].cap ? n0 : sides.s[0].cap;
  if (nsides > 1) {
    n1 = sides.s[1].n_dev ? *sides.s[1].n_dev : sides.s[1].n_host;
    n1 = n1 < sides.s[1].cap ? n1 : sides.s[1].cap;
  }
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wr = wave / WCW, wc = wave % WCW;
  const int iwave = wave;                                      // every wave issues its share of the chunk's DMA pieces
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
  int wid = wid_in;
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

#ifdef DCL_CONV_STAMPS
  int seg__ = -1;
#endif
  while (u < u_end) {
#ifdef DCL_CONV_STAMPS
    ++seg__;
#endif
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
    if (CONV_STAMP_ON)
      g_conv_stamps[CONV_STAMP_SLOT + 7] = ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) |
                                           (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // XCC_ID, HW_ID
#endif
    if (tid < 4) s_kmask[tid] = 0;
    constexpr bool ordered = ORD;                          // (natural order: the slot -> row map is arithmetic, no LDS round trip)
    if (ordered)
      for (int rr = tid; rr < BM; rr += NTHR) s_rows[rr] = row0 + rr < n ? ord.order[row0 + rr] : -1;
    dcl_lds_barrier();
    CONV_STAMP(8);
    // neighbour rows of the offsets this workgroup's chunk range touches (all 27 without split-K), by COLUMNS of three
    // z-neighbours (common.h: dcl_nbr_col -- one mask word + one prefix word per column instead of three dependent loads per
    // neighbour): a work item = (column, tile row); every thread's items are independent loads, all in flight at once
    unsigned mymask = 0;
    const int sx_lo = (j_begin * KC) / CIN;
    const int sx_hi = min(kvol - 1, (nchunks * KC - 1) / CIN);
    if (kvol == 27) {
      unsigned cmask = 0;                                    // columns the segment's steps lie in
      for (int sx = sx_lo; sx <= sx_hi; ++sx) cmask |= 1u << (offset_at(sx, kvol, subm) / 3);
      const int ncols = __builtin_popcount(cmask);
      constexpr int ITEMS = (9 * BM + NTHR - 1) / NTHR;
      const int rr = tid % BM;                               // this thread's tile row, under every column it looks up
      const int orow = ordered ? s_rows[rr] : (row0 + rr < n ? row0 + rr : -1);
      const int4 q4 = orow >= 0 ? dcl_nbr_row(src, orow) : make_int4(0, 0, 0, 0);
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int ci = tid / BM + it * (NTHR / BM);
        if (ci >= ncols) break;
        unsigned m = cmask;
        for (int q = ci; q > 0; --q) m &= m - 1u;
        const int col = __builtin_ctz(m);
        int v[3] = {-1, -1, -1};
        if (orow >= 0) dcl_nbr_col(src, cap, col, orow, q4, v);
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
          Ns[(3 * col + kz) * BM + rr] = v[kz];
          mymask |= (v[kz] >= 0 ? 1u : 0u) << (3 * col + kz);
        }
      }
    } else {
#pragma unroll NW == 4 ? 8 : 4
      for (int e = tid; e < (sx_hi - sx_lo + 1) * BM; e += NTHR) {
        const int si = e / BM, rr = e - si * BM;
        const int k = offset_at(sx_lo + si, kvol, subm);
        const int orow = ordered ? s_rows[rr] : (row0 + rr < n ? row0 + rr : -1);
        const int v = orow >= 0 ? dcl_nbr_at(src, cap, k, orow) : -1;
        Ns[k * BM + rr] = v;
        mymask |= (v >= 0 ? 1u : 0u) << k;
      }
    }
    CONV_STAMP(9);
    // mymask = the offsets under which THIS thread's tile row (tid % BM) has a neighbour: OR over the 32 rows of a wave's row
    // group -> the group's mask (what lets a compute wave skip a chunk's MFMA block), OR over the groups -> the tile's
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) mymask |= __shfl_xor(mymask, d, 64);
    if ((lane & 31) == 0 && mymask) atomicOr(s_kmask + ((tid % BM) >> 5), mymask);
    dcl_lds_barrier();
    CONV_STAMP(10);
    unsigned kmask = 0;
#pragma unroll
    for (int g = 0; g < WR; ++g) kmask |= s_kmask[g];
    // offsets (k) -> steps (visiting order, offset_at): the centre offset moves to step 0 for subm
    auto steps_of = [&](unsigned km) -> unsigned {
      if (!subm) return km;
      const int c = kvol / 2;
      return ((km >> c) & 1u) | ((km & ((1u << c) - 1u)) << 1) | (km & ~((2u << c) - 1u));
    };
    // ---- chunk control, scalar-light.  Steps (= kernel offsets in visiting order, offset_at) own CPK = CIN/KC chunks
    // each, or a chunk spans SPC = KC/CIN steps (CIN = 16).  `smask` marks the steps whose offset has a neighbour in this
    // tile, `wsmask` those with one among this wave's 32 rows; the next used chunk is a find-first-set away.
    constexpr int CPK = CIN >= KC ? CIN / KC : 1, SPC = CIN >= KC ? 1 : KC / CIN;
    const unsigned smask = steps_of(kmask);
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
        const int g = iwave * A_PER + i;
        a_row[i] = g * 8 + (lane >> 3);
        a_chs[i] = ((lane & 7) ^ ((a_row[i] >> 1) & 7)) << 2;
      }
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int g = iwave * B_PER + i;
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
      for (int g = 0; g < B_PER; g += BGRP) conv_glds16_group<BGRP>(psrc + A_PER + g, conv_lds_addr(Bs + (iwave * B_PER + g) * 256));
#pragma unroll
      for (int g = 0; g < A_PER; g += AGRP) conv_glds16_group<AGRP>(psrc + g, conv_lds_addr(As + (iwave * A_PER + g) * 256));
    };
    // steps under which at least one of THIS WAVE's 32 rows has a neighbour (wave-level skip of a chunk's MFMA block)
    const unsigned wsmask = __builtin_amdgcn_readfirstlane(steps_of(s_kmask[wr]));
    CONV_STAMP(11);

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
    if (CONV_STAMP_ON) {
      g_conv_phase[CONV_STAMP_SLOT + 0] = ph_wait; g_conv_phase[CONV_STAMP_SLOT + 1] = ph_bar; g_conv_phase[CONV_STAMP_SLOT + 2] = ph_issue;
      g_conv_phase[CONV_STAMP_SLOT + 3] = ph_mfma; g_conv_phase[CONV_STAMP_SLOT + 4] = ph_n;
      g_conv_phase[CONV_STAMP_SLOT + 5] = ((unsigned long long)(unsigned)tile << 32) | ((unsigned)j_begin << 16) | (unsigned)nchunks;
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
        dcl_lds_barrier();
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
        s_kmask[4] = last;
      }
      __syncthreads();
      CONV_STAMP(5);
      const bool last_arriver = s_kmask[4] != 0;
      if (last_arriver) {
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
      dcl_lds_barrier();
      CONV_STAMP(6);
      continue;
    }
    {
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
    dcl_lds_barrier();
  }
}
// Deferred combine of a stream-K launch (few-row launches: a tile has up to 27 segments, which the last arriver would have
// to add in as many dependent rounds of loads -- here every thread owns one 16-B piece of a tile and has all of its
// segments' loads in flight at once).  Same unit arithmetic as k_sparse_conv_dma; tiles owned by ONE workgroup were
// written by it directly and are skipped.  grid = (tiles_cap, NW*NT*4*64/256), 256 threads.
// tile = item, one 16-B piece per thread (piece0 = blockIdx.y * 256).
template <int WR, int WCW, int NT>
__device__ __forceinline__ void conv_frag_reduce_body(const float *__restrict__ partial, const DclConvSides &sides, int nsides,
                                                      int cout, int C, int G, int min_u, int relu, int item, int piece0) {
  const int tix = (int)threadIdx.x;
  constexpr int NW = WR * WCW, BM = 32 * WR, BN = 32 * NT * WCW;
  const DclConvSide &S0 = sides.s[0];
  int n0 = S0.n_dev ? *S0.n_dev : S0.n_host;
  n0 = n0 < S0.cap ? n0 : S0.cap;
  int n1 = 0;
  if (nsides > 1) {
    n1 = sides.s[1].n_dev ? *sides.s[1].n_dev : sides.s[1].n_host;
    n1 = n1 < sides.s[1].cap ? n1 : sides.s[1].cap;
  }
  const int ncol = cout / BN;
  const int tiles0 = (n0 + BM - 1) / BM * ncol, tiles1 = (n1 + BM - 1) / BM * ncol;
  const int total = (tiles0 + tiles1) * C;
  int U = (total + G - 1) / G;
  if (U < min_u) U = min_u;
  constexpr int NP = NW * NT * 4 * 64;                                       // 16-B pieces of a tile
  const int tile = item;                                                     // numbered over the whole launch: side 0's, then side 1's
  if (tile >= tiles0 + tiles1) return;
  const int w_first = (tile * C) / U, w_last = ((tile + 1) * C - 1) / U;
  if (w_first == w_last) return;                                             // one owner: written by the conv body
  const int second = tile >= tiles0 ? 1 : 0;
  const DclConvSide &S = sides.s[second];
  const int n = second ? n1 : n0, lt = tile - (second ? tiles0 : 0);
  const int blk = lt / ncol, by = lt - blk * ncol;
  const int piece = piece0 + tix;                                            // [wave][t][q][lane] inside the tile
  if (piece >= NP) return;
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
// ---- sparse average pool ------------------------------------------------------------------------
// thread = (output row, 4 channels): rf = #valid offsets (summaryRF.cu:39), then
// out = ((0 + f_k0/rf) + f_k1/rf) + ... in ascending offset order (avgpool.cu:130).
// NTB = threads of the workgroup; item = blockIdx.x of nitems = gridDim.x, row blocks round-robin.
template <int NTB, int PF, bool K27>
__device__ __forceinline__ void avgpool_body(const DclConvSides &sides, int nsides, int c, int kvol_in, int32_t *__restrict__ rf_out,
                                             const int32_t *__restrict__ rf_in, int32_t *s_v /* LDS: 64 * 27 ints */, int item,
                                             int nitems) {
  constexpr int kLookupUnroll = PF == 27 ? 1 : 3;
  const int tix = (int)threadIdx.x;
  const int kvol = K27 ? 27 : kvol_in;                     // K27: the 3^3 window of the network's pools, known to the compiler
  // the c/4 threads of an output row share its 27 neighbour rows through LDS (one lookup per (row, offset) per block).
  // Up to two problems per launch (the two backbones' pools of a level): the row blocks of side 0, then those of side 1;
  // rf_out / rf_in (the op-level API) belong to side 0 of a one-sided launch.
  const DclConvSide &S0 = sides.s[0];
  int n0 = S0.n_dev ? *S0.n_dev : S0.n_host;
  n0 = n0 < S0.cap ? n0 : S0.cap;
  int n1 = 0;
  if (nsides > 1) {
    n1 = sides.s[1].n_dev ? *sides.s[1].n_dev : sides.s[1].n_host;
    n1 = n1 < sides.s[1].cap ? n1 : sides.s[1].cap;
  }
  const int c4 = c >> 2;                                   // a divisor of NTB with NTB / c4 <= 64 (checked by the launcher)
  const int rpb = NTB / c4;                                // output rows per block step
  const int tid = tix;
  const int rr = tid / c4, q = tid - rr * c4;
  const int nb0 = (n0 + rpb - 1) / rpb, nb1 = (n1 + rpb - 1) / rpb;
  for (int bi = item; bi < nb0 + nb1; bi += nitems) {
    const int second = bi >= nb0 ? 1 : 0;
    const DclConvSide &S = sides.s[second];
    const float *__restrict__ feat = S.feat;
    float *__restrict__ out = S.out;
    const int n = second ? n1 : n0, cap = S.cap;
    const int row0 = (bi - (second ? nb0 : 0)) * rpb;
    dcl_lds_barrier();
    if (K27) {                                             // by columns of three z-neighbours (common.h: dcl_nbr_col)
      // (PF = 27 is the kernel of launches of a few thousand rows, which run their code once from a cold instruction
      //  cache: its look-up loop stays rolled -- see the note on code size below)
#pragma unroll kLookupUnroll
      for (int e = tid; e < rpb * 9; e += NTB) {
        const int col = e / rpb, r2 = e - col * rpb;
        int t3[3] = {-1, -1, -1};
        if (row0 + r2 < n) dcl_nbr_col(S.src, cap, col, row0 + r2, dcl_nbr_row(S.src, row0 + r2), t3);
        s_v[r2 * 27 + 3 * col] = t3[0]; s_v[r2 * 27 + 3 * col + 1] = t3[1]; s_v[r2 * 27 + 3 * col + 2] = t3[2];
      }
    } else {
#pragma unroll 1
      for (int e = tid; e < rpb * kvol; e += NTB) {
        const int r2 = e / kvol, k = e - r2 * kvol;
        s_v[r2 * 27 + k] = row0 + r2 < n ? dcl_nbr_at(S.src, cap, k, row0 + r2) : -1;
      }
    }
    dcl_lds_barrier();
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
    // occupancy that 9 lacked in depth -- against 19 / 17 / 16 / 12 with two rounds of 14.)
    // (PF = 27, the whole window in one round, for launches of a few thousand rows: there the launch IS its chain of dependent
    //  loads and nothing else runs on the CU)
    //
    // f / d of the reference's kernel, correctly rounded, in three instructions instead of the IEEE division's dozen: with
    // r = RN(1 / d), q0 = RN(f r), e = f - q0 d (exact in an FMA), RN(q0 + e r) = RN(f / d) for every d of a pooling window
    // (1..27) and every f whose quotient stays clear of the subnormals -- tests/test_pool_division.py walks all 2^23
    // significands of every d -- and a row whose window holds anything else (tiny, huge, inf, nan) is redone with the
    // division itself, neighbour by neighbour.  Why it matters: a pool of a few hundred rows runs its code ONCE, from a
    // cold instruction cache, and the 108 expanded divisions were a third of what it had to fetch.
    const float r = 1.0f / d;
    uint32_t hi = 0u, lo = 0xffffffffu;                    // largest / smallest non-zero |f| of the window, as bit patterns
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
        const float4 g = f[j];
        const uint32_t a0 = __float_as_uint(g.x) & 0x7fffffffu, a1 = __float_as_uint(g.y) & 0x7fffffffu;
        const uint32_t a2 = __float_as_uint(g.z) & 0x7fffffffu, a3 = __float_as_uint(g.w) & 0x7fffffffu;
        hi = max(max(hi, max(a0, a1)), max(a2, a3));       // (rows loaded for missing neighbours included: cheap and safe)
        lo = min(min(lo, min(a0 - 1u, a1 - 1u)), min(a2 - 1u, a3 - 1u));      // (zero wraps to the top and drops out)
        const float q0 = g.x * r, q1 = g.y * r, q2 = g.z * r, q3 = g.w * r;
        const float t0 = __fmaf_rn(__fmaf_rn(-q0, d, g.x), r, q0), t1 = __fmaf_rn(__fmaf_rn(-q1, d, g.y), r, q1);
        const float t2 = __fmaf_rn(__fmaf_rn(-q2, d, g.z), r, q2), t3 = __fmaf_rn(__fmaf_rn(-q3, d, g.w), r, q3);
        acc.x = ok ? acc.x + t0 : acc.x; acc.y = ok ? acc.y + t1 : acc.y;
        acc.z = ok ? acc.z + t2 : acc.z; acc.w = ok ? acc.w + t3 : acc.w;
      }
    }
    if (!(hi <= 0x71800000u && lo >= 0x0d800000u - 1u)) {  // not all of 2^-100 <= |f| <= 2^100 or f == 0
      acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
      for (int k = 0; k < kvol; ++k) {
        const int vk = v[k];
        if (vk < 0) continue;
        const float4 g = reinterpret_cast<const float4 *>(feat + (size_t)vk * c)[q];
        acc.x = acc.x + g.x / d; acc.y = acc.y + g.y / d; acc.z = acc.z + g.z / d; acc.w = acc.w + g.w / d;
      }
    }
    reinterpret_cast<float4 *>(out + (size_t)row * c)[q] = acc;
    if (rf_out && q == 0) rf_out[row] = rf;
  }
}

}  // namespace

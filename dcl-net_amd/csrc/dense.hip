// dense.hip -- the correspondence (cross-attention) contraction, confidence pooling and the
// 9-D -> SO(3) projection of DCL-Net's head.
//
// dcl_cross_attention replaces Aligner.forward + the extra bmm of models/Modules.py:162-169 and
// models/DCL_Net.py:206-215: three cuBLAS batched SGEMMs and a softmax over a MATERIALISED
// (b, Nk, Nq) attention map (4 MiB per crop per direction at N=M=1024, 96 MiB at 12288x2048).
// Here the map never leaves registers: per 32-query x 32-key tile
//     S  = K Q^T            fp32 MFMA 32x32x2, keys on the MFMA row axis, queries on the lane axis
//     P  = exp(S - m_ref)   online softmax over the KEY axis; each lane owns one query column, so
//                           the column max/sum is 16 registers + one lane^32 exchange
//     O += [V_p;V_m] P      the S accumulator registers ARE the B operand of the second MFMA
//                           (register e of lane-half h holds key (e&3)+8(e>>2)+4h), no shuffle/LDS
// with a lazily updated reference maximum (rescale only when the running max grows by > kThr),
// and one final division by the softmax denominator.  Bound: fp32 MFMA (157 TFLOP/s spec);
// algorithmic flop = 2*(dk+dv)*Nq*Nk per crop per direction.
#include "common.h"
#include <atomic>
#include <math.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kKPitch = 68;        // floats per key row of the K tile in LDS: 64 ch + 4 pad (b128 conflict-free)
constexpr float kThr = 20.0f;      // lazy-rescale threshold (e^20 ~ 5e8: far inside fp32 range)

__device__ __forceinline__ int rowmap(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

#ifdef DCL_DIAG   // general-shape predecessors of k_cross_attn_dma: diagnostic library only (A/B references)
// All operands POINT-major: X[(b*n + p)*ld + c]  (the layout the 3-NN interpolation produces and the
// 2-D GEMMs of the MLP stacks consume; the reference's (b,C,n) tensors are transposed views of it).
//
// Workgroup = 4 waves = 128 queries of one crop, one wave per SIMD with the whole register file:
// O (NVT x 16 accumulators) and the wave's Q rows stay in registers for the entire key sweep.  K/V tiles
// of 32 keys are double-buffered in LDS: the global loads of tile t+1 are issued into staging registers
// BEFORE the MFMAs of tile t and written to the other buffer after them, so HBM/L2 latency hides under
// ~12k cycles of matrix work and there is one barrier per tile.
template <int NVT>
__global__ __launch_bounds__(256, 1) void k_cross_attn(
    int nq, int nk, const float *__restrict__ Q, int ldq, const float *__restrict__ K, int ldk,
    const float *__restrict__ V1, int dv1, int ldv1, float *__restrict__ O1, int ldo1,
    const float *__restrict__ V2, int dv2, int ldv2, float *__restrict__ O2, int ldo2) {
  constexpr int T = 256;
  constexpr int DV = NVT * 32;
  constexpr int KT = 32 * kKPitch;            // floats of one K tile  [32 keys][kKPitch]
  constexpr int TILE = KT + 32 * DV;          // + V tile [32 keys][DV]
  constexpr int NVS = (32 * (DV / 4) + T - 1) / T;   // V float4 per thread per tile
  extern __shared__ float attn_lds[];         // [2][TILE]
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int q = blockIdx.x * 128 + wave * 32 + r;
  const bool qlive = q < nq;

  // B operand of S: MFMA step s contracts channels {s, 32+s}; lane half h holds channel 32h+s.
  float Qreg[32];
  {
    const float4 *qp = reinterpret_cast<const float4 *>(Q + ((size_t)b * nq + (qlive ? q : 0)) * ldq + 32 * h);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float4 v = qp[i];
      if (!qlive) v = make_float4(0.f, 0.f, 0.f, 0.f);
      Qreg[4 * i] = v.x; Qreg[4 * i + 1] = v.y; Qreg[4 * i + 2] = v.z; Qreg[4 * i + 3] = v.w;
    }
  }
  f32x16 O[NVT];
#pragma unroll
  for (int t = 0; t < NVT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[t][e] = 0.0f;
  float m_ref = -INFINITY, l_part = 0.0f;

  static_assert((32 * (DV / 4)) % T == 0, "V tile must split evenly over the workgroup");
  // Rows past nk are fetched from the last valid row instead of being zero-filled (branch-free loads): their
  // scores are masked to -inf below, so their P is exactly 0 and finite V rows contribute nothing.
  float4 kst[2], vst[NVS];
  const int last_key = nk - 1;
  auto stage_load = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * T, key = min(kb + (idx >> 4), last_key), c4 = (idx & 15) * 4;
      kst[i] = *reinterpret_cast<const float4 *>(K + ((size_t)b * nk + key) * ldk + c4);
    }
#pragma unroll
    for (int i = 0; i < NVS; ++i) {
      const int idx = tid + i * T, kk = idx / (DV / 4), c4 = (idx - kk * (DV / 4)) * 4;
      const size_t row = (size_t)b * nk + min(kb + kk, last_key);
      const float *src = c4 < dv1 ? V1 + row * ldv1 + c4 : V2 + row * ldv2 + (c4 - dv1);
      vst[i] = *reinterpret_cast<const float4 *>(src);
    }
  };
  auto stage_write = [&](float *buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * T, key = idx >> 4, c4 = (idx & 15) * 4;
      *reinterpret_cast<float4 *>(buf + key * kKPitch + c4) = kst[i];
    }
#pragma unroll
    for (int i = 0; i < NVS; ++i) {
      const int idx = tid + i * T, key = idx / (DV / 4), c4 = (idx - key * (DV / 4)) * 4;
      *reinterpret_cast<float4 *>(buf + KT + key * DV + c4) = vst[i];
    }
  };

  stage_load(0);
  stage_write(attn_lds);
  __syncthreads();
  int cur = 0;
  for (int kb = 0; kb < nk; kb += 32) {
    const bool more = kb + 32 < nk;
    if (more) stage_load(kb + 32);                       // in flight during this tile's MFMAs
    const float *Ks = attn_lds + cur * TILE;
    const float *Vs = Ks + KT;

    // ---- S = K Q^T over 64 channels: A[i=key][k] = K[key][32h+s] ----
    f32x16 S;
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] = 0.0f;
    {
      const float *krow = Ks + r * kKPitch + 32 * h;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 a = *reinterpret_cast<const float4 *>(krow + 4 * i);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, Qreg[4 * i + 0], S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, Qreg[4 * i + 1], S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, Qreg[4 * i + 2], S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, Qreg[4 * i + 3], S, 0, 0, 0);
      }
    }

    // ---- online softmax over keys (column = this lane's query) ----
    float m_tile = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (kb + rowmap(e, h) >= nk) S[e] = -INFINITY;
      m_tile = fmaxf(m_tile, S[e]);
    }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    if (__ballot(m_tile > m_ref + kThr) != 0ull) {       // rare, wave-uniform
      const float m_new = fmaxf(m_ref, m_tile);
      const float f = __expf(m_ref - m_new);             // exp(-inf) = 0 on the first tile
      l_part *= f;
#pragma unroll
      for (int t = 0; t < NVT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[t][e] *= f;
      m_ref = m_new;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      S[e] = __expf(S[e] - m_ref);
      l_part += S[e];
    }

    // ---- O += V^T P : register e of S is the B operand for key rowmap(e,h); A[i=c][k] = V[key][c] ----
#pragma unroll
    for (int t = 0; t < NVT; ++t) {
      const float *vcol = Vs + t * 32 + r + 4 * h * DV;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        O[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vcol[((e & 3) + 8 * (e >> 2)) * DV], S[e], O[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                  // keep the LDS reads of tile t+1.. from piling up in VGPRs
    }
    if (more) stage_write(attn_lds + (cur ^ 1) * TILE);
    __syncthreads();
    cur ^= 1;
  }

  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");     // MFMA -> VALU read of the accumulators
  const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
  if (qlive) {
    const size_t row = (size_t)b * nq + q;
#pragma unroll
    for (int t = 0; t < NVT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = t * 32 + 8 * g + 4 * h;            // 4 consecutive channels: e = 4g .. 4g+3
        float4 v;
        v.x = O[t][4 * g] / l_tot; v.y = O[t][4 * g + 1] / l_tot;
        v.z = O[t][4 * g + 2] / l_tot; v.w = O[t][4 * g + 3] / l_tot;
        if (c < dv1) *reinterpret_cast<float4 *>(O1 + row * ldo1 + c) = v;
        else *reinterpret_cast<float4 *>(O2 + row * ldo2 + (c - dv1)) = v;
      }
  }
}

// Variant for LARGE problems: 8 waves (256 queries) share one single-buffered K/V tile; the wave's Q rows are
// parked in LDS so that two waves fit each SIMD and one wave's softmax / LDS phase overlaps the other's MFMAs.
template <int WAVES, int NVT>
__global__ __launch_bounds__(WAVES * 64, WAVES >= 8 ? 2 : 1) void k_cross_attn_shared(
    int nq, int nk, const float *__restrict__ Q, int ldq, const float *__restrict__ K, int ldk,
    const float *__restrict__ V1, int dv1, int ldv1, float *__restrict__ O1, int ldo1,
    const float *__restrict__ V2, int dv2, int ldv2, float *__restrict__ O2, int ldo2) {
  constexpr int T = WAVES * 64;
  constexpr int DV = NVT * 32;
  extern __shared__ float attn_lds[];
  float *Ks = attn_lds;                       // [32 keys][kKPitch]
  float *Vs = attn_lds + 32 * kKPitch;        // [32 keys][DV]
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int q = blockIdx.x * (WAVES * 32) + wave * 32 + r;
  const bool qlive = q < nq;

  // B operand of S: MFMA step s contracts channels {s, 32+s}; lane half h uses channel 32h+s.
  // The wave's 32 query rows are parked in LDS (same padded layout as the K tile) to keep the
  // register budget at 2 waves/SIMD.
  float *Qs = Vs + 32 * DV + wave * 32 * kKPitch;
  for (int i = lane; i < 32 * 16; i += 64) {
    const int qr = i >> 4, c4 = (i & 15) * 4;
    const int qq = blockIdx.x * (WAVES * 32) + wave * 32 + qr;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qq < nq) v = *reinterpret_cast<const float4 *>(Q + ((size_t)b * nq + qq) * ldq + c4);
    *reinterpret_cast<float4 *>(Qs + qr * kKPitch + c4) = v;
  }
  f32x16 O[NVT];
#pragma unroll
  for (int t = 0; t < NVT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[t][e] = 0.0f;
  float m_ref = -INFINITY, l_part = 0.0f;

  for (int kb = 0; kb < nk; kb += 32) {
    __syncthreads();
    // ---- stage 32 key rows: K (64 ch) and [V1|V2] (DV ch); rows past nk are zero ----
    for (int i = tid; i < 32 * 16; i += T) {
      const int key = i >> 4, c4 = (i & 15) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kb + key < nk) v = *reinterpret_cast<const float4 *>(K + ((size_t)b * nk + kb + key) * ldk + c4);
      *reinterpret_cast<float4 *>(Ks + key * kKPitch + c4) = v;
    }
    for (int i = tid; i < 32 * (DV / 4); i += T) {
      const int key = i / (DV / 4), c4 = (i - key * (DV / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kb + key < nk) {
        const size_t row = (size_t)b * nk + kb + key;
        v = c4 < dv1 ? *reinterpret_cast<const float4 *>(V1 + row * ldv1 + c4)
                     : *reinterpret_cast<const float4 *>(V2 + row * ldv2 + (c4 - dv1));
      }
      *reinterpret_cast<float4 *>(Vs + key * DV + c4) = v;
    }
    __syncthreads();

    // ---- S = K^T Q over 64 channels: A[i=key][k] = K[key][32h+s] ----
    f32x16 S;
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] = 0.0f;
    {
      const float *krow = Ks + r * kKPitch + 32 * h;
      const float *qrow = Qs + r * kKPitch + 32 * h;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 a = *reinterpret_cast<const float4 *>(krow + 4 * i);
        const float4 qv = *reinterpret_cast<const float4 *>(qrow + 4 * i);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qv.x, S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qv.y, S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qv.z, S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qv.w, S, 0, 0, 0);
      }
    }

    // ---- online softmax over keys (column = this lane's query) ----
    float m_tile = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (kb + rowmap(e, h) >= nk) S[e] = -INFINITY;
      m_tile = fmaxf(m_tile, S[e]);
    }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    if (__ballot(m_tile > m_ref + kThr) != 0ull) {       // rare, wave-uniform
      const float m_new = fmaxf(m_ref, m_tile);
      const float f = __expf(m_ref - m_new);             // exp(-inf) = 0 on the first tile
      l_part *= f;
#pragma unroll
      for (int t = 0; t < NVT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[t][e] *= f;
      m_ref = m_new;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      S[e] = __expf(S[e] - m_ref);
      l_part += S[e];
    }

    // ---- O += V^T P : register e of S is the B operand for key rowmap(e,h); A[i=c][k] = V[key][c] ----
#pragma unroll
    for (int t = 0; t < NVT; ++t) {
      const float *vcol = Vs + t * 32 + r + 4 * h * DV;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        O[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vcol[((e & 3) + 8 * (e >> 2)) * DV], S[e], O[t], 0, 0, 0);
    }
  }

  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");     // MFMA -> VALU read of the accumulators
  const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
  if (qlive) {
    const size_t row = (size_t)b * nq + q;
#pragma unroll
    for (int t = 0; t < NVT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = t * 32 + 8 * g + 4 * h;            // 4 consecutive channels: e = 4g .. 4g+3
        float4 v;
        v.x = O[t][4 * g] / l_tot; v.y = O[t][4 * g + 1] / l_tot;
        v.z = O[t][4 * g + 2] / l_tot; v.w = O[t][4 * g + 3] / l_tot;
        if (c < dv1) *reinterpret_cast<float4 *>(O1 + row * ldo1 + c) = v;
        else *reinterpret_cast<float4 *>(O2 + row * ldo2 + (c - dv1)) = v;
      }
  }
}
#endif  // DCL_DIAG

// Variant for LARGE problems with DCL-Net's own channel split (V1 = 256 ch, V2 = 64 ch): the shared-tile kernel above
// plus an asynchronous tile pipeline.  The V tile (40 of the 48 KiB per 32 keys) is double-buffered in LDS and filled
// by LDS-DMA (global_load_lds_dwordx4: no staging registers; each wave-instruction lands 1 KiB = one V1 row or four
// V2 rows), issued right after the barrier that retires the buffer's previous readers, so it has a whole tile of MFMA
// work (~10 us) to land.  K (8 KiB) goes through one float4 register per thread, loaded before P.V and written after it.
// Two barriers per tile as before, but no global-memory latency between them:
//   A (__syncthreads: drains this wave's DMA + K write)  ->  issue DMA V(t+1)  ->  S = K Q^T, softmax
//   B (raw s_barrier, lgkmcnt only: K tile free)         ->  load K(t+1)       ->  O += V^T P  ->  write K(t+1)
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int kAttnPartPitch = 324;                // floats per (key split, query) partial record: 320 channels, m, l, pad

// One LDS-DMA wave-instruction: 64 lanes x 16 B from per-lane global addresses to LDS at (wave-uniform) lds_byte_addr +
// lane*16.  Inline asm on purpose: issued through the builtin, hipcc drains it (vmcnt(0)) before the next ds_read of the
// same LDS array, which would serialise the pipeline; an asm load is not in the compiler's counters, so the kernel waits
// for it itself (s_waitcnt vmcnt(0) before barrier A).  M0 is saved/restored inside the statement (guide section 5.7).
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_byte_addr)
               : "memory");
}
// the same with the source as (wave-uniform base in an SGPR pair) + (per-lane 32-bit byte offset): no vector address arithmetic
__device__ __forceinline__ void glds16_s(unsigned voff, const void *sbase, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const float *p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t *)p);
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k_cross_attn_dma(
    int nq, int nk, const float *__restrict__ Q, int ldq, const float *__restrict__ K, int ldk,
    const float *__restrict__ V1, int ldv1, float *__restrict__ O1, int ldo1,
    const float *__restrict__ V2, int ldv2, float *__restrict__ O2, int ldo2, float *__restrict__ part, int xcd_remap) {
  constexpr int NVT = 10;
  constexpr int KT = 32 * kKPitch;                 // K tile floats
  constexpr int V1T = 32 * 256, V2T = 32 * 64;     // per-buffer floats
  extern __shared__ float attn_lds[];              // [K][V1 x2][V2 x2][Q x8]  (one array: see guide, LDS-DMA traps)
  float *Ks = attn_lds;
  float *V1s = Ks + KT;
  float *V2s = V1s + 2 * V1T;
  // XCD-aware placement: consecutive workgroup ids are dealt round-robin over the 8 XCDs (each with its own L2), so the
  // query blocks of one crop -- which all stream the same K/V -- would land on 8 different L2s.  Renumber bijectively so
  // that the workgroups sharing an XCD walk consecutive (query block, crop) ids: a crop's K/V tiles are then fetched once
  // per XCD group instead of once per query block (a pure speed / traffic choice, never a correctness one).
  int bx, b;
  {
    const int nwg = gridDim.x * gridDim.y, id = blockIdx.x + gridDim.x * blockIdx.y;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int swz = !xcd_remap ? id : (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
    bx = swz % (int)gridDim.x;
    b = swz / (int)gridDim.x;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  constexpr int QB = WAVES * 32;                   // queries per workgroup
  const int q = bx * QB + wave * 32 + r;
  const bool qlive = q < nq;
  float *Qs = V2s + 2 * V2T + wave * 32 * kKPitch;
  for (int i = lane; i < 32 * 16; i += 64) {
    const int qr = i >> 4, c4 = (i & 15) * 4;
    const int qq = bx * QB + wave * 32 + qr;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qq < nq) v = *reinterpret_cast<const float4 *>(Q + ((size_t)b * nq + qq) * ldq + c4);
    *reinterpret_cast<float4 *>(Qs + qr * kKPitch + c4) = v;
  }
  const int last_key = nk - 1;
  const size_t krow0 = (size_t)b * nk;
  // this wave's share of a V tile: RPW = 32/WAVES V1 rows (one 1-KiB DMA each) and the same V2 rows (4 rows per DMA)
  constexpr int RPW = 32 / WAVES;
  auto dma_v = [&](int kb, int buf) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int key = wave * RPW + i;
      const float *src = V1 + (krow0 + min(kb + key, last_key)) * ldv1 + lane * 4;
      glds16(src, lds_addr_of(V1s + buf * V1T + key * 256));
    }
#pragma unroll
    for (int i = 0; i < RPW / 4; ++i) {
      const int key2 = wave * RPW + i * 4 + (lane >> 4);
      const float *src2 = V2 + (krow0 + min(kb + key2, last_key)) * ldv2 + (lane & 15) * 4;
      glds16(src2, lds_addr_of(V2s + buf * V2T + (wave * RPW + i * 4) * 64));
    }
  };
  // this thread's float4(s) of the K tile: 512 per tile over WAVES*64 threads
  constexpr int KPT = 512 / (WAVES * 64);
  auto k_slot = [&](int i, int &key, int &c4) { const int idx = tid + i * WAVES * 64; key = idx >> 4; c4 = (idx & 15) * 4; };
  float4 knext0 = make_float4(0.f, 0.f, 0.f, 0.f), knext1 = knext0;       // named registers: no stack object
  auto load_k = [&](int kb) {
    int key, c4;
    k_slot(0, key, c4);
    knext0 = *reinterpret_cast<const float4 *>(K + (krow0 + min(kb + key, last_key)) * ldk + c4);
    if (KPT > 1) {
      k_slot(1, key, c4);
      knext1 = *reinterpret_cast<const float4 *>(K + (krow0 + min(kb + key, last_key)) * ldk + c4);
    }
  };
  auto store_k = [&]() {
    int key, c4;
    k_slot(0, key, c4);
    *reinterpret_cast<float4 *>(Ks + key * kKPitch + c4) = knext0;
    if (KPT > 1) {
      k_slot(1, key, c4);
      *reinterpret_cast<float4 *>(Ks + key * kKPitch + c4) = knext1;
    }
  };

  f32x16 O[NVT];
#pragma unroll
  for (int t = 0; t < NVT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[t][e] = 0.0f;
  float m_ref = -INFINITY, l_part = 0.0f;

  // key split (gridDim.z > 1): this workgroup owns a contiguous range of 32-key tiles and leaves unnormalised partial
  // sums + (running max, weight sum) per query for k_cross_attn_combine -- fills the GPU when b * nq/128 alone cannot
  const int ntiles = (nk + 31) >> 5;
  const int kb_begin = (int)((long long)blockIdx.z * ntiles / gridDim.z) * 32;
  const int kb_end = min((int)((long long)(blockIdx.z + 1) * ntiles / gridDim.z) * 32, nk);
  load_k(kb_begin);
  store_k();
  dma_v(kb_begin, 0);
  int cur = 0;
  for (int kb = kb_begin; kb < kb_end; kb += 32) {
    const bool more = kb + 32 < kb_end;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this wave's DMA pieces of tile t have landed (a
                                                           // builtin, so hipcc's own counters see the drain too)
    __syncthreads();                                       // A
    if (more) dma_v(kb + 32, cur ^ 1);

    f32x16 S;
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] = 0.0f;
    {
      const float *krow = Ks + r * kKPitch + 32 * h;
      const float *qrow = Qs + r * kKPitch + 32 * h;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 a = *reinterpret_cast<const float4 *>(krow + 4 * i);
        const float4 qv = *reinterpret_cast<const float4 *>(qrow + 4 * i);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qv.x, S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qv.y, S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qv.z, S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qv.w, S, 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0) only: the DMA stays in flight
    __builtin_amdgcn_s_barrier();                          // B: every wave is done with the K tile
    if (more) load_k(kb + 32);

    float m_tile = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (kb + rowmap(e, h) >= nk) S[e] = -INFINITY;
      m_tile = fmaxf(m_tile, S[e]);
    }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    if (__ballot(m_tile > m_ref + kThr) != 0ull) {
      const float m_new = fmaxf(m_ref, m_tile);
      const float f = __expf(m_ref - m_new);
      l_part *= f;
#pragma unroll
      for (int t = 0; t < NVT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[t][e] *= f;
      m_ref = m_new;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      S[e] = __expf(S[e] - m_ref);
      l_part += S[e];
    }

    const float *v1b = V1s + cur * V1T + 4 * h * 256 + r;
    const float *v2b = V2s + cur * V2T + 4 * h * 64 + r;
#pragma unroll
    for (int t = 0; t < NVT; ++t) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int krow = (e & 3) + 8 * (e >> 2);
        const float a = t < 8 ? v1b[krow * 256 + t * 32] : v2b[krow * 64 + (t - 8) * 32];
        O[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, S[e], O[t], 0, 0, 0);
      }
    }
    if (more) store_k();
    cur ^= 1;
  }

  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
  if (qlive && gridDim.z > 1) {
    // partial record of this key range: [320 unnormalised channels | m | l] per (split, query)
    const size_t rows = (size_t)gridDim.y * nq;
    float *P = part + ((size_t)blockIdx.z * rows + (size_t)b * nq + q) * kAttnPartPitch;
#pragma unroll
    for (int t = 0; t < NVT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4 *>(P + t * 32 + 8 * g + 4 * h) =
            make_float4(O[t][4 * g], O[t][4 * g + 1], O[t][4 * g + 2], O[t][4 * g + 3]);
    if (h == 0) { P[320] = m_ref; P[321] = l_tot; }
  } else if (qlive) {
    const size_t row = (size_t)b * nq + q;
#pragma unroll
    for (int t = 0; t < NVT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = t * 32 + 8 * g + 4 * h;
        float4 v;
        v.x = O[t][4 * g] / l_tot; v.y = O[t][4 * g + 1] / l_tot;
        v.z = O[t][4 * g + 2] / l_tot; v.w = O[t][4 * g + 3] / l_tot;
        if (c < 256) *reinterpret_cast<float4 *>(O1 + row * ldo1 + c) = v;
        else *reinterpret_cast<float4 *>(O2 + row * ldo2 + (c - 256)) = v;
      }
  }
}

// ---- the same attention with BOTH products on the bf16 matrix pipe at fp32-sized errors --------------------------------------
// (the scheme of linear_split.hip: every fp32 operand = the exact sum of three bf16 pieces, a product = its six piece products of
//  weight >= 2^-16, fp32 accumulators).  K and V are the shared operands: piece passes (k_attn_split_k / _v; V1's 256 channels can
// come straight from the GEMM that computes them: dcl_linear_split_vpieces_fwd) write each crop's rows as pieces in exactly the
// order the MFMA's A operand wants them -- V per 16-key half tile (lane (r, h) of channel block t: the eight keys (e & 3) + 8 (e >> 2)
// + 4 h of the half tile for channel 32 t + r, the key order the S accumulators already have; 30 KiB per half tile), K per 32-key
// tile (12 KiB) -- so a tile is linear LDS-DMA copies and a fragment ONE ds_read_b128.  Q (per-wave, 32 queries) is read as fp32 and
// split in registers every tile: as pieces it would take 96 KiB of LDS or 48 registers.  P = exp(S - m) is split in registers once
// per tile and wave; the S accumulators become the B operand of P.V as before, piece by piece.  Per 32-key tile and wave: 24 + 120
// bf16 MFMAs (4608 cycles) against 192 fp32 MFMAs (12288).  The sweep's structure is described at k_cross_attn_split.
typedef __bf16 at_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 at_bf16x8 __attribute__((ext_vector_type(8)));
typedef float at_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned at_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAttnHalfBytes = 3 * 320 * 2 * 16;   // one 16-key half tile of V pieces: [piece][channel][lane half][8 keys] bf16 = 30 KiB
constexpr int kAttnKTileBytes = 3 * 32 * 128;      // one 32-key tile of K pieces: [piece][key][8 groups of 8 channels] bf16 = 12 KiB
constexpr int kAttnTileBytes = 2 * kAttnHalfBytes + kAttnKTileBytes;     // scratch per 32-key tile and crop

__device__ __forceinline__ unsigned at_cvt2(float a, float b) {
  const at_f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, at_bf16x2));
}
__device__ __forceinline__ void at_split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {     // x = h + m + l exactly
  // (the empty asm statements keep a pair's subtractions scalar: see linear_split.hip, sp_split2)
  h = at_cvt2(x0, x1);
  float r0 = x0 - __uint_as_float(h << 16);
  asm volatile("" : "+v"(r0));
  float r1 = x1 - __uint_as_float(h & 0xffff0000u);
  asm volatile("" : "+v"(r1));
  m = at_cvt2(r0, r1);
  float s0 = r0 - __uint_as_float(m << 16);
  asm volatile("" : "+v"(s0));
  float s1 = r1 - __uint_as_float(m & 0xffff0000u);
  asm volatile("" : "+v"(s1));
  l = at_cvt2(s0, s1);
}
__device__ __forceinline__ at_bf16x8 at_bf(at_u32x4 v) { return __builtin_bit_cast(at_bf16x8, v); }

// planes[crop][half tile][piece][channel c][slot hs][8] (bf16), slot hs holds lane half h = hs ^ bit 3 of c (bank swizzle), element
// e = key 16 ht + (e & 3) + 8 (e >> 2) + 4 h; keys >= nk are zeros.  One thread per (crop, half tile, c, hs).
__global__ __launch_bounds__(256) void k_attn_split_v(int nk, int nht, const float *__restrict__ V1, int ldv1, const float *__restrict__ V2,
                                                      int ldv2, unsigned *__restrict__ planes, long long total, int c_begin) {
  // (c_begin = 256: the V1 channels' pieces are already there, written by the GEMM that made V1 -- dcl_linear_split_vpieces_fwd)
  const int nc = 320 - c_begin;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int hs = (int)(i & 1);
    const long long q = i >> 1;
    const int c = c_begin + (int)(q % nc);
    const long long cht = q / nc;                          // crop * nht + half tile
    const int ht = (int)(cht % nht), crop = (int)(cht / nht);
    const int h = hs ^ ((c >> 3) & 1);
    const float *col = c < 256 ? V1 + c : V2 + (c - 256);
    const int ld = c < 256 ? ldv1 : ldv2;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int key = 16 * ht + (e & 3) + 8 * (e >> 2) + 4 * h;
      v[e] = key < nk ? col[((size_t)crop * nk + key) * ld] : 0.0f;
    }
    unsigned ph[4], pm[4], pl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) at_split2(v[2 * e], v[2 * e + 1], ph[e], pm[e], pl[e]);
    unsigned *dst = planes + (size_t)cht * (kAttnHalfBytes / 4) + (size_t)(c * 2 + hs) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { dst[e] = ph[e]; dst[320 * 2 * 4 + e] = pm[e]; dst[2 * 320 * 2 * 4 + e] = pl[e]; }
  }
}

// K's pieces: kplanes[crop][32-key tile][piece][key][slot ps][8 channels] (bf16), slot ps holds channel group p = ps ^ ((key >> 1) & 7)
// (bank swizzle: the A-operand fragment of lane (key, h) in k step s is group 2 s + h).  One thread per (crop, tile, key, ps).
__global__ __launch_bounds__(256) void k_attn_split_k(int nk, int ntiles, const float *__restrict__ K, int ldk, unsigned *__restrict__ kplanes,
                                                      long long total) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ps = (int)(i & 7), key = (int)((i >> 3) & 31);
    const long long ct = i >> 8;                           // crop * ntiles + tile
    const int tile = (int)(ct % ntiles), crop = (int)(ct / ntiles);
    const int p = ps ^ ((key >> 1) & 7), kk = tile * 32 + key;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (kk < nk) {
      const float *row = K + ((size_t)crop * nk + kk) * ldk + 8 * p;
      a = *reinterpret_cast<const float4 *>(row);
      b = *reinterpret_cast<const float4 *>(row + 4);
    }
    unsigned ph[4], pm[4], pl[4];
    at_split2(a.x, a.y, ph[0], pm[0], pl[0]);
    at_split2(a.z, a.w, ph[1], pm[1], pl[1]);
    at_split2(b.x, b.y, ph[2], pm[2], pl[2]);
    at_split2(b.z, b.w, ph[3], pm[3], pl[3]);
    unsigned *dst = kplanes + (size_t)ct * (kAttnKTileBytes / 4) + (size_t)(key * 8 + ps) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) { dst[e] = ph[e]; dst[32 * 32 + e] = pm[e]; dst[2 * 32 * 32 + e] = pl[e]; }
  }
}

// The sweep.  Its two halves per tile have opposite appetites -- S + softmax + the splits are vector work (~400 instructions and 24
// MFMAs per wave), P.V is 120 MFMAs and their fragment reads -- and a SIMD holds two of the workgroup's eight waves: wave w and wave
// w + 4.  Waves 0-3 and waves 4-7 therefore run HALF A TILE APART: in every barrier interval one group does S / softmax of a tile while
// the other does P.V (group 0: S(t) in interval 2 t, P.V(t) in 2 t + 1; group 1 one interval later), so a SIMD's matrix pipe and
// vector pipe can work at the same time instead of taking turns.  One barrier per interval.  K pieces (12 KiB per 32-key tile) and
// V pieces (60 KiB) are double-buffered: tile t + 1 of both is fetched at the start of interval 2 t + 1 into the buffers tile t - 1
// left in intervals 2 t - 1 / 2 t.  That fills the LDS (144 KiB), so a wave reads its 32 query rows from global memory (L2) every
// tile instead of keeping them in LDS: 8 KiB per wave and tile.
__global__ __launch_bounds__(512, 2) void k_cross_attn_split(
    int nq, int nk, const float *__restrict__ Q, int ldq, const unsigned char *__restrict__ planes,
    const unsigned char *__restrict__ kplanes, float *__restrict__ O1, int ldo1, float *__restrict__ O2, int ldo2,
    float *__restrict__ part, int xcd_remap, int whatif) {
  constexpr int NVT = 10, WAVES = 8;
  constexpr int VT = 2 * kAttnHalfBytes;           // V pieces of a 32-key tile
  extern __shared__ float attn_lds[];              // [K pieces x2][V pieces x2]
  unsigned char *Kp = reinterpret_cast<unsigned char *>(attn_lds);
  unsigned char *Vp = Kp + 2 * kAttnKTileBytes;
  int bx, b;
  {
    const int nwg = gridDim.x * gridDim.y, id = blockIdx.x + gridDim.x * blockIdx.y;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = id & 7;
    const int swz = !xcd_remap ? id : (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (id >> 3);
    bx = swz % (int)gridDim.x;
    b = swz / (int)gridDim.x;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                       // 0: S in even intervals;  1: S in odd intervals
  const int r = lane & 31, h = lane >> 5;
  constexpr int QB = WAVES * 32;
  const int q = bx * QB + wave * 32 + r;
  const bool qlive = q < nq;
  // this lane's query row (B operand of S: lane (query, h) = channels 16 s + 8 h ..); rows past nq read the last query (never stored)
  const float *qrow = Q + ((size_t)b * nq + min(q, nq - 1)) * ldq + 8 * h;
  const int nht = 2 * ((nk + 31) >> 5);
  const unsigned char *vsrc = planes + (size_t)b * nht * kAttnHalfBytes;             // (wave-uniform: SGPRs)
  const unsigned char *ksrc = kplanes + (size_t)b * (nht >> 1) * kAttnKTileBytes;
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned vp0 = lds_addr_of(reinterpret_cast<const float *>(Vp));
  const unsigned kp0 = lds_addr_of(reinterpret_cast<const float *>(Kp));
  // a tile = 12 one-KiB DMA pieces of K pieces + 60 of V pieces: wave w issues K piece w (and w + 8 if w < 4), V pieces w, w + 8, ...
  // (sources = scalar bases + the one per-lane offset: no vector address arithmetic beside the other group's MFMAs)
  auto dma_tile = [&](int tile, int buf) {
    const unsigned char *ks = ksrc + (size_t)tile * kAttnKTileBytes + wave * 1024;
    const unsigned kd = kp0 + (unsigned)(buf * kAttnKTileBytes + wave * 1024);
    glds16_s(lane16, ks, kd);
    if (wave < 4) glds16_s(lane16, ks + 8 * 1024, kd + 8 * 1024);
    const unsigned char *vs = vsrc + (size_t)tile * VT + wave * 1024;
    const unsigned vd = vp0 + (unsigned)(buf * VT + wave * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (wave + 8 * i < 60) glds16_s(lane16, vs + i * 8192, vd + (unsigned)(i * 8192));
  };

  f32x16 O[NVT];
#pragma unroll
  for (int t = 0; t < NVT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[t][e] = 0.0f;
  float m_ref = -INFINITY, l_part = 0.0f;
  at_u32x4 pp[2][3];                               // the current tile's weights as bf16 pieces (S interval -> P.V interval)
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int p = 0; p < 3; ++p) pp[hf][p] = at_u32x4{0u, 0u, 0u, 0u};

  const int ntiles_all = (nk + 31) >> 5;
  const int t_begin = (int)((long long)blockIdx.z * ntiles_all / gridDim.z);
  const int t_end = (int)((long long)(blockIdx.z + 1) * ntiles_all / gridDim.z);
  const int T = t_end - t_begin;                   // this workgroup's tiles (key split: a contiguous range)
  // fragment addresses in buffer 0: V piece 0 of channel block 0 = slot (h ^ bit 3 of r) of channel r; K = key r's row of channel
  // groups, group 2 s + h in slot (2 s + h) ^ ((r >> 1) & 7)
  const unsigned char *vfrag = Vp + ((r * 2 + (h ^ ((r >> 3) & 1))) << 4);
  const unsigned char *kfrag = Kp + r * 128;
  const int ksw = (r >> 1) & 7;

  // one interval's work of a wave, as two inlined pieces: S / softmax / weights of local tile tl, and P.V of local tile tl
  auto s_phase = [&](int tl) __attribute__((always_inline)) {
    const int buf = tl & 1, kb = (t_begin + tl) * 32;
    // ---- S = K Q^T (64 deep = four k steps, six piece products each, smallest first), softmax, the weights' pieces
    f32x16 S;
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] = 0.0f;
    const unsigned char *kfb = kfrag + buf * kAttnKTileBytes;
    float4 qn0 = *reinterpret_cast<const float4 *>(qrow), qn1 = *reinterpret_cast<const float4 *>(qrow + 4);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const float4 qc0 = qn0, qc1 = qn1;                   // (the next step's eight query values are fetched one step ahead)
      if (st < 3) {
        qn0 = *reinterpret_cast<const float4 *>(qrow + 16 * (st + 1));
        qn1 = *reinterpret_cast<const float4 *>(qrow + 16 * (st + 1) + 4);
      }
      const unsigned char *kf = kfb + (((2 * st + h) ^ ksw) << 4);
      const at_u32x4 kh = *reinterpret_cast<const at_u32x4 *>(kf);
      const at_u32x4 km = *reinterpret_cast<const at_u32x4 *>(kf + 32 * 128);
      const at_u32x4 kl = *reinterpret_cast<const at_u32x4 *>(kf + 2 * 32 * 128);
      at_u32x4 qh, qm, ql;
      unsigned a0, a1, a2;
      at_split2(qc0.x, qc0.y, a0, a1, a2); qh[0] = a0; qm[0] = a1; ql[0] = a2;
      at_split2(qc0.z, qc0.w, a0, a1, a2); qh[1] = a0; qm[1] = a1; ql[1] = a2;
      at_split2(qc1.x, qc1.y, a0, a1, a2); qh[2] = a0; qm[2] = a1; ql[2] = a2;
      at_split2(qc1.z, qc1.w, a0, a1, a2); qh[3] = a0; qm[3] = a1; ql[3] = a2;
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(kl), at_bf(qh), S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(kh), at_bf(ql), S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(km), at_bf(qm), S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(km), at_bf(qh), S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(kh), at_bf(qm), S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(kh), at_bf(qh), S, 0, 0, 0);
    }
    float m_tile = -INFINITY;
    if (kb + 32 > nk) {                                    // (wave-uniform) the ragged last tile: keys past nk score -inf
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (kb + rowmap(e, h) >= nk) S[e] = -INFINITY;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) m_tile = fmaxf(m_tile, S[e]);
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    if (__ballot(m_tile > m_ref + kThr) != 0ull) {
      const float m_new = fmaxf(m_ref, m_tile);
      const float f = __expf(m_ref - m_new);
      l_part *= f;
#pragma unroll
      for (int t = 0; t < NVT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) O[t][e] *= f;
      m_ref = m_new;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      S[e] = __expf(S[e] - m_ref);
      l_part += S[e];
    }
    // half tile A = accumulators 0..7, B = 8..15 (lane half h: keys (e & 3) + 8 (e >> 2) + 4 h)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned ph, pm, pl;
        at_split2(S[8 * hf + 2 * e], S[8 * hf + 2 * e + 1], ph, pm, pl);
        pp[hf][0][e] = ph; pp[hf][1][e] = pm; pp[hf][2][e] = pl;
      }
  };
  auto pv_phase = [&](int tl) __attribute__((always_inline)) {
    // ---- O += V^T P, both half tiles: per channel block and half tile three fragments, six piece products
    const unsigned char *vbuf = vfrag + (tl & 1) * VT;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const unsigned char *vb = vbuf + hf * kAttnHalfBytes;
#pragma unroll
      for (int t = 0; t < NVT; ++t) {
        const at_u32x4 vh = *reinterpret_cast<const at_u32x4 *>(vb + t * 1024);
        const at_u32x4 vm = *reinterpret_cast<const at_u32x4 *>(vb + 320 * 32 + t * 1024);
        const at_u32x4 vl = *reinterpret_cast<const at_u32x4 *>(vb + 2 * 320 * 32 + t * 1024);
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(vl), at_bf(pp[hf][0]), O[t], 0, 0, 0);
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(vh), at_bf(pp[hf][2]), O[t], 0, 0, 0);
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(vm), at_bf(pp[hf][1]), O[t], 0, 0, 0);
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(vm), at_bf(pp[hf][0]), O[t], 0, 0, 0);
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(vh), at_bf(pp[hf][1]), O[t], 0, 0, 0);
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_bf(vh), at_bf(pp[hf][0]), O[t], 0, 0, 0);
      }
    }
  };
  auto interval = [&](int iv) __attribute__((always_inline)) {      // the boundary in front of interval iv
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): this wave's DMA pieces issued an interval ago have landed
    __syncthreads();                                       // everyone's have; everyone is done with the previous interval's reads
    if ((iv & 1) && (iv + 1) / 2 < T && !(whatif & 4)) dma_tile(t_begin + (iv + 1) / 2, ((iv + 1) / 2) & 1);
  };
  if (T > 0) dma_tile(t_begin, 0);
  // (two copies of a straight-line loop rather than one loop with the phase as a branch: with both phases under branches of one loop
  //  body the register allocator spilled 400+ registers)
  if (grp == 0) {
    for (int tl = 0; tl < T; ++tl) {
      interval(2 * tl);
      if (!(whatif & 2)) s_phase(tl);
      interval(2 * tl + 1);
      if (!(whatif & 1)) pv_phase(tl);
    }
    interval(2 * T);
  } else {
    interval(0);
    for (int tl = 0; tl < T; ++tl) {
      interval(2 * tl + 1);
      if (!(whatif & 2)) s_phase(tl);
      interval(2 * tl + 2);
      if (!(whatif & 1)) pv_phase(tl);
    }
  }

  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  const float l_tot = l_part + __shfl_xor(l_part, 32, 64);
  if (qlive && gridDim.z > 1) {
    const size_t rows = (size_t)gridDim.y * nq;
    float *P = part + ((size_t)blockIdx.z * rows + (size_t)b * nq + q) * kAttnPartPitch;
#pragma unroll
    for (int t = 0; t < NVT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4 *>(P + t * 32 + 8 * g + 4 * h) =
            make_float4(O[t][4 * g], O[t][4 * g + 1], O[t][4 * g + 2], O[t][4 * g + 3]);
    if (h == 0) { P[320] = m_ref; P[321] = l_tot; }
  } else if (qlive) {
    const size_t row = (size_t)b * nq + q;
#pragma unroll
    for (int t = 0; t < NVT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = t * 32 + 8 * g + 4 * h;
        float4 v;
        v.x = O[t][4 * g] / l_tot; v.y = O[t][4 * g + 1] / l_tot;
        v.z = O[t][4 * g + 2] / l_tot; v.w = O[t][4 * g + 3] / l_tot;
        if (c < 256) *reinterpret_cast<float4 *>(O1 + row * ldo1 + c) = v;
        else *reinterpret_cast<float4 *>(O2 + row * ldo2 + (c - 256)) = v;
      }
  }
}

// out[row][c] = sum_z e^{m_z - m} O_z[c] / sum_z e^{m_z - m} l_z, m = max_z m_z (splits in index order); thread = 4 channels
__global__ void k_cross_attn_combine(int rows, int nsplit, const float *__restrict__ part, float *__restrict__ O1, int ldo1,
                                     float *__restrict__ O2, int ldo2) {
  const long long total = (long long)rows * 80;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const size_t row = (size_t)(t / 80);
    const int c = (int)(t - (long long)row * 80) * 4;
    float m = -INFINITY;
    for (int z = 0; z < nsplit; ++z) m = fmaxf(m, part[((size_t)z * rows + row) * kAttnPartPitch + 320]);
    float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
    float den = 0.0f;
    for (int z = 0; z < nsplit; ++z) {
      const float *P = part + ((size_t)z * rows + row) * kAttnPartPitch;
      const float f = __expf(P[320] - m);
      const float4 o = *reinterpret_cast<const float4 *>(P + c);
      num.x += f * o.x; num.y += f * o.y; num.z += f * o.z; num.w += f * o.w;
      den += f * P[321];
    }
    const float4 v = make_float4(num.x / den, num.y / den, num.z / den, num.w / den);
    if (c < 256) *reinterpret_cast<float4 *>(O1 + row * ldo1 + c) = v;
    else *reinterpret_cast<float4 *>(O2 + row * ldo2 + (c - 256)) = v;
  }
}

// ---- confidence pooling (models/DCL_Net.py:217-228) ------------------------------------------------
// conf = sigmoid(cat[logit1 (b,n1), logit2 (b,n2)]); w = softmax(conf) over L = n1+n2;
// pooled1[c] = sum_{j<n1} w_j F1[j][c], pooled2[c] = sum_{j<n2} w_{n1+j} F2[j][c] (F point-major),
// wsum1/wsum2 = the two partial weight sums (for applying each side's trailing BatchNorm affine
// after pooling: sum_j w_j (s*x_j + t) = s*pooled + t*wsum).
// Two kernels: (1) one block per crop: sigmoid, max, exp, sum -> w (b,L) + conf + wsum;
// (2) HBM-streaming weighted column sums: block = 256 channels (float4 per lane, 1 KiB per wave per point row)
// x one slice of the point axis; slice partials are combined in a fixed order by the caller (deterministic).
// NT threads per crop: 256 in dcl_conf_pool (the one-launch kernel below repeats that order bit for bit), 1024 in dcl_conf_softmax
// (the forward whose last fuser layer pools: the softmax sits on its critical path, and 14336 logits per crop are 56 strided
// passes of a 256-thread workgroup)
template <int NT>
__device__ __forceinline__ float conf_block_reduce(float v, float *red, int lane, int wave, bool is_max) {
  constexpr int NW = NT / 64;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = is_max ? fmaxf(v, __shfl_xor(v, d, 64)) : v + __shfl_xor(v, d, 64);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r;
  if (NW == 4) {
    r = is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
  } else {
    float t[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) t[i] = red[i];
#pragma unroll
    for (int step = 1; step < NW; step <<= 1)
#pragma unroll
      for (int i = 0; i + step < NW; i += 2 * step) t[i] = is_max ? fmaxf(t[i], t[i + step]) : t[i] + t[i + step];
    r = t[0];
  }
  __syncthreads();
  return r;
}
template <int NT>
__global__ __launch_bounds__(NT) void k_conf_softmax(int n1, int n2, const float *__restrict__ logit1,
                                                     const float *__restrict__ logit2, float *__restrict__ conf,
                                                     float *__restrict__ w, float *__restrict__ wsum) {
  __shared__ float red[NT / 64];
  const int L = n1 + n2;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  float mx = -INFINITY;
  for (int j = tid; j < L; j += NT) {
    const float x = j < n1 ? logit1[(size_t)b * n1 + j] : logit2[(size_t)b * n2 + (j - n1)];
    const float s = 1.0f / (1.0f + expf(-x));
    conf[(size_t)b * L + j] = s;
    mx = fmaxf(mx, s);
  }
  mx = conf_block_reduce<NT>(mx, red, lane, wave, true);
  float sum = 0.0f;
  for (int j = tid; j < L; j += NT) {
    const float e = expf(conf[(size_t)b * L + j] - mx);      // own writes: same thread wrote conf[j]
    w[(size_t)b * L + j] = e;
    sum += e;
  }
  sum = conf_block_reduce<NT>(sum, red, lane, wave, false);
  const float inv = 1.0f / sum;
  float w1 = 0.f, w2 = 0.f;
  for (int j = tid; j < L; j += NT) {
    const float v = w[(size_t)b * L + j] * inv;
    w[(size_t)b * L + j] = v;
    if (j < n1) w1 += v; else w2 += v;
  }
  w1 = conf_block_reduce<NT>(w1, red, lane, wave, false);
  w2 = conf_block_reduce<NT>(w2, red, lane, wave, false);
  if (tid == 0) { wsum[b * 2] = w1; wsum[b * 2 + 1] = w2; }
}

// part[b][slice][c] = sum over this slice's points of w[j] * F[j][c]
// (both directions in one launch: blockIdx.y = direction * nslices + slice)
__global__ __launch_bounds__(256) void k_weighted_colsum(int c, int n1, int n2, int nslices, const float *__restrict__ w,
                                                         const float *__restrict__ F1, int ld1, float *__restrict__ part1,
                                                         const float *__restrict__ F2, int ld2, float *__restrict__ part2) {
  __shared__ float4 red[4][64];
  const int second = (int)blockIdx.y >= nslices ? 1 : 0;
  const int b = blockIdx.z, slice = blockIdx.y - second * nslices;
  const int n = second ? n2 : n1, ld = second ? ld2 : ld1, w_stride = n1 + n2, w_off = second ? n1 : 0;
  const float *__restrict__ F = second ? F2 : F1;
  float *__restrict__ part = second ? part2 : part1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ch = blockIdx.x * 256 + lane * 4;
  const int per = (n + nslices - 1) / nslices;
  const int j0 = slice * per, j1 = min(n, j0 + per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ch < c) {
    const float *wp = w + (size_t)b * w_stride + w_off;
    const float *fp = F + (size_t)b * n * ld + ch;
    for (int j = j0 + wave; j < j1; j += 4) {
      const float wj = wp[j];
      const float4 f = *reinterpret_cast<const float4 *>(fp + (size_t)j * ld);
      acc.x = __fmaf_rn(f.x, wj, acc.x); acc.y = __fmaf_rn(f.y, wj, acc.y);
      acc.z = __fmaf_rn(f.z, wj, acc.z); acc.w = __fmaf_rn(f.w, wj, acc.w);
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && ch < c) {
    float4 o;
    o.x = (red[0][lane].x + red[1][lane].x) + (red[2][lane].x + red[3][lane].x);
    o.y = (red[0][lane].y + red[1][lane].y) + (red[2][lane].y + red[3][lane].y);
    o.z = (red[0][lane].z + red[1][lane].z) + (red[2][lane].z + red[3][lane].z);
    o.w = (red[0][lane].w + red[1][lane].w) + (red[2][lane].w + red[3][lane].w);
    *reinterpret_cast<float4 *>(part + ((size_t)b * nslices + slice) * c + ch) = o;
  }
}

// Both kernels above in one launch for calls of a handful of crops (L = n1 + n2 <= 256 * EPT logits per crop): EVERY workgroup of
// a crop forms the crop's softmax itself -- k_conf_softmax's operations in k_conf_softmax's order, values in registers, the
// normalised weights into LDS -- and then runs k_weighted_colsum's loop on them; the workgroup (0, 0, crop) also writes conf and
// wsum.  2048 logits are 16 transcendentals per thread: cheaper than a launch and its gap on a path that has nothing else to do.
template <int EPT>
__global__ __launch_bounds__(256) void k_conf_pool_small(int c, int n1, int n2, int nslices, const float *__restrict__ logit1,
                                                         const float *__restrict__ logit2, float *__restrict__ conf,
                                                         float *__restrict__ wsum, const float *__restrict__ F1, int ld1,
                                                         float *__restrict__ part1, const float *__restrict__ F2, int ld2,
                                                         float *__restrict__ part2) {
  __shared__ float wl[256 * EPT];
  __shared__ float redf[4], r1[4], r2[4];
  __shared__ float4 red[4][64];
  const int L = n1 + n2;
  const int second = (int)blockIdx.y >= nslices ? 1 : 0;
  const int b = blockIdx.z, slice = blockIdx.y - second * nslices;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
  float sv[EPT];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int j = tid + 256 * k;
    if (j < L) {
      const float x = j < n1 ? logit1[(size_t)b * n1 + j] : logit2[(size_t)b * n2 + (j - n1)];
      sv[k] = 1.0f / (1.0f + expf(-x));
      if (writer) conf[(size_t)b * L + j] = sv[k];
      mx = fmaxf(mx, sv[k]);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
  if (lane == 0) redf[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
  __syncthreads();
  float sum = 0.0f;
#pragma unroll
  for (int k = 0; k < EPT; ++k)
    if (tid + 256 * k < L) {
      sv[k] = expf(sv[k] - mx);
      sum += sv[k];
    }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
  if (lane == 0) redf[wave] = sum;
  __syncthreads();
  sum = (redf[0] + redf[1]) + (redf[2] + redf[3]);
  const float inv = 1.0f / sum;
  float w1 = 0.f, w2 = 0.f;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int j = tid + 256 * k;
    if (j < L) {
      const float v = sv[k] * inv;
      wl[j] = v;
      if (j < n1) w1 += v; else w2 += v;
    }
  }
  if (writer) {                                        // (block-uniform)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { w1 += __shfl_xor(w1, d, 64); w2 += __shfl_xor(w2, d, 64); }
    if (lane == 0) { r1[wave] = w1; r2[wave] = w2; }
  }
  __syncthreads();
  if (writer && tid == 0) { wsum[b * 2] = (r1[0] + r1[1]) + (r1[2] + r1[3]); wsum[b * 2 + 1] = (r2[0] + r2[1]) + (r2[2] + r2[3]); }
  // ---- k_weighted_colsum's body, the weights from LDS
  const int n = second ? n2 : n1, ld = second ? ld2 : ld1, w_off = second ? n1 : 0;
  const float *__restrict__ F = second ? F2 : F1;
  float *__restrict__ part = second ? part2 : part1;
  const int ch = blockIdx.x * 256 + lane * 4;
  const int per = (n + nslices - 1) / nslices;
  const int j0 = slice * per, j1 = min(n, j0 + per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ch < c) {
    const float *wp = wl + w_off;
    const float *fp = F + (size_t)b * n * ld + ch;
    for (int j = j0 + wave; j < j1; j += 4) {
      const float wj = wp[j];
      const float4 f = *reinterpret_cast<const float4 *>(fp + (size_t)j * ld);
      acc.x = __fmaf_rn(f.x, wj, acc.x); acc.y = __fmaf_rn(f.y, wj, acc.y);
      acc.z = __fmaf_rn(f.z, wj, acc.z); acc.w = __fmaf_rn(f.w, wj, acc.w);
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && ch < c) {
    float4 o;
    o.x = (red[0][lane].x + red[1][lane].x) + (red[2][lane].x + red[3][lane].x);
    o.y = (red[0][lane].y + red[1][lane].y) + (red[2][lane].y + red[3][lane].y);
    o.z = (red[0][lane].z + red[1][lane].z) + (red[2][lane].z + red[3][lane].z);
    o.w = (red[0][lane].w + red[1][lane].w) + (red[2][lane].w + red[3][lane].w);
    *reinterpret_cast<float4 *>(part + ((size_t)b * nslices + slice) * c + ch) = o;
  }
}

// ---- ortho9d2matrix (models/DCL_Net.py:15-36) ------------------------------------------------------
// R = U diag(1,1,det(U V^T)) V^T of the column-stacked, normalised raw vectors: one thread per crop,
// one-sided Jacobi SVD in fp64 (the reference calls a batched LAPACK/MAGMA gesdd, ms-scale latency).
// With A V = U Sigma:  R = u1 v1^T + u2 v2^T + det(V) (u1 x u2) v3^T   (sign-ambiguity free).
__device__ void ortho9d_one(const float *__restrict__ o9, float *__restrict__ R, int i) {
  double A[3][3], V[3][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float x = o9[i * 9 + c * 3], y = o9[i * 9 + c * 3 + 1], z = o9[i * 9 + c * 3 + 2];
    const float mag = sqrtf((x * x + y * y) + z * z) + 1e-8f;       // utils/transform3D.py:18-20 (fp32)
    A[0][c] = (double)(x / mag); A[1][c] = (double)(y / mag); A[2][c] = (double)(z / mag);
#pragma unroll
    for (int r = 0; r < 3; ++r) V[r][c] = r == c ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) { alpha += A[r][p] * A[r][p]; beta += A[r][q] * A[r][q]; gamma += A[r][p] * A[r][q]; }
        off = fmax(off, fabs(gamma) / sqrt(alpha * beta + 1e-300));
        if (fabs(gamma) < 1e-300) continue;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const double ap = A[r][p], aq = A[r][q];
          A[r][p] = cs * ap - sn * aq; A[r][q] = sn * ap + cs * aq;
          const double vp = V[r][p], vq = V[r][q];
          V[r][p] = cs * vp - sn * vq; V[r][q] = sn * vp + cs * vq;
        }
      }
    if (off < 1e-15) break;
  }
  // sort the three (column of A, column of V) pairs by descending singular value with explicit swaps -- no
  // dynamically indexed local arrays, so the kernel needs no scratch memory
  double a0[3] = {A[0][0], A[1][0], A[2][0]}, a1[3] = {A[0][1], A[1][1], A[2][1]}, a2[3] = {A[0][2], A[1][2], A[2][2]};
  double v0[3] = {V[0][0], V[1][0], V[2][0]}, v1[3] = {V[0][1], V[1][1], V[2][1]}, v2[3] = {V[0][2], V[1][2], V[2][2]};
  double s0 = sqrt(a0[0] * a0[0] + a0[1] * a0[1] + a0[2] * a0[2]);
  double s1 = sqrt(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]);
  double s2 = sqrt(a2[0] * a2[0] + a2[1] * a2[1] + a2[2] * a2[2]);
#define DCL_SWAP_COLS(sa, sb, aa, ab, va, vb)                                  \
  if (sa < sb) {                                                               \
    double t_ = sa; sa = sb; sb = t_;                                          \
    for (int r_ = 0; r_ < 3; ++r_) {                                           \
      t_ = aa[r_]; aa[r_] = ab[r_]; ab[r_] = t_;                               \
      t_ = va[r_]; va[r_] = vb[r_]; vb[r_] = t_;                               \
    }                                                                          \
  }
  DCL_SWAP_COLS(s0, s1, a0, a1, v0, v1)
  DCL_SWAP_COLS(s1, s2, a1, a2, v1, v2)
  DCL_SWAP_COLS(s0, s1, a0, a1, v0, v1)
#undef DCL_SWAP_COLS
  double u1[3], u2[3], u3[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) { u1[r] = a0[r] / s0; u2[r] = a1[r] / s1; }
  u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
  u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
  u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
  const double detV = v0[0] * (v1[1] * v2[2] - v1[2] * v2[1]) - v0[1] * (v1[0] * v2[2] - v1[2] * v2[0]) +
                      v0[2] * (v1[0] * v2[1] - v1[1] * v2[0]);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      R[i * 9 + r * 3 + c] = (float)(u1[r] * v0[c] + u2[r] * v1[c] + detV * u3[r] * v2[c]);
}
__global__ void k_ortho9d(int b, const float *__restrict__ o9, float *__restrict__ R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b) ortho9d_one(o9, R, i);
}

}  // namespace

// (switches: atomics in the diagnostic library, constants in the product -- common.h)
DCL_HOOK_INT(g_attn_variant, 0);     // 1 = shared-tile 8-wave kernel, 2 = 4-wave register-staged, 3 / 4 = LDS-DMA pipeline with 8 / 4 waves
DCL_HOOK_INT(g_attn_split, 0);       // 0 = automatic key split of small attention launches, n = force n
DCL_HOOK_INT(g_attn_xcd_remap, 1);   // 0 = plain blockIdx order (for traffic comparisons)
#ifdef DCL_DIAG
DCL_API void dcl_debug_attention_split(int n) { g_attn_split = n; }
DCL_API void dcl_debug_attention_xcd_remap(int on) { g_attn_xcd_remap = on; }
DCL_API void dcl_debug_attention_variant(int v) { g_attn_variant = v; }
#endif

DCL_API int dcl_cross_attention(int b, int nq, int nk, const float *Q, int ldq, const float *K, int ldk,
                                const float *V1, int dv1, int ldv1, float *O1, int ldo1, const float *V2, int dv2,
                                int ldv2, float *O2, int ldo2, dclStream_t stream) {
  return dcl_cross_attention_ws(b, nq, nk, Q, ldq, K, ldk, V1, dv1, ldv1, O1, ldo1, V2, dv2, ldv2, O2, ldo2, nullptr, 0,
                                stream);
}

// crops of a PAIR of launches (the two directions side by side) that fill whole rounds of 8-wave workgroups: 256 workgroups per
// round for the pair = 128 / ceil(nq / 256) crops of each direction (32 at nq = 1024); 0 = no such split for this call
DCL_HOOK_INT(g_attn_whatif, 0);      // (diagnostic library: what-if runs of k_cross_attn_split -- 1: no P.V phase, 2: no S phase, 4: no DMA after the first tile)
DCL_HOOK_INT(g_attn_bf16, 1);        // (diagnostic library: dcl_debug_attention_bf16; 0 = fp32 MFMA everywhere)
static bool attn_split_big(long long blocks8, int nk) { return blocks8 >= 64 && blocks8 * dcl_div_up(nk, 32) >= 16384; }
DCL_HOOK_INT(g_attn_pair_split, 1);  // (diagnostic library: dcl_debug_attention_pair_split; 0 = off)
// (nk > 0: the call has scratch for the split-bf16 kernel -- a call that kernel takes as a whole is not cut into rounds + rest: the
//  rest would run on the fp32 4-wave kernel, 2.7 ms for 8 of 24 stress-shape crops)
static int attn_pair_full_crops(int b, int nq, int dv1, int dv2, int concurrent_launches, int nk = 0) {
  if (!g_attn_pair_split || g_attn_variant != 0 || concurrent_launches != 2 || dv1 != 256 || dv2 != 64 || nq <= 0) return 0;
  const int qb8 = dcl_div_up(nq, 256);
  if (nk > 0 && g_attn_bf16 && attn_split_big((long long)b * qb8, nk)) return 0;
  if (qb8 > 128 || 128 % qb8 != 0) return 0;
  const int per_round = 128 / qb8;
  const int full = b / per_round * per_round;
  return full > 0 && full < b ? full : 0;
}
#ifdef DCL_DIAG
DCL_API void dcl_debug_attention_pair_split(int on) { g_attn_pair_split = on; }
#endif

// Key split of a 4-wave attention launch (fewer than 256 eight-wave workgroups): see the launcher below for the model.
static int attn_small_split(int b, int nq, int ntiles) {
  const int forced = g_attn_split;
  if (forced > 0) return forced > 16 ? 16 : forced;
  const long long T = 2ll * b * dcl_div_up(nq, 128);
  const double tiles_us = 1.1 * ntiles;
  double best = 1e30;
  int nsplit = 1;
  for (int z = 1; z <= 16; z *= 2) {
    const double cost = (double)dcl_div_up(T * z, 256) * (tiles_us / z + 4.0) + (z > 1 ? 4.0 + 0.15 * z * (double)b * nq / 1024.0 : 0.0);
    if (cost < best - 1e-9) { best = cost; nsplit = z; }
  }
  if (nsplit > ntiles / 2) nsplit = ntiles / 2;
  return nsplit < 1 ? 1 : nsplit;
}

DCL_API int dcl_cross_attention_scratch_floats(int b, int nq, int64_t *floats_host) {
  // upper bound of what dcl_cross_attention_ws can use for (b, nq): small launches by their split model (up to 16 ways),
  // large ones only while the records stay below 128 Mi floats
  DCL_CHECK_ARG(b >= 0 && nq >= 0 && floats_host);
  const long long per = (long long)b * nq * kAttnPartPitch;
  const long long blocks8 = (long long)b * dcl_div_up(nq > 0 ? nq : 1, 256);
  long long z = 1;
  if (blocks8 < 256) {                                     // (the key count is not known here: the most any count would take)
    for (int ntiles = 2; ntiles <= (1 << 16); ntiles *= 2) z = max(z, (long long)attn_small_split(b, nq, ntiles));
  } else {
    z = 1;
    double best = (double)dcl_div_up(blocks8, 256);
    for (int c = 2; c <= 8; c *= 2) {
      const double cost = (double)dcl_div_up(blocks8 * c, 256) / c;
      if (cost <= best - 0.2 && c * per <= (128ll << 20)) { best = cost; z = c; }
    }
  }
  long long need = z > 1 ? z * per : 0;
  // a pair call that is split into whole rounds + a rest (dcl_cross_attention_ws2): the rest is a call of its own size
  const int full = attn_pair_full_crops(b, nq, 256, 64, 2);
  if (full) {
    int64_t rest = 0;
    const int rc = dcl_cross_attention_scratch_floats(b - full, nq, &rest);
    if (rc) return rc;
    if (rest > need) need = rest;
  }
  *floats_host = need;
  return 0;
}

DCL_API int dcl_cross_attention_ws(int b, int nq, int nk, const float *Q, int ldq, const float *K, int ldk,
                                   const float *V1, int dv1, int ldv1, float *O1, int ldo1, const float *V2, int dv2,
                                   int ldv2, float *O2, int ldo2, float *scratch, int64_t scratch_floats,
                                   dclStream_t stream) {
  return dcl_cross_attention_ws2(b, nq, nk, Q, ldq, K, ldk, V1, dv1, ldv1, O1, ldo1, V2, dv2, ldv2, O2, ldo2, scratch,
                                 scratch_floats, 1, stream);
}

static int attn_dispatch(int b, int nq, int nk, const float *Q, int ldq, const float *K, int ldk, const float *V1, int dv1, int ldv1,
                         float *O1, int ldo1, const float *V2, int dv2, int ldv2, float *O2, int ldo2, float *scratch,
                         int64_t scratch_floats, int concurrent_launches, void *planes, int64_t planes_bytes, dclStream_t stream);

// does a call of this size take the 8-wave workgroup form (see attn_dispatch), the one the split-bf16 kernel exists for?
// (the split kernel is worth its 8-wave workgroups on fewer than 256 of them too when the key axis is long: the key split fills the
//  chip, and the fp32 4-wave kernel it would fall back to is 1.5x slower per flop -- 24 crops of 2048 queries x 12288 keys: 192
//  workgroups, whole forward 15.55 -> 14.0 ms.  The bound: four times the work of a lone 32 x 1024 x 1024 launch, where the two
//  kernels tie.)
static bool attn_takes_w8(int b, int nq, int nk, int concurrent_launches) {
  const long long blocks8 = (long long)b * dcl_div_up(nq, 256);
  const bool pair8 = concurrent_launches == 2 && 2 * blocks8 > 240 && 2 * blocks8 <= 256;
  return g_attn_variant == 3 || (g_attn_variant == 0 && (blocks8 >= 256 || pair8 || attn_split_big(blocks8, nk)));
}
#ifdef DCL_DIAG
DCL_API void dcl_debug_attention_bf16(int on) { g_attn_bf16 = on; }
DCL_API void dcl_debug_attention_whatif(int bits) { g_attn_whatif = bits; }
#endif

DCL_API int dcl_cross_attention_split_crops(int b, int nq, int nk, int concurrent_launches) {
  // how many of the b crops take the split-bf16 kernel when `planes` are handed in: 0, all of them, or the whole rounds of a pair call
  if (b <= 0 || nq <= 0 || nk <= 0 || !g_attn_bf16) return 0;
  int crops = b;
  const int full = attn_pair_full_crops(b, nq, 256, 64, concurrent_launches, nk);
  if (full) crops = full;
  return attn_takes_w8(crops, nq, nk, concurrent_launches) ? crops : 0;
}

DCL_API int64_t dcl_cross_attention_planes_bytes(int b, int nq, int nk, int concurrent_launches) {
  if (b <= 0 || nq <= 0 || nk <= 0 || !g_attn_bf16) return 0;
  int crops = b;
  const int full = attn_pair_full_crops(b, nq, 256, 64, concurrent_launches, nk);
  if (full) crops = full;                                  // (the rest of such a call runs as a small call: 4-wave workgroups)
  if (!attn_takes_w8(crops, nq, nk, concurrent_launches)) return 0;
  return (int64_t)crops * dcl_div_up(nk, 32) * kAttnTileBytes;
}

DCL_API int dcl_cross_attention_ws2(int b, int nq, int nk, const float *Q, int ldq, const float *K, int ldk,
                                    const float *V1, int dv1, int ldv1, float *O1, int ldo1, const float *V2, int dv2,
                                    int ldv2, float *O2, int ldo2, float *scratch, int64_t scratch_floats,
                                    int concurrent_launches, dclStream_t stream) {
  return dcl_cross_attention_ws3(b, nq, nk, Q, ldq, K, ldk, V1, dv1, ldv1, O1, ldo1, V2, dv2, ldv2, O2, ldo2, scratch, scratch_floats,
                                 concurrent_launches, nullptr, 0, stream);
}

DCL_API int dcl_cross_attention_ws3(int b, int nq, int nk, const float *Q, int ldq, const float *K, int ldk,
                                    const float *V1, int dv1, int ldv1, float *O1, int ldo1, const float *V2, int dv2,
                                    int ldv2, float *O2, int ldo2, float *scratch, int64_t scratch_floats,
                                    int concurrent_launches, void *planes, int64_t planes_bytes, dclStream_t stream) {
  DCL_CHECK_ARG(concurrent_launches == 1 || concurrent_launches == 2);
  // A pair of launches whose 8-wave workgroups make one or more WHOLE rounds of the chip plus a rest (40 crops of 1024 x 1024: 1.25
  // rounds): the whole rounds go as they are (8-wave, two waves per SIMD), the rest as the call of that many crops that it is
  // (4-wave workgroups, keys split) -- a quarter-filled last round costs a whole one.  Same results per crop.
  const int full = (b > 0 && Q && K && O1) ? attn_pair_full_crops(b, nq, dv1, dv2, concurrent_launches, planes ? nk : 0) : 0;
  DCL_CHECK_ARG(!(full && !V1));                           // (V1 = NULL is for calls that take the split kernel as a whole: dcl_cross_attention_split_crops)
  if (full) {
    int rc = attn_dispatch(full, nq, nk, Q, ldq, K, ldk, V1, dv1, ldv1, O1, ldo1, V2, dv2, ldv2, O2, ldo2, scratch, scratch_floats,
                           concurrent_launches, planes, planes_bytes, stream);
    if (rc) return rc;
    const size_t qo = (size_t)full * nq, ko = (size_t)full * nk;
    return attn_dispatch(b - full, nq, nk, Q + qo * ldq, ldq, K + ko * ldk, ldk, V1 + ko * ldv1, dv1, ldv1, O1 + qo * ldo1, ldo1,
                         V2 ? V2 + ko * ldv2 : nullptr, dv2, ldv2, O2 ? O2 + qo * ldo2 : nullptr, ldo2, scratch, scratch_floats,
                         concurrent_launches, nullptr, 0, stream);
  }
  return attn_dispatch(b, nq, nk, Q, ldq, K, ldk, V1, dv1, ldv1, O1, ldo1, V2, dv2, ldv2, O2, ldo2, scratch, scratch_floats,
                       concurrent_launches, planes, planes_bytes, stream);
}

static int attn_dispatch(int b, int nq, int nk, const float *Q, int ldq, const float *K, int ldk, const float *V1, int dv1, int ldv1,
                         float *O1, int ldo1, const float *V2, int dv2, int ldv2, float *O2, int ldo2, float *scratch,
                         int64_t scratch_floats, int concurrent_launches, void *planes, int64_t planes_bytes, dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && nq >= 0 && nk > 0 && dv1 > 0 && dv1 % 32 == 0 && dv2 >= 0 && dv2 % 32 == 0);
  if (b == 0 || nq == 0) return 0;
  DCL_CHECK_ARG(Q && K && (V1 || planes) && O1 && (dv2 == 0 || (V2 && O2)) && b <= 65535);
  DCL_CHECK_ARG(ldq >= 64 && ldk >= 64 && ldv1 >= dv1 && ldo1 >= dv1 && (dv2 == 0 || (ldv2 >= dv2 && ldo2 >= dv2)));
  DCL_CHECK_ARG(ldq % 4 == 0 && ldk % 4 == 0 && ldv1 % 4 == 0 && ldo1 % 4 == 0 && ldv2 % 4 == 0 && ldo2 % 4 == 0);
  DCL_CHECK_ARG((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V1 | (uintptr_t)O1 | (uintptr_t)V2 | (uintptr_t)O2) & 15) == 0);
  const bool v1_in_planes = V1 == nullptr;                 // (only the split kernel can take that: checked where it is chosen)
  const int nvt = (dv1 + dv2) / 32;
  DCL_CHECK_ARG(nvt == 1 || nvt == 2 || nvt == 4 || nvt == 8 || nvt == 10);
  hipStream_t s = (hipStream_t)stream;
  // Large grids: 8-wave workgroups sharing a K/V tile (2 waves/SIMD).  Small grids (fewer than one 8-wave
  // workgroup per CU): 4-wave workgroups with double-buffered tiles, one per CU.
  const long long blocks8 = (long long)b * dcl_div_up(nq, 256);
  if (dv1 == 256 && dv2 == 64 && (g_attn_variant == 0 || g_attn_variant == 3 || g_attn_variant == 4)) {
    // LDS-DMA pipeline: 8 waves (256 queries) per workgroup when that fills the chip, else 4 waves (128 queries)
    // Two launches side by side (the two directions of a forward on parallel branches) whose 8-wave workgroups TOGETHER make
    // one round of the chip -- 32 crops of 1024 x 1024, the shipped configuration -- also take the 8-wave form, unsplit: the
    // 4-wave workgroups (402 registers, one wave per SIMD) of the two launches cannot share a CU, so they run one after the
    // other at one wave per SIMD; same-job A/B of the whole forward, 8-wave vs 4-wave: 32 crops 3.813 vs 3.880 ms; 28 (224
    // workgroups) 3.706 vs 3.696, 36: 4.72 vs 4.69, 40: 4.97 vs 4.97, 24: 3.20 vs 3.15, 16: 2.31 vs 2.16 -- hence the window.
    const bool pair8 = concurrent_launches == 2 && 2 * blocks8 > 240 && 2 * blocks8 <= 256;
    const bool split_usable = planes && g_attn_bf16 && (((uintptr_t)planes) & 15) == 0 &&
                              planes_bytes >= (int64_t)b * dcl_div_up(nk, 32) * kAttnTileBytes;
    const bool w8 = g_attn_variant == 3 ||
                    (g_attn_variant == 0 && (blocks8 >= 256 || pair8 || (split_usable && attn_split_big(blocks8, nk))));
    const int W = w8 ? 8 : 4;
    const size_t lds = (size_t)(32 * kKPitch + 2 * 32 * 256 + 2 * 32 * 64 + W * 32 * kKPitch) * sizeof(float);
    if (w8) {
      // wave quantisation: blocks8 workgroups run in ceil(blocks8/256) rounds of one per CU.  When the last round is
      // mostly empty (e.g. 320 workgroups = 2 rounds for 1.25 rounds of work) a key split of Z makes the rounds Z times
      // shorter: cost(Z) = ceil(blocks8*Z/256)/Z full-workgroup times; taken when it saves >= 0.2 of one and the partial
      // records stay below ~512 MiB
      int nsplit = 1;
      if (scratch && !(pair8 && g_attn_split == 0)) {
        if (g_attn_split > 0) {
          nsplit = g_attn_split;
        } else {
          double best = (double)dcl_div_up(blocks8, 256);
          for (int z = 2; z <= 8; z *= 2) {
            const double cost = (double)dcl_div_up(blocks8 * z, 256) / z;
            if (cost <= best - 0.2 && (long long)z * b * nq * kAttnPartPitch <= (128ll << 20)) { best = cost; nsplit = z; }
          }
        }
        const int ntiles = dcl_div_up(nk, 32);
        if (nsplit > 8) nsplit = 8;
        if (nsplit > ntiles / 2) nsplit = ntiles / 2;
        while (nsplit > 1 && (long long)nsplit * b * nq * kAttnPartPitch > scratch_floats) --nsplit;
        if (nsplit < 1) nsplit = 1;
      }
      const int nht = 2 * dcl_div_up(nk, 32);
      const int64_t planes_need = (int64_t)b * (nht / 2) * kAttnTileBytes;
      if (planes && planes_bytes >= planes_need && g_attn_bf16 && (((uintptr_t)planes) & 15) == 0) {
        // P.V on the bf16 matrix pipe at fp32-sized errors: V as three exact bf16 pieces in tile order (one pass), then the sweep
        const int c_begin = V1 ? 0 : 256;                  // V1 == NULL: its pieces are in `planes` already
        const long long total = (long long)b * nht * (320 - c_begin) * 2;
        hipLaunchKernelGGL(k_attn_split_v, dim3(dcl_grid_1d(total, 256)), dim3(256), 0, s, nk, nht, V1, ldv1, V2, ldv2,
                           (unsigned *)planes, total, c_begin);
        unsigned char *kplanes = (unsigned char *)planes + (size_t)b * nht * kAttnHalfBytes;
        const long long ktotal = (long long)b * (nht / 2) * 32 * 8;
        hipLaunchKernelGGL(k_attn_split_k, dim3(dcl_grid_1d(ktotal, 256)), dim3(256), 0, s, nk, nht / 2, K, ldk, (unsigned *)kplanes, ktotal);
        const size_t lds_sp = (size_t)2 * (2 * kAttnHalfBytes + kAttnKTileBytes);
        (void)hipFuncSetAttribute((const void *)k_cross_attn_split, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp);
        hipLaunchKernelGGL(k_cross_attn_split, dim3(dcl_div_up(nq, 256), b, nsplit), dim3(512), lds_sp, s, nq, nk, Q, ldq,
                           (const unsigned char *)planes, (const unsigned char *)kplanes, O1, ldo1, O2, ldo2, scratch,
                           (int)g_attn_xcd_remap, (int)g_attn_whatif);
      } else {
        if (v1_in_planes) {
          dcl_set_error("dcl_cross_attention: V1 = NULL (pieces in `planes`) but this call does not take the split-bf16 kernel");
          return DCL_EINVAL;
        }
        (void)hipFuncSetAttribute((const void *)k_cross_attn_dma<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_cross_attn_dma<8>, dim3(dcl_div_up(nq, 256), b, nsplit), dim3(512), lds, s, nq, nk, Q, ldq, K, ldk,
                           V1, ldv1, O1, ldo1, V2, ldv2, O2, ldo2, scratch, (int)g_attn_xcd_remap);
      }
      if (nsplit > 1)
        hipLaunchKernelGGL(k_cross_attn_combine, dim3(dcl_grid_1d((long long)b * nq * 80, 256)), dim3(256), 0, s, b * nq,
                           nsplit, scratch, O1, ldo1, O2, ldo2);
    } else {
      if (v1_in_planes) {
        dcl_set_error("dcl_cross_attention: V1 = NULL (pieces in `planes`) but this call does not take the split-bf16 kernel");
        return DCL_EINVAL;
      }
      // few workgroups (small batches): split the keys over up to 16 workgroups per query block, >= 2 tiles per split
      int nsplit = 1;
      if (scratch) {
        // Key split of a 4-wave launch (one workgroup per CU): a power of two z (the 32 key tiles of a 1024-key crop divide
        // evenly; 3 or 6 splits measured 2-4 % slower than 2 or 4) by a small time model in microseconds,
        //     rounds(z) * (tiles * 1.1 / z + 4)  +  (z > 1 ? 4 + 0.15 * z * b * nq / 1024 : 0),
        // rounds(z) = ceil(T z / 256) with T = 2 x blocks -- the two directions of a call run side by side (parallel branches
        // of the whole-forward graph, the path every call this small takes); the second term is the combine launch.  Fitted
        // to same-job A/B runs on the whole forward (tools/ab_hook.py dcl_debug_attention_split), best z at 1 / 2 / 4 / 6 / 8 /
        // 12 / 16 / 20 / 32 crops of 1024 x 1024: 16 / 8 / 4 / 2 / 2 / 1 / 1 / 2 / 1 -- what this picks (one crop: 16 splits of two
        // tiles against 8 of four, whole forward 0.414 vs 0.419 ms; two crops 0.547 vs 0.527: not there); against the former rule
        // (fill 256 workgroups per launch) 4 crops -3 %, 6: -1.8 %, 8: -1.3 %, 12: -5.4 %, 16: -1.6 %, 20: -2.5 %.
        nsplit = attn_small_split(b, nq, dcl_div_up(nk, 32));
        const int ntiles = dcl_div_up(nk, 32);
        if (nsplit > 16) nsplit = 16;
        if (nsplit > ntiles / 2) nsplit = ntiles / 2;
        while (nsplit > 1 && (long long)nsplit * b * nq * kAttnPartPitch > scratch_floats) --nsplit;
        if (nsplit < 1) nsplit = 1;
      }
      (void)hipFuncSetAttribute((const void *)k_cross_attn_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k_cross_attn_dma<4>, dim3(dcl_div_up(nq, 128), b, nsplit), dim3(256), lds, s, nq, nk, Q, ldq, K,
                         ldk, V1, ldv1, O1, ldo1, V2, ldv2, O2, ldo2, scratch, (int)g_attn_xcd_remap);
      if (nsplit > 1)
        hipLaunchKernelGGL(k_cross_attn_combine, dim3(dcl_grid_1d((long long)b * nq * 80, 256)), dim3(256), 0, s, b * nq,
                           nsplit, scratch, O1, ldo1, O2, ldo2);
    }
  }
#ifdef DCL_DIAG
  else if ((blocks8 >= 256 && g_attn_variant != 2) || g_attn_variant == 1) {
    const size_t lds = (size_t)(32 * kKPitch + 32 * nvt * 32 + 8 * 32 * kKPitch) * sizeof(float);
#define ATT8(N)                                                                                                \
  do {                                                                                                         \
    (void)hipFuncSetAttribute((const void *)k_cross_attn_shared<8, N>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                       \
    hipLaunchKernelGGL((k_cross_attn_shared<8, N>), dim3(dcl_div_up(nq, 256), b), dim3(512), lds, s, nq, nk, Q, \
                       ldq, K, ldk, V1, dv1, ldv1, O1, ldo1, V2, dv2, ldv2, O2, ldo2);                         \
  } while (0)
    switch (nvt) {
      case 1: ATT8(1); break;
      case 2: ATT8(2); break;
      case 4: ATT8(4); break;
      case 8: ATT8(8); break;
      default: ATT8(10); break;
    }
#undef ATT8
  } else {
    const size_t lds = (size_t)2 * (32 * kKPitch + 32 * nvt * 32) * sizeof(float);
#define ATT(N)                                                                                                 \
  do {                                                                                                         \
    (void)hipFuncSetAttribute((const void *)k_cross_attn<N>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                              (int)lds);                                                                       \
    hipLaunchKernelGGL((k_cross_attn<N>), dim3(dcl_div_up(nq, 128), b), dim3(256), lds, s, nq, nk, Q, ldq, K,   \
                       ldk, V1, dv1, ldv1, O1, ldo1, V2, dv2, ldv2, O2, ldo2);                                 \
  } while (0)
    switch (nvt) {
      case 1: ATT(1); break;
      case 2: ATT(2); break;
      case 4: ATT(4); break;
      case 8: ATT(8); break;
      default: ATT(10); break;
    }
#undef ATT
  }
#else
  else {
    // the product library carries the one kernel the DCL-Net path uses (V = [256 | 64] channels, models/DCL_Net.py:206-215);
    // its general-shape predecessors live in the diagnostic library only
    dcl_set_error("dcl_cross_attention: only dv1 = 256, dv2 = 64 is built into the product library (got %d, %d)", dv1, dv2);
    return DCL_EINVAL;
  }
#endif
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_conf_pool(int b, int c, int n1, int n2, const float *logit1, const float *logit2, const float *F1,
                          int ld1, const float *F2, int ld2, float *conf, float *w_scratch, int nslices,
                          float *part1, float *part2, float *wsum, dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && c > 0 && c % 4 == 0 && n1 > 0 && n2 > 0 && ld1 >= c && ld2 >= c && ld1 % 4 == 0 &&
                ld2 % 4 == 0 && nslices >= 1);
  if (b == 0) return 0;
  DCL_CHECK_ARG(conf && w_scratch && part1 && part2 && wsum && logit1 && F1 && logit2 && F2 && b <= 65535 &&
                nslices <= 32767);
  hipStream_t s = (hipStream_t)stream;
  if (b <= 8 && n1 + n2 <= 256 * 8) {                  // a handful of crops of the shipped size: one launch (k_conf_pool_small)
    hipLaunchKernelGGL(k_conf_pool_small<8>, dim3(dcl_div_up(c, 256), 2 * nslices, b), dim3(256), 0, s, c, n1, n2, nslices, logit1,
                       logit2, conf, wsum, F1, ld1, part1, F2, ld2, part2);
    DCL_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_conf_softmax<256>, dim3(b), dim3(256), 0, s, n1, n2, logit1, logit2, conf, w_scratch, wsum);
  hipLaunchKernelGGL(k_weighted_colsum, dim3(dcl_div_up(c, 256), 2 * nslices, b), dim3(256), 0, s, c, n1, n2, nslices,
                     w_scratch, F1, ld1, part1, F2, ld2, part2);
  DCL_LAUNCH_CHECK();
  return 0;
}

// out[b][ch] = ((sA*P1 + tA*w0) + sB*P2) + tB*w1 with P = slice partials added in a fixed order: the pooled feature of
// both directions behind the fusers' trailing BatchNorms (sum_i w_i (s x_i + t) = s sum w x + t sum w), one launch
__global__ void k_pool_finish(int c, int nslices1, int nslices2, const float *__restrict__ part1, const float *__restrict__ part2,
                              const float *__restrict__ wsum, const float *__restrict__ sA, const float *__restrict__ tA,
                              const float *__restrict__ sB, const float *__restrict__ tB, float *__restrict__ out) {
  const int b = blockIdx.y, ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  // slice s goes to accumulator s % 4 (four independent load chains), the four are combined in a fixed order
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
  const float *q1 = part1 + (size_t)b * nslices1 * c + ch, *q2 = part2 + (size_t)b * nslices2 * c + ch;
  int s = 0;
  for (; s + 4 <= nslices1; s += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a1[j] += q1[(size_t)(s + j) * c];
  }
  for (int j = 0; s < nslices1; ++s, ++j) a1[j] += q1[(size_t)s * c];
  for (s = 0; s + 4 <= nslices2; s += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a2[j] += q2[(size_t)(s + j) * c];
  }
  for (int j = 0; s < nslices2; ++s, ++j) a2[j] += q2[(size_t)s * c];
  const float p1 = (a1[0] + a1[1]) + (a1[2] + a1[3]), p2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  const float w0 = wsum[2 * b], w1 = wsum[2 * b + 1];
  out[(size_t)b * c + ch] = ((sA[ch] * p1 + tA[ch] * w0) + sB[ch] * p2) + tB[ch] * w1;
}

DCL_API int dcl_pool_finish2(int b, int c, int nslices1, int nslices2, const float *part1, const float *part2, const float *wsum,
                             const float *scale1, const float *shift1, const float *scale2, const float *shift2,
                             float *out, dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && c > 0 && nslices1 >= 1 && nslices2 >= 1);
  if (b == 0) return 0;
  DCL_CHECK_ARG(part1 && part2 && wsum && scale1 && shift1 && scale2 && shift2 && out && b <= 65535);
  hipLaunchKernelGGL(k_pool_finish, dim3(dcl_div_up(c, 256), b), dim3(256), 0, (hipStream_t)stream, c, nslices1, nslices2, part1,
                     part2, wsum, scale1, shift1, scale2, shift2, out);
  DCL_LAUNCH_CHECK();
  return 0;
}
DCL_API int dcl_pool_finish(int b, int c, int nslices, const float *part1, const float *part2, const float *wsum,
                            const float *scale1, const float *shift1, const float *scale2, const float *shift2,
                            float *out, dclStream_t stream) {
  return dcl_pool_finish2(b, c, nslices, nslices, part1, part2, wsum, scale1, shift1, scale2, shift2, out, stream);
}

// the softmax half of dcl_conf_pool on its own (k_conf_softmax): conf (b, n1 + n2) = sigmoid(logits), w (b, n1 + n2) = softmax of
// conf over a crop's n1 + n2 points, wsum (b, 2) = the two directions' weight sums -- for the forward whose last fuser layer
// carries the pooling as its epilogue (dcl_linear_pool_fwd) and needs the weights BEFORE that layer
DCL_API int dcl_conf_softmax(int b, int n1, int n2, const float *logit1, const float *logit2, float *conf, float *w, float *wsum,
                             dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && n1 > 0 && n2 > 0);
  if (b == 0) return 0;
  DCL_CHECK_ARG(logit1 && logit2 && conf && w && wsum && b <= 65535);
  if (n1 + n2 > 4096) hipLaunchKernelGGL(k_conf_softmax<1024>, dim3(b), dim3(1024), 0, (hipStream_t)stream, n1, n2, logit1, logit2, conf, w, wsum);
  else hipLaunchKernelGGL(k_conf_softmax<256>, dim3(b), dim3(256), 0, (hipStream_t)stream, n1, n2, logit1, logit2, conf, w, wsum);
  DCL_LAUNCH_CHECK();
  return 0;
}

// ---- first layer of the refiner's MLP_share on the canonicalised points (models/refiner.py:78-80: Conv1d(259 -> 512) on
// cat[xyz, F_Xo_p]): the 256 feature channels' part of it is constant over the refine iterations (`term`, computed once per
// loop by a library GEMM), the xyz part is a K = 3 product -- as a library GEMM + a ReLU pass that was two sweeps over the
// (rows, 512) tensor per iteration (addmm 29 us + clamp 20 us at 32 768 rows); here one: out = relu(term + xyz @ W).
__global__ __launch_bounds__(256) void k_affine3_relu(long long rows, int c4, const float *__restrict__ xyz, const float *__restrict__ W,
                                                      const float4 *__restrict__ term, float4 *__restrict__ out) {
  const long long total = rows * c4;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long row = t / c4;
    const int q = (int)(t - row * c4);
    const float x = xyz[row * 3], y = xyz[row * 3 + 1], z = xyz[row * 3 + 2];
    const float4 wx = reinterpret_cast<const float4 *>(W)[q], wy = reinterpret_cast<const float4 *>(W)[c4 + q],
                 wz = reinterpret_cast<const float4 *>(W)[2 * c4 + q];
    const float4 a = term[t];
    float4 o;                                              // (x Wx + y Wy) + z Wz as an fma chain, then the constant term
    o.x = fmaxf(__fmaf_rn(z, wz.x, __fmaf_rn(y, wy.x, x * wx.x)) + a.x, 0.f);
    o.y = fmaxf(__fmaf_rn(z, wz.y, __fmaf_rn(y, wy.y, x * wx.y)) + a.y, 0.f);
    o.z = fmaxf(__fmaf_rn(z, wz.z, __fmaf_rn(y, wy.z, x * wx.z)) + a.z, 0.f);
    o.w = fmaxf(__fmaf_rn(z, wz.w, __fmaf_rn(y, wy.w, x * wx.w)) + a.w, 0.f);
    out[t] = o;
  }
}

DCL_API int dcl_affine3_relu(int64_t rows, int c, const float *xyz, const float *W3, const float *term, float *out,
                             dclStream_t stream) {
  DCL_CHECK_ARG(rows >= 0 && c > 0 && c % 4 == 0);
  if (rows == 0) return 0;
  DCL_CHECK_ARG(xyz && W3 && term && out);
  hipLaunchKernelGGL(k_affine3_relu, dim3(dcl_grid_1d(rows * (c / 4), 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream,
                     (long long)rows, c / 4, xyz, W3, reinterpret_cast<const float4 *>(term), reinterpret_cast<float4 *>(out));
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_ortho9d_to_matrix(int b, const float *o9, float *R, dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0);
  if (b == 0) return 0;
  DCL_CHECK_ARG(o9 && R);
  hipLaunchKernelGGL(k_ortho9d, dim3(dcl_div_up(b, 64)), dim3(64), 0, (hipStream_t)stream, b, o9, R);
  DCL_LAUNCH_CHECK();
  return 0;
}

// ---- pose heads for a handful of crops --------------------------------------------------------------------------
// regressor_rot / regressor_trans (models/DCL_Net.py:139-151,231-235: Conv1d k1 1024 -> 512 -> 128 -> {9,3}, ReLU after
// the first two) on the pooled (b,1024) feature.  With a few crops these are row-vector x matrix products: as library
// GEMMs they are 6 launches + the copy / activation kernels the row-vector path of the library wrapper adds (14-16
// kernels on the critical path of a one-image forward).  Two launches here, both heads in each:
//   k_heads_l1: h1[head][crop][512]; workgroup = (64 outputs, crop, head), 16 k-slices of 64 terms, LDS reduce
//   k_heads_l23: h2 = relu(W2 h1 + b2) (128), out = W3 h2 + b3 (9 | 3); workgroup = (crop, head)
// Weights are the transposed (in, out) matrices the dense pipeline keeps (row k contiguous over outputs): coalesced.
namespace {
struct HeadWeights {
  const float *w1[2], *b1[2], *w2[2], *b2[2], *w3[2], *b3[2];   // [0] rotation (9 outputs), [1] translation (3)
};

// (k-slices: a thread's dot product is a chain of dependent FMAs fed by strided weight loads; with 4 slices of 256 terms a
// launch spent 15 us waiting on 32 rounds of loads -- 16 slices of 64 terms need 4-8 rounds.  The slice sums are added in a
// fixed tree order.)
constexpr int kHeadSlices = 16;
__device__ __forceinline__ float heads_tree16(const float (*part)[64], int o) {
  float v[kHeadSlices];
#pragma unroll
  for (int i = 0; i < kHeadSlices; ++i) v[i] = part[i][o];
#pragma unroll
  for (int d = 1; d < kHeadSlices; d <<= 1)
#pragma unroll
    for (int i = 0; i < kHeadSlices; i += 2 * d) v[i] = v[i] + v[i + d];
  return v[0];
}

// the pooled feature still as the pooling's slice partials (calls of a handful of crops: k_pool_finish folded into this launch)
struct PoolParts {
  const float *part1, *part2, *wsum, *sA, *tA, *sB, *tB;
  int nslices;
};
// out[ch] of k_pool_finish for (crop, ch), the same operations in the same order
__device__ __forceinline__ float pool_finish_one(const PoolParts &P, int c, int b, int ch) {
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
  const float *q1 = P.part1 + (size_t)b * P.nslices * c + ch, *q2 = P.part2 + (size_t)b * P.nslices * c + ch;
  int s = 0;
  for (; s + 4 <= P.nslices; s += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a1[j] += q1[(size_t)(s + j) * c];
      a2[j] += q2[(size_t)(s + j) * c];
    }
  }
  for (int j = 0; s < P.nslices; ++s, ++j) {
    a1[j] += q1[(size_t)s * c];
    a2[j] += q2[(size_t)s * c];
  }
  const float p1 = (a1[0] + a1[1]) + (a1[2] + a1[3]), p2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  const float w0 = P.wsum[2 * b], w1 = P.wsum[2 * b + 1];
  return ((P.sA[ch] * p1 + P.tA[ch] * w0) + P.sB[ch] * p2) + P.tB[ch] * w1;
}

template <bool FROM_PARTS>
__global__ __launch_bounds__(1024) void k_heads_l1(int b, const float *__restrict__ x, const PoolParts P, HeadWeights hw,
                                                   float *__restrict__ h1) {
  __shared__ float part[kHeadSlices][64];
  __shared__ float xs[FROM_PARTS ? 1024 : 1];
  const int head = blockIdx.z, crop = blockIdx.y, o0 = blockIdx.x * 64;
  const int o = threadIdx.x & 63, ks = threadIdx.x >> 6;                   // 16 slices of 64 terms
  const float *w = hw.w1[head] + (size_t)(ks * 64) * 512 + o0 + o;
  const float *xv = x + (size_t)crop * 1024 + ks * 64;
  if constexpr (FROM_PARTS) {
    xs[threadIdx.x] = pool_finish_one(P, 1024, crop, (int)threadIdx.x);    // (every workgroup of the crop forms the 1024 inputs itself)
    __syncthreads();
    xv = xs + ks * 64;
  }
  // (all 64 weight loads of the slice in flight: in rounds of 16 a call of one crop waited four round trips here)
  float wv[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) wv[i] = w[(size_t)i * 512];
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) acc = __fmaf_rn(xv[i], wv[i], acc);
  part[ks][o] = acc;
  __syncthreads();
  if (ks == 0) {
    const float v = heads_tree16(part, o) + hw.b1[head][o0 + o];
    h1[((size_t)head * b + crop) * 512 + o0 + o] = fmaxf(v, 0.f);
  }
}

__global__ __launch_bounds__(1024) void k_heads_l23(int b, const float *__restrict__ h1, HeadWeights hw,
                                                    float *__restrict__ o9, float *__restrict__ t3, float *__restrict__ R) {
  __shared__ float xs[512], part[8][128], h2[128], part3[16][12], o9s[12];
  const int head = blockIdx.y, crop = blockIdx.x;
  const float *xin = h1 + ((size_t)head * b + crop) * 512;
  for (int i = threadIdx.x; i < 512; i += 1024) xs[i] = xin[i];
  __syncthreads();
  const int o = threadIdx.x & 127, ks = threadIdx.x >> 7;                  // 8 slices of 64 terms
  const float *w = hw.w2[head] + (size_t)(ks * 64) * 128 + o;
  float wv[64];                                                            // (as in layer 1: one round of loads)
#pragma unroll
  for (int i = 0; i < 64; ++i) wv[i] = w[(size_t)i * 128];
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) acc = __fmaf_rn(xs[ks * 64 + i], wv[i], acc);
  part[ks][o] = acc;
  __syncthreads();
  if (ks == 0)
    h2[o] = fmaxf((((part[0][o] + part[1][o]) + (part[2][o] + part[3][o])) + ((part[4][o] + part[5][o]) + (part[6][o] + part[7][o]))) +
                      hw.b2[head][o], 0.f);
  __syncthreads();
  const int nout = head == 0 ? 9 : 3;
  const int o3 = threadIdx.x & 15, k3 = threadIdx.x >> 4;                  // layer 3: 16 slices of 8 terms per output
  if (k3 < 16 && o3 < nout) {
    const float *w3 = hw.w3[head] + (size_t)(k3 * 8) * nout + o3;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) a = __fmaf_rn(h2[k3 * 8 + i], w3[(size_t)i * nout], a);
    part3[k3][o3] = a;
  }
  __syncthreads();
  if ((int)threadIdx.x < nout) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = part3[i][threadIdx.x];
#pragma unroll
    for (int d = 1; d < 16; d <<= 1)
#pragma unroll
      for (int i = 0; i < 16; i += 2 * d) v[i] = v[i] + v[i + d];
    const float a = v[0] + hw.b3[head][threadIdx.x];
    if (head == 0) { o9[(size_t)crop * 9 + threadIdx.x] = a; o9s[threadIdx.x] = a; }
    else t3[(size_t)crop * 3 + threadIdx.x] = a;
  }
  if (head == 0 && R != nullptr) {                     // ortho9d2matrix of this crop right here: one launch less on the path
    __syncthreads();
    if (threadIdx.x == 0) ortho9d_one(o9s, R + (size_t)crop * 9, 0);
  }
}

// ---- the confidence regressor as ONE launch (models/DCL_Net.py:115-126, 217-218: Head_MultiLayerPerceptron [128, 128, 128, 1],
// ReLU, ReLU, none) ---------------------------------------------------------------------------------------------------------
// logit[m] = w3 . relu(W2t^T relu(W1t^T x[m] + b1) + b2) + b3.  Three library GEMMs of K = 128 are three launches of 9-14 us
// with nothing to do (M x 128 x 128: 0.03-1 GFLOP) and two round trips of the hidden rows through memory; here a workgroup
// keeps both 128 x 128 filters in LDS (136 KB with the row tile: one workgroup per CU), walks 32-row tiles, and a tile's hidden
// rows never leave the CU: wave w owns output columns 32w..32w+31 (64 fp32 MFMAs 32x32x2 per layer), layer 1's rows go
// through the x tile's LDS, layer 2's are dotted with w3 in registers and summed over the columns in a fixed order.
constexpr int kMlpXP = 130, kMlpWP = 33;               // LDS pitches: x rows (even / odd k on even / odd banks), filter rows
__global__ __launch_bounds__(256) void k_mlp128_to1(int M, const float *__restrict__ x, long long ldx,
                                                    const float *__restrict__ W1t, const float *__restrict__ b1,
                                                    const float *__restrict__ W2t, const float *__restrict__ b2,
                                                    const float *__restrict__ w3, long long ldw3, const float *__restrict__ b3,
                                                    float *__restrict__ out) {
  extern __shared__ float mlp_lds[];
  float *Ws1 = mlp_lds;                                // [4 waves][128 k][33]
  float *Ws2 = Ws1 + 4 * 128 * kMlpWP;
  float *xs = Ws2 + 4 * 128 * kMlpWP;                  // [32 rows][130]
  float *red = xs + 32 * kMlpXP;                       // [4 waves][32 rows]
  const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
  // the filters: wave w stages its 32 columns of each -- 16 float4 loads per lane and filter, all of a filter's in flight; the
  // second filter's are asked for before the first tile's layer 1 and parked in LDS after it (a call of one crop is one tile
  // per workgroup: the 64 KB would otherwise be a second exposed round trip)
  float4 wreg[16];
  auto ask = [&](const float *__restrict__ Wt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 16; ++i) wreg[i] = *reinterpret_cast<const float4 *>(Wt + (size_t)(i * 8 + (lane >> 3)) * 128 + 32 * w + (lane & 7) * 4);
  };
  auto park = [&](float *Ws) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float *d = Ws + (w * 128 + i * 8 + (lane >> 3)) * kMlpWP + (lane & 7) * 4;
      d[0] = wreg[i].x; d[1] = wreg[i].y; d[2] = wreg[i].z; d[3] = wreg[i].w;
    }
  };
  // a tile's x rows: thread = (row, 16 floats); the first tile's are asked for together with the first filter
  const int ntiles = (M + 31) >> 5;
  float4 xr[4];
  auto ask_x = [&](int t) __attribute__((always_inline)) {
    const int row = t * 32 + (tid >> 3);
    const float *src = x + (size_t)row * ldx + (tid & 7) * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) xr[q] = row < M ? *reinterpret_cast<const float4 *>(src + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  if ((int)blockIdx.x < ntiles) ask_x((int)blockIdx.x);
  ask(W1t);
  park(Ws1);
  ask(W2t);
  bool w2_parked = false;
  const float bias1 = b1[32 * w + r], bias2 = b2[32 * w + r], w3c = w3[(size_t)(32 * w + r) * ldw3], bias3 = b3[0];
  for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
    const int row0 = t * 32;
    dcl_lds_barrier();                                 // the previous tile's readers of xs / red are done
    {
      float *dst = xs + (tid >> 3) * kMlpXP + (tid & 7) * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) { dst[4 * q] = xr[q].x; dst[4 * q + 1] = xr[q].y; dst[4 * q + 2] = xr[q].z; dst[4 * q + 3] = xr[q].w; }
    }
    if (t + (int)gridDim.x < ntiles) ask_x(t + (int)gridDim.x);     // the next tile's rows travel under this tile's layers
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    {
      const float *arow = xs + r * kMlpXP + h, *brow = Ws1 + (w * 128 + h) * kMlpWP + r;
#pragma unroll 16
      for (int k0 = 0; k0 < 128; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[k0], brow[k0 * kMlpWP], acc, 0, 0, 0);
    }
    if (!w2_parked) { park(Ws2); w2_parked = true; }   // (wave w's own columns: read by wave w only, after the barriers below)
    dcl_lds_barrier();                                 // every wave has read the x tile: it becomes the hidden tile
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = (e & 3) + 8 * (e >> 2) + 4 * h;
      xs[m * kMlpXP + 32 * w + r] = fmaxf(acc[e] + bias1, 0.0f);
    }
    dcl_lds_barrier();
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    {
      const float *arow = xs + r * kMlpXP + h, *brow = Ws2 + (w * 128 + h) * kMlpWP + r;
#pragma unroll 16
      for (int k0 = 0; k0 < 128; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[k0], brow[k0 * kMlpWP], acc, 0, 0, 0);
    }
    // layer 3: this lane's column of the 16 rows it holds, times w3; summed over the 32 columns of the wave (lanes of one half)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = fmaxf(acc[e] + bias2, 0.0f) * w3c;
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
      if (r == 0) red[w * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] = v;
    }
    dcl_lds_barrier();
    if (tid < 32 && row0 + tid < M) out[row0 + tid] = ((red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid])) + bias3;
  }
}
}  // namespace

DCL_API int dcl_mlp128_to1(const float *x, int64_t ldx, int M, const float *W1t, const float *b1, const float *W2t,
                           const float *b2, const float *w3, int64_t ldw3, const float *b3, float *out, dclStream_t stream) {
  DCL_CHECK_ARG(M >= 0 && ldx >= 128 && ldx % 4 == 0 && ldw3 >= 1);
  if (M == 0) return 0;
  DCL_CHECK_ARG(x && W1t && b1 && W2t && b2 && w3 && b3 && out);
  DCL_CHECK_ARG((((uintptr_t)x | (uintptr_t)W1t | (uintptr_t)W2t) & 15) == 0);
  const size_t lds = (size_t)(2 * 4 * 128 * kMlpWP + 32 * kMlpXP + 4 * 32) * sizeof(float);
  (void)hipFuncSetAttribute((const void *)k_mlp128_to1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int tiles = (M + 31) / 32;
  hipLaunchKernelGGL(k_mlp128_to1, dim3(tiles < 256 ? tiles : 256), dim3(256), lds, (hipStream_t)stream, M, x, (long long)ldx, W1t, b1,
                     W2t, b2, w3, (long long)ldw3, b3, out);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_pose_heads(int b, const float *pooled, const float *const *rot_layers, const float *const *trans_layers,
                           float *h1_scratch, float *o9, float *trans, float *R, dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && b <= 65535);
  if (b == 0) return 0;
  DCL_CHECK_ARG(pooled && rot_layers && trans_layers && h1_scratch && o9 && trans);
  HeadWeights hw;
  for (int h = 0; h < 2; ++h) {
    const float *const *L = h == 0 ? rot_layers : trans_layers;         // {W1t, b1, W2t, b2, W3t, b3}
    for (int i = 0; i < 6; ++i) DCL_CHECK_ARG(L[i] != nullptr);
    hw.w1[h] = L[0]; hw.b1[h] = L[1]; hw.w2[h] = L[2]; hw.b2[h] = L[3]; hw.w3[h] = L[4]; hw.b3[h] = L[5];
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_heads_l1<false>, dim3(512 / 64, b, 2), dim3(1024), 0, s, b, pooled, PoolParts{}, hw, h1_scratch);
  hipLaunchKernelGGL(k_heads_l23, dim3(b, 2), dim3(1024), 0, s, b, h1_scratch, hw, o9, trans, R);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_pose_heads_parts(int b, int nslices, const float *part1, const float *part2, const float *wsum,
                                 const float *scale1, const float *shift1, const float *scale2, const float *shift2,
                                 const float *const *rot_layers, const float *const *trans_layers, float *h1_scratch, float *o9,
                                 float *trans, float *R, dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && b <= 65535 && nslices >= 1);
  if (b == 0) return 0;
  DCL_CHECK_ARG(part1 && part2 && wsum && scale1 && shift1 && scale2 && shift2 && rot_layers && trans_layers && h1_scratch && o9 && trans);
  HeadWeights hw;
  for (int h = 0; h < 2; ++h) {
    const float *const *L = h == 0 ? rot_layers : trans_layers;         // {W1t, b1, W2t, b2, W3t, b3}
    for (int i = 0; i < 6; ++i) DCL_CHECK_ARG(L[i] != nullptr);
    hw.w1[h] = L[0]; hw.b1[h] = L[1]; hw.w2[h] = L[2]; hw.b2[h] = L[3]; hw.w3[h] = L[4]; hw.b3[h] = L[5];
  }
  const PoolParts P{part1, part2, wsum, scale1, shift1, scale2, shift2, nslices};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_heads_l1<true>, dim3(512 / 64, b, 2), dim3(1024), 0, s, b, nullptr, P, hw, h1_scratch);
  hipLaunchKernelGGL(k_heads_l23, dim3(b, 2), dim3(1024), 0, s, b, h1_scratch, hw, o9, trans, R);
  DCL_LAUNCH_CHECK();
  return 0;
}

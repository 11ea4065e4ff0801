// linear_group.hip -- several INDEPENDENT per-point linear layers in ONE launch, for calls of a handful of crops.
//
// The reference runs every Conv1d(k=1) / 1x1x1 Conv3d of its MLP stacks as its own cuDNN launch (models/Modules.py:58-97,
// 173-201; call sites models/DCL_Net.py:188-235).  With one image's crops (one to a few thousand rows) each of those GEMMs
// is a few microseconds of work behind a launch, and the whole-forward hipGraph is a chain of such nodes: what a call costs
// is the NUMBER of nodes on its longest path.  Layers that do not depend on each other -- the four second layers of a side's
// disengage stacks, a confidence-MLP layer beside the fuser layer of the same depth -- are therefore issued as one launch:
// a table of problems  y_j[M_j x N_j] = act(x_j[M_j x K_j] Wt_j[K_j x N_j] + bias_j),  row-major, free row pitches, every
// workgroup computing one 64 x 64 tile of one problem.  Large batches keep the library GEMMs (dcl_linear_fwd): their 256 x 256
// tiles are what reaches the MFMA peak; this kernel is sized for latency.
//
// Kernel: 4 waves (2 x 2), each a 32 x 32 fp32 MFMA tile (v_mfma_f32_32x32x2f32), K in chunks of 32; x rows and Wt rows go
// global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction) into a FIVE-stage ring, four chunks in
// flight (counted vmcnt + raw s_barrier: a __syncthreads() would drain the DMAs), LDS images XOR-swizzled exactly like the
// sparse-conv kernel's (sparse_conv.hip).  Rows >= M and columns >= N read a zero line and are not stored.  Per element the
// sum runs over k ascending inside one accumulator -- the order of a plain dot product.
#include <hip/hip_runtime.h>
#include "common.h"

namespace {

constexpr int kLgBM = 64, kLgBN = 64, kLgKC = 32, kLgStages = 5;   // 80 KiB of LDS: two workgroups per CU
constexpr int kLgAT = kLgBM * kLgKC, kLgBT = kLgKC * kLgBN, kLgST = kLgAT + kLgBT;   // floats per stage (16 KiB)

__device__ __attribute__((aligned(256))) float g_lg_zero[256];      // the zero line (static storage: all zero)

typedef __attribute__((address_space(3))) void lg_lds_void_t;
typedef float lg_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned lg_lds_addr(const float *p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lg_lds_void_t *)p);
}
// two pieces whose LDS destinations are 1 KiB apart under ONE M0 value (see conv_glds16_group, sparse_conv.hip): the second
// source pointer is pre-decremented by 1 KiB because the instruction's offset field moves the global address too
__device__ __forceinline__ void lg_glds16_pair(const float *g0, const float *g1_minus_1k, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "global_load_lds_dwordx4 %2, off offset:1024\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g0), "v"(g1_minus_1k), "s"(lds_byte_addr) : "memory");
}

__global__ __launch_bounds__(256) void k_linear_group(const DclLinearJobs jobs, int njobs) {
  extern __shared__ __attribute__((aligned(16))) float lg_lds[];    // kLgStages x [A 64x32 | B 32x64]
  // which problem, which tile (a handful of problems: a scalar walk)
  int wid = blockIdx.x, j = 0;
  for (; j < njobs; ++j) {
    const int t = ((jobs.job[j].M + kLgBM - 1) / kLgBM) * ((jobs.job[j].N + kLgBN - 1) / kLgBN);
    if (wid < t) break;
    wid -= t;
  }
  if (j >= njobs) return;
  const DclLinearJob &J = jobs.job[j];
  const int M = J.M, N = J.N, K = J.K;
  const int ncol = (N + kLgBN - 1) / kLgBN;
  const int row0 = (wid / ncol) * kLgBM, col0 = (wid % ncol) * kLgBN;
  const float *__restrict__ x = J.x;
  const float *__restrict__ W = J.Wt;
  const long long ldx = J.ldx, ldw = J.ldw;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
  const float *zero = g_lg_zero;

  // this wave's four DMA pieces per chunk: A pieces 2w, 2w+1 (8 rows of 128 B each), W pieces 2w, 2w+1 (4 rows of 256 B each)
  const float *asrc[2], *bsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int arow = (2 * wave + i) * 8 + (lane >> 3);                          // tile row
    const int achs = ((lane & 7) ^ ((arow >> 1) & 7)) << 2;                     // source column inside the chunk (swizzle)
    asrc[i] = row0 + arow < M ? x + (size_t)(row0 + arow) * ldx + achs : nullptr;
    const int bkk = (2 * wave + i) * 4 + (lane >> 4);                           // W row inside the chunk
    const int pcol = lane & 15;
    const int bcol = col0 + ((pcol ^ (((bkk >> 2) & 1) << 3)) << 2);            // half swap of rows with bit 2 set
    bsrc[i] = bcol < N ? W + (size_t)bkk * ldw + bcol : nullptr;                // (a row of Wt holds N rounded up to 4 floats)
  }
  auto issue = [&](int chunk, int stage) {
    float *As = lg_lds + stage * kLgST, *Bs = As + kLgAT;
    const float *a0 = asrc[0] ? asrc[0] + chunk * kLgKC : zero;
    const float *a1 = (asrc[1] ? asrc[1] + chunk * kLgKC : zero) - 256;
    const float *b0 = bsrc[0] ? bsrc[0] + (size_t)chunk * kLgKC * ldw : zero;
    const float *b1 = (bsrc[1] ? bsrc[1] + (size_t)chunk * kLgKC * ldw : zero) - 256;
    lg_glds16_pair(b0, b1, lg_lds_addr(Bs + (2 * wave) * 256));
    lg_glds16_pair(a0, a1, lg_lds_addr(As + (2 * wave) * 256));
  };

  lg_f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  const int nchunks = K / kLgKC;
  constexpr int D = kLgStages - 1;                       // chunks in flight (the one being waited for included)
#pragma unroll
  for (int c = 0; c < D; ++c)
    if (c < nchunks) issue(c, c);
  for (int c = 0; c < nchunks; ++c) {
    // chunk c has landed once only the pieces of the chunks behind it (4 per wave and chunk) are outstanding
    const int behind = nchunks - 1 - c < D - 1 ? nchunks - 1 - c : D - 1;
    if (behind >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (behind == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // everyone's pieces of chunk c are in; chunk c - 1's stage has no reader left
    if (c + D < nchunks) issue(c + D, (c + D) % kLgStages);
    const float *As = lg_lds + (c % kLgStages) * kLgST, *Bs = As + kLgAT;
    const float *arow = As + (wr * 32 + r) * kLgKC;
    const int sw = (r >> 1) & 7;
#pragma unroll
    for (int i = 0; i < kLgKC / 8; ++i) {
      const float4 a = *reinterpret_cast<const float4 *>(arow + (((2 * i + h) ^ sw) << 2));
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float bv = Bs[(8 * i + 4 * h + q) * kLgBN + ((wc * 32 + r) ^ (32 * h))];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv, acc, 0, 0, 0);
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // MFMA -> VALU read of the accumulator
  const int co = col0 + wc * 32 + r;
  if (co < N) {
    const float bias = J.bias ? J.bias[co] : 0.0f;
    float *__restrict__ y = J.y;
    const long long ldy = J.ldy;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int orow = row0 + wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (orow < M) {
        float v = acc[e] + bias;
        if (J.relu) v = fmaxf(v, 0.0f);
        y[(size_t)orow * ldy + co] = v;
      }
    }
  }
}

}  // namespace

DCL_API int dcl_linear_group_fwd(const DclLinearJob *jobs_host, int njobs, dclStream_t stream) {
  DCL_CHECK_ARG(jobs_host && njobs >= 1 && njobs <= DCL_LINEAR_MAX_JOBS);
  DclLinearJobs jobs;
  long long tiles = 0;
  for (int j = 0; j < njobs; ++j) {
    const DclLinearJob &J = jobs_host[j];
    DCL_CHECK_ARG(J.x && J.Wt && J.y && J.M >= 1 && J.N >= 1 && J.K >= kLgKC && J.K % kLgKC == 0);
    // 16-byte LDS-DMA pieces: aligned bases and pitches; Wt columns come in pieces of 4, so a row of Wt holds N rounded up
    // to 4 floats (a layer with a lone output column passes a zero-padded Wt)
    DCL_CHECK_ARG(J.ldx % 4 == 0 && J.ldw % 4 == 0 && ((size_t)J.x & 15) == 0 && ((size_t)J.Wt & 15) == 0);
    DCL_CHECK_ARG(J.ldx >= J.K && J.ldw >= (J.N + 3) / 4 * 4 && J.ldy >= J.N);
    jobs.job[j] = J;
    tiles += (long long)((J.M + kLgBM - 1) / kLgBM) * ((J.N + kLgBN - 1) / kLgBN);
  }
  DCL_CHECK_ARG(tiles <= 65535 * 16);
  const size_t lds = (size_t)kLgStages * kLgST * sizeof(float);
  (void)hipFuncSetAttribute((const void *)k_linear_group, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_linear_group, dim3((unsigned)tiles), dim3(256), lds, (hipStream_t)stream, jobs, njobs);
  DCL_LAUNCH_CHECK();
  return 0;
}

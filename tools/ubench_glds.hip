// ubench_glds.hip -- what does ONE wave pay to ISSUE a gathered 1-KiB LDS-DMA piece (global_load_lds_dwordx4, 8 feature rows of
// 128 B), and does the price depend on how M0 is handled?  (tools/ only; hipcc --offload-arch=gfx950 -O3 ubench_glds.hip)
//   variant 0: the conv kernel's statement (save M0, set M0, s_nop, load, restore M0) per piece
//   variant 1: ONE M0 per 4 pieces, the pieces' LDS offsets in the instruction's offset field (source pointer pre-decremented)
//   variant 2: set M0 per piece, no save / restore
//   variant 3: as 0 but the sources are contiguous (no gather)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ unsigned lds_addr(const float *p) { return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t *)p); }

__device__ __forceinline__ void glds_v0(const void *g, unsigned l) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(l) : "memory");
}
__device__ __forceinline__ void glds_v2(const void *g, unsigned l) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(l) : "memory");
}
__device__ __forceinline__ void glds_setm0(unsigned l) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(l) : "memory"); }
template <int OFF>
__device__ __forceinline__ void glds_off(const void *g) {
  asm volatile("global_load_lds_dwordx4 %0, off offset:%1" : : "v"(g), "i"(OFF) : "memory");
}

constexpr int PIECES = 20;                         // per chunk, as the 128 x 32 tile (16 A + 4 W)
template <int VAR>
__global__ __launch_bounds__(256) void k(const float *feat, const int *rows, int nrows, int iters, int issuers, unsigned long long *out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long t_issue = 0, t_wait = 0;
  const int per = PIECES / issuers;                // pieces this wave issues per chunk
  for (int it = 0; it < iters; ++it) {
    const float *src[PIECES];
    if (wave < issuers) {
#pragma unroll
      for (int p = 0; p < PIECES; ++p) {
        const int piece = wave * per + p;
        const int slot = ((it * PIECES + piece) * 8 + (lane >> 3)) % nrows;
        const int row = VAR == 3 ? ((it * PIECES + piece) * 8 + (lane >> 3)) % nrows : rows[slot];
        src[p] = feat + (size_t)row * 32 + (lane & 7) * 4;
      }
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < issuers) {
      float *base = lds + (it & 1) * PIECES * 256 + wave * per * 256;
      if constexpr (VAR == 0 || VAR == 3) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) if (p < per) glds_v0(src[p], lds_addr(base + p * 256));
      } else if constexpr (VAR == 2) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) if (p < per) glds_v2(src[p], lds_addr(base + p * 256));
      } else {
#pragma unroll
        for (int p = 0; p < PIECES; p += 4) {
          if (p < per) {
            glds_setm0(lds_addr(base + p * 256));
            glds_off<0>(src[p]);
            if (p + 1 < per) glds_off<1024>((const char *)src[p + 1] - 1024);
            if (p + 2 < per) glds_off<2048>((const char *)src[p + 2] - 2048);
            if (p + 3 < per) glds_off<3072>((const char *)src[p + 3] - 3072);
          }
        }
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    t_issue += t1 - t0;
    t_wait += t2 - t1;
    __syncthreads();
  }
  // checksum of what landed (keeps the DMAs honest), by wave 0
  float s = 0.f;
  for (int i = threadIdx.x; i < 2 * PIECES * 256; i += 256) s += lds[i];
  if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = t_issue; out[blockIdx.x * 4 + 1] = t_wait; out[blockIdx.x * 4 + 2] = (unsigned long long)(s != 12345.f); }
}

template <int VAR>
static void run(const float *feat, const int *rows, int nrows, int grid, int issuers, unsigned long long *dout, std::vector<float> &h_feat,
                std::vector<int> &h_rows) {
  const int iters = 200;
  const size_t ldsb = 2 * PIECES * 1024;
  hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), ldsb, 0, feat, rows, nrows, iters, issuers, dout);
  hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), ldsb, 0, feat, rows, nrows, iters, issuers, dout);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
  double a = 0, w = 0;
  for (int i = 0; i < grid; ++i) { a += h[i * 4]; w += h[i * 4 + 1]; }
  const double per_wave = PIECES / issuers;
  printf("variant %d grid %4d issuers %d: issue %.0f cycles per chunk (%.0f per piece), wait %.0f\n", VAR, grid, issuers,
         a / grid / iters, a / grid / iters / per_wave, w / grid / iters);
}

int main() {
  const int nrows = 100000;
  std::vector<float> h_feat((size_t)nrows * 32);
  for (size_t i = 0; i < h_feat.size(); ++i) h_feat[i] = (float)(i % 97) * 0.01f;
  std::vector<int> h_rows(nrows);
  unsigned s = 12345;
  for (int i = 0; i < nrows; ++i) { s = s * 1664525u + 1013904223u; h_rows[i] = (int)((s >> 8) % nrows); }
  float *feat; int *rows; unsigned long long *dout;
  hipMalloc(&feat, h_feat.size() * 4); hipMalloc(&rows, nrows * 4); hipMalloc(&dout, 4096 * 4 * 8);
  hipMemcpy(feat, h_feat.data(), h_feat.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(rows, h_rows.data(), nrows * 4, hipMemcpyHostToDevice);
  for (int grid : {1, 256, 512})
    for (int issuers : {1, 4}) {
      run<0>(feat, rows, nrows, grid, issuers, dout, h_feat, h_rows);
      run<1>(feat, rows, nrows, grid, issuers, dout, h_feat, h_rows);
      run<2>(feat, rows, nrows, grid, issuers, dout, h_feat, h_rows);
      run<3>(feat, rows, nrows, grid, issuers, dout, h_feat, h_rows);
    }
  return 0;
}

// Sustained fp32 MFMA rate by instruction shape (v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32), operands in registers, random
// data, every CU busy: does the chip hold a different clock for the two shapes under load (MI355X_MICROARCH.md: DVFS item 7 found
// 1.12-1.15x for the bf16 shapes)?   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_mfma_f32 tools/ubench_mfma_f32.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xffff]; b[i] = in[(t * 8 + 4 + i) & 0xffff]; }
  float acc_sum = 0.f;
  if (SHAPE == 32) {
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + j) & 3], b[u], c[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc_sum += c[j][e];
  } else {
    f32x4 c[8];
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) c[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + j) & 3], b[u], c[j], 0, 0, 0);
    }
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) acc_sum += c[j][e];
  }
  out[t] = acc_sum;
}
int main() {
  float *in, *out;
  hipMalloc(&in, 65536 * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
  std::vector<float> h(65536);
  for (int i = 0; i < 65536; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(in, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu)
    for (int shape : {32, 16, 32, 16}) {
      const int iters = 20000, grid = 256 * wg_per_cu;
      auto run = [&]() { if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(grid), dim3(256), 0, 0, in, out, iters); else hipLaunchKernelGGL(k<16>, dim3(grid), dim3(256), 0, 0, in, out, iters); };
      for (int w = 0; w < 20; ++w) run();                                  // ~2 s of back-to-back launches first
      hipEventRecord(e0); for (int r = 0; r < 10; ++r) run(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
      const double flop = (double)grid * 4 * iters * (shape == 32 ? 16 * 4096.0 : 32 * 2048.0);
      printf("shape %dx%d, %d wave(s) per SIMD: %.2f ms  %.1f TFLOP/s\n", shape, shape, wg_per_cu, ms, flop / ms / 1e9);
    }
  return 0;
}

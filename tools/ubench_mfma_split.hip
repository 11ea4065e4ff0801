// What a "three bf16 pieces per fp32 operand, six products" GEMM body could sustain on this chip, before building one:
//   mode 0  bare v_mfma_f32_32x32x16_bf16 loop, operands in registers (random bits)
//   mode 1  the six-product body of a 64-row x 128-column wave tile, all 3+3 operand planes in registers (no loads, no VALU)
//   modes 3 / 4 / 5  mode 2 without the split / with A from a per-wave LDS tile instead of global memory / without the B reads
//   mode 2  mode 1's MFMAs + what a real loop does per 16-deep step: the wave's 64x16 fp32 A values from global memory (two
//           dwordx4 per row block and lane), split into three bf16 planes in registers (the VALU work), the B planes of four
//           column blocks by ds_read_b128 from an LDS tile
// fp32-equivalent TFLOP/s = bf16 MFMA flop / 6.   hipcc --offload-arch=gfx950 -O3 -o tools/_bin/ubench_mfma_split tools/ubench_mfma_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// mode 6: the six-product body on v_mfma_f32_16x16x32_bf16, the same 64 x 128 wave tile as 4 x 8 blocks of 16 x 16 (operands in
// registers): MI355X_MICROARCH.md reports 1.12-1.15 x the 32x32x16 shape's FLOP/s in bare loops on random data (the clock the chip holds)
__global__ __launch_bounds__(256, 2) void k16(const unsigned *wbits, float *out, long long *clk, int iters) {
  const int t = threadIdx.x;
  f32x4 c[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) c[i][j][e] = 0.f;
  u32x4 ap[2][3], bp[2][3];
  for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) for (int e = 0; e < 4; ++e) {
    ap[i][p][e] = wbits[(t * 24 + i * 12 + p * 4 + e) & 0xffff];
    bp[i][p][e] = wbits[(t * 24 + 7000 + i * 12 + p * 4 + e) & 0xffff];
  }
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[i & 1][2]), __builtin_bit_cast(bf16x8, bp[j & 1][0]), c[i][j], 0, 0, 0);
        c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[i & 1][0]), __builtin_bit_cast(bf16x8, bp[j & 1][2]), c[i][j], 0, 0, 0);
        c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[i & 1][1]), __builtin_bit_cast(bf16x8, bp[j & 1][1]), c[i][j], 0, 0, 0);
        c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[i & 1][1]), __builtin_bit_cast(bf16x8, bp[j & 1][0]), c[i][j], 0, 0, 0);
        c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[i & 1][0]), __builtin_bit_cast(bf16x8, bp[j & 1][1]), c[i][j], 0, 0, 0);
        c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ap[i & 1][0]), __builtin_bit_cast(bf16x8, bp[j & 1][0]), c[i][j], 0, 0, 0);
      }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) sum += c[i][j][e];
  out[blockIdx.x * 256 + t] = sum;
  if (t == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// truncating exact split of a pair: x = h + m + l with h, m, l each 8 significant bits (24 = 8 + 8 + 8)
__device__ __forceinline__ void split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
  const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float *x, const unsigned *wbits, float *out, long long *clk, int ldx, int K, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[3 * 128 * 32 / 2];                // three planes of [128 cols][32 k] bf16
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 3 * 128 * 32 / 2; i += 256) lds[i] = wbits[(i * 7 + blockIdx.x) & 0xffff];
  __syncthreads();
  f32x16 c[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[i][j][e] = 0.f;
  u32x4 ap[2][3], bp[4][3];
  for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) for (int e = 0; e < 4; ++e) ap[i][p][e] = wbits[(t * 24 + i * 12 + p * 4 + e) & 0xffff];
  for (int j = 0; j < 4; ++j) for (int p = 0; p < 3; ++p) for (int e = 0; e < 4; ++e) bp[j][p][e] = wbits[(t * 48 + 9000 + j * 12 + p * 4 + e) & 0xffff];
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][u % 3]), as_bf(bp[j][u / 2]), c[i][j], 0, 0, 0);
  } else if (MODE == 1) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][2]), as_bf(bp[j][0]), c[i][j], 0, 0, 0);
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][0]), as_bf(bp[j][2]), c[i][j], 0, 0, 0);
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][1]), as_bf(bp[j][1]), c[i][j], 0, 0, 0);
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][1]), as_bf(bp[j][0]), c[i][j], 0, 0, 0);
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][0]), as_bf(bp[j][1]), c[i][j], 0, 0, 0);
          c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][0]), as_bf(bp[j][0]), c[i][j], 0, 0, 0);
        }
  } else {
    constexpr bool SPLIT = MODE != 3, A_LDS = MODE == 4, B_LDS = MODE != 5;
    __shared__ __attribute__((aligned(16))) float alds[4 * 64 * 16 * 2];
    if (A_LDS) { for (int i = t; i < 4 * 64 * 16 * 2; i += 256) alds[i] = x[(i * 13 + blockIdx.x * 77) & 0xfffff]; __syncthreads(); }
    const float *xr = x + ((long long)blockIdx.x * 256 + wave * 64 + (lane & 31)) * ldx + (lane >> 5) * 8;
    const int steps = K / 16;
    f32x4 a[2][2], an[2][2];
    for (int i = 0; i < 2; ++i) for (int q = 0; q < 2; ++q) a[i][q] = *(const f32x4 *)(xr + (long long)i * 32 * ldx + q * 4);
    for (int it = 0; it < iters; ++it)
      for (int s = 0; s < steps; ++s) {
        const int sn = (s + 1 == steps) ? 0 : s + 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            if (A_LDS) {                                                                 // the wave's own [64][16] fp32 tile, 16-byte pieces swizzled by row
              const int row = i * 32 + (lane & 31), piece = ((lane >> 5) * 2 + q) ^ ((row >> 1) & 3);
              an[i][q] = *(const f32x4 *)&alds[(wave * 2 + (sn & 1)) * 1024 + row * 16 + piece * 4];
            } else an[i][q] = *(const f32x4 *)(xr + (long long)i * 32 * ldx + sn * 16 + q * 4);
          }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            unsigned h, m, l;
            if (SPLIT) {
              split2(a[i][q][0], a[i][q][1], h, m, l); ap[i][0][q * 2] = h; ap[i][1][q * 2] = m; ap[i][2][q * 2] = l;
              split2(a[i][q][2], a[i][q][3], h, m, l); ap[i][0][q * 2 + 1] = h; ap[i][1][q * 2 + 1] = m; ap[i][2][q * 2 + 1] = l;
            } else {
              ap[i][0][q * 2] = __float_as_uint(a[i][q][0]); ap[i][1][q * 2] = __float_as_uint(a[i][q][1]);
              ap[i][0][q * 2 + 1] = __float_as_uint(a[i][q][2]); ap[i][1][q * 2 + 1] = __float_as_uint(a[i][q][3]);
            }
          }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = j * 32 + (lane & 31), piece = ((lane >> 5) + 2 * (s & 1)) ^ ((col >> 2) & 3);
#pragma unroll
          for (int p = 0; p < 3; ++p) if (B_LDS) bp[j][p] = *(const u32x4 *)&lds[p * 2048 + col * 16 + piece * 4];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][2]), as_bf(bp[j][0]), c[i][j], 0, 0, 0);
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][0]), as_bf(bp[j][2]), c[i][j], 0, 0, 0);
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][1]), as_bf(bp[j][1]), c[i][j], 0, 0, 0);
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][1]), as_bf(bp[j][0]), c[i][j], 0, 0, 0);
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][0]), as_bf(bp[j][1]), c[i][j], 0, 0, 0);
            c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ap[i][0]), as_bf(bp[j][0]), c[i][j], 0, 0, 0);
          }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 2; ++q) a[i][q] = an[i][q];
      }
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) sum += c[i][j][e];
  out[blockIdx.x * 256 + t] = sum;
  if (t == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
  const int K = 512, ldx = 512, grid = 512;
  float *x, *out; unsigned *w; long long *clk;
  const size_t xn = (size_t)grid * 256 * ldx;
  hipMalloc(&x, xn * 4); hipMalloc(&out, grid * 256 * 4); hipMalloc(&w, 65536 * 4); hipMalloc(&clk, grid * 16);
  std::vector<float> hx(xn); std::vector<unsigned> hw(65536);
  unsigned s = 12345;
  for (size_t i = 0; i < xn; ++i) { s = s * 1664525u + 1013904223u; hx[i] = (float)(s >> 8) / 16777216.f - 0.5f; }
  for (int i = 0; i < 65536; ++i) {                                                        // random bf16 pairs of moderate magnitude
    s = s * 1664525u + 1013904223u; const unsigned lo = 0x3c00u + ((s >> 8) & 0x3ffu) + ((s >> 30) << 15);
    s = s * 1664525u + 1013904223u; const unsigned hi = 0x3c00u + ((s >> 8) & 0x3ffu) + ((s >> 30) << 15);
    hw[i] = lo | (hi << 16);
  }
  hipMemcpy(x, hx.data(), xn * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), 65536 * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<long long> hc(grid * 2);
  for (int mode : {1, 6, 2, 4, 1, 6, 2, 4}) {
    const int iters = (mode >= 2 && mode != 6) ? 40 : (mode == 6 ? 500 : 2000);
    auto run = [&]() {
      if (mode == 6) hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, w, out, clk, iters);
      else if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, x, w, out, clk, ldx, K, iters);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, x, w, out, clk, ldx, K, iters);
      else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, x, w, out, clk, ldx, K, iters);
      else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, x, w, out, clk, ldx, K, iters);
      else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, x, w, out, clk, ldx, K, iters);
      else hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, x, w, out, clk, ldx, K, iters);
    };
    for (int wu = 0; wu < 300; ++wu) run();
    hipEventRecord(e0); for (int r = 0; r < 20; ++r) run(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    hipMemcpy(hc.data(), clk, grid * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < grid; ++i) ghz.push_back((double)hc[i * 2] / (double)hc[i * 2 + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double nmfma = (double)grid * 4 * iters * ((mode >= 2 && mode != 6) ? (K / 16) * 48.0 : (mode == 6 ? 192.0 : 48.0));
    const double flop = nmfma * (mode == 6 ? 16384.0 : 32768.0);
    printf("mode %d: %.3f ms  bf16 %.0f TFLOP/s  fp32-equivalent (six products) %.1f TFLOP/s  in-kernel clock %.2f GHz  cycles per MFMA per SIMD %.1f\n",
           mode, ms, flop / ms / 1e9, flop / 6 / ms / 1e9, ghz[grid / 2], (double)hc[0] / (nmfma / grid / 4) * (grid >= 512 ? 0.5 : 1.0));
  }
  return 0;
}

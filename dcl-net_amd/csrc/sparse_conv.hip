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
__global__ void k_sparse_conv_valu(const float *__restrict__ feat, const int32_t *__restrict__ nbr, int cap,
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
      const int v = nbr[(size_t)k * cap + row];
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

// ---- MFMA kernel: Cin % 8 == 0, Cout == 32*NT --------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_sparse_conv_mfma(
    const float *__restrict__ feat, const int32_t *__restrict__ nbr, int cap, const int32_t *__restrict__ n_out_dev,
    int n_out_host, const float *__restrict__ W, int cin, int kvol, int subm, const float *__restrict__ scale,
    const float *__restrict__ shift, int relu, float *__restrict__ out) {
  constexpr int cout = 32 * NT;
  int n = n_out_dev ? *n_out_dev : n_out_host;
  n = n < cap ? n : cap;
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  const int waves_per_block = blockDim.x >> 6;
  const int ntiles = (n + 31) >> 5;
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
      const int v = valid ? nbr[(size_t)k * cap + row] : -1;
      if (__ballot(v >= 0) == 0ull) continue;                       // nobody in this tile uses offset k
      const float *fp = feat + (size_t)(v >= 0 ? v : 0) * cin + h * 4;
      const float *wp = W + (size_t)k * cin * cout + r;
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
    // C/D layout: col = lane&31 (cout within tile), row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = j * 32 + r;
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

// ---- sparse average pool ------------------------------------------------------------------------
// thread = (output row, 4 channels): rf = #valid offsets (summaryRF.cu:39), then
// out = ((0 + f_k0/rf) + f_k1/rf) + ... in ascending offset order (avgpool.cu:130).
__global__ void k_sparse_avgpool(const float *__restrict__ feat, const int32_t *__restrict__ nbr, int cap,
                                 const int32_t *__restrict__ n_out_dev, int n_out_host, int c, int kvol,
                                 float *__restrict__ out, int32_t *__restrict__ rf_out) {
  int n = n_out_dev ? *n_out_dev : n_out_host;
  n = n < cap ? n : cap;
  const int c4 = c >> 2;
  const long long total = (long long)n * c4;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(t / c4);
    const int q = (int)(t - (long long)row * c4);
    int rf = 0;
    for (int k = 0; k < kvol; ++k) rf += nbr[(size_t)k * cap + row] >= 0;
    const float d = (float)rf;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < kvol; ++k) {
      const int v = nbr[(size_t)k * cap + row];
      if (v < 0) continue;
      const float4 f = reinterpret_cast<const float4 *>(feat + (size_t)v * c)[q];
      acc.x = acc.x + f.x / d; acc.y = acc.y + f.y / d; acc.z = acc.z + f.z / d; acc.w = acc.w + f.w / d;
    }
    reinterpret_cast<float4 *>(out + (size_t)row * c)[q] = acc;
    if (rf_out && q == 0) rf_out[row] = rf;
  }
}

__global__ void k_sparse_avgpool_scalar(const float *__restrict__ feat, const int32_t *__restrict__ nbr, int cap,
                                        const int32_t *__restrict__ n_out_dev, int n_out_host, int c, int kvol,
                                        float *__restrict__ out, int32_t *__restrict__ rf_out) {
  int n = n_out_dev ? *n_out_dev : n_out_host;
  n = n < cap ? n : cap;
  const long long total = (long long)n * c;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(t / c);
    const int ch = (int)(t - (long long)row * c);
    int rf = 0;
    for (int k = 0; k < kvol; ++k) rf += nbr[(size_t)k * cap + row] >= 0;
    const float d = (float)rf;
    float acc = 0.f;
    for (int k = 0; k < kvol; ++k) {
      const int v = nbr[(size_t)k * cap + row];
      if (v >= 0) acc = acc + feat[(size_t)v * c + ch] / d;
    }
    out[t] = acc;
    if (rf_out && ch == 0) rf_out[row] = rf;
  }
}

}  // namespace

static int g_force_valu = 0;   // test hook: dcl_debug_force_valu_conv(1) routes every conv through the VALU kernel
DCL_API void dcl_debug_force_valu_conv(int on) { g_force_valu = on; }

DCL_API int dcl_sparse_conv_fwd(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev,
                                int n_out_host, const float *W, int cin, int cout, int kvol, int subm,
                                const float *scale, const float *shift, int relu, float *out,
                                dclStream_t stream) {
  DCL_CHECK_ARG(feat && nbr && W && out && cap > 0 && cin > 0 && cout > 0 && kvol > 0 && kvol <= 27);
  DCL_CHECK_ARG((scale == nullptr) == (shift == nullptr));
  DCL_CHECK_ARG(n_out_dev || (n_out_host >= 0 && n_out_host <= cap));
  const int rows = n_out_dev ? cap : n_out_host;
  if (rows == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const bool mfma_ok = !g_force_valu && (cin % 8 == 0) && (cout % 32 == 0) && cout <= 256 &&
                       (cout == 32 || cout == 64 || cout == 128 || cout == 256);
  if (mfma_ok) {
    const int ntiles = dcl_div_up(rows, 32);
    const int blocks = dcl_grid_1d(ntiles, 4, 256 * 8);
#define LAUNCH_MFMA(NT)                                                                                       \
  hipLaunchKernelGGL((k_sparse_conv_mfma<NT>), dim3(blocks), dim3(256), 0, s, feat, nbr, cap, n_out_dev,      \
                     n_out_host, W, cin, kvol, subm, scale, shift, relu, out)
    switch (cout) {
      case 32: LAUNCH_MFMA(1); break;
      case 64: LAUNCH_MFMA(2); break;
      case 128: LAUNCH_MFMA(4); break;
      default: LAUNCH_MFMA(8); break;
    }
#undef LAUNCH_MFMA
  } else {
    hipLaunchKernelGGL(k_sparse_conv_valu, dim3(dcl_grid_1d((long long)rows * cout, 256)), dim3(256), 0, s, feat,
                       nbr, cap, n_out_dev, n_out_host, W, cin, cout, kvol, subm, scale, shift, relu, out);
  }
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_sparse_avgpool_fwd(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev,
                                   int n_out_host, int c, int kvol, float *out, int32_t *rf,
                                   dclStream_t stream) {
  DCL_CHECK_ARG(feat && nbr && out && cap > 0 && c > 0 && kvol > 0 && kvol <= 27);
  DCL_CHECK_ARG(n_out_dev || (n_out_host >= 0 && n_out_host <= cap));
  const int rows = n_out_dev ? cap : n_out_host;
  if (rows == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (c % 4 == 0)
    hipLaunchKernelGGL(k_sparse_avgpool, dim3(dcl_grid_1d((long long)rows * (c / 4), 256)), dim3(256), 0, s, feat,
                       nbr, cap, n_out_dev, n_out_host, c, kvol, out, rf);
  else
    hipLaunchKernelGGL(k_sparse_avgpool_scalar, dim3(dcl_grid_1d((long long)rows * c, 256)), dim3(256), 0, s, feat,
                       nbr, cap, n_out_dev, n_out_host, c, kvol, out, rf);
  DCL_LAUNCH_CHECK();
  return 0;
}

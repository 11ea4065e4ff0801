// backward.hip -- training-side kernels of the ops on DCL-Net's path:
//
//   voxelize_bp            libs/pointgroup_ops/src/voxelize/voxelize.cu:35-50
//   indice_conv backward   libs/spconv/include/spconv/spconv_ops.h:351-438 (per offset: dW[k] = X_k^T dO_k, dX += dO_k W[k]^T)
//   indice_avgpool bwd     libs/spconv/src/spconv/avgpool.cu:178-206, pool_ops.h:211-246 (din[i] += dout[o] / rf[o])
//   three_interpolate grad libs/pointnet_sp/src/interpolate_gpu.cu:124-148
//
// MI355X mapping.  The reference scatters (per-offset gather -> GEMM -> scatter-add, or atomics); here everything that
// can be a GATHER is one: the forward rulebook nbr[k][o] = i is transposed once into inv[k][i] = o (every (k, i) has at
// most one o), after which  dX = sparse_conv_fwd(dO, inv, W^T)  reuses the forward MFMA kernel unchanged and the pooling
// gradient is a 27-term gather in the reference's ascending-offset order (bit-exact).  dW is a rows-contracted MFMA GEMM per
// offset: both operands are read straight from global memory with the 32 lanes of a half-wave on 32 consecutive channels
// (coalesced 128-B segments), split over row ranges into partial sums that a second kernel adds in split order
// (deterministic; the reference's cuBLAS order is unspecified, so parity is by tolerance).  The interpolation gradient
// keeps the reference's atomic scatter (its summation order is unspecified there too).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_fill_i32(int32_t *__restrict__ p, long long n, int32_t v) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) p[t] = v;
}

__global__ void k_rulebook_transpose(const int32_t *__restrict__ nbr, int cap_out, const int32_t *__restrict__ n_out_dev,
                                     int n_out_host, int kvol, int32_t *__restrict__ inv, int cap_in) {
  int n = n_out_dev ? *n_out_dev : n_out_host;
  n = n < cap_out ? n : cap_out;
  const long long total = (long long)n * kvol;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(t / n), o = (int)(t - (long long)k * n);
    const int i = nbr[(size_t)k * cap_out + o];
    if (i >= 0 && i < cap_in) inv[(size_t)k * cap_in + i] = o;
  }
}

// din[i] = ((0 + dout[o_k0]/rf[o_k0]) + dout[o_k1]/rf[o_k1]) + ...  ascending offsets (avgpool.cu:204, pool_ops.h:222)
__global__ void k_sparse_avgpool_bwd(const float *__restrict__ dout, const int32_t *__restrict__ inv, int cap_in, int n_in,
                                     const int32_t *__restrict__ rf, int c, int kvol, float *__restrict__ din) {
  const int c4 = c >> 2;
  const long long total = (long long)n_in * c4;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(t / c4), q = (int)(t - (long long)row * c4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < kvol; ++k) {
      const int o = inv[(size_t)k * cap_in + row];
      if (o < 0) continue;
      const float d = (float)rf[o];
      const float4 g = reinterpret_cast<const float4 *>(dout + (size_t)o * c)[q];
      acc.x = acc.x + g.x / d; acc.y = acc.y + g.y / d; acc.z = acc.z + g.z / d; acc.w = acc.w + g.w / d;
    }
    reinterpret_cast<float4 *>(din + (size_t)row * c)[q] = acc;
  }
}

__global__ void k_three_interp_grad_sp(int c, int n, const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
                                       const float *__restrict__ weight, float *__restrict__ grad_points) {
  const long long total = (long long)n * c;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(t / c), ch = (int)(t - (long long)p * c);
    const float g = grad_out[t];
#pragma unroll
    for (int j = 0; j < 3; ++j) atomicAdd(grad_points + (size_t)idx[p * 3 + j] * c + ch, g * weight[p * 3 + j]);
  }
}

__global__ void k_voxelize_bp(int n_rows, int max_active, int c, const float *__restrict__ d_out,
                              const int32_t *__restrict__ rules, int average, float *__restrict__ d_feats) {
  const long long total = (long long)n_rows * c;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(t / c), plane = (int)(t - (long long)row * c);
    const int32_t *r = rules + (size_t)row * (max_active + 1);
    const int na = r[0];
    const float mult = (average && na > 0) ? 1.0f / (float)na : 1.0f;
    const float v = mult * d_out[t];
    for (int i = 1; i <= na; ++i) atomicAdd(d_feats + (size_t)r[i] * c + plane, v);
  }
}

// ---- dW[k] = sum_o X[nbr[k][o]]^T dO[o]: one wave = (32*TM) x (32*TN) block of dW[k] over a row range
template <int TM, int TN>
__global__ __launch_bounds__(64) void k_conv_wgrad_mfma(const float *__restrict__ feat, const int32_t *__restrict__ nbr,
                                                        int cap, int n_out, const float *__restrict__ dout, int cin,
                                                        int cout, int rows_per_split, float *__restrict__ partial) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int tiles_n = cout / (32 * TN);
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int k = blockIdx.y;
  const int ci0 = tm * 32 * TM, co0 = tn * 32 * TN;
  const int row_lo = blockIdx.z * rows_per_split, row_hi = min(row_lo + rows_per_split, n_out);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
  for (int o0 = row_lo; o0 < row_hi; o0 += 8) {
    // 4 MFMA steps of 2 rows each; lane half h owns row o0 + 2s + h
    int v[4];
    float av[4][TM], bv[4][TN];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int o = o0 + 2 * s + h;
      v[s] = o < row_hi ? nbr[(size_t)k * cap + o] : -1;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int o = o0 + 2 * s + h;
#pragma unroll
      for (int a = 0; a < TM; ++a) av[s][a] = v[s] >= 0 ? feat[(size_t)v[s] * cin + ci0 + 32 * a + r] : 0.0f;
#pragma unroll
      for (int b = 0; b < TN; ++b) bv[s][b] = v[s] >= 0 ? dout[(size_t)o * cout + co0 + 32 * b + r] : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][a], bv[s][b], acc[a][b], 0, 0, 0);
  }
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  float *P = partial + ((size_t)blockIdx.z * gridDim.y + k) * cin * cout;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = ci0 + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * h;       // C/D layout: row = (e&3)+8(e>>2)+4h, col = lane&31
        P[(size_t)ci * cout + co0 + 32 * b + r] = acc[a][b][e];
      }
}

// any channel counts: thread = one (ci, co) element of dW[k], running sum over the split's rows in row order
__global__ void k_conv_wgrad_valu(const float *__restrict__ feat, const int32_t *__restrict__ nbr, int cap, int n_out,
                                  const float *__restrict__ dout, int cin, int cout, int rows_per_split,
                                  float *__restrict__ partial) {
  const int k = blockIdx.y;
  const int row_lo = blockIdx.z * rows_per_split, row_hi = min(row_lo + rows_per_split, n_out);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= cin * cout) return;
  const int ci = e / cout, co = e - ci * cout;
  float acc = 0.0f;
  for (int o = row_lo; o < row_hi; ++o) {
    const int v = nbr[(size_t)k * cap + o];
    if (v >= 0) acc = __fmaf_rn(feat[(size_t)v * cin + ci], dout[(size_t)o * cout + co], acc);
  }
  partial[((size_t)blockIdx.z * gridDim.y + k) * cin * cout + e] = acc;
}

__global__ void k_wgrad_reduce(const float *__restrict__ partial, int nsplit, long long n, float *__restrict__ dW) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    float a = partial[t];
    for (int z = 1; z < nsplit; ++z) a = a + partial[(size_t)z * n + t];
    dW[t] = a;
  }
}

}  // namespace

DCL_API int dcl_rulebook_transpose(const int32_t *nbr, int cap_out, const int32_t *n_out_dev, int n_out_host, int kvol,
                                   int32_t *inv, int cap_in, dclStream_t stream) {
  DCL_CHECK_ARG(nbr && inv && cap_out > 0 && cap_in > 0 && kvol > 0 && kvol <= 27 && (n_out_dev || n_out_host >= 0));
  hipStream_t s = (hipStream_t)stream;
  const long long ni = (long long)kvol * cap_in;
  hipLaunchKernelGGL(k_fill_i32, dim3(dcl_grid_1d(ni, 256)), dim3(256), 0, s, inv, ni, -1);
  const int rows = n_out_dev ? cap_out : n_out_host;
  if (rows > 0)
    hipLaunchKernelGGL(k_rulebook_transpose, dim3(dcl_grid_1d((long long)rows * kvol, 256)), dim3(256), 0, s, nbr, cap_out,
                       n_out_dev, n_out_host, kvol, inv, cap_in);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_sparse_avgpool_bwd(const float *dout, const int32_t *inv, int cap_in, int n_in, const int32_t *rf, int c,
                                   int kvol, float *din, dclStream_t stream) {
  DCL_CHECK_ARG(n_in >= 0 && c > 0 && c % 4 == 0 && kvol > 0 && kvol <= 27 && cap_in >= n_in);
  if (n_in == 0) return 0;
  DCL_CHECK_ARG(dout && inv && rf && din);
  hipLaunchKernelGGL(k_sparse_avgpool_bwd, dim3(dcl_grid_1d((long long)n_in * (c / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, dout, inv, cap_in, n_in, rf, c, kvol, din);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_three_interpolate_grad_sp(int c, int n, int m, const float *grad_out, const int32_t *idx, const float *weight,
                                          float *grad_points_zeroed, dclStream_t stream) {
  DCL_CHECK_ARG(c >= 0 && n >= 0 && m >= 0);
  if (c == 0 || n == 0) return 0;
  DCL_CHECK_ARG(grad_out && idx && weight && grad_points_zeroed && m > 0);
  hipLaunchKernelGGL(k_three_interp_grad_sp, dim3(dcl_grid_1d((long long)n * c, 256)), dim3(256), 0, (hipStream_t)stream, c,
                     n, grad_out, idx, weight, grad_points_zeroed);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_voxelize_bp(const float *d_out, const int32_t *rules, float *d_feats_zeroed, int n_rows, int max_active, int c,
                            int average, dclStream_t stream) {
  DCL_CHECK_ARG(n_rows >= 0 && max_active >= 0 && c > 0);
  if (n_rows == 0) return 0;
  DCL_CHECK_ARG(d_out && rules && d_feats_zeroed);
  hipLaunchKernelGGL(k_voxelize_bp, dim3(dcl_grid_1d((long long)n_rows * c, 256)), dim3(256), 0, (hipStream_t)stream, n_rows,
                     max_active, c, d_out, rules, average, d_feats_zeroed);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_sparse_conv_wgrad_splits(int n_out, int32_t *splits_host) {
  DCL_CHECK_ARG(n_out >= 0 && splits_host);
  int s = dcl_div_up(n_out > 0 ? n_out : 1, 2048);
  *splits_host = s > 64 ? 64 : s;
  return 0;
}

DCL_API int dcl_sparse_conv_wgrad(const float *feat, const int32_t *nbr, int cap, int n_out, const float *dout, int cin,
                                  int cout, int kvol, float *partial /* splits*kvol*cin*cout */, float *dW,
                                  dclStream_t stream) {
  DCL_CHECK_ARG(n_out >= 0 && cap >= n_out && cin > 0 && cout > 0 && kvol > 0 && kvol <= 27 && dW && partial);
  DCL_CHECK_ARG(n_out == 0 || (feat && nbr && dout));
  hipStream_t s = (hipStream_t)stream;
  int splits;
  (void)dcl_sparse_conv_wgrad_splits(n_out, &splits);
  const int rps = dcl_div_up(dcl_div_up(n_out > 0 ? n_out : 1, splits), 8) * 8;
  const long long n = (long long)kvol * cin * cout;
  if (cin % 64 == 0 && cout % 64 == 0)
    hipLaunchKernelGGL((k_conv_wgrad_mfma<2, 2>), dim3((cin / 64) * (cout / 64), kvol, splits), dim3(64), 0, s, feat, nbr, cap,
                       n_out, dout, cin, cout, rps, partial);
  else if (cin % 32 == 0 && cout % 32 == 0)
    hipLaunchKernelGGL((k_conv_wgrad_mfma<1, 1>), dim3((cin / 32) * (cout / 32), kvol, splits), dim3(64), 0, s, feat, nbr, cap,
                       n_out, dout, cin, cout, rps, partial);
  else
    hipLaunchKernelGGL(k_conv_wgrad_valu, dim3(dcl_div_up(cin * cout, 256), kvol, splits), dim3(256), 0, s, feat, nbr, cap,
                       n_out, dout, cin, cout, rps, partial);
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(dcl_grid_1d(n, 256)), dim3(256), 0, s, partial, splits, n, dW);
  DCL_LAUNCH_CHECK();
  return 0;
}

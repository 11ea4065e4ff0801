// pointnet.hip -- PointNet++ set-abstraction primitives (libs/pointnet_lib/src/*.cu).
//
//   ball_query   ball_query_gpu.cu:9-45     first `nsample` indices (ascending) with d2 < r2
//   group_points group_points_gpu.cu:47-66  out[b,c,p,s] = points[b,c,idx[b,p,s]]
//   gather_points sampling_gpu.cu:8-24      out[b,c,m]   = points[b,c,idx[b,m]]
//   furthest_point_sampling sampling_gpu.cu:86-253
//
// MI355X mapping.  ball_query: one centre per lane, candidate index wave-uniform (scalar loads
// broadcast the candidate), wave-level early exit by ballot once all 64 lanes are full; hits are
// staged in LDS ([slot][thread], +1 padded) so the final (B,M,nsample) rows leave as coalesced
// 256-B stores instead of one dword per lane per hit.  group_points/gather_points: pure HBM
// streaming, write-dominated -- each lane owns 4 consecutive outputs (one 16-B store), reuses its
// 4 indices over a chunk of channels, source rows stay in L2.  FPS: one workgroup per cloud,
// points + running min-distance held in REGISTERS for the whole run (the reference re-reads
// dataset[] and temp[] from global memory every iteration), wave argmax by DPP/shuffle, one LDS
// exchange + two barriers per iteration.
#include "common.h"
#include <math.h>

namespace {

// ------------------------------------------------------------------------------------ ball query
template <int T>
__global__ __launch_bounds__(T) void k_ball_query(int n, int m, float radius2, int nsample,
                                                  const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                                                  int32_t *__restrict__ idx) {
  extern __shared__ int32_t bq_lds[];          // hits[nsample][T+1], then cnt[T]
  int32_t *hits = bq_lds;
  int32_t *cnts = bq_lds + (size_t)nsample * (T + 1);
  const int bs = blockIdx.y;
  const int t = threadIdx.x;
  const int p = blockIdx.x * T + t;
  const bool live = p < m;
  const float *c = new_xyz + ((size_t)bs * m + (live ? p : 0)) * 3;
  const float cx = c[0], cy = c[1], cz = c[2];
  const float *X = xyz + (size_t)bs * n * 3;
  int cnt = live ? 0 : nsample;
  for (int k = 0; k < n; ++k) {
    if (__ballot(cnt < nsample) == 0ull) break;          // every centre of this wave is full
    const float d2 = dcl_dist2(cx, cy, cz, X[k * 3], X[k * 3 + 1], X[k * 3 + 2]);
    if (d2 < radius2 && cnt < nsample) {
      hits[(size_t)cnt * (T + 1) + t] = k;
      ++cnt;
    }
  }
  cnts[t] = live ? cnt : 0;
  __syncthreads();
  // coalesced write-out: pad with the first hit (ball_query_gpu.cu:35-39), zeros if none
  const int p0 = blockIdx.x * T;
  const int rows = min(T, m - p0);
  int32_t *o = idx + ((size_t)bs * m + p0) * nsample;
  for (int e = t; e < rows * nsample; e += T) {
    const int r = e / nsample, s = e - r * nsample;
    const int cr = cnts[r];
    int v = 0;
    if (cr > 0) v = hits[(size_t)(s < cr ? s : 0) * (T + 1) + r];
    o[e] = v;
  }
}

// ---------------------------------------------------------------------------- group / gather
// thread = 4 consecutive outputs of one (b, p*ns+s) run, looped over a chunk of channels.
template <int CCHUNK>
__global__ void k_group_points(int c, int n, int nps /* npoints*nsample */, const float *__restrict__ points,
                               const int32_t *__restrict__ idx, float *__restrict__ out) {
  const int bs = blockIdx.z;
  const int c0 = blockIdx.y * CCHUNK;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;        // quad index
  if (q * 4 >= nps) return;
  const int4 id = reinterpret_cast<const int4 *>(idx + (size_t)bs * nps)[q];
  const float *P = points + ((size_t)bs * c + c0) * n;
  float *O = out + ((size_t)bs * c + c0) * nps;
#pragma unroll
  for (int j = 0; j < CCHUNK; ++j) {
    if (c0 + j >= c) break;
    float4 v;
    v.x = P[id.x]; v.y = P[id.y]; v.z = P[id.z]; v.w = P[id.w];
    reinterpret_cast<float4 *>(O)[q] = v;
    P += n; O += nps;
  }
}

__global__ void k_group_points_scalar(int c, int n, int nps, const float *__restrict__ points,
                                      const int32_t *__restrict__ idx, float *__restrict__ out) {
  const int bs = blockIdx.z, ch = blockIdx.y;
  const float *P = points + ((size_t)bs * c + ch) * n;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nps; e += gridDim.x * blockDim.x)
    out[((size_t)bs * c + ch) * nps + e] = P[idx[(size_t)bs * nps + e]];
}

// ------------------------------------------------------------------------------------------ FPS
// Tie rule of the reference: block size TR = min(2^floor(log2 N), 1024); thread t scans
// k = t, t+TR, ... keeping the earliest strict maximum; the shared-memory tree keeps the LEFT
// operand on ties, which over all levels selects, among equal maxima, the thread with the smallest
// bit-reversed id (the last level compares slot 0 vs slot 1, i.e. the id's LSB decides first).
// The reduction below uses that total order explicitly: larger d wins, then smaller brev(t).
struct Cand { float d; int i; unsigned key; };
__device__ __forceinline__ Cand better(const Cand &a, const Cand &b) {
  return (b.d > a.d || (b.d == a.d && b.key < a.key)) ? b : a;
}

template <int R>   // R = points per thread (registers), block = TR threads
__global__ __launch_bounds__(1024) void k_fps(int n, int m, int TR, int log2TR, const float *__restrict__ dataset,
                                              float *__restrict__ temp, int32_t *__restrict__ idxs) {
  __shared__ float s_d[16];
  __shared__ int s_i[16];
  __shared__ unsigned s_k[16];
  __shared__ float s_old[3];
  __shared__ int s_oldi;
  const int bs = blockIdx.x;
  const int t = threadIdx.x;
  const float *X = dataset + (size_t)bs * n * 3;
  float *tp = temp + (size_t)bs * n;
  int32_t *out = idxs + (size_t)bs * m;
  float px[R], py[R], pz[R], td[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int k = t + r * TR;
    if (k < n) { px[r] = X[k * 3]; py[r] = X[k * 3 + 1]; pz[r] = X[k * 3 + 2]; td[r] = tp[k]; }
    else { px[r] = py[r] = pz[r] = 0.f; td[r] = 0.f; }
  }
  const unsigned key = __brev((unsigned)t) >> (32 - (log2TR > 0 ? log2TR : 1));
  if (t == 0) { out[0] = 0; s_old[0] = X[0]; s_old[1] = X[1]; s_old[2] = X[2]; }
  __syncthreads();
  const int nw = TR >> 6 ? TR >> 6 : 1;
  for (int j = 1; j < m; ++j) {
    const float x1 = s_old[0], y1 = s_old[1], z1 = s_old[2];
    Cand c; c.d = -1.0f; c.i = 0; c.key = log2TR > 0 ? key : 0u;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int k = t + r * TR;
      if (k < n) {
        const float d = dcl_dist2(px[r], py[r], pz[r], x1, y1, z1);
        const float d2 = fminf(d, td[r]);
        td[r] = d2;
        if (d2 > c.d) { c.d = d2; c.i = k; }
      }
    }
    // wave argmax
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      Cand o;
      o.d = __shfl_xor(c.d, s, 64); o.i = __shfl_xor(c.i, s, 64); o.key = __shfl_xor(c.key, s, 64);
      c = better(c, o);
    }
    __syncthreads();                                   // previous iteration's readers of s_* are done
    if ((t & 63) == 0) { s_d[t >> 6] = c.d; s_i[t >> 6] = c.i; s_k[t >> 6] = c.key; }
    __syncthreads();
    if (t < 64) {
      Cand w;
      if (t < nw) { w.d = s_d[t]; w.i = s_i[t]; w.key = s_k[t]; } else { w.d = -2.0f; w.i = 0; w.key = 0xffffffffu; }
#pragma unroll
      for (int s = 8; s >= 1; s >>= 1) {
        Cand o;
        o.d = __shfl_xor(w.d, s, 64); o.i = __shfl_xor(w.i, s, 64); o.key = __shfl_xor(w.key, s, 64);
        w = better(w, o);
      }
      if (t == 0) {
        s_oldi = w.i; out[j] = w.i;
        s_old[0] = X[w.i * 3]; s_old[1] = X[w.i * 3 + 1]; s_old[2] = X[w.i * 3 + 2];
      }
    }
    __syncthreads();
  }
  // the reference leaves the running min-distances in temp[]
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int k = t + r * TR;
    if (k < n) tp[k] = td[r];
  }
  (void)s_oldi;
}

// generic fallback: any n, distances kept in global temp[] (like the reference)
__global__ __launch_bounds__(1024) void k_fps_generic(int n, int m, int TR, int log2TR,
                                                      const float *__restrict__ dataset, float *__restrict__ temp,
                                                      int32_t *__restrict__ idxs) {
  __shared__ float s_d[16];
  __shared__ int s_i[16];
  __shared__ unsigned s_k[16];
  __shared__ int s_old;
  const int bs = blockIdx.x, t = threadIdx.x;
  const float *X = dataset + (size_t)bs * n * 3;
  float *tp = temp + (size_t)bs * n;
  int32_t *out = idxs + (size_t)bs * m;
  const unsigned key = log2TR > 0 ? __brev((unsigned)t) >> (32 - log2TR) : 0u;
  if (t == 0) { out[0] = 0; s_old = 0; }
  __syncthreads();
  const int nw = TR >> 6 ? TR >> 6 : 1;
  for (int j = 1; j < m; ++j) {
    const int old = s_old;
    const float x1 = X[old * 3], y1 = X[old * 3 + 1], z1 = X[old * 3 + 2];
    Cand c; c.d = -1.0f; c.i = 0; c.key = key;
    if (t < TR)
      for (int k = t; k < n; k += TR) {
        const float d2 = fminf(dcl_dist2(X[k * 3], X[k * 3 + 1], X[k * 3 + 2], x1, y1, z1), tp[k]);
        tp[k] = d2;
        if (d2 > c.d) { c.d = d2; c.i = k; }
      }
    else { c.d = -2.0f; c.key = 0xffffffffu; }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      Cand o;
      o.d = __shfl_xor(c.d, s, 64); o.i = __shfl_xor(c.i, s, 64); o.key = __shfl_xor(c.key, s, 64);
      c = better(c, o);
    }
    __syncthreads();
    if ((t & 63) == 0) { s_d[t >> 6] = c.d; s_i[t >> 6] = c.i; s_k[t >> 6] = c.key; }
    __syncthreads();
    if (t < 64) {
      Cand w;
      if (t < nw) { w.d = s_d[t]; w.i = s_i[t]; w.key = s_k[t]; } else { w.d = -2.0f; w.i = 0; w.key = 0xffffffffu; }
#pragma unroll
      for (int s = 8; s >= 1; s >>= 1) {
        Cand o;
        o.d = __shfl_xor(w.d, s, 64); o.i = __shfl_xor(w.i, s, 64); o.key = __shfl_xor(w.key, s, 64);
        w = better(w, o);
      }
      if (t == 0) { s_old = w.i; out[j] = w.i; }
    }
    __syncthreads();
  }
}

int fps_block_size(int n) {            // opt_n_threads, libs/pointnet_lib/src/cuda_utils.h:10-14
  const int pow_2 = (int)(log((double)n) / log(2.0));
  int t = 1 << pow_2;
  if (t > 1024) t = 1024;
  return t < 1 ? 1 : t;
}

}  // namespace

DCL_API int dcl_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz,
                           int32_t *idx, dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && nsample > 0);
  if (b == 0 || m == 0) return 0;
  DCL_CHECK_ARG(new_xyz && idx && (n == 0 || xyz) && b <= 65535);
  hipStream_t s = (hipStream_t)stream;
  const float r2 = radius * radius;
  // LDS: nsample*(T+1) + T ints; pick the largest T in {256,128,64} that fits 160 KiB / 2 blocks
  const size_t budget = 80 * 1024;
  int T = 256;
  while (T > 64 && ((size_t)nsample * (T + 1) + T) * 4 > budget) T >>= 1;
  const size_t lds = ((size_t)nsample * (T + 1) + T) * 4;
  DCL_CHECK_ARG(lds <= 160 * 1024);
#define BQ(TT)                                                                                                 \
  do {                                                                                                         \
    if (lds > 48 * 1024)                                                                                       \
      (void)hipFuncSetAttribute((const void *)k_ball_query<TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((k_ball_query<TT>), dim3(dcl_div_up(m, TT), b), dim3(TT), lds, s, n, m, r2, nsample,     \
                       new_xyz, xyz, idx);                                                                     \
  } while (0)
  if (T == 256) BQ(256); else if (T == 128) BQ(128); else BQ(64);
#undef BQ
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_group_points(int b, int c, int n, int npoints, int nsample, const float *points, const int32_t *idx,
                             float *out, dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
  const long long nps = (long long)npoints * nsample;
  if (b == 0 || c == 0 || nps == 0) return 0;
  DCL_CHECK_ARG(points && idx && out && b <= 65535 && c <= 65535 && nps < (1ll << 31));
  hipStream_t s = (hipStream_t)stream;
  if (nps % 4 == 0) {
    constexpr int CC = 8;
    hipLaunchKernelGGL((k_group_points<CC>), dim3(dcl_div_up(nps / 4, 256), dcl_div_up(c, CC), b), dim3(256), 0, s, c,
                       n, (int)nps, points, idx, out);
  } else {
    hipLaunchKernelGGL(k_group_points_scalar, dim3(dcl_grid_1d(nps, 256, 1024), c, b), dim3(256), 0, s, c, n, (int)nps,
                       points, idx, out);
  }
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_gather_points(int b, int c, int n, int npoints, const float *points, const int32_t *idx, float *out,
                              dclStream_t stream) {
  // gather_points is group_points with nsample == 1 (sampling_gpu.cu:8-24 vs group_points_gpu.cu:47-66)
  return dcl_group_points(b, c, n, npoints, 1, points, idx, out, stream);
}

DCL_API int dcl_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp, int32_t *idxs,
                                        dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && n >= 0 && m >= 0);
  if (b == 0 || m == 0) return 0;
  DCL_CHECK_ARG(n > 0 && dataset && temp && idxs);
  hipStream_t s = (hipStream_t)stream;
  const int TR = fps_block_size(n);
  int log2TR = 0;
  while ((1 << log2TR) < TR) ++log2TR;
  const int R = dcl_div_up(n, TR);
  const int block = TR < 64 ? 64 : TR;
  if (TR >= 64 && R <= 16) {
#define FPS(RR) hipLaunchKernelGGL((k_fps<RR>), dim3(b), dim3(TR), 0, s, n, m, TR, log2TR, dataset, temp, idxs)
    if (R <= 1) FPS(1); else if (R <= 2) FPS(2); else if (R <= 4) FPS(4); else if (R <= 8) FPS(8);
    else if (R <= 12) FPS(12); else FPS(16);
#undef FPS
  } else {
    hipLaunchKernelGGL(k_fps_generic, dim3(b), dim3(block), 0, s, n, m, TR, log2TR, dataset, temp, idxs);
  }
  DCL_LAUNCH_CHECK();
  return 0;
}

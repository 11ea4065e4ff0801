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
// FOUR lanes per centre.  Per 256-candidate tile (staged in LDS as x[]/y[]/z[] arrays, broadcast read, packed f32x2 math)
// each of the 4 lanes tests a contiguous 64-candidate quarter and builds a 64-bit hit mask -- branch-free, no
// stores; the quarter masks, taken in order, list the hits in ascending index, so the group's leader appends the
// first `nsample` of them to its LDS column ([slot][centre], +1 padded) at a cost proportional to the HITS, not the
// candidates.  A workgroup (one wave, 16 centres) leaves as soon as all its centres are full; rows go out coalesced,
// padded with the first hit (ball_query_gpu.cu:35-39).
constexpr int kBQTile = 256;
constexpr int kBQCentres = 16;                 // centres per workgroup = one wave: it leaves as soon as ITS 16 are full
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void k_ball_query(int n, int m, float radius2, int nsample,
                                                   const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                                                   int32_t *__restrict__ idx) {
  extern __shared__ int32_t bq_lds[];          // hits[nsample][17], cnt[16], tile x[256] y[256] z[256]
  constexpr int P = kBQCentres + 1;
  int32_t *hits = bq_lds;
  int32_t *cnts = bq_lds + (size_t)nsample * P;
  float *tx = reinterpret_cast<float *>(bq_lds + (((size_t)nsample * P + kBQCentres + 3) & ~(size_t)3));
  float *ty = tx + kBQTile, *tz = ty + kBQTile;
  const int bs = blockIdx.y;
  const int t = threadIdx.x;
  const int sub = t & 3, cs = t >> 2;
  const int p = blockIdx.x * kBQCentres + cs;
  const bool live = p < m;
  const float *c = new_xyz + ((size_t)bs * m + (live ? p : 0)) * 3;
  const f32x2 cx = {c[0], c[0]}, cy = {c[1], c[1]}, cz = {c[2], c[2]};
  const float *X = xyz + (size_t)bs * n * 3;
  int cnt = live ? 0 : nsample;                // kept identical in the 4 lanes of a group
  for (int base = 0; base < n; base += kBQTile) {
    if (__ballot(cnt < nsample) == 0ull) break;          // all 16 centres of this wave are full
    const int tn = min(kBQTile, n - base);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kBQTile / 64; ++i) {
      const int j = t + i * 64;
      float x = 3.0e38f, y = 3.0e38f, z = 3.0e38f;       // padding rows: never within any radius
      if (j < tn) { const float *q = X + (size_t)(base + j) * 3; x = q[0]; y = q[1]; z = q[2]; }
      tx[j] = x; ty[j] = y; tz[j] = z;
    }
    __syncthreads();
    // two candidates per packed instruction; same fma association as dcl_dist2, per component
    unsigned mlo = 0u, mhi = 0u;
    const f32x2 *px = reinterpret_cast<const f32x2 *>(tx + sub * 64);
    const f32x2 *py = reinterpret_cast<const f32x2 *>(ty + sub * 64);
    const f32x2 *pz = reinterpret_cast<const f32x2 *>(tz + sub * 64);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const f32x2 dx = cx - px[i], dy = cy - py[i], dz = cz - pz[i];
      const f32x2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
      const unsigned b0 = d2.x < radius2 ? 1u : 0u, b1 = d2.y < radius2 ? 1u : 0u;
      if (i < 16) mlo |= (b0 << (2 * i)) | (b1 << (2 * i + 1));
      else mhi |= (b0 << (2 * i - 32)) | (b1 << (2 * i - 31));
    }
    const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
    // the leader (sub 0) walks the 4 quarter masks in index order
    const int lead = t & ~3;
    unsigned long long mq[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) mq[s] = __shfl(mask, lead + s, 64);
    if (sub == 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        unsigned long long mm = mq[s];
        while (mm != 0ull && cnt < nsample) {
          const int bit = __ffsll((long long)mm) - 1;
          mm &= mm - 1ull;
          hits[(size_t)cnt * P + cs] = base + s * 64 + bit;
          ++cnt;
        }
      }
    }
    cnt = __shfl(cnt, lead, 64);
  }
  if (sub == 0) cnts[cs] = live ? cnt : 0;
  __syncthreads();
  const int p0 = blockIdx.x * kBQCentres;
  const int rows = min(kBQCentres, m - p0);
  int32_t *o = idx + ((size_t)bs * m + p0) * nsample;
  for (int e = t; e < rows * nsample; e += 64) {
    const int r = e / nsample, s = e - r * nsample;
    const int cr = cnts[r];
    int v = 0;
    if (cr > 0) v = hits[(size_t)(s < cr ? s : 0) * P + r];
    o[e] = v;
  }
}

// ---------------------------------------------------------------------------- group / gather
// LDS-staged gather: a workgroup owns CC channel rows of one cloud (CC*N floats in LDS, filled with coalesced 16-B
// loads) and streams the whole (npoints*nsample) index list against them: indices are read once per CC channels,
// every output leaves as a 16-B store, and the random reads hit LDS instead of the vector-memory path.
template <int CC>
__global__ __launch_bounds__(1024) void k_group_points_lds(int c, int n, int nps, const float *__restrict__ points,
                                                          const int32_t *__restrict__ idx, float *__restrict__ out) {
  extern __shared__ float gp_lds[];            // [CC][n]
  const int bs = blockIdx.z;
  const int c0 = blockIdx.y * CC;
  const int ncc = min(CC, c - c0);
  const float *P = points + ((size_t)bs * c + c0) * n;
  for (int j = threadIdx.x; j < ncc * n; j += 1024) gp_lds[j] = P[j];
  __syncthreads();
  const int nq = nps >> 2;
  const int4 *I = reinterpret_cast<const int4 *>(idx + (size_t)bs * nps);
  float *O = out + ((size_t)bs * c + c0) * nps;
#pragma unroll 2
  for (int q = blockIdx.x * 1024 + threadIdx.x; q < nq; q += gridDim.x * 1024) {
    const int4 id = I[q];
#pragma unroll
    for (int j = 0; j < CC; ++j) {
      if (j >= ncc) break;
      const float *row = gp_lds + j * n;
      float4 v;
      v.x = row[id.x]; v.y = row[id.y]; v.z = row[id.z]; v.w = row[id.w];
      reinterpret_cast<float4 *>(O + (size_t)j * nps)[q] = v;
    }
  }
}

// direct-gather variant (rows too long for LDS, or npoints*nsample not a multiple of 4)
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
  const size_t lds = ((((size_t)nsample * (kBQCentres + 1) + kBQCentres + 3) & ~(size_t)3)) * 4 + (size_t)kBQTile * 12;
  DCL_CHECK_ARG(lds <= 160 * 1024);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void *)k_ball_query, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_ball_query, dim3(dcl_div_up(m, kBQCentres), b), dim3(64), lds, s, n, m, r2, nsample, new_xyz,
                     xyz, idx);
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
  if (nps % 4 == 0 && n <= 36 * 1024 && nps >= 4096) {
    // LDS-staged rows: as many channel rows per workgroup as fit ~144 KiB (1 workgroup of 512 threads per CU)
    const int cc = (int)((144 * 1024) / ((size_t)n * 4));
    const size_t lds = (size_t)(cc >= 4 ? 4 : cc >= 3 ? 3 : cc >= 2 ? 2 : 1) * n * 4;
    const int ccu = cc >= 4 ? 4 : cc >= 3 ? 3 : cc >= 2 ? 2 : 1;
    const int ychunks = dcl_div_up(c, ccu);
    // enough x-blocks to give every CU work, few enough that the row fill stays a small fraction
    int xb = dcl_div_up(256 * 2, ychunks * b);
    if (xb < 1) xb = 1;
    const int max_xb = dcl_div_up(nps / 4, 1024 * 8);
    if (xb > max_xb) xb = max_xb > 0 ? max_xb : 1;
#define GPL(CCU)                                                                                               \
  do {                                                                                                         \
    (void)hipFuncSetAttribute((const void *)k_group_points_lds<CCU>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                       \
    hipLaunchKernelGGL((k_group_points_lds<CCU>), dim3(xb, ychunks, b), dim3(1024), lds, s, c, n, (int)nps, points, \
                       idx, out);                                                                              \
  } while (0)
    if (ccu == 4) GPL(4); else if (ccu == 3) GPL(3); else if (ccu == 2) GPL(2); else GPL(1);
#undef GPL
  } else if (nps % 4 == 0) {
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

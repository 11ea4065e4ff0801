// pointnet.hip -- PointNet++ set-abstraction primitives (libs/pointnet_lib/src/*.cu).
//
//   ball_query   ball_query_gpu.cu:9-45     first `nsample` indices (ascending) with d2 < r2
//   group_points group_points_gpu.cu:47-66  out[b,c,p,s] = points[b,c,idx[b,p,s]]
//   gather_points sampling_gpu.cu:8-24      out[b,c,m]   = points[b,c,idx[b,m]]
//   furthest_point_sampling sampling_gpu.cu:86-253
//
// MI355X mapping.  ball_query: 16 lanes per centre over LDS-staged 1024-candidate super-tiles, branch-free packed
// distance tests into 64-bit hit masks, prefix-sum compaction, workgroup-level early exit once its 16 centres are
// full; hits are staged in LDS ([slot][centre], +1 padded) so the final (B,M,nsample) rows leave as coalesced stores.
// group_points/gather_points: pure HBM streaming, write-dominated -- channel rows of a cloud staged in LDS, each lane
// owns 4 consecutive outputs (one 16-B store).  FPS: one workgroup per cloud, points + running min-distance held in
// REGISTERS for the whole run (the reference re-reads dataset[] and temp[] from global memory every iteration), wave
// argmax by DPP/shuffle, one LDS exchange + two barriers per iteration.
#include "common.h"
#include <atomic>
#include <math.h>

namespace {

// ------------------------------------------------------------------------------------ ball query
// A workgroup (4 waves) owns 16 centres; SIXTEEN lanes per centre.  The cloud is walked in super-tiles of 1024
// candidates staged in LDS as x[]/y[]/z[] arrays (16 segments of 64, pitch 66 floats: the 16 segment readers of a
// ds_read_b64 fall on 16 different bank pairs, the 4 centres sharing a segment broadcast).  Each lane tests ITS 64-candidate
// segment with packed f32x2 math and builds a 64-bit hit mask -- branch-free, no stores: the hit bit is the sign of
// (d2 - r2), shifted in with one v_alignbit per candidate.  A 16-lane prefix sum over the popcounts gives every segment
// its first output slot, so all 16 lanes append their own hits in parallel (cost ~ hits/16, not candidates) to the LDS
// column of the centre ([slot][centre], +1 padded).  The workgroup leaves as soon as all its 16 centres are full; rows go
// out coalesced, padded with the first hit (ball_query_gpu.cu:35-39).  Sparse clouds (few hits, whole-cloud scans) keep
// 4x the lanes per centre of the earlier one-wave layout busy, which is what bounded the kernel (tail of slow waves).
constexpr int kBQSuper = 1024;                 // candidates per staged super-tile
constexpr int kBQSegPitch = 66;                // floats between the 64-candidate segments
constexpr int kBQCentres = 16;                 // centres per workgroup
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_ball_query(int n, int m, float radius2, int nsample,
                                                    const float *__restrict__ new_xyz, const float *__restrict__ xyz,
                                                    int32_t *__restrict__ idx) {
  extern __shared__ int32_t bq_lds[];          // hits[nsample][17], cnt[16], flags[4], x[16*66] y[] z[]
  constexpr int P = kBQCentres + 1;
  int32_t *hits = bq_lds;
  int32_t *cnts = bq_lds + (size_t)nsample * P;
  int32_t *flags = cnts + kBQCentres;
  float *tx = reinterpret_cast<float *>(bq_lds + (((size_t)nsample * P + kBQCentres + 4 + 3) & ~(size_t)3));
  float *ty = tx + 16 * kBQSegPitch, *tz = ty + 16 * kBQSegPitch;
  const int bs = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int seg = t & 15, cs = t >> 4;
  const int p = blockIdx.x * kBQCentres + cs;
  const bool live = p < m;
  const float *c = new_xyz + ((size_t)bs * m + (live ? p : 0)) * 3;
  const f32x2 cx = {c[0], c[0]}, cy = {c[1], c[1]}, cz = {c[2], c[2]};
  const f32x2 r2v = {radius2, radius2};
  const float *X = xyz + (size_t)bs * n * 3;
  const bool vec_ok = ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
  int cnt = live ? 0 : nsample;                // kept identical in the 16 lanes of a centre
  for (int base = 0; base < n; base += kBQSuper) {
    const unsigned long long open = __ballot(cnt < nsample);
    if (lane == 0) flags[wave] = open != 0ull;
    __syncthreads();                                     // also: everyone is done with the previous super-tile
    if ((flags[0] | flags[1] | flags[2] | flags[3]) == 0) break;          // all 16 centres are full
    {
      // stage 4 consecutive candidates per thread (12 floats); rows past n: never within any radius
      const int j0 = base + 4 * t;
      float v[12];
      if (vec_ok && j0 + 4 <= n) {
        const float4 *q = reinterpret_cast<const float4 *>(X + (size_t)j0 * 3);
        const float4 a = q[0], b4 = q[1], c4 = q[2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
        v[8] = c4.x; v[9] = c4.y; v[10] = c4.z; v[11] = c4.w;
      } else {
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = (j0 + k / 3 < n) ? X[(size_t)j0 * 3 + k] : 3.0e38f;
      }
      const int o = (t >> 4) * kBQSegPitch + ((4 * t) & 63);             // segment of candidate 4t, offset inside it
#pragma unroll
      for (int u = 0; u < 4; ++u) { tx[o + u] = v[3 * u]; ty[o + u] = v[3 * u + 1]; tz[o + u] = v[3 * u + 2]; }
    }
    __syncthreads();
    // two candidates per packed instruction; same fma association as dcl_dist2, per component
    unsigned mlo = 0u, mhi = 0u;
    const f32x2 *px = reinterpret_cast<const f32x2 *>(tx + seg * kBQSegPitch);
    const f32x2 *py = reinterpret_cast<const f32x2 *>(ty + seg * kBQSegPitch);
    const f32x2 *pz = reinterpret_cast<const f32x2 *>(tz + seg * kBQSegPitch);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const f32x2 dx = cx - px[i], dy = cy - py[i], dz = cz - pz[i];
      const f32x2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
      const f32x2 sg = d2 - r2v;                         // sign bit set <=> d2 < r2 (exact for finite/inf operands)
      if (i < 16) {
        mlo = __builtin_amdgcn_alignbit(mlo, __float_as_uint(sg.x), 31);
        mlo = __builtin_amdgcn_alignbit(mlo, __float_as_uint(sg.y), 31);
      } else {
        mhi = __builtin_amdgcn_alignbit(mhi, __float_as_uint(sg.x), 31);
        mhi = __builtin_amdgcn_alignbit(mhi, __float_as_uint(sg.y), 31);
      }
    }
    unsigned long long mask = ((unsigned long long)__brev(mhi) << 32) | __brev(mlo);   // bit j = candidate j of the segment
    // first output slot of this segment: 16-lane exclusive prefix of the hit counts
    const int pc = __popcll(mask);
    int incl = pc;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      const int up = __shfl_up(incl, d, 16);
      if (seg >= d) incl += up;
    }
    const int total = __shfl(incl, 15, 16);
    int slot = cnt + incl - pc;
    const int j0 = base + seg * 64;
    while (mask != 0ull && slot < nsample) {
      const int bit = __ffsll((long long)mask) - 1;
      mask &= mask - 1ull;
      hits[(size_t)slot * P + cs] = j0 + bit;
      ++slot;
    }
    cnt = min(cnt + total, nsample);
  }
  if (seg == 0) cnts[cs] = live ? cnt : 0;
  __syncthreads();
  const int p0 = blockIdx.x * kBQCentres;
  const int rows = min(kBQCentres, m - p0);
  int32_t *o = idx + ((size_t)bs * m + p0) * nsample;
  for (int e = t; e < rows * nsample; e += 256) {
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
template <int CC, bool NT>
__global__ __launch_bounds__(1024) void k_group_points_lds(int c, int n, int nps, int oc, const float *__restrict__ points,
                                                          const int32_t *__restrict__ idx, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float gp_lds[];            // [CC][n]
  const int bs = blockIdx.z;
  const int c0 = blockIdx.y * CC;
  const int ncc = min(CC, c - c0);
  const int nthr = blockDim.x;
  const float *P = points + ((size_t)bs * c + c0) * n;
  const int tot = ncc * n;
  if ((reinterpret_cast<uintptr_t>(P) & 15) == 0) {
    for (int j = threadIdx.x; j < (tot >> 2); j += nthr)
      reinterpret_cast<float4 *>(gp_lds)[j] = reinterpret_cast<const float4 *>(P)[j];
    for (int j = (tot & ~3) + threadIdx.x; j < tot; j += nthr) gp_lds[j] = P[j];
  } else {
    for (int j = threadIdx.x; j < tot; j += nthr) gp_lds[j] = P[j];
  }
  __syncthreads();
  const int nq = nps >> 2;
  const int4 *I = reinterpret_cast<const int4 *>(idx + (size_t)bs * nps);
  float *O = out + ((size_t)bs * oc + c0) * nps;          // oc = channels per batch entry of the output tensor (>= c)
#pragma unroll 2
  for (int q = blockIdx.x * nthr + threadIdx.x; q < nq; q += gridDim.x * nthr) {
    const int4 id = I[q];
#pragma unroll
    for (int j = 0; j < CC; ++j) {
      if (j >= ncc) break;
      const float *row = gp_lds + j * n;
      float4 v;
      v.x = row[id.x]; v.y = row[id.y]; v.z = row[id.z]; v.w = row[id.w];
      float4 *dst = reinterpret_cast<float4 *>(O + (size_t)j * nps) + q;
      if (NT) {
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        const f32x4_t nv = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(nv, reinterpret_cast<f32x4_t *>(dst));
      } else {
        *dst = v;
      }
    }
  }
}

// direct-gather variant (rows too long for LDS, or npoints*nsample not a multiple of 4)
template <int CCHUNK>
__global__ void k_group_points(int c, int n, int nps /* npoints*nsample */, int oc, const float *__restrict__ points,
                               const int32_t *__restrict__ idx, float *__restrict__ out) {
  const int bs = blockIdx.z;
  const int c0 = blockIdx.y * CCHUNK;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;        // quad index
  if (q * 4 >= nps) return;
  const int4 id = reinterpret_cast<const int4 *>(idx + (size_t)bs * nps)[q];
  const float *P = points + ((size_t)bs * c + c0) * n;
  float *O = out + ((size_t)bs * oc + c0) * nps;
#pragma unroll
  for (int j = 0; j < CCHUNK; ++j) {
    if (c0 + j >= c) break;
    float4 v;
    v.x = P[id.x]; v.y = P[id.y]; v.z = P[id.z]; v.w = P[id.w];
    reinterpret_cast<float4 *>(O)[q] = v;
    P += n; O += nps;
  }
}

__global__ void k_group_points_scalar(int c, int n, int nps, int oc, const float *__restrict__ points,
                                      const int32_t *__restrict__ idx, float *__restrict__ out) {
  const int bs = blockIdx.z, ch = blockIdx.y;
  const float *P = points + ((size_t)bs * c + ch) * n;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nps; e += gridDim.x * blockDim.x)
    out[((size_t)bs * oc + ch) * nps + e] = P[idx[(size_t)bs * nps + e]];
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

// Register-resident kernel.  The (larger d, then smaller brev(t)) order is the order of the pair
//   (bits(d), (TR-1 - brev(t)) << 22 | k)        (d >= 0, so its bit pattern is monotonic; k < 2^22)
// reduced with DPP inside the waves, one LDS exchange of the waves' winners, ONE barrier per iteration (the exchange buffer
// alternates), then every wave reduces the <= 16 wave winners itself, so nobody waits for a broadcast.  The winner's
// coordinates come from an LDS copy of the cloud (n*12 B, up to ~13k points; beyond that from global memory) instead of a
// dependent global load on the critical path.
// 32-bit maxima by DPP: every lane of a 16-lane row gets the row's maximum (the compiler folds the DPP move into v_max_u32),
// then the four rows' values meet in scalar registers.  The (d, key) order of the reduction is taken in two 32-bit passes --
// first the largest distance bits, then, among the lanes that hold it, the largest low word -- which is the same total order
// as one unsigned 64-bit maximum at under half the vector instructions (a 64-bit step is two DPP moves, a 64-bit compare and
// two selects).
__device__ __forceinline__ unsigned row_max_u32(unsigned v) {
  unsigned o;
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); v = o > v ? o : v;    // quad_perm [1,0,3,2]
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); v = o > v ? o : v;    // quad_perm [2,3,0,1]
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); v = o > v ? o : v;   // row_half_mirror
  o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true); v = o > v ? o : v;   // row_mirror
  return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {      // -> the wave's maximum, uniform (scalar)
  v = row_max_u32(v);
  const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}

template <int R, bool PTS_IN_LDS>   // R = points per thread (registers), block = TR threads (TR >= 64)
__global__ __launch_bounds__(1024) void k_fps(int n, int m, int TR, int log2TR, const float *__restrict__ dataset,
                                              float *__restrict__ temp, int32_t *__restrict__ idxs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fps_lds[];
  unsigned long long *xch = reinterpret_cast<unsigned long long *>(fps_lds);        // [2][64] row winners
  float *pts = reinterpret_cast<float *>(fps_lds + 2 * 64 * sizeof(unsigned long long));   // [n][3] when PTS_IN_LDS
  const int bs = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float *X = dataset + (size_t)bs * n * 3;
  float *tp = temp + (size_t)bs * n;
  int32_t *out = idxs + (size_t)bs * m;
  constexpr int R2 = (R + 1) / 2;
  f32x2 px[R2], py[R2], pz[R2], td[R2];                // points r and r+1 share a register pair (packed math below)
#pragma unroll
  for (int r = 0; r < 2 * R2; ++r) {
    const int k = t + r * TR;
    float x = 0.f, y = 0.f, z = 0.f, d = -1.0f;          // slots past n: min-distance -1 never beats a real one (>= 0)
    if (r < R && k < n) { x = X[k * 3]; y = X[k * 3 + 1]; z = X[k * 3 + 2]; d = tp[k]; }
    px[r >> 1][r & 1] = x; py[r >> 1][r & 1] = y; pz[r >> 1][r & 1] = z; td[r >> 1][r & 1] = d;
  }
  if (PTS_IN_LDS)
    for (int j = t; j < 3 * n; j += TR) pts[j] = X[j];
  for (int q = t; q < 128; q += TR) xch[q] = 0ull;    // waves that do not exist never win
  const unsigned key = __brev((unsigned)t) >> (32 - log2TR);
  const unsigned low_base = ((unsigned)(TR - 1) - key) << 22;
  if (t == 0) out[0] = 0;
  __syncthreads();
  float x1 = X[0], y1 = X[1], z1 = X[2];
  for (int j = 1; j < m; ++j) {
    float bd = -1.0f;
    int br = 0;                                          // the thread's best slot r (its point index is t + r * TR)
    {
      // two points per packed instruction, dcl_dist2's association per component
      const f32x2 x2 = {x1, x1}, y2 = {y1, y1}, z2 = {z1, z1};
#pragma unroll
      for (int q = 0; q < R2; ++q) {
        const f32x2 dx = px[q] - x2, dy = py[q] - y2, dz = pz[q] - z2;
        const f32x2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
        // (v_min_f32 directly: fminf() would first canonicalise both operands -- one more instruction per point in a loop
        // that is bound by vector-instruction issue; distances and the running minima are never NaN)
        float da, db;
        asm("v_min_f32 %0, %1, %2" : "=v"(da) : "v"(d.x), "v"(td[q].x));
        asm("v_min_f32 %0, %1, %2" : "=v"(db) : "v"(d.y), "v"(td[q].y));
        td[q].x = da; td[q].y = db;                      // branch-free: padded slots carry -1 and never win
        if (da > bd) { bd = da; br = 2 * q; }            // (slot numbers are inline constants: no index arithmetic per point)
        if (db > bd) { bd = db; br = 2 * q + 1; }
      }
    }
    // reduction in the reference's order -- larger d, then smaller brev(t) -- in two 32-bit passes (see row_max_u32): inside
    // the wave, ONE exchange of the 16 waves' (d, low) pairs through LDS and ONE barrier (the buffer alternates), then every
    // wave finishes on the 16 pairs itself, so nobody waits for a broadcast.  d >= 0: its bit pattern is monotonic; a thread
    // without a live point (bd = -1) enters as (0, 0) and cannot win: every live low word is > 0.
    const bool has = bd >= 0.0f;
    const unsigned hi = has ? __float_as_uint(bd) : 0u;
    const unsigned lo = has ? (low_base | (unsigned)(t + br * TR)) : 0u;
    const unsigned whi = wave_max_u32(hi);
    const unsigned wlo = wave_max_u32(hi == whi ? lo : 0u);
    unsigned long long *buf = xch + (j & 1) * 64;
    if (lane == 0) buf[wave] = ((unsigned long long)whi << 32) | wlo;
    __syncthreads();
    const unsigned long long e = buf[lane & 15];         // the 16 waves' pairs in every row
    const unsigned ehi = (unsigned)(e >> 32), elo = (unsigned)e;
    const unsigned ghi = row_max_u32(ehi);
    const unsigned glo = row_max_u32(ehi == ghi ? elo : 0u);
    const int wi = (int)(glo & 0x3fffffu);
    if (PTS_IN_LDS) { x1 = pts[wi * 3]; y1 = pts[wi * 3 + 1]; z1 = pts[wi * 3 + 2]; }
    else { x1 = X[wi * 3]; y1 = X[wi * 3 + 1]; z1 = X[wi * 3 + 2]; }
    if (t == 0) out[j] = wi;
  }
  // the reference leaves the running min-distances in temp[]
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int k = t + r * TR;
    if (k < n) tp[k] = td[r >> 1][r & 1];
  }
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
  const size_t lds = ((((size_t)nsample * (kBQCentres + 1) + kBQCentres + 4 + 3) & ~(size_t)3)) * 4 +
                     (size_t)3 * 16 * kBQSegPitch * 4;
  DCL_CHECK_ARG(lds <= 160 * 1024);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void *)k_ball_query, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_ball_query, dim3(dcl_div_up(m, kBQCentres), b), dim3(256), lds, s, n, m, r2, nsample, new_xyz,
                     xyz, idx);
  DCL_LAUNCH_CHECK();
  return 0;
}

#ifdef DCL_DIAG
static std::atomic<int> g_gp_cfg[4] = {{0}, {0}, {0}, {0}};   // tuning hook: rows per workgroup, x-blocks, threads, stores (2 = plain); 0 = default
DCL_API void dcl_debug_group_points_cfg(int cc, int xb, int threads, int nontemporal) {
  g_gp_cfg[0] = cc; g_gp_cfg[1] = xb; g_gp_cfg[2] = threads; g_gp_cfg[3] = nontemporal;
}
#else
static constexpr int g_gp_cfg[4] = {0, 0, 0, 0};               // product: the built-in choices
#endif

DCL_API int dcl_group_points(int b, int c, int n, int npoints, int nsample, const float *points, const int32_t *idx,
                             float *out, dclStream_t stream) {
  return dcl_group_points_into(b, c, n, npoints, nsample, points, idx, out, c, stream);
}

DCL_API int dcl_group_points_into(int b, int c, int n, int npoints, int nsample, const float *points, const int32_t *idx,
                                  float *out, int oc, dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0 && oc >= c);
  const long long nps = (long long)npoints * nsample;
  if (b == 0 || c == 0 || nps == 0) return 0;
  DCL_CHECK_ARG(points && idx && out && b <= 65535 && c <= 65535 && nps < (1ll << 31));
  hipStream_t s = (hipStream_t)stream;
  if (nps % 4 == 0 && n <= 36 * 1024 && nps >= 4096) {
    // LDS-staged rows: ~48 KiB of channel rows per workgroup, so that 2 workgroups of 1024 threads share a CU and one
    // fills its rows while the other streams (measured best at the north-star shape: 1 row of 12288 floats, 5.97 TB/s)
    const int cc = (int)((48 * 1024) / ((size_t)n * 4));
    int ccu = cc >= 4 ? 4 : cc >= 3 ? 3 : cc >= 2 ? 2 : 1;
    if (g_gp_cfg[0] > 0 && g_gp_cfg[0] <= 4 && (size_t)g_gp_cfg[0] * n * 4 <= 144 * 1024) ccu = (int)g_gp_cfg[0];
    const size_t lds = (size_t)ccu * n * 4;
    const int threads = g_gp_cfg[2] > 0 ? (int)g_gp_cfg[2] : 1024;
    const int ychunks = dcl_div_up(c, ccu);
    // enough x-blocks to give every CU work, few enough that the row fill stays a small fraction
    int xb = dcl_div_up(256 * 2, ychunks * b);
    if (xb < 1) xb = 1;
    const int max_xb = dcl_div_up(nps / 4, 1024 * 8);
    if (xb > max_xb) xb = max_xb > 0 ? max_xb : 1;
    if (g_gp_cfg[1] > 0) xb = (int)g_gp_cfg[1];
    const bool nt = g_gp_cfg[3] != 2;                  // nontemporal (streaming) stores unless the hook says 2 = plain
#define GPL(CCU, NTB)                                                                                               \
  do {                                                                                                         \
    (void)hipFuncSetAttribute((const void *)k_group_points_lds<CCU, NTB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                       \
    hipLaunchKernelGGL((k_group_points_lds<CCU, NTB>), dim3(xb, ychunks, b), dim3(threads), lds, s, c, n, (int)nps, oc, points, \
                       idx, out);                                                                              \
  } while (0)
#define GPL2(CCU) do { if (nt) GPL(CCU, true); else GPL(CCU, false); } while (0)
    if (ccu == 4) GPL2(4); else if (ccu == 3) GPL2(3); else if (ccu == 2) GPL2(2); else GPL2(1);
#undef GPL2
#undef GPL
  } else if (nps % 4 == 0) {
    constexpr int CC = 8;
    hipLaunchKernelGGL((k_group_points<CC>), dim3(dcl_div_up(nps / 4, 256), dcl_div_up(c, CC), b), dim3(256), 0, s, c,
                       n, (int)nps, oc, points, idx, out);
  } else {
    hipLaunchKernelGGL(k_group_points_scalar, dim3(dcl_grid_1d(nps, 256, 1024), c, b), dim3(256), 0, s, c, n, (int)nps,
                       oc, points, idx, out);
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
    const bool in_lds = (size_t)n * 12 + 1024 <= 156 * 1024;
    const size_t lds = 1024 + (in_lds ? (size_t)n * 12 : 0);
#define FPS_L(RR, L)                                                                                                  \
  do {                                                                                                                \
    if (lds > 48 * 1024)                                                                                              \
      (void)hipFuncSetAttribute((const void *)k_fps<RR, L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
    hipLaunchKernelGGL((k_fps<RR, L>), dim3(b), dim3(TR), lds, s, n, m, TR, log2TR, dataset, temp, idxs);             \
  } while (0)
#define FPS(RR) do { if (in_lds) FPS_L(RR, true); else FPS_L(RR, false); } while (0)
    if (R <= 1) FPS(1); else if (R <= 2) FPS(2); else if (R <= 4) FPS(4); else if (R <= 8) FPS(8);
    else if (R <= 12) FPS(12); else FPS(16);
#undef FPS_L
#undef FPS
  } else {
    hipLaunchKernelGGL(k_fps_generic, dim3(b), dim3(block), 0, s, n, m, TR, log2TR, dataset, temp, idxs);
  }
  DCL_LAUNCH_CHECK();
  return 0;
}

// interp.hip -- 3-nearest-neighbour search and inverse-distance interpolation.
//
// Replaces libs/pointnet_sp/src/interpolate_gpu.cu:9-122 (flat variants with a batch column, the
// ones DCL_Net.forward uses through models/Modules.py:213-227) and the batched variants + knn of
// libs/pointnet_lib/src/interpolate_gpu.cu:9-189.
//
// three_nn: one query per lane, kept in registers; the candidate range is wave-uniform, candidates
// are staged through a small LDS tile and broadcast-read.  With `known_seg` a query scans only its own crop's rows (the reference
// scans the whole batch and skips foreign rows: O(b^2)); results are identical because rows are
// visited in ascending k either way.  Tie rule = the reference's strict '<' cascade.
// three_interpolate: lanes run over channels of one point (coalesced row gathers + coalesced
// stores; the reference maps threads to points and is fully uncoalesced).
#include "common.h"
#include <atomic>
#include <math.h>

namespace {

struct Best3 {
  float d1, d2, d3;
  int i1, i2, i3;
  __device__ __forceinline__ void init() { d1 = d2 = d3 = INFINITY; i1 = i2 = i3 = 0; }   // (float)1e40 == +inf
  __device__ __forceinline__ void push(float d, int k) {
    if (d < d1) { d3 = d2; i3 = i2; d2 = d1; i2 = i1; d1 = d; i1 = k; }
    else if (d < d2) { d3 = d2; i3 = i2; d2 = d; i2 = k; }
    else if (d < d3) { d3 = d; i3 = k; }
  }
};

// Branch-free top-3: a candidate is the 64-bit key (bits(d2) << 32) | k.  d2 >= +0 so its bit pattern orders like
// the float, and equal distances order by k -- exactly the reference's strict-'<' cascade over ascending k.
// Empty slots hold (bits(+inf) << 32) | 0, i.e. dist2 = (float)1e40 = +inf, idx = 0.
typedef unsigned long long u64;
constexpr u64 kEmptyKey = 0x7f80000000000000ull;
struct Top3 {
  u64 k1, k2, k3;
  __device__ __forceinline__ void init() { k1 = k2 = k3 = kEmptyKey; }
  __device__ __forceinline__ void push(u64 key) {
    u64 t = key < k1 ? key : k1; key = key < k1 ? k1 : key; k1 = t;
    t = key < k2 ? key : k2; key = key < k2 ? k2 : key; k2 = t;
    k3 = key < k3 ? key : k3;
  }
};
__device__ __forceinline__ u64 make_key(float d, int k) { return ((u64)__float_as_uint(d) << 32) | (unsigned)k; }

// 16 queries per wave, 4 lanes per query: lane sub-id s scans candidates j = s (mod 4) of the wave-uniform range,
// staged through an LDS tile; the 4 partial top-3 lists are merged with two shuffle rounds (exact: keys are a
// total order).  No global-memory latency and no divergent branch in the inner loop.
constexpr int kNNTile = 512;
// voxel centre of a grid row (Ops_tensor2points, models/Modules.py:204-211): fp32, left to right
__device__ __forceinline__ float4 voxel_centre(const int4 v, float ve, float off, float half) {
  float4 c;
  c.x = (float)v.x;
  c.y = ((float)v.y * ve + off) + half;
  c.z = ((float)v.z * ve + off) + half;
  c.w = ((float)v.w * ve + off) + half;
  return c;
}

// FROM_INDICES: `known` holds the voxel rows (b,x,y,z) i32 and the centres are formed while the tile is staged
template <bool FROM_INDICES>
__global__ __launch_bounds__(64) void k_three_nn_sp(int n, int m, const float4 *__restrict__ unknown,
                                                    const float4 *__restrict__ known, float *__restrict__ dist2,
                                                    int32_t *__restrict__ idx, const int32_t *__restrict__ known_seg,
                                                    int nbatch, int seg_stride, float ve, float off) {
  __shared__ float4 tile[kNNTile];
  const int lane = threadIdx.x;
  const int sub = lane & 3;
  const int p = blockIdx.x * 16 + (lane >> 2);
  const bool live = p < n;
  float4 u = make_float4(-1.f, 0.f, 0.f, 0.f);
  if (live) u = unknown[p];
  int lo = 0, hi = m;
  if (known_seg) {
    const int bi = (int)u.x;
    if (live && bi >= 0 && bi < nbatch && (float)bi == u.x) {
      lo = known_seg[(size_t)bi * seg_stride]; hi = known_seg[(size_t)(bi + 1) * seg_stride];
    } else { lo = 0x7fffffff; hi = 0; }
  } else if (!live) { lo = 0x7fffffff; hi = 0; }
  int wlo = lo, whi = hi;                               // wave-uniform scan range = union of the lanes' ranges
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    wlo = min(wlo, __shfl_xor(wlo, d, 64));
    whi = max(whi, __shfl_xor(whi, d, 64));
  }
  wlo = __builtin_amdgcn_readfirstlane(wlo);
  whi = __builtin_amdgcn_readfirstlane(whi);
  Top3 b; b.init();
  for (int base = wlo; base < whi; base += kNNTile) {
    const int cnt = min(kNNTile, whi - base);
    __syncthreads();
    for (int j = lane; j < cnt; j += 64)
      tile[j] = FROM_INDICES ? voxel_centre(reinterpret_cast<const int4 *>(known)[base + j], ve, off, 0.5f * ve)
                             : known[base + j];
    __syncthreads();
#pragma unroll 4
    for (int j = sub; j < cnt; j += 4) {
      const float4 q = tile[j];
      const float d = dcl_dist2(u.y, u.z, u.w, q.y, q.z, q.w);
      b.push(q.x == u.x ? make_key(d, base + j) : ~0ull);            // interpolate_gpu.cu:36-38
    }
  }
  // merge the 4 sub-lane lists
#pragma unroll
  for (int d = 1; d <= 2; d <<= 1) {
    const u64 o1 = __shfl_xor(b.k1, d, 64), o2 = __shfl_xor(b.k2, d, 64), o3 = __shfl_xor(b.k3, d, 64);
    b.push(o1); b.push(o2); b.push(o3);
  }
  if (live && sub == 0) {
    dist2[p * 3 + 0] = __uint_as_float((unsigned)(b.k1 >> 32));
    dist2[p * 3 + 1] = __uint_as_float((unsigned)(b.k2 >> 32));
    dist2[p * 3 + 2] = __uint_as_float((unsigned)(b.k3 >> 32));
    idx[p * 3 + 0] = (int)(unsigned)b.k1; idx[p * 3 + 1] = (int)(unsigned)b.k2; idx[p * 3 + 2] = (int)(unsigned)b.k3;
  }
}

// Exact 3-NN against one level of the backbone's occupancy grid (S <= 32, word-aligned z rows), one thread per query:
// only the 5x5x5 cells around the query's own cell are visited (their rows come from the bitmask rank, their centres
// from the cell coordinates -- the same fp32 expressions as voxel_centre / dcl_dist2, so keys are bit-identical to the
// brute-force kernel's).  Every unvisited voxel lies >= `bound` away along one axis; if the third-best distance found
// is below that (with a 1e-4 margin for rounding) the answer is the global one, ties included (keys order (d, row)
// exactly like the sequential scan of interpolate_gpu.cu:36-38).  Otherwise -- isolated queries, queries outside
// the grid -- the thread scans its crop's rows.
// mask_lk / pre_lk + wbase: where the window lookups read the occupancy words and rank prefixes from -- the global
// arrays (wbase = 0) or the LDS copy of the point's crop (wbase = first word of the crop), see k_three_nn_grid_levels
// LPQ lanes per query (1 or 4, consecutive lanes; `sub` = the lane's share): with 4, the 25 (x, y) rows of the window and the
// rows of a fallback scan are dealt round-robin to the lanes, every lane keeps its own three best keys (and prunes with
// them), and two shuffle rounds merge the lists -- keys are a total order, so the merged triple is the sequential one.
// Few queries (the reference's 1024 points per crop at bs 32, one-image calls) leave a one-thread-per-query launch at the
// latency of its longest thread; many queries (12288 points per crop) fill the GPU either way and skip the merge.
template <int LPQ>
__device__ __forceinline__ void three_nn_grid_point(int p, int sub, const float4 u, const int4 *__restrict__ indices,
                                                    const uint32_t *__restrict__ mask,
                                                    const int32_t *__restrict__ wprefix, int nbatch, int S, int wpc,
                                                    float ve, float off, float *__restrict__ dist2,
                                                    int32_t *__restrict__ idx, int force_scan,
                                                    const uint32_t *mask_lk, const int32_t *pre_lk, int wbase) {
  const float half = 0.5f * ve;
  Top3 b; b.init();
  const int bi = (int)u.x;
  if (bi >= 0 && bi < nbatch && (float)bi == u.x) {
    const float pc[3] = {u.y, u.z, u.w};
    int lo[3], hi[3], cc[3];
    float bound = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float f = floorf((pc[a] - off) / ve);
      const int ci = f >= (float)(S - 1) ? S - 1 : (f > 0.0f ? (int)f : 0);            // NaN -> 0
      cc[a] = ci;
      lo[a] = max(ci - 2, 0); hi[a] = min(ci + 2, S - 1);
      if (ci - 3 >= 0) bound = fminf(bound, pc[a] - (((float)(ci - 3) * ve + off) + half));
      if (ci + 3 <= S - 1) bound = fminf(bound, (((float)(ci + 3) * ve + off) + half) - pc[a]);
    }
    const uint32_t zmask = (hi[2] - lo[2] == 31) ? 0xffffffffu : ((1u << (hi[2] - lo[2] + 1)) - 1u);
    // The window is visited from the point's own cell outwards, and a plane / a z-row is skipped once it cannot hold
    // anything better than the third best key so far: dcl_dist2 = fma(dz,dz, fma(dx,dx, dy*dy)) >= fma(dx,dx, dy*dy)
    // >= dx*dx in floating point too (dz*dz, dy*dy >= 0 and rounding is monotone), so "row bound > d3" excludes every key
    // of the row, ties included (an equal distance is not pruned).  Until three candidates exist d3 reads as NaN: no pruning.
    // The set of the three smallest keys does not depend on the visiting order.
    if (!force_scan && LPQ > 1) {
      const int order[5] = {0, -1, 1, -2, 2};
#pragma unroll 2
      for (int t = sub; t < 25; t += LPQ) {              // row t = (x-plane t / 5, y-row t % 5), centre-out in both
        const int x = cc[0] + order[t / 5], y = cc[1] + order[t % 5];
        if (x < lo[0] || x > hi[0] || y < lo[1] || y > hi[1]) continue;
        const float qx = ((float)x * ve + off) + half, qy = ((float)y * ve + off) + half;
        const float dx = u.y - qx, dy = u.z - qy;
        if (__fmaf_rn(dx, dx, dy * dy) > __uint_as_float((unsigned)(b.k3 >> 32))) continue;
        const int lin0 = ((bi * S + x) * S + y) * S;
        const int w = (lin0 >> 5) - wbase, sh = lin0 & 31;
        const uint32_t m = mask_lk[w];
        uint32_t bits = (m >> (sh + lo[2])) & zmask;
        if (bits == 0u) continue;
        const int pre = pre_lk[w];
        while (bits) {
          const int tz = __builtin_ctz(bits);
          bits &= bits - 1u;
          const int z = lo[2] + tz, pos = sh + z;
          const int row = pre + __popc(m & ((1u << pos) - 1u));
          const float qz = ((float)z * ve + off) + half;
          b.push(make_key(dcl_dist2(u.y, u.z, u.w, qx, qy, qz), row));
        }
      }
#pragma unroll
      for (int d = 1; d < LPQ; d <<= 1) {
        const u64 o1 = __shfl_xor(b.k1, d, 64), o2 = __shfl_xor(b.k2, d, 64), o3 = __shfl_xor(b.k3, d, 64);
        b.push(o1); b.push(o2); b.push(o3);
      }
    }
    if (!force_scan && LPQ == 1) {
      const int order[5] = {0, -1, 1, -2, 2};
#pragma unroll
      for (int ix = 0; ix < 5; ++ix) {
        const int x = cc[0] + order[ix];
        if (x < lo[0] || x > hi[0]) continue;
        const float qx = ((float)x * ve + off) + half;
        const float dx = u.y - qx;
        if (dx * dx > __uint_as_float((unsigned)(b.k3 >> 32))) continue;
        // the (up to) five z-rows of this x-plane: mask words and rank prefixes are fetched together (independent
        // loads) before any bit is examined -- one exposed memory latency per plane instead of two per row
        uint32_t mw[5];
        int pw[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int yy = cc[1] + order[j];
          const int y = yy < lo[1] ? lo[1] : (yy > hi[1] ? hi[1] : yy);
          const int w = ((((bi * S + x) * S + y) * S) >> 5) - wbase;
          mw[j] = mask_lk[w];
          pw[j] = pre_lk[w];
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int y = cc[1] + order[j];
          if (y < lo[1] || y > hi[1]) continue;
          const int lin0 = ((bi * S + x) * S + y) * S;
          const int sh = lin0 & 31;
          const uint32_t m = mw[j];
          uint32_t bits = (m >> (sh + lo[2])) & zmask;
          if (bits == 0u) continue;
          const float qy = ((float)y * ve + off) + half;
          const float dy = u.z - qy;
          if (__fmaf_rn(dx, dx, dy * dy) > __uint_as_float((unsigned)(b.k3 >> 32))) continue;
          const int pre = pw[j];
          while (bits) {
            const int t = __builtin_ctz(bits);
            bits &= bits - 1u;
            const int z = lo[2] + t, pos = sh + z;
            const int row = pre + __popc(m & ((1u << pos) - 1u));
            const float qz = ((float)z * ve + off) + half;
            b.push(make_key(dcl_dist2(u.y, u.z, u.w, qx, qy, qz), row));
          }
        }
      }
    }
    const float d3 = __uint_as_float((unsigned)(b.k3 >> 32));
    const bool certified = !force_scan && (bound == INFINITY || d3 < bound * bound * 0.9999f) && !(bound < 0.0f);
    if (!certified) {                                    // (uniform over a query's lanes: they hold the same merged keys)
      b.init();
      const int r0 = wprefix[(size_t)bi * wpc], r1 = wprefix[(size_t)(bi + 1) * wpc];
#pragma unroll 4
      for (int j = r0 + sub; j < r1; j += LPQ) {
        const float4 q = voxel_centre(indices[j], ve, off, half);
        b.push(q.x == u.x ? make_key(dcl_dist2(u.y, u.z, u.w, q.y, q.z, q.w), j) : ~0ull);
      }
#pragma unroll
      for (int d = 1; d < LPQ; d <<= 1) {
        const u64 o1 = __shfl_xor(b.k1, d, 64), o2 = __shfl_xor(b.k2, d, 64), o3 = __shfl_xor(b.k3, d, 64);
        b.push(o1); b.push(o2); b.push(o3);
      }
    }
  }
  if (sub != 0) return;
  dist2[p * 3 + 0] = __uint_as_float((unsigned)(b.k1 >> 32));
  dist2[p * 3 + 1] = __uint_as_float((unsigned)(b.k2 >> 32));
  dist2[p * 3 + 2] = __uint_as_float((unsigned)(b.k3 >> 32));
  idx[p * 3 + 0] = (int)(unsigned)b.k1; idx[p * 3 + 1] = (int)(unsigned)b.k2; idx[p * 3 + 2] = (int)(unsigned)b.k3;
}

__global__ __launch_bounds__(256) void k_three_nn_grid(int n, const float4 *__restrict__ unknown,
                                                       const int4 *__restrict__ indices, const uint32_t *__restrict__ mask,
                                                       const int32_t *__restrict__ wprefix, int nbatch, int S, int wpc,
                                                       float ve, float off, float *__restrict__ dist2,
                                                       int32_t *__restrict__ idx, int force_scan) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) three_nn_grid_point<1>(p, 0, unknown[p], indices, mask, wprefix, nbatch, S, wpc, ve, off, dist2, idx, force_scan, mask, wprefix, 0);
}

// all 4 levels of the read-out in one launch (blockIdx.y = level); dist2 / idx are level-major blocks of n*3
// Window lookups are per-lane gathers of single words: from global memory every lane of a load hits its own cache line
// (2 x 25 such loads per point and level: the kernel was bound by the request rate, not by arithmetic).  A crop's
// occupancy words and rank prefixes of one level are at most 2 x 4 KB (32^3 cells), so a workgroup whose 256 points
// belong to one crop -- the runner's point rows are crop-contiguous -- stages them in LDS first.
constexpr int kNnLdsWords = 1024;
template <int LPQ>
__global__ __launch_bounds__(256) void k_three_nn_grid_levels(int n, const float4 *__restrict__ unknown,
                                                              const DclReadoutLevels L, int nbatch, float off,
                                                              float *__restrict__ dist2, int32_t *__restrict__ idx,
                                                              int force_scan) {
  __shared__ uint32_t s_mask[kNnLdsWords];
  __shared__ int32_t s_pre[kNnLdsWords];
  constexpr int QPB = 256 / LPQ;                                            // queries per workgroup
  const int p = blockIdx.x * QPB + threadIdx.x / LPQ, sub = threadIdx.x % LPQ, m = blockIdx.y;
  const int wpc = L.wpc[m];
  const float4 u = p < n ? unknown[p] : make_float4(-1.f, 0.f, 0.f, 0.f);
  const float4 u0 = unknown[blockIdx.x * QPB];                              // uniform: the block's first point
  const int b0 = (int)u0.x;
  const bool crop_ok = b0 >= 0 && b0 < nbatch && (float)b0 == u0.x && wpc <= kNnLdsWords;
  const int same = __syncthreads_and((p >= n || u.x == u0.x) ? 1 : 0);
  const bool staged = crop_ok && same;
  if (staged) {
    for (int i = threadIdx.x; i < wpc; i += 256) {
      s_mask[i] = L.mask[m][(size_t)b0 * wpc + i];
      s_pre[i] = L.wprefix[m][(size_t)b0 * wpc + i];
    }
    __syncthreads();
  }
  if (p < n)
    three_nn_grid_point<LPQ>(p, sub, u, reinterpret_cast<const int4 *>(L.indices[m]), L.mask[m], L.wprefix[m], nbatch, L.S[m], wpc,
                        L.ve[m], off, dist2 + (size_t)m * n * 3, idx + (size_t)m * n * 3, force_scan,
                        staged ? s_mask : L.mask[m], staged ? s_pre : L.wprefix[m], staged ? b0 * wpc : 0);
}

// workgroup = kInterpPts points x all 4 levels: the three weights of a (point, level) pair -- square roots and
// correctly rounded divisions -- are formed ONCE into LDS (they used to be recomputed by each of the level's C/4
// threads: most of the kernel's VALU work), then the 480 channels stream out as 16-B nontemporal stores (the rows are
// read once, much later, by the disengage GEMM: no point in allocating them in L2).  Same arithmetic, same bits.
constexpr int kInterpPts = 32;
// one block of kInterpPts points (p0 ..) by NT threads: weights into LDS, then the 480 channels
template <int NT>
__device__ __forceinline__ void interpolate_block(int n, int p0, const DclReadoutLevels &L, const int32_t *__restrict__ idx,
                                                  const float *__restrict__ dist2, float *__restrict__ out, int ld,
                                                  float (*s_w)[3], int32_t (*s_i)[3]) {
  const int q0 = L.c[0] >> 2, q1 = q0 + (L.c[1] >> 2), q2 = q1 + (L.c[2] >> 2), qn = q2 + (L.c[3] >> 2);
  if (threadIdx.x < kInterpPts * 4) {
    const int pl = threadIdx.x / 4, m = threadIdx.x & 3, p = p0 + pl;
    if (p < n) {
      const size_t o = ((size_t)m * n + p) * 3;
      const float r0 = 1.0f / (sqrtf(dist2[o]) + 1e-8f), r1 = 1.0f / (sqrtf(dist2[o + 1]) + 1e-8f),
                  r2 = 1.0f / (sqrtf(dist2[o + 2]) + 1e-8f);
      const float norm = (r0 + r1) + r2;
      s_w[threadIdx.x][0] = r0 / norm; s_w[threadIdx.x][1] = r1 / norm; s_w[threadIdx.x][2] = r2 / norm;
      s_i[threadIdx.x][0] = idx[o]; s_i[threadIdx.x][1] = idx[o + 1]; s_i[threadIdx.x][2] = idx[o + 2];
    }
  }
  __syncthreads();
  const int npts = n - p0 < kInterpPts ? n - p0 : kInterpPts;
  for (int t = threadIdx.x; t < npts * qn; t += NT) {
    const int pl = t / qn;
    int q = t - pl * qn;
    const int m = q < q0 ? 0 : (q < q1 ? 1 : (q < q2 ? 2 : 3));
    q -= m == 0 ? 0 : (m == 1 ? q0 : (m == 2 ? q1 : q2));
    const int e = pl * 4 + m;
    const float w0 = s_w[e][0], w1 = s_w[e][1], w2 = s_w[e][2];
    const int c = L.c[m];
    const float *__restrict__ F = L.feats[m];
    const float4 a = reinterpret_cast<const float4 *>(F + (size_t)s_i[e][0] * c)[q];
    const float4 b = reinterpret_cast<const float4 *>(F + (size_t)s_i[e][1] * c)[q];
    const float4 d = reinterpret_cast<const float4 *>(F + (size_t)s_i[e][2] * c)[q];
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v;
    v.x = dcl_wsum3(w0, a.x, w1, b.x, w2, d.x);
    v.y = dcl_wsum3(w0, a.y, w1, b.y, w2, d.y);
    v.z = dcl_wsum3(w0, a.z, w1, b.z, w2, d.z);
    v.w = dcl_wsum3(w0, a.w, w1, b.w, w2, d.w);
    __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(out + (size_t)(p0 + pl) * ld + L.col[m]) + q);
  }
}
__global__ __launch_bounds__(256) void k_three_interpolate_levels(int n, const DclReadoutLevels L,
                                                                  const int32_t *__restrict__ idx,
                                                                  const float *__restrict__ dist2, float *__restrict__ out,
                                                                  int ld) {
  __shared__ float s_w[kInterpPts * 4][3];
  __shared__ int32_t s_i[kInterpPts * 4][3];
  for (int p0 = blockIdx.x * kInterpPts; p0 < n; p0 += gridDim.x * kInterpPts) {
    __syncthreads();
    interpolate_block<256>(n, p0, L, idx, dist2, out, ld, s_w, s_i);
  }
}

// The whole read-out of 32 points in ONE workgroup of 1024 threads (calls of up to kNnCoop8MaxQueries points, where the two
// launches above are a search of 8 lanes per query and level followed by an interpolation that waits for it): threads 256 m ..
// 256 m + 255 search level m exactly as k_three_nn_grid_levels<8> does (the crop's occupancy words and rank prefixes of all four
// levels staged in LDS), a barrier, then all 1024 threads interpolate the block's 32 x 480 channels -- dist2 / idx still go to
// memory (callers may ask for them) and come back from the CU's own cache.  One launch and one launch gap less per backbone.
__global__ __launch_bounds__(1024) void k_readout_levels(int n, const float4 *__restrict__ unknown, const DclReadoutLevels L,
                                                         int nbatch, float off, float *__restrict__ dist2,
                                                         int32_t *__restrict__ idx, int force_scan, float *__restrict__ out, int ld) {
  __shared__ uint32_t s_mask[4][kNnLdsWords];
  __shared__ int32_t s_pre[4][kNnLdsWords];
  __shared__ float s_w[kInterpPts * 4][3];
  __shared__ int32_t s_i[kInterpPts * 4][3];
  constexpr int LPQ = 8, QPB = 256 / LPQ;
  static_assert(QPB == kInterpPts, "one block of the interpolation per block of the search");
  const int m = (int)threadIdx.x >> 8, tl = (int)threadIdx.x & 255;
  const int p0 = blockIdx.x * QPB, p = p0 + tl / LPQ, sub = tl % LPQ;
  const int wpc = L.wpc[m];
  const float4 u = p < n ? unknown[p] : make_float4(-1.f, 0.f, 0.f, 0.f);
  const float4 u0 = unknown[p0];                                            // uniform: the block's first point
  const int b0 = (int)u0.x;
  const bool crop_ok = b0 >= 0 && b0 < nbatch && (float)b0 == u0.x;
  const int same = __syncthreads_and((p >= n || u.x == u0.x) ? 1 : 0);
  const bool staged = crop_ok && same && wpc <= kNnLdsWords;                // (per level: wave-group uniform)
  if (crop_ok && same) {                                                    // (block-uniform: the barrier below is reached by all)
    if (staged)
      for (int i = tl; i < wpc; i += 256) {
        s_mask[m][i] = L.mask[m][(size_t)b0 * wpc + i];
        s_pre[m][i] = L.wprefix[m][(size_t)b0 * wpc + i];
      }
    __syncthreads();
  }
  if (p < n)
    three_nn_grid_point<LPQ>(p, sub, u, reinterpret_cast<const int4 *>(L.indices[m]), L.mask[m], L.wprefix[m], nbatch, L.S[m], wpc,
                             L.ve[m], off, dist2 + (size_t)m * n * 3, idx + (size_t)m * n * 3, force_scan,
                             staged ? s_mask[m] : L.mask[m], staged ? s_pre[m] : L.wprefix[m], staged ? b0 * wpc : 0);
  __syncthreads();                                                          // the block's dist2 / idx are written (and visible)
  interpolate_block<1024>(n, p0, L, idx, dist2, out, ld, s_w, s_i);
}

// voxel centres (Ops_tensor2points, models/Modules.py:204-211): fp32, left to right.
__global__ void k_voxel_centres(const int4 *__restrict__ indices, const int32_t *__restrict__ n_dev, int n_host,
                                float ve, float off, float4 *__restrict__ centres) {
  const int n = n_dev ? *n_dev : n_host;
  const float half = 0.5f * ve;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    centres[i] = voxel_centre(indices[i], ve, off, half);
}

template <bool FROM_DIST2>
__global__ void k_three_interpolate_sp(int c, int n, const float *__restrict__ points, const int32_t *__restrict__ idx,
                                       const float *__restrict__ wsrc, float *__restrict__ out, int out_stride) {
  const int c4 = c >> 2;
  const long long total = (long long)n * c4;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(t / c4);
    const int q = (int)(t - (long long)p * c4);
    const int i0 = idx[p * 3], i1 = idx[p * 3 + 1], i2 = idx[p * 3 + 2];
    float w0 = wsrc[p * 3], w1 = wsrc[p * 3 + 1], w2 = wsrc[p * 3 + 2];
    if (FROM_DIST2) {                                   // models/Modules.py:221-224, pointnet2_utils.py:31
      const float r0 = 1.0f / (sqrtf(w0) + 1e-8f), r1 = 1.0f / (sqrtf(w1) + 1e-8f), r2 = 1.0f / (sqrtf(w2) + 1e-8f);
      const float norm = (r0 + r1) + r2;
      w0 = r0 / norm; w1 = r1 / norm; w2 = r2 / norm;
    }
    const float4 a = reinterpret_cast<const float4 *>(points + (size_t)i0 * c)[q];
    const float4 b = reinterpret_cast<const float4 *>(points + (size_t)i1 * c)[q];
    const float4 d = reinterpret_cast<const float4 *>(points + (size_t)i2 * c)[q];
    float4 o;
    o.x = dcl_wsum3(w0, a.x, w1, b.x, w2, d.x);
    o.y = dcl_wsum3(w0, a.y, w1, b.y, w2, d.y);
    o.z = dcl_wsum3(w0, a.z, w1, b.z, w2, d.z);
    o.w = dcl_wsum3(w0, a.w, w1, b.w, w2, d.w);
    reinterpret_cast<float4 *>(out + (size_t)p * out_stride)[q] = o;
  }
}

template <bool FROM_DIST2>
__global__ void k_three_interpolate_sp_scalar(int c, int n, const float *__restrict__ points,
                                              const int32_t *__restrict__ idx, const float *__restrict__ wsrc,
                                              float *__restrict__ out, int out_stride) {
  const long long total = (long long)n * c;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(t / c);
    const int ch = (int)(t - (long long)p * c);
    float w0 = wsrc[p * 3], w1 = wsrc[p * 3 + 1], w2 = wsrc[p * 3 + 2];
    if (FROM_DIST2) {
      const float r0 = 1.0f / (sqrtf(w0) + 1e-8f), r1 = 1.0f / (sqrtf(w1) + 1e-8f), r2 = 1.0f / (sqrtf(w2) + 1e-8f);
      const float norm = (r0 + r1) + r2;
      w0 = r0 / norm; w1 = r1 / norm; w2 = r2 / norm;
    }
    out[(size_t)p * out_stride + ch] =
        dcl_wsum3(w0, points[(size_t)idx[p * 3] * c + ch], w1, points[(size_t)idx[p * 3 + 1] * c + ch], w2,
                  points[(size_t)idx[p * 3 + 2] * c + ch]);
  }
}

// ---- batched (pointnet_lib) variants -----------------------------------------------------------
__global__ __launch_bounds__(256) void k_three_nn(int n, int m, const float *__restrict__ unknown,
                                                  const float *__restrict__ known, float *__restrict__ dist2,
                                                  int32_t *__restrict__ idx) {
  const int bs = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p < n;
  const float *u = unknown + ((size_t)bs * n + (live ? p : 0)) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  const float *K = known + (size_t)bs * m * 3;
  Best3 b; b.init();
  for (int k = 0; k < m; ++k)                           // uniform address -> scalar loads
    b.push(dcl_dist2(ux, uy, uz, K[k * 3], K[k * 3 + 1], K[k * 3 + 2]), k);
  if (live) {
    const size_t o = ((size_t)bs * n + p) * 3;
    dist2[o] = b.d1; dist2[o + 1] = b.d2; dist2[o + 2] = b.d3;
    idx[o] = b.i1; idx[o + 1] = b.i2; idx[o + 2] = b.i3;
  }
}

// ---- pruned exact search for k = 1 and k = 3 (knn with k = 1 is what DCL-Net's get_cano_label calls; three_nn is k = 3):
// the cloud's known points are bucketed along the axis of their largest extent (256 buckets, counting sort in LDS, exact
// per-bucket minimum / maximum coordinate); the workgroup's QUERIES are sorted by their bucket too, so that the 64 queries of
// a wave sit next to each other on the axis and walk the buckets TOGETHER: first the contiguous run of buckets the wave's own
// queries fall into, then outwards in both directions, a group of buckets at a time, until no lane of the wave can still be
// beaten (the squared coordinate gap to the next bucket exceeds its worst kept distance).  Every lane tests every point the
// wave visits -- a superset of what it needs -- so the point reads are LDS broadcasts and the control flow is uniform.  (Round 4
// let every lane walk its own buckets: 64 divergent walks per wave, 0.196 ms at B = 32, n = 12288, m = 2048; the plain scan 0.55.)
// Exact: candidates are 64-bit (d2, index) keys -- the reference's strict-'<' cascade over ascending k is the lexicographic
// order of those keys, so the scan order is free -- and d2 = fma(dz,dz, fma(dx,dx, dy*dy)) >= fl(gap*gap) for the gap along ANY
// one axis (rounding is monotonic), so a bucket whose nearest coordinate is further than the worst kept distance cannot
// contribute, ties included (the walk stops on a strictly greater bound only).
#ifdef DCL_DIAG
// diagnostic library: (query, point) distance evaluations the bucketed search EXECUTES (every lane of a wave tests every point
// the wave visits) -- bench.py prices the kernel against these, not against the brute-force pair count it no longer performs
__device__ unsigned long long g_nn_tests_executed;
#endif
constexpr int kNNBuckets = 256;
constexpr int kNNThreads = 512;           // QPT queries per thread: the bucket build (per workgroup) is shared by 512 * QPT queries
template <int KB, int QPT>
__global__ __launch_bounds__(kNNThreads) void k_nn_bucketed(int n, int m, const float *__restrict__ unknown,
                                                     const float *__restrict__ known, float *__restrict__ dist2,
                                                     int32_t *__restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) float nn_lds[];
  float4 *pts = reinterpret_cast<float4 *>(nn_lds);                        // [m] (x, y, z, index bits), bucket by bucket
  float4 *sq = pts + m;                                                    // [512 * QPT] the workgroup's queries (x, y, z, local index bits), bucket by bucket
  int *start = reinterpret_cast<int *>(sq + kNNThreads * QPT);             // [kNNBuckets + 1]
  int *fill = start + kNNBuckets + 1;                                      // [kNNBuckets] counts, then fill cursors (points, then queries)
  unsigned *bmin = reinterpret_cast<unsigned *>(fill + kNNBuckets);        // [kNNBuckets] monotone-uint minimum coordinate
  unsigned *bmax = bmin + kNNBuckets;                                      // [kNNBuckets]
  float *red = reinterpret_cast<float *>(bmax + kNNBuckets);               // [6][NWV] wave partials of the extent
  constexpr int NWV = kNNThreads / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bs = blockIdx.y;
  const float *K = known + (size_t)bs * m * 3;
  // 1. extent of the known points along the three axes
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = tid; i < m; i += kNNThreads)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float c = K[i * 3 + a];
      lo[a] = fminf(lo[a], c);
      hi[a] = fmaxf(hi[a], c);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64));
    }
  if (lane == 0)
#pragma unroll
    for (int a = 0; a < 3; ++a) { red[a * NWV + wave] = lo[a]; red[3 * NWV + a * NWV + wave] = hi[a]; }
  if (tid < kNNBuckets) {
    fill[tid] = 0;
    bmin[tid] = 0xffffffffu;
    bmax[tid] = 0u;
  }
  __syncthreads();
  int axis = 0;
  float cmin = 0.f, inv_w = 0.f;
  {
    float best = -1.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float l = INFINITY, h = -INFINITY;
#pragma unroll
      for (int w = 0; w < NWV; ++w) {
        l = fminf(l, red[a * NWV + w]);
        h = fmaxf(h, red[3 * NWV + a * NWV + w]);
      }
      if (h - l > best) { best = h - l; axis = a; cmin = l; }
    }
    inv_w = best > 0.f ? (float)kNNBuckets / best : 0.f;
  }
  auto bucket_of = [&](float c) -> int {                                   // monotone non-decreasing in c
    const float f = (c - cmin) * inv_w;
    int q = f > 0.f ? (int)fminf(f, (float)(kNNBuckets - 1)) : 0;
    return q;
  };
  auto mono = [](float c) -> unsigned {                                    // float order -> unsigned order
    const unsigned u = __float_as_uint(c);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  };
  auto unmono = [](unsigned u) -> float { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); };
  // exclusive scan of the 256 counts in fill[] -> cursors in fill[] (and start[] when `keep`); waves 0-3 hold the counts
  auto scan_counts = [&](bool keep) {
    const int cnt = tid < kNNBuckets ? fill[tid] : 0;
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    __syncthreads();                                                       // (everyone has read its count, and the extent partials)
    if (lane == 63 && wave < kNNBuckets / 64) red[wave] = __int_as_float(inc);
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < kNNBuckets / 64; ++w) base += w < wave ? __float_as_int(red[w]) : 0;
    if (tid < kNNBuckets) {
      fill[tid] = base + inc - cnt;                                        // fill cursor
      if (keep) {
        start[tid] = base + inc - cnt;
        if (tid == kNNBuckets - 1) start[kNNBuckets] = base + inc;
      }
    }
    __syncthreads();
  };
  // 2. counting sort of the known points into the buckets (the order inside a bucket is free: keys are a total order)
  for (int i = tid; i < m; i += kNNThreads) {
    const float c = K[i * 3 + axis];
    const int q = bucket_of(c);
    atomicAdd(&fill[q], 1);
    atomicMin(&bmin[q], mono(c));
    atomicMax(&bmax[q], mono(c));
  }
  __syncthreads();
  scan_counts(true);
  for (int i = tid; i < m; i += kNNThreads) {
    const float x = K[i * 3], y = K[i * 3 + 1], z = K[i * 3 + 2];
    const float c = axis == 0 ? x : (axis == 1 ? y : z);
    const int pos = atomicAdd(&fill[bucket_of(c)], 1);
    pts[pos] = make_float4(x, y, z, __int_as_float(i));
  }
  __syncthreads();
  // 3. this workgroup's queries, sorted by bucket (same counting sort; fill[] is free again)
  const int q_base = blockIdx.x * kNNThreads * QPT;
  const int nq = min(kNNThreads * QPT, n - q_base);
  const float *U = unknown + ((size_t)bs * n + q_base) * 3;
  if (tid < kNNBuckets) fill[tid] = 0;
  __syncthreads();
  float qx[QPT], qy[QPT], qz[QPT];
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    const int p = j * kNNThreads + tid;
    qx[j] = qy[j] = qz[j] = 0.f;
    if (p < nq) {
      qx[j] = U[p * 3]; qy[j] = U[p * 3 + 1]; qz[j] = U[p * 3 + 2];
      atomicAdd(&fill[bucket_of(axis == 0 ? qx[j] : (axis == 1 ? qy[j] : qz[j]))], 1);
    }
  }
  __syncthreads();
  scan_counts(false);
#pragma unroll
  for (int j = 0; j < QPT; ++j) {
    const int p = j * kNNThreads + tid;
    if (p < nq) {
      const int pos = atomicAdd(&fill[bucket_of(axis == 0 ? qx[j] : (axis == 1 ? qy[j] : qz[j]))], 1);
      sq[pos] = make_float4(qx[j], qy[j], qz[j], __int_as_float(p));
    }
  }
  __syncthreads();
  // 4. the sorted queries, 64 neighbours on the axis per wave
#pragma unroll 1
  for (int j = 0; j < QPT; ++j) {
    const int sidx = j * kNNThreads + tid;
    if (__ballot(sidx < nq) == 0ull) break;                                // (whole waves leave together: no barrier below)
    const bool valid = sidx < nq;
    const float4 me = sq[valid ? sidx : 0];
    const float ux = me.x, uy = me.y, uz = me.z;
    const float uc = axis == 0 ? ux : (axis == 1 ? uy : uz);
    u64 key[KB];
#pragma unroll
    for (int t = 0; t < KB; ++t) key[t] = kEmptyKey;
    float worst = INFINITY;                                                // distance of the last kept key
    auto consider = [&](float d, float w) {
      if (d > worst) return;                                               // cannot enter (an equal distance may: lower index)
      u64 c = make_key(d, __float_as_int(w));
#pragma unroll
      for (int t = 0; t < KB; ++t) {
        const u64 lo2 = c < key[t] ? c : key[t];
        c = c < key[t] ? key[t] : c;
        key[t] = lo2;
      }
      worst = __uint_as_float((unsigned)(key[KB - 1] >> 32));
    };
    auto scan_range = [&](int i0, int i1) {                                // every lane tests points [i0, i1): uniform bounds, broadcast reads
      int i = i0;
#ifdef DCL_DIAG
      if (lane == 0 && i1 > i0) atomicAdd(&g_nn_tests_executed, (unsigned long long)(i1 - i0) * 64ull);
#endif
      for (; i + 8 <= i1; i += 8) {                                        // eight points' reads and distances in flight, entered in order
        float4 v[8];
        float d[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = pts[i + e];
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = dcl_dist2(ux, uy, uz, v[e].x, v[e].y, v[e].z);
        const float dm = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
        if (__ballot(dm <= worst) == 0ull) continue;                       // nobody in the wave can use any of the eight
#pragma unroll
        for (int e = 0; e < 8; ++e) consider(d[e], v[e].w);
      }
      for (; i + 4 <= i1; i += 4) {
        const float4 v0 = pts[i], v1 = pts[i + 1], v2 = pts[i + 2], v3 = pts[i + 3];
        const float d0 = dcl_dist2(ux, uy, uz, v0.x, v0.y, v0.z), d1 = dcl_dist2(ux, uy, uz, v1.x, v1.y, v1.z);
        const float d2 = dcl_dist2(ux, uy, uz, v2.x, v2.y, v2.z), d3 = dcl_dist2(ux, uy, uz, v3.x, v3.y, v3.z);
        if (__ballot(fminf(fminf(d0, d1), fminf(d2, d3)) <= worst) == 0ull) continue;
        consider(d0, v0.w); consider(d1, v1.w); consider(d2, v2.w); consider(d3, v3.w);
      }
      for (; i < i1; ++i) {
        const float4 v = pts[i];
        consider(dcl_dist2(ux, uy, uz, v.x, v.y, v.z), v.w);
      }
    };
    // the wave's own buckets: one contiguous run of the sorted points
    const int q0 = bucket_of(uc);
    int qlo = valid ? q0 : kNNBuckets, qhi = valid ? q0 : -1;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      qlo = min(qlo, __shfl_xor(qlo, d, 64));
      qhi = max(qhi, __shfl_xor(qhi, d, 64));
    }
    qlo = __builtin_amdgcn_readfirstlane(qlo);
    qhi = __builtin_amdgcn_readfirstlane(qhi);
    scan_range(start[qlo], start[qhi + 1]);
    // outwards, kStep buckets at a time: the group is scanned if ANY lane could still be beaten by its nearest bucket
    constexpr int kStep = 4;
    for (int q = qhi + 1; q < kNNBuckets; q += kStep) {
      const int qe = min(q + kStep, kNNBuckets);
      if (start[qe] == start[q]) continue;
      int qn = q;                                                          // first non-empty bucket of the group: its minimum bounds the group
      while (start[qn + 1] == start[qn]) ++qn;
      const float gap = unmono(bmin[qn]) - uc;                             // every point from here on is at least this far along the axis
      const bool done = !valid || (gap > 0.f && gap * gap > worst);
      if (__ballot(!done) == 0ull) break;
      scan_range(start[q], start[qe]);
    }
    for (int q = qlo - 1; q >= 0; q -= kStep) {
      const int qb = max(q - kStep + 1, 0);
      if (start[q + 1] == start[qb]) continue;
      int qn = q;                                                          // last non-empty bucket of the group
      while (start[qn + 1] == start[qn]) --qn;
      const float gap = uc - unmono(bmax[qn]);
      const bool done = !valid || (gap > 0.f && gap * gap > worst);
      if (__ballot(!done) == 0ull) break;
      scan_range(start[qb], start[q + 1]);
    }
    if (valid) {
      const size_t o = ((size_t)bs * n + q_base + __float_as_int(me.w)) * KB;
#pragma unroll
      for (int t = 0; t < KB; ++t) {
        dist2[o + t] = __uint_as_float((unsigned)(key[t] >> 32));
        idx[o + t] = (int)(unsigned)key[t];
      }
    }
  }
}
constexpr int kNNBucketedMaxKnown = 8192;       // 128 KiB of staged points
static size_t nn_bucketed_lds(int m, int qpt) {
  return (size_t)m * 16 + (size_t)kNNThreads * qpt * 16 + (size_t)(4 * kNNBuckets + 1 + 6 * (kNNThreads / 64) + 4) * 4;
}
// queries per thread: as many as keep >= 256 workgroups in the launch and fit the LDS beside the staged points (more queries per
// workgroup = tighter waves on the axis and fewer bucket builds)
DCL_HOOK_INT(g_nn_qpt, 0);            // (diagnostic library) queries per thread of the bucketed search: 0 = automatic
static int nn_bucketed_qpt(int b, int n, int m) {
  if (g_nn_qpt >= 1 && g_nn_qpt <= 4 && nn_bucketed_lds(m, (int)g_nn_qpt) <= 150 * 1024) return (int)g_nn_qpt;
  int qpt = 1;
  for (int c = 2; c <= 4; ++c)
    if ((long long)b * dcl_div_up(n, kNNThreads * c) >= 256 && nn_bucketed_lds(m, c) <= 150 * 1024) qpt = c;
  return qpt;
}

// knn, k <= 200 (interpolate_gpu.cu:9-57): sorted list with strict-'<' insertion, one query per
// thread, list kept in a per-thread LDS column (conflict-free: slot j of thread t at j*T + t).
constexpr int kKnnThreads = 64;
__global__ __launch_bounds__(kKnnThreads) void k_knn(int n, int m, int k, const float *__restrict__ unknown,
                                                     const float *__restrict__ known, float *__restrict__ dist2,
                                                     int32_t *__restrict__ idx) {
  extern __shared__ float knn_lds[];
  float *bd = knn_lds;                                   // [k][T]
  int *bi = reinterpret_cast<int *>(knn_lds + (size_t)k * kKnnThreads);
  const int t = threadIdx.x;
  const int bs = blockIdx.y;
  const int p = blockIdx.x * kKnnThreads + t;
  const bool live = p < n;
  const float *u = unknown + ((size_t)bs * n + (live ? p : 0)) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  const float *K = known + (size_t)bs * m * 3;
  for (int j = 0; j < k; ++j) { bd[j * kKnnThreads + t] = INFINITY; bi[j * kKnnThreads + t] = 0; }
  for (int i = 0; i < m; ++i) {
    const float d = dcl_dist2(ux, uy, uz, K[i * 3], K[i * 3 + 1], K[i * 3 + 2]);
    if (!(d < bd[(k - 1) * kKnnThreads + t])) continue;  // cannot enter the list
    int j = k - 1;                                       // shift larger entries down, stable
    while (j > 0 && d < bd[(j - 1) * kKnnThreads + t]) {
      bd[j * kKnnThreads + t] = bd[(j - 1) * kKnnThreads + t];
      bi[j * kKnnThreads + t] = bi[(j - 1) * kKnnThreads + t];
      --j;
    }
    bd[j * kKnnThreads + t] = d;
    bi[j * kKnnThreads + t] = i;
  }
  if (live)
    for (int j = 0; j < k; ++j) {
      dist2[((size_t)bs * n + p) * k + j] = bd[j * kKnnThreads + t];
      idx[((size_t)bs * n + p) * k + j] = bi[j * kKnnThreads + t];
    }
}

__global__ void k_three_interpolate(int c, int m, int n, const float *__restrict__ points,
                                    const int32_t *__restrict__ idx, const float *__restrict__ weight,
                                    float *__restrict__ out) {
  const int bs = blockIdx.z, ch = blockIdx.y;
  const float *P = points + ((size_t)bs * c + ch) * m;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const size_t o = ((size_t)bs * n + p) * 3;
    out[((size_t)bs * c + ch) * n + p] =
        dcl_wsum3(weight[o], P[idx[o]], weight[o + 1], P[idx[o + 1]], weight[o + 2], P[idx[o + 2]]);
  }
}

}  // namespace

// internal: known_seg[b * seg_stride] (lets the backbone runner use a wprefix array in place)
int dcl_three_nn_sp_strided(int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx,
                            const int32_t *known_seg, int nbatch, int seg_stride, dclStream_t stream) {
  DCL_CHECK_ARG(n >= 0 && m >= 0);
  if (n == 0) return 0;
  DCL_CHECK_ARG(unknown && dist2 && idx && (m == 0 || known) && (!known_seg || (nbatch > 0 && seg_stride > 0)));
  hipLaunchKernelGGL(k_three_nn_sp<false>, dim3(dcl_div_up(n, 16)), dim3(64), 0, (hipStream_t)stream, n, m,
                     reinterpret_cast<const float4 *>(unknown), reinterpret_cast<const float4 *>(known), dist2, idx,
                     known_seg, nbatch, seg_stride, 0.0f, 0.0f);
  DCL_LAUNCH_CHECK();
  return 0;
}

// internal: the known set is given as voxel rows (b,x,y,z) i32; their centres idx*ve + off + ve/2 are formed in the kernel
DCL_HOOK_INT(g_nn_grid, 1);   // (atomic in the diagnostic library, constant in the product) 0 = brute-force scan per crop for every level, 2 = grid kernel with the scan forced, 3 / 4 / 5 = one / four / eight lanes per query
constexpr int kNnCoopMaxQueries = 1 << 17;   // read-outs of up to this many points search with four lanes per query,
constexpr int kNnCoop8MaxQueries = 40960;    // up to this many (bs 40 x 1024 points) with eight
DCL_HOOK_INT(g_nn_batched_mode, 0);   // (diagnostic library) 1 = the batched three_nn / knn as plain scans (A/B of the bucketed search)
#ifdef DCL_DIAG
DCL_API void dcl_debug_three_nn_grid(int mode) { g_nn_grid = mode; }
DCL_API void dcl_debug_nn_batched_mode(int mode) { g_nn_batched_mode = mode; }
DCL_API void dcl_debug_nn_qpt(int q) { g_nn_qpt = q; }
DCL_API unsigned long long dcl_debug_nn_tests_executed(int reset) {
  unsigned long long v = 0;
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_nn_tests_executed), sizeof(v));
  if (reset) {
    const unsigned long long z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nn_tests_executed), &z, sizeof(z));
  }
  return v;
}

#endif

// known_mask / S (optional): the level's occupancy bits and grid size; with them, levels of S = 16 or 32 go through the
// grid-pruned kernel (same results)
int dcl_three_nn_sp_voxels(int n, int m, const float *unknown, const int32_t *known_indices, float ve, float off,
                           float *dist2, int32_t *idx, const int32_t *known_seg, int nbatch, int seg_stride,
                           const uint32_t *known_mask, int S, dclStream_t stream) {
  DCL_CHECK_ARG(n >= 0 && m >= 0);
  if (n == 0) return 0;
  DCL_CHECK_ARG(unknown && dist2 && idx && (m == 0 || known_indices) && (!known_seg || (nbatch > 0 && seg_stride > 0)));
  if (g_nn_grid && known_mask && known_seg && known_indices && (S == 16 || S == 32) && seg_stride == S * S * S / 32 &&
      ve > 0.0f) {
    hipLaunchKernelGGL(k_three_nn_grid, dim3(dcl_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n,
                       reinterpret_cast<const float4 *>(unknown), reinterpret_cast<const int4 *>(known_indices), known_mask,
                       known_seg, nbatch, S, seg_stride, ve, off, dist2, idx, g_nn_grid == 2 ? 1 : 0);
    DCL_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_three_nn_sp<true>, dim3(dcl_div_up(n, 16)), dim3(64), 0, (hipStream_t)stream, n, m,
                     reinterpret_cast<const float4 *>(unknown), reinterpret_cast<const float4 *>(known_indices), dist2, idx,
                     known_seg, nbatch, seg_stride, ve, off);
  DCL_LAUNCH_CHECK();
  return 0;
}

// internal (backbone.hip): the whole read-out of a backbone pass in two launches -- 3-NN searches of all 4 levels
// (grid-pruned, exact), then the interpolation of all 4 levels into the row buffer.  `fused_ok` says whether the levels
// qualify (grids of 4..32 cells per axis, channel counts and columns in float4 units); otherwise the caller goes level
// by level through dcl_three_nn_sp_voxels / dcl_three_interpolate_dist2_sp.
bool dcl_internal_readout_fused_ok(const DclReadoutLevels &L, int ld, bool need_search) {
  if (ld % 4 != 0) return false;
  for (int m = 0; m < 4; ++m) {
    if (L.c[m] <= 0 || L.c[m] % 4 != 0 || L.col[m] % 4 != 0 || L.col[m] + L.c[m] > ld) return false;
    if (!need_search) continue;
    const int S = L.S[m];
    if (!g_nn_grid || !(S == 4 || S == 8 || S == 16 || S == 32) || L.wpc[m] != S * S * S / 32 || !(L.ve[m] > 0.0f) ||
        !L.mask[m] || !L.wprefix[m] || !L.indices[m])
      return false;
  }
  return true;
}

int dcl_internal_readout_neighbours(int n, const float *points_b4, const DclReadoutLevels &L, int nbatch, float off,
                                    float *dist2, int32_t *idx, dclStream_t stream) {
  DCL_CHECK_ARG(n > 0 && points_b4 && dist2 && idx && nbatch > 0);
  // several lanes per query while the launch would otherwise be latency-bound (see three_nn_grid_point): measured on the
  // bs-32 sets, us for 1 / 4 / 8 lanes: 32768 points 137 / 44 / 36, 65536 points 141 / 53 / 59, 393216 points 172 / 186 / --.
  // hook: 3 / 4 / 5 force the one- / four- / eight-lane variant
  const int mode = g_nn_grid;
  const int lpq = mode == 3 ? 1 : mode == 4 ? 4 : mode == 5 ? 8 : n <= kNnCoop8MaxQueries ? 8 : n <= kNnCoopMaxQueries ? 4 : 1;
  if (lpq == 8)
    hipLaunchKernelGGL(k_three_nn_grid_levels<8>, dim3(dcl_div_up(n, 32), 4), dim3(256), 0, (hipStream_t)stream, n,
                       reinterpret_cast<const float4 *>(points_b4), L, nbatch, off, dist2, idx, mode == 2 ? 1 : 0);
  else if (lpq == 4)
    hipLaunchKernelGGL(k_three_nn_grid_levels<4>, dim3(dcl_div_up(n, 64), 4), dim3(256), 0, (hipStream_t)stream, n,
                       reinterpret_cast<const float4 *>(points_b4), L, nbatch, off, dist2, idx, mode == 2 ? 1 : 0);
  else
    hipLaunchKernelGGL(k_three_nn_grid_levels<1>, dim3(dcl_div_up(n, 256), 4), dim3(256), 0, (hipStream_t)stream, n,
                       reinterpret_cast<const float4 *>(points_b4), L, nbatch, off, dist2, idx, mode == 2 ? 1 : 0);
  DCL_LAUNCH_CHECK();
  return 0;
}

// search + interpolation of all levels in one launch (k_readout_levels); false = the caller takes the two launches
bool dcl_internal_readout_one_launch_ok(int n) { return g_nn_grid == 1 && n <= kNnCoop8MaxQueries; }
int dcl_internal_readout_one_launch(int n, const float *points_b4, const DclReadoutLevels &L, int nbatch, float off, float *dist2,
                                    int32_t *idx, float *out, int ld, dclStream_t stream) {
  DCL_CHECK_ARG(n > 0 && points_b4 && dist2 && idx && out && nbatch > 0 && dcl_internal_readout_one_launch_ok(n));
  hipLaunchKernelGGL(k_readout_levels, dim3(dcl_div_up(n, kInterpPts)), dim3(1024), 0, (hipStream_t)stream, n,
                     reinterpret_cast<const float4 *>(points_b4), L, nbatch, off, dist2, idx, 0, out, ld);
  DCL_LAUNCH_CHECK();
  return 0;
}

int dcl_internal_readout_interpolate(int n, const DclReadoutLevels &L, const int32_t *idx, const float *dist2, float *out,
                                     int ld, dclStream_t stream) {
  DCL_CHECK_ARG(n > 0 && idx && dist2 && out);
  const long long quads = (long long)n * ((L.c[0] + L.c[1] + L.c[2] + L.c[3]) / 4);
  (void)quads;
  hipLaunchKernelGGL(k_three_interpolate_levels, dim3(dcl_grid_1d(n, kInterpPts, 256 * 32)), dim3(256), 0,
                     (hipStream_t)stream, n, L, idx, dist2, out, ld);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_three_nn_sp(int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx,
                            const int32_t *known_seg, int nbatch, dclStream_t stream) {
  return dcl_three_nn_sp_strided(n, m, unknown, known, dist2, idx, known_seg, nbatch, 1, stream);
}

DCL_API int dcl_voxel_centres(const int32_t *indices, const int32_t *n_dev, int n_host, float ve, float off,
                              float *centres, dclStream_t stream) {
  DCL_CHECK_ARG(indices && centres && n_host >= 0);
  if (n_host == 0) return 0;
  hipLaunchKernelGGL(k_voxel_centres, dim3(dcl_grid_1d(n_host, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const int4 *>(indices), n_dev, n_host, ve, off,
                     reinterpret_cast<float4 *>(centres));
  DCL_LAUNCH_CHECK();
  return 0;
}

template <bool FROM_DIST2>
static int launch_interp_sp(int c, int n, const float *points, const int32_t *idx, const float *w, float *out,
                            int out_stride, hipStream_t s) {
  if (c % 4 == 0 && out_stride % 4 == 0)
    hipLaunchKernelGGL((k_three_interpolate_sp<FROM_DIST2>), dim3(dcl_grid_1d((long long)n * (c / 4), 256)),
                       dim3(256), 0, s, c, n, points, idx, w, out, out_stride);
  else
    hipLaunchKernelGGL((k_three_interpolate_sp_scalar<FROM_DIST2>), dim3(dcl_grid_1d((long long)n * c, 256)),
                       dim3(256), 0, s, c, n, points, idx, w, out, out_stride);
  return 0;
}

DCL_API int dcl_three_interpolate_sp(int c, int m, int n, const float *points, const int32_t *idx,
                                     const float *weight, float *out, int out_stride, dclStream_t stream) {
  DCL_CHECK_ARG(c > 0 && m >= 0 && n >= 0 && out_stride >= c);
  if (n == 0) return 0;
  DCL_CHECK_ARG(points && idx && weight && out);
  launch_interp_sp<false>(c, n, points, idx, weight, out, out_stride, (hipStream_t)stream);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_three_interpolate_dist2_sp(int c, int m, int n, const float *points, const int32_t *idx,
                                           const float *dist2, float *out, int out_stride, dclStream_t stream) {
  DCL_CHECK_ARG(c > 0 && m >= 0 && n >= 0 && out_stride >= c);
  if (n == 0) return 0;
  DCL_CHECK_ARG(points && idx && dist2 && out);
  launch_interp_sp<true>(c, n, points, idx, dist2, out, out_stride, (hipStream_t)stream);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx,
                         dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && n >= 0 && m >= 0);
  if (b == 0 || n == 0) return 0;
  DCL_CHECK_ARG(unknown && dist2 && idx && (m == 0 || known) && b <= 65535);
  if (m >= 64 && m <= kNNBucketedMaxKnown && g_nn_batched_mode == 0) {        // bucketed exact search (same results as the scan)
    const int qpt = nn_bucketed_qpt(b, n, m);
    const size_t lds = nn_bucketed_lds(m, qpt);
#define NNB_LAUNCH(KB_, Q_)                                                                                                       \
    do {                                                                                                                        \
      (void)hipFuncSetAttribute((const void *)k_nn_bucketed<KB_, Q_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);      \
      hipLaunchKernelGGL((k_nn_bucketed<KB_, Q_>), dim3(dcl_div_up(n, kNNThreads * Q_), b), dim3(kNNThreads), lds, (hipStream_t)stream, \
                         n, m, unknown, known, dist2, idx);                                                                     \
    } while (0)
    if (qpt == 4) NNB_LAUNCH(3, 4); else if (qpt == 3) NNB_LAUNCH(3, 3); else if (qpt == 2) NNB_LAUNCH(3, 2); else NNB_LAUNCH(3, 1);
    DCL_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_three_nn, dim3(dcl_div_up(n, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, unknown, known,
                     dist2, idx);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_knn(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2, int32_t *idx,
                    dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && n >= 0 && m >= 0 && k >= 1 && k <= 200);
  if (b == 0 || n == 0) return 0;
  DCL_CHECK_ARG(unknown && dist2 && idx && (m == 0 || known) && b <= 65535);
  if (k == 1 && m >= 64 && m <= kNNBucketedMaxKnown && g_nn_batched_mode == 0) {
    const int qpt = nn_bucketed_qpt(b, n, m);
    const size_t lds = nn_bucketed_lds(m, qpt);
    if (qpt == 4) NNB_LAUNCH(1, 4); else if (qpt == 3) NNB_LAUNCH(1, 3); else if (qpt == 2) NNB_LAUNCH(1, 2); else NNB_LAUNCH(1, 1);
    DCL_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = (size_t)k * kKnnThreads * 8;
  hipLaunchKernelGGL(k_knn, dim3(dcl_div_up(n, kKnnThreads), b), dim3(kKnnThreads), lds, (hipStream_t)stream, n, m, k,
                     unknown, known, dist2, idx);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                                  const float *weight, float *out, dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && c >= 0 && n >= 0 && m >= 0);
  if (b == 0 || c == 0 || n == 0) return 0;
  DCL_CHECK_ARG(points && idx && weight && out && b <= 65535 && c <= 65535);
  hipLaunchKernelGGL(k_three_interpolate, dim3(dcl_grid_1d(n, 256, 64), c, b), dim3(256), 0, (hipStream_t)stream, c, m,
                     n, points, idx, weight, out);
  DCL_LAUNCH_CHECK();
  return 0;
}

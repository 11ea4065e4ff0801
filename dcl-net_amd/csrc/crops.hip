// crops.hip -- the crop builder that runs in front of DCL_Net.forward, on the device.
//
// The reference builds every object crop of an image on the CPU inside the DataLoader workers
// (YCBV/dataloader_test_YCBV.py:124-183; LineMOD's loader has the same arithmetic): mask the detection box, back-project
// the depth pixels, subtract the centroid, drop points outside the voxel grid, sample N of them, emit the (N,7) feature
// rows and the integer voxel coordinates that voxelize_idx consumes.  Here the data-dependent part runs on the device in
// the reference's ORDER (ascending flat pixel index inside the box) and with the reference's float32 / float64 arithmetic
// step by step, so that the results are bit-identical:
//
//   dcl_crop_points   box mask -> ordered compaction -> back-projection -> sequential float32 centroid (numpy's
//                     mean(axis=0) is a row-order running sum, verified in tests) -> box filter -> ordered compaction
//   [host: np.random.choice(count, N) -- the sampling indices stay the caller's, it owns the RNG stream]
//   dcl_crop_sample   gather the sampled points, write feats rows [1,r,g,b,x,y,z] and (batch,x,y,z) voxel coordinates
//
// dcl_crop_points is three launches (round 5; until then ONE workgroup per instance walked its box chunk by chunk, 159 us for
// the six instances of a frame on a 256-CU chip):
//   k_crop_mask      one workgroup per 4096-pixel chunk of a box: mask, count, ordered compaction through a decoupled
//                    look-back over the instance's earlier chunks (lower workgroup ids: running or done), back-projection
//   k_crop_centroid  one workgroup per instance: the only part that is sequential by contract -- the row-order float32 sum
//                    (three lanes, one coordinate each, LDS-fed) -- then, all threads, the in-grid count per 4096-row chunk and
//                    the chunks' exclusive output offsets
//   k_crop_keep      one workgroup per 4096-row chunk: filter, ordered compaction at the chunk's offset, centring
#include "common.h"

namespace {

constexpr int kCropThreads = 1024;
constexpr int kCropChunk = 4096;                // pixels / rows per workgroup step: 4 consecutive per thread

// exclusive prefix of one small count per thread over the workgroup (order = thread id); returns the block total
__device__ __forceinline__ int block_excl_scan(int v, int *s_wave /* [16] */, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  __syncthreads();                             // s_wave may still be read from the previous call
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < kCropThreads / 64; ++w) {
    const int t = s_wave[w];
    if (w < wave) base += t;
    total += t;
  }
  return base + incl - v;
}

struct CropCam { float cx, cy, fx, fy, scale, post_div; };

// scratch of an instance (ints): [0] n (masked pixels), [1] rows kept, [2] filter applied, [3] reserved,
// [4 .. 4 + nch): status of the mask chunks (count + 1; 0 = not yet published), [4 + nch .. 4 + 2 nch): output offset of the row chunks
__host__ __device__ inline int crop_ws_ints(int cap) { return 4 + 2 * ((cap + kCropChunk - 1) / kCropChunk); }

// ---- 1. masked pixels of the box in flat order (dataloader_test_YCBV.py:128-133), back-projection (:147-154)
__global__ __launch_bounds__(kCropThreads) void k_crop_mask(
    const uint16_t *__restrict__ depth, const int32_t *__restrict__ label, const uint8_t *__restrict__ rgb, int H, int W,
    int rgb_channels, const int32_t *__restrict__ boxes /* (n,4) rmin,rmax,cmin,cmax */, const int32_t *__restrict__ obj_ids,
    CropCam cam, double mean_r, double mean_g, double mean_b, int cap, int nch, float *__restrict__ raw_xyz,
    float *__restrict__ raw_rgb, int32_t *__restrict__ ws) {
  __shared__ int s_wave[kCropThreads / 64];
  __shared__ int s_base;
  const int inst = blockIdx.x / nch, chunk = blockIdx.x - inst * nch, t = threadIdx.x;
  const int rmin = boxes[inst * 4], rmax = boxes[inst * 4 + 1], cmin = boxes[inst * 4 + 2], cmax = boxes[inst * 4 + 3];
  const int bh = max(rmax - rmin, 0), bw = max(cmax - cmin, 0);
  const int area = min(bh * bw, cap);
  const int obj = obj_ids[inst];
  int32_t *status = ws + (size_t)inst * crop_ws_ints(cap) + 4;
  float *rx = raw_xyz + (size_t)inst * cap * 3, *rc = raw_rgb + (size_t)inst * cap * 3;
  const int base = chunk * kCropChunk;
  bool keep[4];
  int cnt = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int f = base + 4 * t + u;
    keep[u] = false;
    if (f < area) {
      const int r = rmin + f / bw, c = cmin + f % bw;
      if (r >= 0 && r < H && c >= 0 && c < W)
        keep[u] = label[(size_t)r * W + c] == obj && depth[(size_t)r * W + c] != 0;
    }
    cnt += keep[u];
  }
  int total;
  const int mine = block_excl_scan(cnt, s_wave, total);
  // publish this chunk's count, then add up the earlier chunks' (decoupled look-back: they belong to lower workgroup ids)
  if (t == 0) __hip_atomic_store(status + chunk, total + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (t < 64) {
    int sum = 0;
    for (int c0 = 0; c0 < chunk; c0 += 64) {
      const int c = c0 + t;
      int v = 1;
      if (c < chunk)
        while ((v = __hip_atomic_load(status + c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(1);
      sum += v - 1;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (t == 0) s_base = sum;
  }
  __syncthreads();
  int o = s_base + mine;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (!keep[u]) continue;
    const int f = base + 4 * t + u;
    const int r = rmin + f / bw, c = cmin + f % bw;
    const size_t pix = (size_t)r * W + c;
    const float pt2 = (float)depth[pix] / cam.scale;
    const float pt0 = ((float)c - cam.cx) * pt2 / cam.fx;
    const float pt1 = ((float)r - cam.cy) * pt2 / cam.fy;
    // LineMOD's loader converts millimetres afterwards, `cloud = cloud / 1000.0` (LM/dataloader_test_LM.py:160); x / 1.0f
    // is the identity for the YCB-V loader
    rx[(size_t)o * 3] = pt0 / cam.post_div; rx[(size_t)o * 3 + 1] = pt1 / cam.post_div; rx[(size_t)o * 3 + 2] = pt2 / cam.post_div;
    // img/255.0 in float32, minus the float64 mean, rounded to float32 when the FloatTensor is made (:143-145,168)
    const uint8_t *px = rgb + pix * rgb_channels;
    rc[(size_t)o * 3] = (float)((double)((float)px[0] / 255.0f) - mean_r);
    rc[(size_t)o * 3 + 1] = (float)((double)((float)px[1] / 255.0f) - mean_g);
    rc[(size_t)o * 3 + 2] = (float)((double)((float)px[2] / 255.0f) - mean_b);
    ++o;
  }
}

// ---- 2. centroid = np.mean(cloud, axis=0): running float32 sum in row order, one division (:156); in-grid counts (:160-163)
__global__ __launch_bounds__(kCropThreads) void k_crop_centroid(int cap, int nch, float hx, float hy, float hz, int min_valid,
                                                                int always_filter, const float *__restrict__ raw_xyz,
                                                                float *__restrict__ centroid, int32_t *__restrict__ counts,
                                                                int32_t *__restrict__ ws) {
  __shared__ int s_wave[kCropThreads / 64];
  __shared__ float s_stage[2][kCropChunk * 3];
  __shared__ float s_cen[3];
  __shared__ int s_cnt[1024];                    // in-grid rows of every 4096-row chunk (cap <= 4 Mi pixels)
  const int inst = blockIdx.x, t = threadIdx.x;
  int32_t *w = ws + (size_t)inst * crop_ws_ints(cap);
  const float *rx = raw_xyz + (size_t)inst * cap * 3;
  int n = 0;
  for (int c = t; c < nch; c += kCropThreads) n += w[4 + c] - 1;
  {
    int total;
    (void)block_excl_scan(n, s_wave, total);
    n = total;
  }
  if (n == 0) {                                 // empty mask: the reference skips the instance (:135-143)
    if (t < 3) { counts[inst * 3 + t] = 0; centroid[inst * 3 + t] = 0.0f; }
    if (t == 0) { w[0] = 0; w[1] = 0; w[2] = 0; }
    return;
  }
  // the sum is sequential by contract (row order, one rounding per row): a chain of n dependent adds per coordinate.  Everything
  // else is taken off the chain: the rest of the workgroup stages chunk j + 1 into LDS (one array per coordinate) while three
  // lanes add chunk j -- 16-byte LDS reads, the next 64 rows in registers before the current 64 are added
  float acc = 0.0f;
  const int nchunk = (n + kCropChunk - 1) / kCropChunk;
  auto stage = [&](int buf, int first_row, int rows, int j0, int step) {
    float *dst = &s_stage[buf][0];
    for (int j = j0; j < rows * 3; j += step) {
      const int row = j / 3, c = j - 3 * row;
      dst[c * kCropChunk + row] = rx[(size_t)first_row * 3 + j];
    }
  };
  stage(0, 0, min(kCropChunk, n), t, kCropThreads);
  __syncthreads();
  for (int ch = 0; ch < nchunk; ++ch) {
    const int base = ch * kCropChunk, rows = min(kCropChunk, n - base);
    if (t >= 64) {                              // the other waves: next chunk -> the other buffer
      const int nb = base + kCropChunk;
      if (nb < n) stage((ch + 1) & 1, nb, min(kCropChunk, n - nb), t - 64, kCropThreads - 64);
    } else if (t < 3) {
      const float *col = &s_stage[ch & 1][t * kCropChunk];
      const float4 *col4 = reinterpret_cast<const float4 *>(col);
      const int nb64 = rows >> 5;                          // blocks of 32 rows
      // two register sets in turn (A: even blocks, B: odd ones; 2 x 32 registers -- the 1024-thread workgroup has 128): the loads of one set fly under the adds of the other
      float4 va[8], vb[8];
      auto load16 = [&](float4 (&v)[8], int blk) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = col4[blk * 8 + q];
      };
      auto add16 = [&](const float4 (&v)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { acc = acc + v[q].x; acc = acc + v[q].y; acc = acc + v[q].z; acc = acc + v[q].w; }
      };
      if (nb64 > 0) load16(va, 0);
      int b64 = 0;
      for (; b64 + 2 <= nb64; b64 += 2) {
        load16(vb, b64 + 1);
        add16(va);
        if (b64 + 2 < nb64) load16(va, b64 + 2);
        add16(vb);
      }
      if (b64 < nb64) add16(va);
      for (int i = nb64 << 5; i < rows; ++i) acc = acc + col[i];
    }
    __syncthreads();
  }
  if (t < 3) { const float cen = acc / (float)n; s_cen[t] = cen; centroid[inst * 3 + t] = cen; }
  __syncthreads();
  const float cx = s_cen[0], cy = s_cen[1], cz = s_cen[2];
  // points inside the voxel grid, per 4096-row chunk
  int valid = 0;
  for (int ch = 0; ch < nchunk; ++ch) {
    int v = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = ch * kCropChunk + 4 * t + u;
      if (i < n) {
        const float x = rx[(size_t)i * 3] - cx, y = rx[(size_t)i * 3 + 1] - cy, z = rx[(size_t)i * 3 + 2] - cz;
        v += fabsf(x) < hx && fabsf(y) < hy && fabsf(z) < hz;
      }
    }
    int total;
    (void)block_excl_scan(v, s_wave, total);
    if (t == 0) s_cnt[ch] = total;
    valid += total;
  }
  __syncthreads();
  const bool filter = valid > min_valid || always_filter;   // `if valid_num > 32` (:163); LM eval mode filters always (:197)
  if (t == 0) {
    int off = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
      w[4 + nch + ch] = off;
      off += filter ? s_cnt[ch] : min(kCropChunk, n - ch * kCropChunk);
    }
    w[0] = n; w[1] = off; w[2] = filter ? 1 : 0;
    counts[inst * 3] = n; counts[inst * 3 + 1] = valid; counts[inst * 3 + 2] = off;
  }
}

// ---- 3. keep the points inside the grid (in order), centred (:160-165)
__global__ __launch_bounds__(kCropThreads) void k_crop_keep(int cap, int nch, float hx, float hy, float hz,
                                                            const float *__restrict__ raw_xyz, const float *__restrict__ raw_rgb,
                                                            const float *__restrict__ centroid, float *__restrict__ out_xyz,
                                                            float *__restrict__ out_rgb, const int32_t *__restrict__ ws) {
  __shared__ int s_wave[kCropThreads / 64];
  const int inst = blockIdx.x / nch, chunk = blockIdx.x - inst * nch, t = threadIdx.x;
  const int32_t *w = ws + (size_t)inst * crop_ws_ints(cap);
  const int n = w[0];
  if (chunk * kCropChunk >= n) return;
  const bool filter = w[2] != 0;
  const float cx = centroid[inst * 3], cy = centroid[inst * 3 + 1], cz = centroid[inst * 3 + 2];
  const float *rx = raw_xyz + (size_t)inst * cap * 3, *rc = raw_rgb + (size_t)inst * cap * 3;
  float *ox = out_xyz + (size_t)inst * cap * 3, *oc = out_rgb + (size_t)inst * cap * 3;
  bool keep[4];
  float p[4][3];
  int cnt = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = chunk * kCropChunk + 4 * t + u;
    keep[u] = false;
    if (i < n) {
      p[u][0] = rx[(size_t)i * 3] - cx; p[u][1] = rx[(size_t)i * 3 + 1] - cy; p[u][2] = rx[(size_t)i * 3 + 2] - cz;
      keep[u] = !filter || (fabsf(p[u][0]) < hx && fabsf(p[u][1]) < hy && fabsf(p[u][2]) < hz);
    }
    cnt += keep[u];
  }
  int total;
  int o = w[4 + nch + chunk] + block_excl_scan(cnt, s_wave, total);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (!keep[u]) continue;
    const int i = chunk * kCropChunk + 4 * t + u;
#pragma unroll
    for (int j = 0; j < 3; ++j) { ox[(size_t)o * 3 + j] = p[u][j]; oc[(size_t)o * 3 + j] = rc[(size_t)i * 3 + j]; }
    ++o;
  }
}

// feats row [1, r, g, b, x, y, z] and voxel coordinate row [batch, ix, iy, iz] of every sampled point (:170-176,186-190):
//   voxel = trunc((xyz + half_extent0) / unit) in float32, clamped to [0, limit-1] first when the crop had <= 32 points
//   inside the grid.  sample_idx == nullptr: identity (template clouds, :179-182).
__global__ void k_crop_sample(int n_inst, int npoint, int cap, const float *__restrict__ xyz, const float *__restrict__ rgb,
                              const int64_t *__restrict__ sample_idx, const int32_t *__restrict__ counts, int min_valid,
                              float half0, float ux, float uy, float uz, float limit, float *__restrict__ feats,
                              int64_t *__restrict__ coords) {
  const long long total = (long long)n_inst * npoint;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int inst = (int)(e / npoint);
    const long long i = sample_idx ? sample_idx[e] : (e - (long long)inst * npoint);
    const float *p = xyz + ((size_t)inst * cap + i) * 3, *c = rgb + ((size_t)inst * cap + i) * 3;
    const float x = p[0], y = p[1], z = p[2];
    float *f = feats + e * 7;
    f[0] = 1.0f; f[1] = c[0]; f[2] = c[1]; f[3] = c[2]; f[4] = x; f[5] = y; f[6] = z;
    float vx = (x + half0) / ux, vy = (y + half0) / uy, vz = (z + half0) / uz;
    if (counts && counts[inst * 3 + 1] <= min_valid) {
      vx = fminf(fmaxf(vx, 0.0f), limit - 1.0f); vy = fminf(fmaxf(vy, 0.0f), limit - 1.0f);
      vz = fminf(fmaxf(vz, 0.0f), limit - 1.0f);
    }
    int64_t *o = coords + e * 4;
    o[0] = inst; o[1] = (int64_t)vx; o[2] = (int64_t)vy; o[3] = (int64_t)vz;
  }
}

}  // namespace

DCL_API int dcl_crop_points_ws_ints(int n_inst, int cap, int64_t *ints_host) {
  DCL_CHECK_ARG(n_inst >= 0 && cap > 0 && ints_host);
  *ints_host = (int64_t)n_inst * crop_ws_ints(cap);
  return 0;
}

DCL_API int dcl_crop_points(const uint16_t *depth, const int32_t *label, const uint8_t *rgb, int H, int W, int rgb_channels,
                            int n_inst, const int32_t *boxes, const int32_t *obj_ids, const float *cam_host /*6*/,
                            const double *rgb_mean_host /*3*/, const float *half_extent_host /*3*/, int min_valid,
                            int always_filter, int cap,
                            float *raw_xyz, float *raw_rgb, float *out_xyz, float *out_rgb, float *centroid,
                            int32_t *counts, int32_t *ws, dclStream_t stream) {
  DCL_CHECK_ARG(n_inst >= 0 && H > 0 && W > 0 && rgb_channels >= 3 && cap > 0 && cap <= 1024 * kCropChunk);
  if (n_inst == 0) return 0;
  DCL_CHECK_ARG(depth && label && rgb && boxes && obj_ids && cam_host && rgb_mean_host && half_extent_host && raw_xyz &&
                raw_rgb && out_xyz && out_rgb && centroid && counts && ws);
  const CropCam cam = {cam_host[0], cam_host[1], cam_host[2], cam_host[3], cam_host[4], cam_host[5]};
  DCL_CHECK_ARG(cam.scale != 0.0f && cam.post_div != 0.0f);
  const int nch = dcl_div_up(cap, kCropChunk);
  DCL_CHECK_ARG((long long)n_inst * nch < (1ll << 31));
  hipStream_t s = (hipStream_t)stream;
  dcl_internal_zero_words(ws, (long long)n_inst * crop_ws_ints(cap), s);       // chunk statuses: 0 = not yet published
  hipLaunchKernelGGL(k_crop_mask, dim3(n_inst * nch), dim3(kCropThreads), 0, s, depth, label, rgb, H, W, rgb_channels, boxes,
                     obj_ids, cam, rgb_mean_host[0], rgb_mean_host[1], rgb_mean_host[2], cap, nch, raw_xyz, raw_rgb, ws);
  hipLaunchKernelGGL(k_crop_centroid, dim3(n_inst), dim3(kCropThreads), 0, s, cap, nch, half_extent_host[0], half_extent_host[1],
                     half_extent_host[2], min_valid, always_filter, raw_xyz, centroid, counts, ws);
  hipLaunchKernelGGL(k_crop_keep, dim3(n_inst * nch), dim3(kCropThreads), 0, s, cap, nch, half_extent_host[0], half_extent_host[1],
                     half_extent_host[2], raw_xyz, raw_rgb, centroid, out_xyz, out_rgb, ws);
  DCL_LAUNCH_CHECK();
  return 0;
}

DCL_API int dcl_crop_sample(int n_inst, int npoint, int cap, const float *xyz, const float *rgb, const int64_t *sample_idx,
                            const int32_t *counts, int min_valid, float half_extent0, const float *unit_host /*3*/,
                            int voxel_limit, float *feats, int64_t *coords, dclStream_t stream) {
  DCL_CHECK_ARG(n_inst >= 0 && npoint >= 0 && cap > 0 && voxel_limit > 0);
  if (n_inst == 0 || npoint == 0) return 0;
  DCL_CHECK_ARG(xyz && rgb && unit_host && feats && coords);
  hipLaunchKernelGGL(k_crop_sample, dim3(dcl_grid_1d((long long)n_inst * npoint, 256)), dim3(256), 0, (hipStream_t)stream,
                     n_inst, npoint, cap, xyz, rgb, sample_idx, counts, min_valid, half_extent0, unit_host[0],
                     unit_host[1], unit_host[2], (float)voxel_limit, feats, coords);
  DCL_LAUNCH_CHECK();
  return 0;
}

// voxelize.hip -- point -> voxel feature pooling (replaces voxelize_fp_cuda_,
// libs/pointgroup_ops/src/voxelize/voxelize.cu:9-31).
//
// One thread per (voxel row, plane).  The reference accumulates with atomicAdd from one
// thread per plane, i.e. a serial fp32 sum in rule order: ((0 + m*x1) + m*x2) + ...;
// this kernel keeps that order in a register and stores once (no atomics, no pre-zeroing).
// HBM-bound: 4*(V*(1+maxActive) + N*C + V*C) bytes; rows are consecutive so the rule reads
// of neighbouring threads of a row hit the same cache line.
#include "common.h"

__global__ void voxelize_fp_kernel(const float *__restrict__ feats, const int32_t *__restrict__ rules,
                                   float *__restrict__ out, int n_rows, int max_active, int n_planes,
                                   int average) {
  const long long total = (long long)n_rows * n_planes;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(t / n_planes);
    const int plane = (int)(t - (long long)row * n_planes);
    const int32_t *r = rules + (size_t)row * (max_active + 1);
    const int n_active = r[0];
    const float mult = (average && n_active > 0) ? 1.0f / (float)n_active : 1.0f;
    float acc = 0.0f;
    for (int i = 1; i <= n_active; ++i) acc = acc + mult * feats[(size_t)r[i] * n_planes + plane];
    out[t] = acc;
  }
}

DCL_API int dcl_voxelize_fp(const float *feats, const int32_t *rules, float *out, int n_rows,
                            int max_active, int n_planes, int average, dclStream_t stream) {
  DCL_CHECK_ARG(n_rows >= 0 && max_active >= 0 && n_planes > 0);
  if (n_rows == 0) return 0;
  DCL_CHECK_ARG(feats && rules && out);
  const long long total = (long long)n_rows * n_planes;
  hipLaunchKernelGGL(voxelize_fp_kernel, dim3(dcl_grid_1d(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, feats, rules, out, n_rows, max_active, n_planes, average);
  DCL_LAUNCH_CHECK();
  return 0;
}

// ---- input staging / result hand-over of the whole-forward hipGraph (models/DCL_Net.py::forward_graphed): the loader's
// tensors go into the graph's static buffers and the graph's outputs into the caller's tensors with ONE launch each,
// instead of one copy / fill kernel per tensor (a one-crop forward is ~0.8 ms: ten extra launches are 5 % of it).
// job j: dst (rows_dst x cols_dst, dense) <- src (rows_src x cols_src at src_pitch), zero padded; src == NULL: fill.
namespace {
struct PadCopyJobs {
  DclPadCopyJob job[DCL_PAD_COPY_MAX_JOBS];
};
__global__ void k_pad_copy_many(const PadCopyJobs jobs, unsigned vec_mask) {
  const DclPadCopyJob &q = jobs.job[blockIdx.y];
  if ((vec_mask >> blockIdx.y) & 1u) {            // every width / pitch a multiple of 4 elements, 16-B aligned: 16-B moves
    const int cd = q.cols_dst >> 2, cs = q.cols_src >> 2, sp = q.src_pitch >> 2, dp = q.dst_pitch >> 2;
    const long long total = (long long)q.rows_dst * cd;
    int4 *dst = reinterpret_cast<int4 *>(q.dst);
    const int4 *src = reinterpret_cast<const int4 *>(q.src);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
      const int r = (int)(t / cd), c = (int)(t - (long long)r * cd);
      int4 v = make_int4(0, 0, 0, 0);
      if (r < q.rows_src && c < cs) v = src[(size_t)r * sp + c];
      dst[dp > 0 ? (size_t)r * dp + c : (size_t)t] = v;
    }
    return;
  }
  const long long total = (long long)q.rows_dst * q.cols_dst;
  int32_t *dst = reinterpret_cast<int32_t *>(q.dst);
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(t / q.cols_dst), c = (int)(t - (long long)r * q.cols_dst);
    int32_t v = q.fill_value;
    if (q.src && r < q.rows_src && c < q.cols_src) {
      const size_t i = (size_t)r * q.src_pitch + c;
      v = q.src_is_i64 ? (int32_t) reinterpret_cast<const int64_t *>(q.src)[i] : reinterpret_cast<const int32_t *>(q.src)[i];
    } else if (q.src) {
      v = 0;
    }
    dst[q.dst_pitch > 0 ? (size_t)r * q.dst_pitch + c : (size_t)t] = v;
  }
}
}  // namespace

DCL_API int dcl_pad_copy_many(const DclPadCopyJob *jobs_host, int njobs, dclStream_t stream) {
  DCL_CHECK_ARG(jobs_host && njobs >= 1 && njobs <= DCL_PAD_COPY_MAX_JOBS);
  PadCopyJobs pack;
  long long most = 1;
  unsigned vec_mask = 0;
  for (int j = 0; j < njobs; ++j) {
    const DclPadCopyJob &q = jobs_host[j];
    if (q.src && !q.src_is_i64 && ((q.cols_dst | q.cols_src | q.src_pitch | q.dst_pitch) & 3) == 0 &&
        ((((uintptr_t)q.dst) | ((uintptr_t)q.src)) & 15) == 0)
      vec_mask |= 1u << j;
    DCL_CHECK_ARG(q.dst && q.rows_dst >= 0 && q.cols_dst >= 0 && q.rows_src >= 0 && q.cols_src >= 0 &&
                  q.src_pitch >= q.cols_src && (q.dst_pitch == 0 || q.dst_pitch >= q.cols_dst));
    pack.job[j] = q;
    const long long t = (long long)q.rows_dst * q.cols_dst;
    if (t > most) most = t;
  }
  hipLaunchKernelGGL(k_pad_copy_many, dim3(dcl_grid_1d(most / 4 + 1, 256, 4096), njobs), dim3(256), 0, (hipStream_t)stream,
                     pack, vec_mask);
  DCL_LAUNCH_CHECK();
  return 0;
}

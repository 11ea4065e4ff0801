// metric.hip -- fused ADD-S / ADD distances (the step right after the forward in the eval harness).
//
// LineMOD variant (tools/test_LM.py:123-135): non-symmetric objects use ADD = mean_i |pred_i - gt_i| (corresponding
// points), symmetric ones ADD-S; `sym` selects per object inside the same launch (workgroup-uniform branch).
//
// tools/test_YCBV_stage1.py:186-189 poses the class cloud (2620 points) by the predicted and by the ground-truth pose and
// takes mean_i min_j |pred_i - gt_j| through a materialised (b, P, P, 3) difference tensor: 82 MB per object.
// Here one workgroup per (object, 256-point slice of i): the gt-posed cloud is staged once per workgroup in LDS
// (P*12 B = 31 KB), every lane keeps one pred point in registers and scans the LDS copy (broadcast reads), the
// per-point minima are reduced in the workgroup and accumulated per object in fixed slice order by a second tiny
// kernel (deterministic).  sqrt is applied to the minimum squared distance (monotone, same argmin as min of norms).
#include "common.h"
#include <math.h>

namespace {

__device__ __forceinline__ void pose_point(const float *R, const float *t, float x, float y, float z, float &ox,
                                           float &oy, float &oz) {
  // row-vector form of bmm(cld, R^T) + t: out[c] = x*R[c][0] + y*R[c][1] + z*R[c][2] + t[c]
  ox = __fmaf_rn(z, R[2], __fmaf_rn(y, R[1], x * R[0])) + t[0];
  oy = __fmaf_rn(z, R[5], __fmaf_rn(y, R[4], x * R[3])) + t[1];
  oz = __fmaf_rn(z, R[8], __fmaf_rn(y, R[7], x * R[6])) + t[2];
}

__global__ __launch_bounds__(256) void k_adds_partial(int P, const float *__restrict__ cld, const int32_t *__restrict__ cls,
                                                      const float *__restrict__ Rp, const float *__restrict__ tp,
                                                      const float *__restrict__ Rg, const float *__restrict__ tg,
                                                      float *__restrict__ partial, int nslices,
                                                      const int32_t *__restrict__ sym, int all_mode) {
  extern __shared__ float gt[];                 // [P][3] posed by the ground truth
  __shared__ float red[4];
  const int obj = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
  const float *C = cld + (size_t)(cls ? cls[obj] : obj) * P * 3;
  const float *R1 = Rp + obj * 9, *t1 = tp + obj * 3, *R2 = Rg + obj * 9, *t2 = tg + obj * 3;
  for (int j = tid; j < P; j += 256) {
    float x, y, z;
    pose_point(R2, t2, C[j * 3], C[j * 3 + 1], C[j * 3 + 2], x, y, z);
    gt[j * 3] = x; gt[j * 3 + 1] = y; gt[j * 3 + 2] = z;
  }
  __syncthreads();
  const int i = slice * 256 + tid;
  float best = 0.0f;
  if (i < P) {
    float px, py, pz;
    pose_point(R1, t1, C[i * 3], C[i * 3 + 1], C[i * 3 + 2], px, py, pz);
    const bool nearest = sym ? sym[obj] != 0 : all_mode != 0;     // ADD-S: nearest gt point; ADD: the corresponding one
    float m = INFINITY;
    if (nearest) {
      for (int j = 0; j < P; ++j) {
        const float dx = px - gt[j * 3], dy = py - gt[j * 3 + 1], dz = pz - gt[j * 3 + 2];
        m = fminf(m, __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx)));
      }
    } else {
      const float dx = px - gt[i * 3], dy = py - gt[i * 3 + 1], dz = pz - gt[i * 3 + 2];
      m = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx));
    }
    best = sqrtf(m);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) best += __shfl_xor(best, d, 64);
  if ((tid & 63) == 0) red[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) partial[(size_t)obj * nslices + slice] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void k_adds_finish(int b, int P, int nslices, const float *__restrict__ partial, float *__restrict__ out) {
  const int obj = blockIdx.x * blockDim.x + threadIdx.x;
  if (obj >= b) return;
  float s = 0.0f;
  for (int k = 0; k < nslices; ++k) s += partial[(size_t)obj * nslices + k];
  out[obj] = s / (float)P;
}

}  // namespace

static int add_launch(int b, int P, const float *cld, const int32_t *cls, const float *R_pred, const float *t_pred,
                      const float *R_gt, const float *t_gt, float *partial_scratch, float *out, const int32_t *sym,
                      int all_mode, dclStream_t stream);

DCL_API int dcl_add_s(int b, int P, const float *cld, const int32_t *cls, const float *R_pred, const float *t_pred,
                      const float *R_gt, const float *t_gt, float *partial_scratch, float *out, dclStream_t stream) {
  return add_launch(b, P, cld, cls, R_pred, t_pred, R_gt, t_gt, partial_scratch, out, nullptr, 1, stream);
}

DCL_API int dcl_add(int b, int P, const float *cld, const int32_t *cls, const float *R_pred, const float *t_pred,
                    const float *R_gt, const float *t_gt, float *partial_scratch, float *out, dclStream_t stream) {
  return add_launch(b, P, cld, cls, R_pred, t_pred, R_gt, t_gt, partial_scratch, out, nullptr, 0, stream);
}

DCL_API int dcl_add_by_symmetry(int b, int P, const float *cld, const int32_t *cls, const int32_t *sym_flag,
                                const float *R_pred, const float *t_pred, const float *R_gt, const float *t_gt,
                                float *partial_scratch, float *out, dclStream_t stream) {
  DCL_CHECK_ARG(sym_flag || b == 0);
  return add_launch(b, P, cld, cls, R_pred, t_pred, R_gt, t_gt, partial_scratch, out, sym_flag, 0, stream);
}

static int add_launch(int b, int P, const float *cld, const int32_t *cls, const float *R_pred, const float *t_pred,
                      const float *R_gt, const float *t_gt, float *partial_scratch, float *out, const int32_t *sym,
                      int all_mode, dclStream_t stream) {
  DCL_CHECK_ARG(b >= 0 && P > 0 && (size_t)P * 12 <= 150 * 1024);
  if (b == 0) return 0;
  DCL_CHECK_ARG(cld && R_pred && t_pred && R_gt && t_gt && partial_scratch && out && b <= 65535);
  hipStream_t s = (hipStream_t)stream;
  const int nslices = dcl_div_up(P, 256);
  const size_t lds = (size_t)P * 12;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void *)k_adds_partial, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_adds_partial, dim3(nslices, b), dim3(256), lds, s, P, cld, cls, R_pred, t_pred, R_gt, t_gt,
                     partial_scratch, nslices, sym, all_mode);
  hipLaunchKernelGGL(k_adds_finish, dim3(dcl_div_up(b, 64)), dim3(64), 0, s, b, P, nslices, partial_scratch, out);
  DCL_LAUNCH_CHECK();
  return 0;
}

// linear.cpp -- the per-point linear layers (Conv1d k=1 / 1x1x1 Conv3d with folded BN, reference models/Modules.py:58-97,
// 173-201) as plain library GEMMs: y = act(x Wt + bias) through hipBLASLt with the bias / ReLU epilogue, called from the
// C-ABI so that x, Wt and y may be column blocks of wider row-major buffers (leading dimensions are free).  The point of
// the free ldy: the disengage layers write straight into the column blocks of the fuser / confidence inputs that the
// correspondence attention fills from the other side, so no copy kernel sits between them.
//
// Row-major y[M x N] = x[M x K] Wt[K x N] is the column-major product Y'[N x M] = W'[N x K] X'[K x M] of the same
// buffers (ld = row pitch), so no transposes are requested; the bias runs along the rows of Y' (= output channels).
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-version.h>

#include <mutex>
#include <unordered_map>

#include "common.h"

namespace {

struct PlanKey {
  int64_t M, N, K, ldx, ldw, ldy;
  int epilogue;
  int device;                          // plans (heuristics) belong to the device whose handle produced them
  bool operator==(const PlanKey &o) const {
    return M == o.M && N == o.N && K == o.K && ldx == o.ldx && ldw == o.ldw && ldy == o.ldy && epilogue == o.epilogue &&
           device == o.device;
  }
};
struct PlanKeyHash {
  size_t operator()(const PlanKey &k) const {
    size_t h = 1469598103934665603ull;
    for (int64_t v : {k.M, k.N, k.K, k.ldx, k.ldw, k.ldy, (int64_t)k.epilogue, (int64_t)k.device}) h = (h ^ (size_t)v) * 1099511628211ull;
    return h;
  }
};
struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t w = nullptr, x = nullptr, y = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;                       // workspace the heuristic reported for `algo`: 0, or get_plan has refused it
};

std::mutex g_mu;
std::unordered_map<int, hipblasLtHandle_t> g_handles;    // one handle per device (created with that device current)
std::unordered_map<PlanKey, Plan, PlanKeyHash> g_plans;
bool g_version_checked = false;

#define LT_CHECK(call)                                                                                   \
  do {                                                                                                   \
    hipblasStatus_t st__ = (call);                                                                       \
    if (st__ != HIPBLAS_STATUS_SUCCESS) {                                                                \
      dcl_set_error("%s: %s failed with hipblasStatus %d", __func__, #call, (int)st__);                  \
      return DCL_EINVAL;                                                                                 \
    }                                                                                                    \
  } while (0)

// the handle of the CURRENT device (a process may drive several GPUs; the library binds a handle to the device that was
// current when it was created)
int get_handle(int device, hipblasLtHandle_t *out) {
  auto it = g_handles.find(device);
  if (it == g_handles.end()) {
    hipblasLtHandle_t h = nullptr;
    LT_CHECK(hipblasLtCreate(&h));
    if (!g_version_checked) {
      // compiled against the system ROCm's hipBLASLt headers (by-value hipblasLtMatmulAlgo_t, epilogue enums); at run time
      // the soname binds to the copy the process has already mapped (PyTorch's).  Minor versions have proved layout-
      // compatible; another MAJOR version is refused loudly (the version word is major * 100000 or * 10000 + ...)
      int ver = 0;
      if (hipblasLtGetVersion(h, &ver) == HIPBLAS_STATUS_SUCCESS && ver > 0 && ver / 100000 != HIPBLASLT_VERSION_MAJOR &&
          ver / 10000 != HIPBLASLT_VERSION_MAJOR) {
        dcl_set_error("dcl_linear_fwd: hipBLASLt version word %d at run time, headers are %d.%d.%d: rebuild against it", ver,
                      HIPBLASLT_VERSION_MAJOR, HIPBLASLT_VERSION_MINOR, HIPBLASLT_VERSION_PATCH);
        return DCL_EINVAL;
      }
      g_version_checked = true;
    }
    it = g_handles.emplace(device, h).first;
  }
  *out = it->second;
  return 0;
}

int get_plan(const PlanKey &key, hipblasLtHandle_t g_handle, Plan **out) {
  auto it = g_plans.find(key);
  if (it != g_plans.end()) {
    *out = &it->second;
    return 0;
  }
  Plan p;
  LT_CHECK(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  const int32_t opn = HIPBLAS_OP_N;
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opn, sizeof(opn)));
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opn, sizeof(opn)));
  const uint32_t epi = (uint32_t)key.epilogue;
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
  if (key.epilogue & HIPBLASLT_EPILOGUE_BIAS) {
    const int32_t bt = HIP_R_32F;
    LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
  }
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.w, HIP_R_32F, (uint64_t)key.N, (uint64_t)key.K, key.ldw));
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.x, HIP_R_32F, (uint64_t)key.K, (uint64_t)key.M, key.ldx));
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.y, HIP_R_32F, (uint64_t)key.N, (uint64_t)key.M, key.ldy));
  hipblasLtMatmulPreference_t pref = nullptr;
  LT_CHECK(hipblasLtMatmulPreferenceCreate(&pref));
  // The library's fp32 kernels on gfx950 are stream-K hybrids: when a problem's tiles do not divide into whole rounds of the
  // chip (33 crops of 1024 points: M = 33792) the heuristic's first choice hands the leftover tiles around between workgroups
  // through the workspace, SPINNING on flags -- which needs all of its workgroups resident at once.  A forward may run GEMMs of
  // its two branches side by side on two streams: two such kernels each holding part of the chip wait for workgroups the
  // other one keeps out -- the GPU hangs (seen at 25 / 33 crops, whole-forward graph and launch by launch alike).  So the
  // library is never OFFERED a workspace: the heuristic is queried with a maximum of 0 bytes, hipblasLtMatmul is called with
  // (NULL, 0), and a candidate that asks for workspace all the same is skipped.  If nothing is left the call FAILS (loudly,
  // DCL_EINVAL) -- it never falls through to a workspace-exchanging algorithm.
  const uint64_t max_ws = 0;
  LT_CHECK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws)));
  constexpr int kCand = 64;
  hipblasLtMatmulHeuristicResult_t res[kCand];
  int found = 0;
  hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_handle, p.desc, p.w, p.x, p.y, p.y, pref, kCand, res, &found);
  hipblasLtMatmulPreferenceDestroy(pref);
  int pick = -1;
  if (st == HIPBLAS_STATUS_SUCCESS)
    for (int i = 0; i < found; ++i)
      if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize == 0) { pick = i; break; }
  if (pick < 0) {
    hipblasLtMatrixLayoutDestroy(p.w); hipblasLtMatrixLayoutDestroy(p.x); hipblasLtMatrixLayoutDestroy(p.y);
    hipblasLtMatmulDescDestroy(p.desc);
    dcl_set_error("dcl_linear_fwd: hipBLASLt offers no algorithm WITHOUT workspace for M=%lld N=%lld K=%lld (status %d, %d "
                  "candidates): refusing to run a workspace-exchanging stream-K kernel (two of them side by side hang the GPU)",
                  (long long)key.M, (long long)key.N, (long long)key.K, (int)st, found);
    return DCL_EINVAL;
  }
  p.algo = res[pick].algo;
  p.ws = res[pick].workspaceSize;
  auto ins = g_plans.emplace(key, p);
  *out = &ins.first->second;
  return 0;
}

}  // namespace

DCL_API int dcl_linear_fwd(const float *x, int64_t ldx, const float *Wt, int64_t ldw, const float *bias, float *y, int64_t ldy,
                           int M, int N, int K, int relu, void *workspace, int64_t workspace_bytes, dclStream_t stream) {
  DCL_CHECK_ARG(M >= 0 && N > 0 && K > 0 && x && Wt && y && ldx >= K && ldw >= N && ldy >= N && workspace_bytes >= 0);
  DCL_CHECK_ARG(workspace_bytes == 0 || workspace);
  if (M == 0) return 0;
  int epilogue = HIPBLASLT_EPILOGUE_DEFAULT;
  if (bias && relu) epilogue = HIPBLASLT_EPILOGUE_RELU_BIAS;
  else if (bias) epilogue = HIPBLASLT_EPILOGUE_BIAS;
  else if (relu) epilogue = HIPBLASLT_EPILOGUE_RELU;
  std::lock_guard<std::mutex> lock(g_mu);                  // plans and the descriptor's bias pointer are shared state
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) {
    dcl_set_error("dcl_linear_fwd: hipGetDevice failed");
    return DCL_EINVAL;
  }
  hipblasLtHandle_t g_handle = nullptr;
  int rc = get_handle(device, &g_handle);
  if (rc) return rc;
  Plan *p = nullptr;
  rc = get_plan(PlanKey{M, N, K, ldx, ldw, ldy, epilogue, device}, g_handle, &p);
  if (rc) return rc;
  if (bias) LT_CHECK(hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
  const float one = 1.0f, zero = 0.0f;
  LT_CHECK(hipblasLtMatmul(g_handle, p->desc, &one, Wt, p->w, x, p->x, &zero, y, p->y, y, p->y, &p->algo, nullptr,
                           0, (hipStream_t)stream));
  return 0;
}

#ifdef DCL_DIAG
// Diagnostic: bytes of workspace the algorithm dcl_linear_fwd takes for y[M x N] = relu(x[M x K] Wt + bias) asks for: 0 by
// construction (get_plan refuses everything else), -1 when the library offers no such algorithm (dcl_linear_fwd fails then).
DCL_API long long dcl_debug_linear_plan_workspace(int M, int N, int K) {
  std::lock_guard<std::mutex> lock(g_mu);
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return -1;
  hipblasLtHandle_t h = nullptr;
  if (get_handle(device, &h)) return -1;
  Plan *p = nullptr;
  if (get_plan(PlanKey{M, N, K, K, N, N, (int)HIPBLASLT_EPILOGUE_RELU_BIAS, device}, h, &p)) return -1;
  const size_t ws = p->ws;                                 // what the heuristic reported for the algorithm taken
  return (long long)ws;
}
// Diagnostic: the library's first `ncand` heuristic candidates for y[M x N] = relu(x[M x K] Wt[K x N] + bias) (dense pitches)
// with 32 MiB of workspace on offer, each timed on the current device ALONE on the GPU (3 warm-up runs, then 10 runs between
// events): ms_out[i] = mean milliseconds of candidate i, ws_out[i] = the workspace it asks for (dcl_linear_fwd takes the first
// one with ws_out == 0 of the zero-workspace query; candidates with ws_out > 0 are the ones it refuses), *found_out = how many
// there were.  tools/gemm_candidates.py.
DCL_API int dcl_debug_linear_candidates(int M, int N, int K, int ncand, float *ms_out, long long *ws_out, int *found_out) {
  DCL_CHECK_ARG(M > 0 && N > 0 && K > 0 && ncand >= 1 && ncand <= 64 && ms_out && found_out);
  std::lock_guard<std::mutex> lock(g_mu);
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return DCL_EINVAL;
  hipblasLtHandle_t h = nullptr;
  int rc = get_handle(device, &h);
  if (rc) return rc;
  float *x = nullptr, *w = nullptr, *y = nullptr, *bias = nullptr;
  void *ws = nullptr;
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t lw = nullptr, lx = nullptr, ly = nullptr;
  hipblasLtMatmulPreference_t pref = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const size_t ws_bytes = 32u << 20;
  rc = DCL_EINVAL;
  do {                                                     // one exit: everything acquired so far is released below
    if (hipMalloc(&x, sizeof(float) * (size_t)M * K) != hipSuccess || hipMalloc(&w, sizeof(float) * (size_t)K * N) != hipSuccess ||
        hipMalloc(&y, sizeof(float) * (size_t)M * N) != hipSuccess || hipMalloc(&bias, sizeof(float) * N) != hipSuccess ||
        hipMalloc(&ws, ws_bytes) != hipSuccess) {
      dcl_set_error("dcl_debug_linear_candidates: out of memory");
      break;
    }
    (void)hipMemset(x, 0, sizeof(float) * (size_t)M * K); (void)hipMemset(w, 0, sizeof(float) * (size_t)K * N); (void)hipMemset(bias, 0, sizeof(float) * N);
    const int32_t opn = HIPBLAS_OP_N, bt = HIP_R_32F;
    const uint32_t epi = HIPBLASLT_EPILOGUE_RELU_BIAS;
    const uint64_t max_ws = ws_bytes;
    int found = 0;
    hipblasLtMatmulHeuristicResult_t res[64];
    if (hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opn, sizeof(opn)) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opn, sizeof(opn)) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatrixLayoutCreate(&lw, HIP_R_32F, (uint64_t)N, (uint64_t)K, N) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatrixLayoutCreate(&lx, HIP_R_32F, (uint64_t)K, (uint64_t)M, K) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatrixLayoutCreate(&ly, HIP_R_32F, (uint64_t)N, (uint64_t)M, N) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws)) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatmulAlgoGetHeuristic(h, desc, lw, lx, ly, ly, pref, ncand, res, &found) != HIPBLAS_STATUS_SUCCESS) {
      dcl_set_error("dcl_debug_linear_candidates: hipBLASLt set-up failed");
      break;
    }
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
    const float one = 1.0f, zero = 0.0f;
    for (int i = 0; i < found; ++i) {
      ms_out[i] = -1.0f;
      if (ws_out) ws_out[i] = (long long)res[i].workspaceSize;
      bool ok = true;
      for (int r = 0; r < 13 && ok; ++r) {
        if (r == 3) (void)hipEventRecord(e0, nullptr);
        ok = hipblasLtMatmul(h, desc, &one, w, lw, x, lx, &zero, y, ly, y, ly, &res[i].algo, ws, res[i].workspaceSize, nullptr) == HIPBLAS_STATUS_SUCCESS;
      }
      (void)hipEventRecord(e1, nullptr);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      if (ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ms_out[i] = ms / 10.0f;
    }
    *found_out = found;
    rc = 0;
  } while (0);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (pref) hipblasLtMatmulPreferenceDestroy(pref);
  if (lw) hipblasLtMatrixLayoutDestroy(lw);
  if (lx) hipblasLtMatrixLayoutDestroy(lx);
  if (ly) hipblasLtMatrixLayoutDestroy(ly);
  if (desc) hipblasLtMatmulDescDestroy(desc);
  (void)hipFree(x); (void)hipFree(w); (void)hipFree(y); (void)hipFree(bias); (void)hipFree(ws);
  return rc;
}
#endif

// common.h -- shared helpers of libdclnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dclnet_hip.h"

#define DCL_API extern "C" __attribute__((visibility("default")))

// A/B and tuning switches.  The PRODUCT library (libdclnet_hip.so) is built without DCL_DIAG: every switch is a compile-time
// constant, no dcl_debug_* symbol is exported, superseded kernel variants are not compiled and nothing reads the environment.
// The DIAGNOSTIC library (make diag -> tests/_diag/libdclnet_hip_diag.so, -DDCL_DIAG) turns them into process-wide atomics
// set through the dcl_debug_* entry points of include/dclnet_hip.h; tests and tools/ select kernel variants through it.
#ifdef DCL_DIAG
#include <atomic>
#define DCL_HOOK_INT(name, dflt) static std::atomic<int> name{dflt}
#else
#define DCL_HOOK_INT(name, dflt) static constexpr int name = dflt
#endif

// Launch census (diagnostic library only): every hipLaunchKernelGGL of the library notes the kernel's host stub before it
// launches; dcl_debug_launch_census() lists demangled kernel names with their launch counts.  tests/test_kernel_census.py
// uses it to prove that every kernel a committed profile names was launched by a parity test.
#ifdef DCL_DIAG
void dcl_diag_note_launch(const void *host_stub);
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                    \
  do {                                                         \
    dcl_diag_note_launch((const void *)(kernelName));          \
    hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__);     \
  } while (0)
#endif

void dcl_set_error(const char *fmt, ...);

#define DCL_CHECK_ARG(cond)                                                          \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      dcl_set_error("%s: invalid argument: %s", __func__, #cond);                    \
      return DCL_EINVAL;                                                             \
    }                                                                                \
  } while (0)

#define DCL_LAUNCH_CHECK()                                                           \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      dcl_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__));      \
      return (int)e__;                                                               \
    }                                                                                \
  } while (0)

static inline int dcl_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int dcl_grid_1d(long long work, int block, int max_blocks = 256 * 16) {
  long long g = (work + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// Squared distance under the pinned contraction policy (DESIGN.md "Floating point"):
// dx*dx + dy*dy + dz*dz evaluated as fma(dz,dz, fma(dx,dx, dy*dy)).
__device__ __forceinline__ float dcl_dist2(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, dy * dy));
}
// Split-K sparse-conv launches combine their partial sums inside the launch: every (tile, split) workgroup publishes its
// partial tile, takes a ticket on the tile's counter and the last arriver adds the partials in split order (deterministic)
// and runs the epilogue.  The counters are the first kConvCounterWords int32 of the caller's split-K scratch; they must be
// zero before a launch and are left zero by it.  (32768: a capacity-mode launch is planned by its CAPACITY in tiles -- 40 crops'
// level 1 are 10240 -- and one with more tiles than counters is never split: one workgroup per capacity tile, mostly empty.)
constexpr int kConvCounterWords = 32768;
void dcl_internal_zero_words(void *p, long long nwords, hipStream_t s);

// Where a kernel gets nbr[k][o] (the input row feeding output row o under kernel offset k) from: an explicit gather table
// (dcl_rulebook_gather; the spconv shim and the training path keep it) or, inside the native backbone runner, the input
// set's occupancy grid directly -- p = o*stride - pad + k looked up by bitmask rank (no table, no k_build_nbr launch).
struct DclNbrSrc {
  const int32_t *nbr;              // explicit table, row stride `cap`; nullptr = implicit
  const int32_t *out_indices;      // (rows,4) [b,x,y,z] of the output set
  const uint32_t *in_mask;         // input set occupancy bits
  const int32_t *in_wprefix;       // exclusive popcount prefix per mask word
  const int32_t *in_perm;          // rank -> feature row (level 0) or nullptr
  int S_in, stride, pad;
};
// The 8 active sets of a backbone pass (conv / pool set of every level) for the batched scan + enumerate launches of the
// geometry stage (rulebook.hip: dcl_internal_scan_enumerate_sets).
struct DclGeoSets {
  const uint32_t *mask[8];
  int32_t *wprefix[8];      // nwords + 1 entries each
  int32_t *indices[8];      // (cap, 4) rows [b,x,y,z]
  int32_t *n_out[8];        // live row count (device)
  int32_t *block_sums[8];   // scratch: one entry per 1024-word scan block
  int nwords[8], S[8], cap[8];
  int32_t *zero_words;      // optional: 16 int32 the first batched launch zeroes (tickets of the row-order launch of the pass)
};
// PG_OP.voxelize_fp of a pass's points riding on the one-launch geometry stage (rulebook.hip: k_geometry_small)
struct DclVoxelizeRider {
  const float *feats;
  const int32_t *rules;
  float *out;
  int rows, max_active, planes, average;
};
// Row ordering of one conv layer (row_order.hip): tile slot i of the launch computes output row order[i]; bal[t] = number of
// used kernel-offset steps of the 128-row tiles in front of tile t (bal[ntiles] = all of them), smask[t] = tile t's step mask.
struct DclRowOrder {
  const int32_t *order;
  const int32_t *bal;
  const uint32_t *smask;
};
// One problem of a (possibly grouped) sparse-conv / avg-pool launch.  DCL-Net's two backbones (observed crops / template
// clouds) run the same layer shapes on different active sets with different weights: a grouped launch deals the tiles of
// both over the same resident workgroup slots, so the fixed part of a launch (ramp, tail, combine) is paid once per layer
// instead of once per layer and side.
struct DclConvSide {
  const float *feat;
  DclNbrSrc src;
  const int32_t *n_dev;         // live output rows (device-visible) or nullptr -> n_host
  const float *W, *scale, *shift;
  float *out;
  DclRowOrder ord;
  int cap, n_host;
  int form_rows;                // rows by which a launch picks among kernel FORMS with different summation orders (the stem's one /
                                // four lanes per row): data-independent (crops x the per-crop mean of the level) inside the backbone
                                // runner, so that the same batch takes the same form launch by launch and under graph capture;
                                // 0 = none (op-level calls: the row count itself)
};
// the problems of one dcl_linear_group_fwd launch (kernel argument)
struct DclLinearJobs {
  DclLinearJob job[DCL_LINEAR_MAX_JOBS];
};

struct DclConvSides {
  DclConvSide s[2];
};
struct DclOrderJob {
  const int32_t *out_indices;   // (rows, 4) [b,x,y,z] of the layer's output set
  const int32_t *n_dev;         // live row count (device-visible) or nullptr -> n_host
  const uint32_t *in_mask;      // occupancy bits of the layer's INPUT set (k3, s1, p1: same grid side as the output set)
  uint32_t *rowmask;            // scratch [cap]: 27-bit neighbour mask per row
  int32_t *order;               // out [cap]
  int32_t *tile_cnt;            // scratch [ceil(cap/128)]
  int32_t *bal;                 // out [ceil(cap/128) + 1]
  uint32_t *smask;              // out [ceil(cap/128)]
  int32_t *ticket;              // one int32, zero before the launch (left zero)
  int n_host, cap, S_in, subm;
};
constexpr int DCL_ORDER_MAX_JOBS = 8;
struct DclOrderJobs {
  DclOrderJob job[DCL_ORDER_MAX_JOBS];
};
// The 4 pooled levels of a backbone pass as seen by the point read-out (interp.hip: dcl_internal_readout_*): occupancy
// bits + ranks + rows for the 3-NN searches, features for the interpolation; out columns col[m] .. col[m]+c[m].
struct DclReadoutLevels {
  const int32_t *indices[4];
  const uint32_t *mask[4];
  const int32_t *wprefix[4];
  const float *feats[4];
  int S[4], wpc[4], c[4], col[4];
  float ve[4];
};
// How an LDS-DMA conv launch is decomposed (sparse_conv.hip: plan_conv_dma) -- whole tiles, aligned split-K or stream-K over
// G workgroups / work items, the combine inside the launch (tile tickets) or deferred to k_conv_frag_reduce.
struct DclConvPlan {
  int stream_k, aligned_ns, G, use_bal, deferred, split, counters, keep_order, tiles, nchunks;
};
// Which kernel a conv layer takes (sparse_conv.hip: conv_choose): the family and, for the LDS-DMA kernel, the tile shape
// (WR x WCW compute waves of 32 rows x 32*NT channels).
enum { DCL_CONV_GENERIC = 0, DCL_CONV_STEM = 1, DCL_CONV_WLDS = 2, DCL_CONV_DMA = 3 };
struct DclConvChoice {
  int family, WR, WCW, NT;
};
#if defined(__HIPCC__)
// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also waits for vmcnt(0): a loop that ends
// an iteration with global stores and opens the next with a barrier (to reuse LDS) stalls there until its own stores have
// LANDED, ~1 us each time.  Use where the barrier orders LDS accesses only; LDS-DMA loads (counted by vmcnt) need their own
// explicit wait, as in the conv kernel's chunk loop.
__device__ __forceinline__ void dcl_lds_barrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int dcl_nbr_at(const DclNbrSrc &s, int cap, int k, int row) {
  if (s.nbr) return s.nbr[(size_t)k * cap + row];
  const int4 q = reinterpret_cast<const int4 *>(s.out_indices)[row];
  const int kx = k / 9, ky = (k - 9 * kx) / 3, kz = k - 9 * kx - 3 * ky;             // 3x3x3 offsets: k = kz + 3 ky + 9 kx
  const int px = q.y * s.stride - s.pad + kx, py = q.z * s.stride - s.pad + ky, pz = q.w * s.stride - s.pad + kz;
  if ((unsigned)px >= (unsigned)s.S_in || (unsigned)py >= (unsigned)s.S_in || (unsigned)pz >= (unsigned)s.S_in) return -1;
  const int lin = ((q.x * s.S_in + px) * s.S_in + py) * s.S_in + pz;
  const int w = lin >> 5;
  const uint32_t m = s.in_mask[w];
  const uint32_t bit = 1u << (lin & 31);
  if (!(m & bit)) return -1;
  const int r = s.in_wprefix[w] + __popc(m & (bit - 1));
  return s.in_perm ? s.in_perm[r] : r;
}
// The three z-neighbours (kz = 0, 1, 2: k = 3 * col + kz, col = 3 kx + ky) of output row `row` at once.  In the implicit
// form the three cells are consecutive bits of the input set's occupancy mask: ONE mask word and ONE prefix word serve all
// three (two of each where the run crosses a 32-cell boundary: z = 31 | 32 of a 64-wide grid), against three dependent
// loads per neighbour taken one by one -- the look-ups of a conv tile drop from 81 loads per row to ~20, all independent.
// q = dcl_nbr_row(s, row): the output row's (b, x, y, z), loaded ONCE for all of its columns.
__device__ __forceinline__ int4 dcl_nbr_row(const DclNbrSrc &s, int row) {
  return s.nbr ? make_int4(0, 0, 0, 0) : reinterpret_cast<const int4 *>(s.out_indices)[row];
}
__device__ __forceinline__ void dcl_nbr_col(const DclNbrSrc &s, int cap, int col, int row, const int4 q, int (&v)[3]) {
  if (s.nbr) {
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) v[kz] = s.nbr[(size_t)(3 * col + kz) * cap + row];
    return;
  }
  // (branch-free but for the rare second word: an out-of-grid column reads word 0 and keeps nothing.  Early returns cost
  //  an exec-mask save / restore each and kept a caller's unrolled items from overlapping; in a launch of a few tiles the
  //  look-up code runs once, from a cold instruction cache, and its size is its time)
  const int S = s.S_in;
  const int kx = (col * 11) >> 5, ky = col - 3 * kx;                   // col / 3 for col < 9
  const int px = q.y * s.stride - s.pad + kx, py = q.z * s.stride - s.pad + ky, pz0 = q.w * s.stride - s.pad;
  const int z_lo = pz0 < 0 ? 0 : pz0, z_hi = pz0 + 2 > S - 1 ? S - 1 : pz0 + 2;
  const bool any = (unsigned)px < (unsigned)S && (unsigned)py < (unsigned)S && z_lo <= z_hi;
  const int base = any ? ((q.x * S + px) * S + py) * S : 0;
  const int w_lo = any ? (base + z_lo) >> 5 : 0, w_hi = any ? (base + z_hi) >> 5 : 0;
  const uint32_t m_lo = s.in_mask[w_lo];
  const int p_lo = s.in_wprefix[w_lo];
  uint32_t m_hi = m_lo;
  int p_hi = p_lo;
  if (w_hi != w_lo) { m_hi = s.in_mask[w_hi]; p_hi = s.in_wprefix[w_hi]; }
#pragma unroll
  for (int kz = 0; kz < 3; ++kz) {
    const int pz = pz0 + kz, lin = base + pz;
    const bool hi = (lin >> 5) != w_lo;
    const uint32_t m = hi ? m_hi : m_lo;
    const uint32_t bit = 1u << (lin & 31);
    const bool present = any && pz >= z_lo && pz <= z_hi && (m & bit);
    const int r = (hi ? p_hi : p_lo) + __popc(m & (bit - 1));
    v[kz] = present ? r : -1;
  }
  if (s.in_perm) {
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      const int r = s.in_perm[v[kz] < 0 ? 0 : v[kz]];
      v[kz] = v[kz] < 0 ? -1 : r;
    }
  }
}
#endif

// a*b + c*d + e*f under the same policy.
__device__ __forceinline__ float dcl_wsum3(float a, float b, float c, float d, float e, float f) {
  return __fmaf_rn(e, f, __fmaf_rn(a, b, c * d));
}

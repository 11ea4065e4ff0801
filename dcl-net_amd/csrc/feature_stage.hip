// feature_stage.hip -- the sparse feature stage of DCL-Net's backbones as ONE persistent launch.
//
// Reference: Backbone_SPCONV.forward (models/Modules.py:153-159) = 4 x [SparseConv3d k3 s1 p1 + BN + ReLU, SubMConv3d k3 + BN +
// ReLU, SparseAvgPool3d k3 s2 p1], each layer ~81 launches + a host sync there (spconv_ops.h:284-344, pool_ops.h:170-208).
// The per-layer kernels of sparse_conv.hip made that 12 launches (+ up to 7 combine launches for one-image calls) per
// backbone.  What is left per launch is not arithmetic: ramp and drain of the grid, the neighbour table and first operand
// fetch of every workgroup, and a tail of unequal workgroup lives -- 40-60 us of a ~100 us launch at 32 crops, and nearly all
// of a 10-15 us launch at one crop.
//
// Here every layer of BOTH backbones is a PHASE of one launch.  The launch's work items -- exactly the workgroups the
// per-layer launches would have had, computing the same segments in the same order (conv_body.h) -- are numbered in phase
// order; the grid is one workgroup of 512 threads per item, and a workgroup draws its item's number from one ticket counter
// when it starts.  An item meets its producers through completion counters per group of 1024 output rows (conv_body.h:
// DclStageIo): before its first operand fetch -- AFTER it has built its neighbour table, which needs the geometry only -- it
// waits for the input rows its tile can touch (the x-slabs of its own row range, +-1); when it is done it releases and reports
// the rows it has written.  So a layer starts under the tail of the previous one wherever the rows it needs are there, the
// other backbone's items fill the rest, and there is no grid-wide barrier anywhere (MI355X_MICROARCH.md prices one at 6-14 us
// with 512 workgroups, several times a kernel boundary) and no kernel boundary either.
//
// No co-residency assumption: tickets are drawn in phase order by RUNNING workgroups and an item waits only for rows of
// earlier phases, so the oldest unfinished item always runs -- workgroups beyond the resident 2 x 256 simply queue in the
// dispatcher, two such launches may interleave on the GPU.  Every poll is bounded all the same: a wait that exceeds its
// budget sets the stage's status word (and the caller's) and the item goes on without waiting; the host sees the status after
// the call, discards the results and falls back to the per-layer launches.  The kernel never hangs.
// (A dequeue LOOP inside persistent workgroups was the first form: with the bodies inlined under a switch inside that loop
// the compiler hoisted their loop-invariant values over it and the eight-wave tiles, which run at their 128 registers,
// spilled; as non-inlined functions they pay callee-saved spills.  One item per workgroup has neither problem, and the
// hardware dispatcher is the queue.)
#include "common.h"
#include "conv_body.h"

int dcl_internal_feature_stage_run(const DclStagePhase *phases_host, int nphases, void *table_dev, int32_t *sync_dev,
                                   long long sync_words, int32_t *status_ext, int spin_limit, dclStream_t stream);
size_t dcl_internal_feature_stage_table_bytes();

namespace {

constexpr int kStageThreads = 512;
// dynamic LDS: the largest phase body -- the 128 x 128 LDS-DMA tile: [2 stages: A 128x32 | B 32x128][Ns 27x128][kmask 4][rows 128]
constexpr int kStageLdsFloats = 2 * (128 * 32 + 32 * 128) + 27 * 128 + 4 + 128;
static_assert(27 * 16 * 32 <= kStageLdsFloats, "the 16-channel filter fits");

// LDS of the launch: the dynamic part is the largest body's need, s_ctl = [0] ticket, [4..] the item's finished-row notes
extern __shared__ __attribute__((aligned(16))) float stage_lds[];
__shared__ int32_t s_ctl[4 + 1 + 2 * kStageSigMax];

// One item of one phase: the body of the per-layer kernel it mirrors, instantiated for 512 threads and the staged hand-off.
#define STAGE_ARGS const DCL_CONST_AS DclStagePhase *P, int32_t *sync, int spin_limit, int item
#define STAGE_IO()                                                            \
  DclConvSides sides;  /* never read: staged bodies fetch their problem through io.P */ \
  DclStageIo io;                                                              \
  io.P = P;                                                                   \
  io.sync = sync;                                                             \
  io.spin_limit = spin_limit;                                                 \
  int32_t *s_sig = s_ctl + 4;                                                 \
  const int items = P->items

__device__ __forceinline__ void stage_item_stem(STAGE_ARGS) {
  STAGE_IO();
  conv_stem_body<7, 16, kStageThreads, true>(sides, 1, 27, P->subm, P->relu, stage_lds, item, items, &io, s_sig);
}
template <bool SUBM>
__device__ __forceinline__ void stage_item_wlds16(STAGE_ARGS) {
  STAGE_IO();
  conv_wlds_body<16, 32, SUBM, kStageThreads, true>(sides, 1, P->relu, stage_lds, item, items, 0, &io, s_sig);
}
__device__ __forceinline__ void stage_item_pool(STAGE_ARGS) {
  STAGE_IO();
  avgpool_body<kStageThreads, true>(sides, 1, P->cout, 27, nullptr, nullptr, reinterpret_cast<int32_t *>(stage_lds), item, items, &io,
                                    s_sig);
}
template <int WR, int WCW, int NT>
__device__ __forceinline__ void stage_item_reduce(STAGE_ARGS) {
  STAGE_IO();
  conv_frag_reduce_body<WR, WCW, NT, kStageThreads, true>(P->partial, sides, 1, P->cout, P->nchunks, P->conv_items, P->stream_k, P->relu,
                                                          item, items, 0, &io, s_sig);
}
template <int CIN, int WR, int WCW, int NT, bool ORD>
__device__ __forceinline__ void stage_item_dma(STAGE_ARGS) {
  STAGE_IO();
  conv_dma_body<CIN, WR, WCW, NT, ORD, kStageThreads, true>(sides, 1, P->cout, 27, P->subm, 0, nullptr, P->stream_k, P->aligned_ns,
                                                            P->xcd_remap, nullptr, P->use_bal, stage_lds, item, items, &io, s_sig);
}

#ifdef DCL_DIAG
// diagnostic library only (tools/stage_debug.py): per item its 100 MHz start / end stamps and which phase it belonged to
constexpr int kStageStampItems = 1 << 16;
__device__ unsigned long long g_stage_stamps[kStageStampItems * 4];
#endif

__global__ __launch_bounds__(kStageThreads, 4) void k_feature_stage(const DclStagePhase *table_g, int nphases, int total_items,
                                                                    int32_t *__restrict__ sync, int spin_limit) {
  const DCL_CONST_AS DclStagePhase *table = (const DCL_CONST_AS DclStagePhase *)table_g;
  if (threadIdx.x == 0) {
    s_ctl[4] = 0;
    s_ctl[0] = __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  {
    const int ticket = __builtin_amdgcn_readfirstlane(s_ctl[0]);
    if (ticket >= total_items) return;
    int p = 0;
    while (p + 1 < nphases && table[p + 1].first_item <= ticket) ++p;
    const DCL_CONST_AS DclStagePhase *P = table + p;
    const int item = ticket - P->first_item;
    const int kind = P->kind;
#ifdef DCL_DIAG
    if (threadIdx.x == 0 && ticket < kStageStampItems) {
      g_stage_stamps[ticket * 4 + 0] = __builtin_amdgcn_s_memrealtime();
      g_stage_stamps[ticket * 4 + 2] = (unsigned long long)p;
      g_stage_stamps[ticket * 4 + 3] = ((unsigned long long)kind << 32) | (unsigned)(P->cin * 1000 + P->WR * 100 + P->WCW * 10 + P->NT);
    }
#endif
#define STAGE_CALL(f) f(P, sync, spin_limit, item)
    if (kind == DCL_PH_STEM) {
      STAGE_CALL(stage_item_stem);
    } else if (kind == DCL_PH_WLDS16) {
      if (P->subm) STAGE_CALL(stage_item_wlds16<true>);
      else STAGE_CALL(stage_item_wlds16<false>);
    } else if (kind == DCL_PH_POOL) {
      STAGE_CALL(stage_item_pool);
    } else if (kind == DCL_PH_REDUCE) {
      if (P->WR == 4) STAGE_CALL((stage_item_reduce<4, 1, 1>));
      else if (P->NT == 2) STAGE_CALL((stage_item_reduce<2, 2, 2>));
      else STAGE_CALL((stage_item_reduce<2, 2, 1>));
    } else {
      const int shape = P->cin * 1000 + P->WR * 100 + P->WCW * 10 + P->NT;
      const bool ord = P->ord != 0;
#ifdef STAGE_ONLY_SHAPE   /* (build experiment: register use of one tile shape alone) */
      switch (shape == STAGE_ONLY_SHAPE ? shape : 0) {
#else
      switch (shape) {
#endif
        case 16411: STAGE_CALL((stage_item_dma<16, 4, 1, 1, false>)); break;
        case 32411: STAGE_CALL((stage_item_dma<32, 4, 1, 1, false>)); break;
        case 32421: STAGE_CALL((stage_item_dma<32, 4, 2, 1, false>)); break;
        case 32221: STAGE_CALL((stage_item_dma<32, 2, 2, 1, false>)); break;
        case 64421: if (ord) STAGE_CALL((stage_item_dma<64, 4, 2, 1, true>)); else STAGE_CALL((stage_item_dma<64, 4, 2, 1, false>)); break;
        case 64221: STAGE_CALL((stage_item_dma<64, 2, 2, 1, false>)); break;
        case 64422: if (ord) STAGE_CALL((stage_item_dma<64, 4, 2, 2, true>)); else STAGE_CALL((stage_item_dma<64, 4, 2, 2, false>)); break;
        case 64222: STAGE_CALL((stage_item_dma<64, 2, 2, 2, false>)); break;
        case 128422: if (ord) STAGE_CALL((stage_item_dma<128, 4, 2, 2, true>)); else STAGE_CALL((stage_item_dma<128, 4, 2, 2, false>)); break;
        case 128222: STAGE_CALL((stage_item_dma<128, 2, 2, 2, false>)); break;
        default: break;                                   // (the host refuses shapes outside this list)
      }
    }
#undef STAGE_CALL
#ifdef DCL_DIAG
    if (threadIdx.x == 0 && ticket < kStageStampItems) g_stage_stamps[ticket * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
  }
}

// table + sync area are set up by kernels (a captured hipGraph replays them; ROCm 7.2's memset / memcpy nodes are avoided on
// this path): every launch writes up to kChunk phases of the table, the first one also zeroes the sync
// words (ticket, status, all counters).
constexpr int kChunk = 14;
struct StageChunk {
  DclStagePhase ph[kChunk];
};
static_assert(sizeof(StageChunk) <= 3968, "a chunk of phases must fit the kernel-argument segment");

__global__ void k_stage_prepare(const StageChunk chunk, int first, int count, DclStagePhase *table, int32_t *sync, long long sync_words,
                                int32_t *status_ext) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int W = sizeof(DclStagePhase) / 4;
  if (t < (long long)count * W) {
    const int32_t *src = reinterpret_cast<const int32_t *>(&chunk);
    reinterpret_cast<int32_t *>(table + first)[t] = src[t];
  }
  // sync words: [0] ticket, [1] status, [2..3] where a timed-out wait reports to the caller (status_ext), then the counters
  for (long long i = t; i < sync_words; i += (long long)gridDim.x * blockDim.x) {
    int32_t v = 0;
    if (i == 2) v = (int32_t)(unsigned)((unsigned long long)status_ext & 0xffffffffull);
    if (i == 3) v = (int32_t)(unsigned)((unsigned long long)status_ext >> 32);
    sync[i] = v;
  }
}

bool stage_shape_ok(const DclStagePhase &P) {
  if (P.kind != DCL_PH_DMA) return true;
  const int shape = P.cin * 1000 + P.WR * 100 + P.WCW * 10 + P.NT;
  switch (shape) {
    case 16411: case 32411: case 32421: case 32221: case 64221: case 64222: case 128222: return P.ord == 0;
    case 64421: case 64422: case 128422: return true;
    default: return false;
  }
}

}  // namespace

size_t dcl_internal_feature_stage_table_bytes() { return sizeof(DclStagePhase) * DCL_STAGE_MAX_PHASES; }

// phases_host: the phases in queue order with first_item / items filled in.  table_dev: DCL_STAGE_MAX_PHASES phases of
// device memory, sync_dev: sync_words ints (head + every counter the phases refer to), status_ext: device-visible int32 a
// timed-out wait sets to 1 (nullptr: none).  Enqueue only.  The grid is one workgroup per item: a workgroup draws its ticket
// when it starts, so tickets are held by running workgroups only, in phase order.
int dcl_internal_feature_stage_run(const DclStagePhase *phases_host, int nphases, void *table_dev, int32_t *sync_dev,
                                   long long sync_words, int32_t *status_ext, int spin_limit, dclStream_t stream) {
  DCL_CHECK_ARG(phases_host && nphases >= 1 && nphases <= DCL_STAGE_MAX_PHASES && table_dev && sync_dev &&
                sync_words >= DCL_STAGE_SYNC_HEAD && spin_limit >= 1);
  hipStream_t s = (hipStream_t)stream;
  int total = 0;
  for (int p = 0; p < nphases; ++p) {
    DCL_CHECK_ARG(stage_shape_ok(phases_host[p]) && phases_host[p].first_item == total && phases_host[p].items >= 0);
    total += phases_host[p].items;
  }
  DclStagePhase *table = reinterpret_cast<DclStagePhase *>(table_dev);
  for (int first = 0; first < nphases; first += kChunk) {
    StageChunk chunk;
    const int count = nphases - first < kChunk ? nphases - first : kChunk;
    for (int i = 0; i < count; ++i) chunk.ph[i] = phases_host[first + i];
    for (int i = count; i < kChunk; ++i) chunk.ph[i] = phases_host[first];
    const long long zero_words = first == 0 ? sync_words : 0;
    const long long table_words = (long long)count * (long long)(sizeof(DclStagePhase) / 4);
    const long long work = zero_words > table_words ? zero_words : table_words;
    hipLaunchKernelGGL(k_stage_prepare, dim3(dcl_grid_1d(work, 256, 1024)), dim3(256), 0, s, chunk, first, count, table, sync_dev,
                       zero_words, status_ext);
  }
  if (total > 0) {
    const size_t lds = (size_t)kStageLdsFloats * sizeof(float);
    (void)hipFuncSetAttribute((const void *)k_feature_stage, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_feature_stage, dim3(total), dim3(kStageThreads), lds, s, table, nphases, total, sync_dev, spin_limit);
  }
  DCL_LAUNCH_CHECK();
  return 0;
}

#ifdef DCL_DIAG
// (start, end, phase, kind << 32 | shape) of the first n items of the last staged launch
extern "C" __attribute__((visibility("default"))) int dcl_debug_stage_stamps(unsigned long long *host, int n_items) {
  if (n_items > kStageStampItems) n_items = kStageStampItems;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stage_stamps), sizeof(unsigned long long) * 4 * n_items);
}
#endif

// backbone.hip -- native runner of one DCL-Net sparse backbone + its point-feature read-out.
//
// The reference drives this part from Python: Backbone_SPCONV.forward (models/Modules.py:153-159) ->
// 8 x SparseConvolution.forward + 4 x SparseAvgPool.forward (spconv/conv.py:114-174, pool.py:225-247), then
// Ops_GetPointFeat_spconv.forward (Modules.py:236-251): ~2.5k kernel launches and 32 host syncs per forward.
// Here the whole chain is three C calls per backbone (host cost ~ a few us per enqueued kernel, no Python
// in the loop), with a single host read-back of the 8 level sizes in between:
//
//   dcl_backbone_geometry   occupied voxels -> all 8 active sets (conv/pool x 4 levels), counts on device
//   [host reads the 8 counts -- the only synchronisation of the forward]
//   dcl_backbone_features   rulebooks + [conv+BN+ReLU, subm conv+BN+ReLU, avg-pool] x 4
//   dcl_point_features      4 x (voxel centres, crop-local 3-NN, weighted interpolation) -> (n, 480)
//
// Workspaces are caller-provided (PyTorch is the allocator); their layout is computed here and queried with
// dcl_backbone_ws_bytes / dcl_backbone_ws2_bytes.
#include "common.h"

int dcl_internal_grid_from_indices(const int32_t *indices, const int32_t *n_rows_dev, int n_rows, int batch_lo, int batch,
                                   int S, uint32_t *mask, int32_t *wprefix, int32_t *perm, int32_t *scratch,
                                   dclStream_t stream);
int dcl_internal_sparse_conv_fwd(const float *feat, const DclNbrSrc &nbr, int cap, const int32_t *n_out_dev,
                                 int n_out_host, const float *W, int cin, int cout, int kvol, int subm, const float *scale,
                                 const float *shift, int relu, float *out, float *scratch, int64_t scratch_floats,
                                 dclStream_t stream, int counters_ready = 0, const DclRowOrder *ord = nullptr);
int dcl_internal_sparse_conv_fwd_sides(const DclConvSides &sides, int nsides, int cin, int cout, int kvol, int subm, int relu,
                                       float *scratch, int64_t scratch_floats, dclStream_t stream, int counters_ready = 0,
                                       int *counters_state = nullptr);
int dcl_internal_sparse_avgpool_fwd_sides(const DclConvSides &sides, int nsides, int c, int kvol, int32_t *rf,
                                          const int32_t *rf_in, dclStream_t stream);
int dcl_internal_conv_split_cap(long long rows);
int dcl_internal_order_rows(const DclOrderJobs &jobs, int njobs, dclStream_t stream);
int dcl_internal_out_mask_k3(const uint32_t *in_mask, int batch, int S_in, int stride, uint32_t *out_mask,
                             dclStream_t stream);
int dcl_internal_scan_enumerate_sets(const DclGeoSets &g, int nsets, dclStream_t stream);
bool dcl_internal_mask_chain_ok(int S);
int dcl_internal_mask_chain(const uint32_t *mask0, int batch, const DclGeoSets &g, dclStream_t stream);
bool dcl_internal_geometry_small_ok(int batch, int S, int rows);
int dcl_internal_geometry_small(const int32_t *occ, const int32_t *n_dev, int n_host, int batch_lo, int batch, uint32_t *mask0,
                                int32_t *wprefix0, int32_t *perm0, int32_t *comm, const DclGeoSets &g, const DclVoxelizeRider *vx,
                                dclStream_t stream);
bool dcl_internal_readout_fused_ok(const DclReadoutLevels &L, int ld, bool need_search);
int dcl_internal_readout_neighbours(int n, const float *points_b4, const DclReadoutLevels &L, int nbatch, float off,
                                    float *dist2, int32_t *idx, dclStream_t stream);
bool dcl_internal_readout_one_launch_ok(int n);
int dcl_internal_readout_one_launch(int n, const float *points_b4, const DclReadoutLevels &L, int nbatch, float off, float *dist2,
                                    int32_t *idx, float *out, int ld, dclStream_t stream);
int dcl_internal_readout_interpolate(int n, const DclReadoutLevels &L, const int32_t *idx, const float *dist2, float *out,
                                     int ld, dclStream_t stream);
int dcl_internal_sparse_avgpool_fwd(const float *feat, const DclNbrSrc &nbr, int cap, const int32_t *n_out_dev,
                                    int n_out_host, int c, int kvol, float *out, int32_t *rf, dclStream_t stream);
int dcl_three_nn_sp_voxels(int n, int m, const float *unknown, const int32_t *known_indices, float ve, float off,
                           float *dist2, int32_t *idx, const int32_t *known_seg, int nbatch, int seg_stride,
                           const uint32_t *known_mask, int S, dclStream_t stream);
int dcl_three_nn_sp_strided(int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx,
                            const int32_t *known_seg, int nbatch, int seg_stride, dclStream_t stream);

#include <stdlib.h>
namespace {
DCL_HOOK_INT(g_geo_chain, 1);   // (diagnostic library only) 0 = the 8 masks of a pass as 8 chained launches, 2 = one-launch mask chain but never the one-launch geometry stage

#ifdef DCL_DIAG
// debugging aids of the diagnostic library: DCL_DBG_FEATURE_STEPS=N enqueues only the first N kernels of
// dcl_backbone_features*; DCL_EXPLICIT_NBR=1 feeds the kernels from explicit gather tables (A/B of the implicit rulebooks)
inline int dbg_steps() { const char *e = getenv("DCL_DBG_FEATURE_STEPS"); return e ? atoi(e) : 1 << 30; }
inline bool dbg_explicit_nbr() { const char *e = getenv("DCL_EXPLICIT_NBR"); return e != nullptr && atoi(e) != 0; }
#else
constexpr int dbg_steps() { return 1 << 30; }                   // the product library reads no environment
constexpr bool dbg_explicit_nbr() { return false; }
#endif

constexpr int kLevels = 4;
constexpr size_t kAlign = 256;

inline size_t align_up(size_t x) { return (x + kAlign - 1) / kAlign * kAlign; }
inline long long words(int batch, int S) { return ((long long)batch * S * S * S + 31) / 32; }

struct SetLayout {
  int S, cap;
  size_t mask, wprefix, indices;      // byte offsets in the geometry workspace
};

// row order of one conv layer (row_order.hip): outputs order / bal / smask + the pass's scratch
struct OrderLayout {
  bool on;
  size_t order, bal, smask, tile_cnt, rowmask;
};

struct GeoLayout {
  size_t mask0, wprefix0, perm0;
  SetLayout conv[kLevels], pool[kLevels];
  OrderLayout ord[kLevels][2];          // [level][0 = the dilating conv, 1 = the submanifold conv] on the level's conv set
  size_t comm;                          // exchange words of the one-launch geometry stage (k_geometry_small)
  size_t tickets;                       // 16 int32: in-launch hand-off tickets of the row-order launch (zeroed every pass)
  size_t scratch, total;
};

// Row ordering pays where launches are MFMA-bound on their tiles: from 14 crops on (one-image calls are latency-bound
// few-row launches, which take no order), for the layers of the two deep levels (64- and 128-channel inputs: the dilating
// convs skip 35-45 % of their issued chunks there, the submanifold ones 10-15 %).  Level 1 (32 channels) was measured both
// ways: its launches are bound by the per-tile fixed cost, not by MFMA work -- ordered 55.9 / 89.6 us, natural 53 / 83 --
// and its two sets are the largest ones to sort; level 0 runs the stem and a 16-channel layer (no used-chunk dealing).
DCL_HOOK_INT(kOrderMinBatch, 14);   // same-job A/B, order on vs off: 10 crops +0.9 %, 12: +1.5 %, 13: +0.7 %, 14: -1.2 %, 16: -1.5 %, 32: -3.2 % (diagnostic library: dcl_debug_order_min_batch; a huge value = off)
DCL_HOOK_INT(kOrderMinLevel, 2);    // first level whose conv layers get a row order (diagnostic library: dcl_debug_order_min_level)
inline bool order_layer(int batch, int m, int which) { return batch >= kOrderMinBatch && m >= kOrderMinLevel && (which == 0 || which == 1); }

bool make_geo_layout(int batch, int S, int V0, GeoLayout *L) {
  if (batch <= 0 || S < 16 || (S & (S - 1)) || S > 64 || V0 < 0) return false;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  const long long nw0 = words(batch, S);
  L->mask0 = take(sizeof(uint32_t) * nw0);
  L->wprefix0 = take(sizeof(int32_t) * (nw0 + 1));
  L->perm0 = take(sizeof(int32_t) * (size_t)(V0 > 0 ? V0 : 1));
  long long cap_prev = V0 > 0 ? V0 : 1;
  int s = S;
  for (int m = 0; m < kLevels; ++m) {
    const long long bound_c = (long long)batch * s * s * s;
    long long cap_c = cap_prev * 27 < bound_c ? cap_prev * 27 : bound_c;
    if (cap_c < 1) cap_c = 1;
    const long long nwc = words(batch, s);
    L->conv[m] = {s, (int)cap_c, take(sizeof(uint32_t) * nwc), take(sizeof(int32_t) * (nwc + 1)),
                  take(sizeof(int32_t) * 4 * (size_t)cap_c)};
    const int sp = s / 2;
    const long long bound_p = (long long)batch * sp * sp * sp;
    long long cap_p = cap_c * 8 < bound_p ? cap_c * 8 : bound_p;
    if (cap_p < 1) cap_p = 1;
    const long long nwp = words(batch, sp);
    L->pool[m] = {sp, (int)cap_p, take(sizeof(uint32_t) * nwp), take(sizeof(int32_t) * (nwp + 1)),
                  take(sizeof(int32_t) * 4 * (size_t)cap_p)};
    cap_prev = cap_p;
    s = sp;
  }
  for (int m = 0; m < kLevels; ++m)
    for (int q = 0; q < 2; ++q) {
      OrderLayout &o = L->ord[m][q];
      o.on = order_layer(batch, m, q);
      if (!o.on) continue;
      const size_t cap = (size_t)L->conv[m].cap, tiles = (cap + 127) / 128;
      o.order = take(sizeof(int32_t) * cap);
      o.bal = take(sizeof(int32_t) * (tiles + 2));
      o.smask = take(sizeof(uint32_t) * (tiles + 1));
      o.tile_cnt = take(sizeof(int32_t) * (tiles + 1));
      o.rowmask = take(sizeof(uint32_t) * cap);
    }
  L->tickets = take(sizeof(int32_t) * 16);
  L->comm = take(sizeof(int32_t) * 64 * 16);                   // exchange words of the one-launch geometry (small batches)
  // scan scratch: block sums of the input grid, then of the 8 generated sets (batched scan, one slice per set)
  size_t blocks = (size_t)(nw0 + 1023) / 1024 + 2;
  for (int m = 0; m < kLevels; ++m)
    blocks += (size_t)(words(batch, L->conv[m].S) + 1023) / 1024 + 1 + (size_t)(words(batch, L->pool[m].S) + 1023) / 1024 + 1;
  L->scratch = take(sizeof(int32_t) * blocks);
  L->total = off;
  return true;
}

struct FeatLayout {
  size_t nbr, x1, x2, scratch, scratch_floats, total;
};

bool make_feat_layout(const int32_t *counts, const int *chan /*9*/, FeatLayout *L) {
  size_t max_rows = 1, x1 = 4, x2 = 4, scratch = 0;
  for (int m = 0; m < kLevels; ++m) {
    const size_t nc = (size_t)(counts[2 * m] > 0 ? counts[2 * m] : 0), np = (size_t)(counts[2 * m + 1] > 0 ? counts[2 * m + 1] : 0);
    if (nc > max_rows) max_rows = nc;
    if (np > max_rows) max_rows = np;
    if (nc * chan[2 * m + 1] * 4 > x1) x1 = nc * chan[2 * m + 1] * 4;
    if (nc * chan[2 * m + 2] * 4 > x2) x2 = nc * chan[2 * m + 2] * 4;
    // stream-K scratch of the two convs: two partial-tile slots per workgroup of the 512-slot grid (see launch_conv_dma)
    for (int q = 1; q <= 2; ++q) {
      const size_t cout = (size_t)chan[2 * m + q];
      if (cout % 32 != 0 || nc == 0) continue;
      const size_t need = (size_t)2 * 512 * 128 * (cout < 128 ? cout : 128);
      if (need > scratch) scratch = need;
    }
  }
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  L->nbr = take(sizeof(int32_t) * 27 * max_rows);
  L->x1 = take(x1);
  L->x2 = take(x2);
  if (scratch) scratch += kConvCounterWords;               // tile tickets of the in-launch split-K combine (common.h)
  L->scratch = take(scratch * sizeof(float));
  L->scratch_floats = scratch;
  L->total = off;
  return true;
}

template <typename T>
inline T *at(void *base, size_t off) { return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + off); }

}  // namespace

DCL_API int dcl_backbone_ws_bytes(int batch, int S, int V0, int64_t *bytes_host) {
  GeoLayout L;
  DCL_CHECK_ARG(bytes_host && make_geo_layout(batch, S, V0, &L));
  *bytes_host = (int64_t)L.total;
  return 0;
}

static int backbone_geometry(const int32_t *occ, const int32_t *V0_dev, int V0, int batch_lo, int batch, int S, void *ws,
                             int64_t ws_bytes, int32_t *counts_dev, dclStream_t stream, const DclVoxelizeRider *vx = nullptr);

DCL_API int dcl_backbone_geometry(const int32_t *occ, int V0, int batch, int S, void *ws, int64_t ws_bytes,
                                  int32_t *counts_dev, dclStream_t stream) {
  return backbone_geometry(occ, nullptr, V0, 0, batch, S, ws, ws_bytes, counts_dev, stream);
}

// batch window: only the voxels of crops batch_lo .. batch_lo+batch-1 of `occ` enter this pass (re-based to crop 0), so
// several passes over sub-batches can share one occupied-voxel array / one voxelised feature array (pipelined forward)
DCL_API int dcl_backbone_geometry_window(const int32_t *occ, int V0, int batch_lo, int batch, int S, void *ws,
                                         int64_t ws_bytes, int32_t *counts_dev, dclStream_t stream) {
  DCL_CHECK_ARG(batch_lo >= 0);
  return backbone_geometry(occ, nullptr, V0, batch_lo, batch, S, ws, ws_bytes, counts_dev, stream);
}

// capacity mode: occ has V0_cap rows of which the first *V0_dev are live (enqueue-only, shapes independent of the data)
DCL_API int dcl_backbone_geometry_cap(const int32_t *occ, const int32_t *V0_dev, int V0_cap, int batch, int S, void *ws,
                                      int64_t ws_bytes, int32_t *counts_dev, dclStream_t stream) {
  DCL_CHECK_ARG(V0_dev);
  return backbone_geometry(occ, V0_dev, V0_cap, 0, batch, S, ws, ws_bytes, counts_dev, stream);
}

// the same + PG_OP.voxelize_fp of the pass's points (out[row][plane], rules = v2p maps of `rows` rows with max_active + 1 columns):
// a pass that takes the one-launch geometry stage carries it in that launch, any other issues it as a launch of its own
DCL_API int dcl_backbone_geometry_cap_vox(const int32_t *occ, const int32_t *V0_dev, int V0_cap, int batch, int S, void *ws,
                                          int64_t ws_bytes, int32_t *counts_dev, const float *feats, const int32_t *rules,
                                          float *vox_out, int rows, int max_active, int planes, int average, dclStream_t stream) {
  DCL_CHECK_ARG(V0_dev && rows >= 0 && max_active >= 0 && planes > 0 && (rows == 0 || (feats && rules && vox_out)));
  const DclVoxelizeRider vx{feats, rules, vox_out, rows, max_active, planes, average};
  return backbone_geometry(occ, V0_dev, V0_cap, 0, batch, S, ws, ws_bytes, counts_dev, stream, &vx);
}

static int backbone_geometry(const int32_t *occ, const int32_t *V0_dev, int V0, int batch_lo, int batch, int S, void *ws,
                             int64_t ws_bytes, int32_t *counts_dev, dclStream_t stream, const DclVoxelizeRider *vx) {
  GeoLayout L;
  DCL_CHECK_ARG(ws && counts_dev && make_geo_layout(batch, S, V0, &L) && ws_bytes >= (int64_t)L.total);
  DCL_CHECK_ARG(V0 == 0 || occ);
  int32_t *scratch = at<int32_t>(ws, L.scratch);
  // a handful of crops on 64^3 grids: the whole stage is ONE launch (rulebook.hip: k_geometry_small)
  const bool one_launch = dcl_internal_geometry_small_ok(batch, S, V0) && g_geo_chain == 1;
  int rc = 0;
  if (!one_launch) {
    rc = dcl_internal_grid_from_indices(occ, V0_dev, V0, batch_lo, batch, S, at<uint32_t>(ws, L.mask0),
                                        at<int32_t>(ws, L.wprefix0), at<int32_t>(ws, L.perm0), scratch, stream);
    if (rc) return rc;
  }
  // the 8 masks depend only on each other (bit-parallel dilation / stride-2 reduction of the previous mask): one
  // workgroup per crop walks the chain in LDS (64^3 grids; other sizes chain 8 launches), then all 8 sets are ranked and
  // decoded in three batched launches
  DclGeoSets g{};
  g.zero_words = at<int32_t>(ws, L.tickets);
  size_t so = (size_t)(words(batch, S) + 1023) / 1024 + 2;
  const uint32_t *in_mask = at<uint32_t>(ws, L.mask0);
  int s = S;
  const bool fused_chain = dcl_internal_mask_chain_ok(S) && g_geo_chain != 0;
  for (int m = 0; m < kLevels; ++m) {
    const SetLayout *sets[2] = {&L.conv[m], &L.pool[m]};
    for (int q = 0; q < 2; ++q) {
      const SetLayout &t = *sets[q];
      if (!fused_chain && !one_launch) {
        rc = dcl_internal_out_mask_k3(in_mask, batch, s, q == 0 ? 1 : 2, at<uint32_t>(ws, t.mask), stream);
        if (rc) return rc;
      }
      const int i = 2 * m + q, nw = (int)words(batch, t.S);
      g.mask[i] = at<uint32_t>(ws, t.mask);
      g.wprefix[i] = at<int32_t>(ws, t.wprefix);
      g.indices[i] = at<int32_t>(ws, t.indices);
      g.n_out[i] = counts_dev + i;
      g.block_sums[i] = scratch + so;
      g.nwords[i] = nw; g.S[i] = t.S; g.cap[i] = t.cap;
      so += (size_t)(nw + 1023) / 1024 + 1;
      in_mask = g.mask[i];
      s = t.S;
    }
  }
  if (one_launch) {
    rc = dcl_internal_geometry_small(occ, V0_dev, V0, batch_lo, batch, at<uint32_t>(ws, L.mask0), at<int32_t>(ws, L.wprefix0),
                                     at<int32_t>(ws, L.perm0), at<int32_t>(ws, L.comm), g, vx, stream);
    if (rc) return rc;
    vx = nullptr;                                      // done: rode on the stage's launch
  } else {
    if (fused_chain) {
      rc = dcl_internal_mask_chain(at<uint32_t>(ws, L.mask0), batch, g, stream);
      if (rc) return rc;
    }
    rc = dcl_internal_scan_enumerate_sets(g, 2 * kLevels, stream);
    if (rc) return rc;
  }
  if (vx && vx->rows > 0) {                            // (the separate-launch stage: the voxelisation as its own launch)
    rc = dcl_voxelize_fp(vx->feats, vx->rules, vx->out, vx->rows, vx->max_active, vx->planes, vx->average, stream);
    if (rc) return rc;
  }
  // row orders of the MFMA-bound conv layers (depends on the geometry only): all jobs of the pass in five launches
  DclOrderJobs jobs{};
  int njobs = 0;
  for (int m = 0; m < kLevels; ++m)
    for (int q = 0; q < 2; ++q) {
      const OrderLayout &o = L.ord[m][q];
      if (!o.on) continue;
      DclOrderJob &j = jobs.job[njobs++];
      j.out_indices = at<int32_t>(ws, L.conv[m].indices);
      j.n_dev = counts_dev + 2 * m;
      j.n_host = 0;
      j.cap = L.conv[m].cap;
      j.S_in = L.conv[m].S;
      j.subm = q;
      j.in_mask = q ? at<uint32_t>(ws, L.conv[m].mask) : at<uint32_t>(ws, L.pool[m - 1].mask);   // order_layer: m >= 1
      j.rowmask = at<uint32_t>(ws, o.rowmask);
      j.order = at<int32_t>(ws, o.order);
      j.tile_cnt = at<int32_t>(ws, o.tile_cnt);
      j.bal = at<int32_t>(ws, o.bal);
      j.smask = at<uint32_t>(ws, o.smask);
      j.ticket = at<int32_t>(ws, L.tickets) + njobs - 1;
    }
  return njobs ? dcl_internal_order_rows(jobs, njobs, stream) : 0;
}

#ifdef DCL_DIAG
DCL_API void dcl_debug_order_min_batch(int n) { kOrderMinBatch = n; }
DCL_API void dcl_debug_order_min_level(int m) { kOrderMinLevel = m; }
// Test hook: 1 (default) = one-launch mask chain on 64^3 grids, 0 = the 8 chained launches (A/B of the two paths).
DCL_API int dcl_debug_geometry_chain(int mode) {
  g_geo_chain.store(mode);
  return 0;
}
#endif

DCL_API int dcl_backbone_ws2_bytes(const int32_t *counts_host, const int32_t *channels_host, int64_t *bytes_host) {
  FeatLayout F;
  DCL_CHECK_ARG(counts_host && channels_host && bytes_host && make_feat_layout(counts_host, channels_host, &F));
  *bytes_host = (int64_t)F.total;
  return 0;
}

// weights[2m], weights[2m+1]: (27,Cin,Cout) of module m's conv / subm conv; scale/shift: folded BatchNorm1d.
// level_out[m]: caller-allocated (counts[2m+1], channels[2m+2]).
// One backbone of the feature stage: its geometry, inputs, parameters and outputs.  DCL-Net runs TWO backbones of the same
// shape (observed crops / template clouds, separate weights) per forward: backbone_features takes one or both and issues
// every layer as ONE launch over both sides' tiles (DclConvSides), so a forward has 8 conv + 4 pool launches, not 16 + 8.
namespace {
struct SideArgs {
  int V0;
  void *ws;
  const int32_t *counts_host, *counts_dev;
  const float *vox_feats;
  const float *const *weights, *const *scales, *const *shifts;
  void *ws2;
  int64_t ws2_bytes;
  float *const *level_out;
};
}  // namespace

// Capacity mode (whole-forward graph): how many rows the conv launches of level m will really see, roughly -- the measured
// per-crop means of the dilated sets on 64^3 grids (3217 / 1537 / 729 / 382 voxels) with a margin.  A hint for
// the few-row decisions of the conv launcher (a capacity says little about the fill), never a bound.
static int expect_rows(int batch, int level, int cap) {
  static const int per_crop[kLevels] = {3600, 1800, 900, 450};
  const long long e = (long long)batch * per_crop[level];
  return (int)(e < cap ? e : cap);
}

static int backbone_features(const SideArgs *sd, int nsides, int batch, int S, const int32_t *channels_host, dclStream_t stream) {
  DCL_CHECK_ARG(sd && nsides >= 1 && nsides <= 2 && channels_host);
  GeoLayout L[2];
  FeatLayout F[2];
  for (int i = 0; i < nsides; ++i) {
    const SideArgs &a = sd[i];
    DCL_CHECK_ARG(a.ws && a.ws2 && a.counts_host && a.weights && a.scales && a.shifts && a.level_out);
    DCL_CHECK_ARG(make_geo_layout(batch, S, a.V0, &L[i]) && make_feat_layout(a.counts_host, channels_host, &F[i]) &&
                  a.ws2_bytes >= (int64_t)F[i].total);
    DCL_CHECK_ARG((a.counts_dev != nullptr) == (sd[0].counts_dev != nullptr));
    for (int m = 0; m < kLevels; ++m)
      DCL_CHECK_ARG(a.counts_host[2 * m] <= L[i].conv[m].cap && a.counts_host[2 * m + 1] <= L[i].pool[m].cap);
  }
  // split-K scratch (partial-tile slots + tile tickets) of the launches: side 0's; the tickets are zeroed once per pass -- by
  // the first launch that draws one (a pass of few-row launches, whose combine is a launch of its own, zeroes nothing) -- and
  // every split launch leaves them zero again
  float *scr = F[0].scratch_floats ? at<float>(sd[0].ws2, F[0].scratch) : nullptr;
  const int64_t scr_floats = (int64_t)F[0].scratch_floats;
  int tickets_zero = 0;
  int steps_left = dbg_steps();
  const bool explicit_nbr = dbg_explicit_nbr() && nsides == 1;
#define DBG_STEP() do { if (--steps_left < 0) return 0; } while (0)
  const float *x[2];
  const uint32_t *in_mask[2];
  const int32_t *in_wp[2], *in_perm[2];
  for (int i = 0; i < nsides; ++i) {
    x[i] = sd[i].vox_feats;
    in_mask[i] = at<uint32_t>(sd[i].ws, L[i].mask0);
    in_wp[i] = at<int32_t>(sd[i].ws, L[i].wprefix0);
    in_perm[i] = at<int32_t>(sd[i].ws, L[i].perm0);
  }
  int s = S, rc;
  for (int m = 0; m < kLevels; ++m) {
    const int c0 = channels_host[2 * m], c1 = channels_host[2 * m + 1], c2 = channels_host[2 * m + 2];
    // The kernels derive their neighbour rows from the input set's occupancy grid themselves (DclNbrSrc, no gather
    // table and no k_build_nbr launch); the diagnostic library's DCL_EXPLICIT_NBR=1 keeps the table path for A/B runs.
    auto source = [&](int i, const int32_t *out_idx, const int32_t *n_dev, int n_host, const uint32_t *mask, const int32_t *wp,
                      const int32_t *perm, int stride, DclNbrSrc *src) -> int {
      if (explicit_nbr) {
        int32_t *nbr = at<int32_t>(sd[i].ws2, F[i].nbr);
        const int rc2 = dcl_rulebook_gather(out_idx, n_dev, n_dev ? 0 : n_host, mask, wp, perm, batch, s, 3, stride, 1, nbr,
                                            n_host, stream);
        *src = DclNbrSrc{nbr, nullptr, nullptr, nullptr, nullptr, 0, 1, 0};
        return rc2;
      }
      *src = DclNbrSrc{nullptr, out_idx, mask, wp, perm, s, stride, 1};
      return 0;
    };
    auto order_of = [&](int i, int which) -> DclRowOrder {
      const OrderLayout &o = L[i].ord[m][which];
      if (!o.on || explicit_nbr) return DclRowOrder{nullptr, nullptr, nullptr};
      return DclRowOrder{at<int32_t>(sd[i].ws, o.order), at<int32_t>(sd[i].ws, o.bal), at<uint32_t>(sd[i].ws, o.smask)};
    };
    // conv (k3,s1,p1): out rows = conv set, inputs looked up in the previous level's set; then the submanifold conv on the
    // conv set; then the pool onto the pool set -- each ONE launch over the sides that have rows at this level
    for (int stage = 0; stage < 3; ++stage) {
      DclConvSides cs{};
      int ns = 0;
      for (int i = 0; i < nsides; ++i) {
        const SideArgs &a = sd[i];
        const SetLayout &c = L[i].conv[m], &p = L[i].pool[m];
        const int nc = a.counts_host[2 * m], np = a.counts_host[2 * m + 1];     // live counts, or capacities in capacity mode
        const int32_t *nc_dev = a.counts_dev ? a.counts_dev + 2 * m : nullptr;
        const int32_t *np_dev = a.counts_dev ? a.counts_dev + 2 * m + 1 : nullptr;
        float *x1 = at<float>(a.ws2, F[i].x1), *x2 = at<float>(a.ws2, F[i].x2);
        if (stage < 2 ? nc <= 0 : np <= 0) continue;
        DclConvSide &Sd = cs.s[ns];
        if (stage == 0) {
          rc = source(i, at<int32_t>(a.ws, c.indices), nc_dev, nc, in_mask[i], in_wp[i], in_perm[i], 1, &Sd.src);
          if (rc) return rc;
          Sd.feat = x[i]; Sd.out = x1; Sd.cap = nc; Sd.n_dev = nc_dev; Sd.n_host = nc_dev ? expect_rows(batch, m, nc) : nc;
          Sd.form_rows = expect_rows(batch, m, 0x7fffffff);
          Sd.W = a.weights[2 * m]; Sd.scale = a.scales[2 * m]; Sd.shift = a.shifts[2 * m];
          Sd.ord = order_of(i, 0);
        } else if (stage == 1) {
          rc = source(i, at<int32_t>(a.ws, c.indices), nc_dev, nc, at<uint32_t>(a.ws, c.mask), at<int32_t>(a.ws, c.wprefix),
                      nullptr, 1, &Sd.src);
          if (rc) return rc;
          Sd.feat = x1; Sd.out = x2; Sd.cap = nc; Sd.n_dev = nc_dev; Sd.n_host = nc_dev ? expect_rows(batch, m, nc) : nc;
          Sd.form_rows = 0;
          Sd.W = a.weights[2 * m + 1]; Sd.scale = a.scales[2 * m + 1]; Sd.shift = a.shifts[2 * m + 1];
          Sd.ord = order_of(i, 1);
        } else {
          rc = source(i, at<int32_t>(a.ws, p.indices), np_dev, np, at<uint32_t>(a.ws, c.mask), at<int32_t>(a.ws, c.wprefix),
                      nullptr, 2, &Sd.src);
          if (rc) return rc;
          Sd.feat = x2; Sd.out = a.level_out[m]; Sd.cap = np; Sd.n_dev = np_dev;
          Sd.n_host = np_dev ? (expect_rows(batch, m, 4 * np) + 3) / 4 : np;      // (a pooled set holds about a quarter of its conv set's rows)
          Sd.ord = DclRowOrder{nullptr, nullptr, nullptr};
          Sd.form_rows = 0;
        }
        ++ns;
      }
      if (ns == 0) continue;
      DBG_STEP();
      if (stage == 0) rc = dcl_internal_sparse_conv_fwd_sides(cs, ns, c0, c1, 27, 0, 1, scr, scr_floats, stream, 0, &tickets_zero);
      else if (stage == 1) rc = dcl_internal_sparse_conv_fwd_sides(cs, ns, c1, c2, 27, 1, 1, scr, scr_floats, stream, 0, &tickets_zero);
      else rc = dcl_internal_sparse_avgpool_fwd_sides(cs, ns, c2, 27, nullptr, nullptr, stream);
      if (rc) return rc;
    }
    for (int i = 0; i < nsides; ++i) {
      x[i] = sd[i].level_out[m];
      in_mask[i] = at<uint32_t>(sd[i].ws, L[i].pool[m].mask);
      in_wp[i] = at<int32_t>(sd[i].ws, L[i].pool[m].wprefix);
      in_perm[i] = nullptr;
    }
    s = L[0].pool[m].S;
  }
#undef DBG_STEP
  return 0;
}

DCL_API int dcl_backbone_features(const int32_t *occ, int V0, int batch, int S, void *ws, const int32_t *counts_host,
                                  const int32_t *channels_host, const float *vox_feats, const float *const *weights,
                                  const float *const *scales, const float *const *shifts, void *ws2, int64_t ws2_bytes,
                                  float *const *level_out, dclStream_t stream) {
  (void)occ;
  DCL_CHECK_ARG(counts_host);
  const SideArgs a{V0, ws, counts_host, nullptr, vox_feats, weights, scales, shifts, ws2, ws2_bytes, level_out};
  return backbone_features(&a, 1, batch, S, channels_host, stream);
}

// capacity mode: every buffer is sized by the layout's capacities (dcl_backbone_caps), the live row counts are read
// from counts_dev by the kernels; no host value depends on the data (graph-capture safe).
DCL_API int dcl_backbone_caps(int batch, int S, int V0_cap, int32_t *caps_host /* i32[8]: conv1,pool1,... */) {
  GeoLayout L;
  DCL_CHECK_ARG(caps_host && make_geo_layout(batch, S, V0_cap, &L));
  for (int m = 0; m < kLevels; ++m) { caps_host[2 * m] = L.conv[m].cap; caps_host[2 * m + 1] = L.pool[m].cap; }
  return 0;
}

DCL_API int dcl_backbone_features_cap(const int32_t *occ, int V0_cap, int batch, int S, void *ws,
                                      const int32_t *counts_dev, const int32_t *channels_host, const float *vox_feats,
                                      const float *const *weights, const float *const *scales,
                                      const float *const *shifts, void *ws2, int64_t ws2_bytes,
                                      float *const *level_out, dclStream_t stream) {
  (void)occ;
  int32_t caps[8];
  DCL_CHECK_ARG(counts_dev);
  int rc = dcl_backbone_caps(batch, S, V0_cap, caps);
  if (rc) return rc;
  const SideArgs a{V0_cap, ws, caps, counts_dev, vox_feats, weights, scales, shifts, ws2, ws2_bytes, level_out};
  return backbone_features(&a, 1, batch, S, channels_host, stream);
}

// Both backbones of a forward in one feature stage: per-side arrays of 2 (index 0 / 1 = the two sides, e.g. observed /
// template).  counts_dev != NULL selects capacity mode for both sides (V0 are then capacities, counts_host is ignored).
DCL_API int dcl_backbone_features_pair(int batch, int S, const int32_t *channels_host, const int32_t *V0, void *const *ws,
                                       const int32_t *const *counts_host, const int32_t *const *counts_dev,
                                       const float *const *vox_feats, const float *const *const *weights,
                                       const float *const *const *scales, const float *const *const *shifts, void *const *ws2,
                                       const int64_t *ws2_bytes, float *const *const *level_out, dclStream_t stream) {
  DCL_CHECK_ARG(V0 && ws && vox_feats && weights && scales && shifts && ws2 && ws2_bytes && level_out &&
                (counts_dev || counts_host));
  int32_t caps[2][8];
  SideArgs a[2];
  for (int i = 0; i < 2; ++i) {
    const int32_t *ch = counts_host ? counts_host[i] : nullptr;
    const int32_t *cd = counts_dev ? counts_dev[i] : nullptr;
    if (cd) {
      const int rc = dcl_backbone_caps(batch, S, V0[i], caps[i]);
      if (rc) return rc;
      ch = caps[i];
    }
    DCL_CHECK_ARG(ch);
    a[i] = SideArgs{V0[i], ws[i], ch, cd, vox_feats[i], weights[i], scales[i], shifts[i], ws2[i], ws2_bytes[i], level_out[i]};
  }
  return backbone_features(a, 2, batch, S, channels_host, stream);
}

// Byte offsets of level m's rows / prefix inside the geometry workspace (for callers that want the voxel ids).
DCL_API int dcl_backbone_level_info(int batch, int S, int V0, int level, int64_t *indices_off_host,
                                    int64_t *wprefix_off_host, int32_t *S_level_host) {
  GeoLayout L;
  DCL_CHECK_ARG(level >= 0 && level < kLevels && make_geo_layout(batch, S, V0, &L));
  if (indices_off_host) *indices_off_host = (int64_t)L.pool[level].indices;
  if (wprefix_off_host) *wprefix_off_host = (int64_t)L.pool[level].wprefix;
  if (S_level_host) *S_level_host = L.pool[level].S;
  return 0;
}

// Ops_GetPointFeat_spconv.forward (models/Modules.py:236-251) over the 4 pooled levels.
// points_b4 (n,4) [b,x,y,z]; level_feats[m] (counts[2m+1], channels[2m+2]); out (n, ld) with the 4 levels'
// channels side by side (ld >= sum); tmp: caller scratch of n*24 + max_level_rows*16 bytes.
static int point_features(int n, const float *points_b4, int batch, int S, int V0, void *ws,
                          const int32_t *counts_host, const int32_t *counts_dev, const int32_t *channels_host,
                          const float *const *level_feats, const float *voxel_extent_host, float offset, float *out,
                          int ld, void *tmp, int64_t tmp_bytes, dclStream_t stream);

DCL_API int dcl_point_features(int n, const float *points_b4, int batch, int S, int V0, void *ws,
                               const int32_t *counts_host, const int32_t *channels_host,
                               const float *const *level_feats, const float *voxel_extent_host /*4*/, float offset,
                               float *out, int ld, void *tmp, int64_t tmp_bytes, dclStream_t stream) {
  return point_features(n, points_b4, batch, S, V0, ws, counts_host, nullptr, channels_host, level_feats,
                        voxel_extent_host, offset, out, ld, tmp, tmp_bytes, stream);
}

DCL_API int dcl_point_features_cap(int n, const float *points_b4, int batch, int S, int V0_cap, void *ws,
                                   const int32_t *counts_dev, const int32_t *channels_host,
                                   const float *const *level_feats, const float *voxel_extent_host, float offset,
                                   float *out, int ld, void *tmp, int64_t tmp_bytes, dclStream_t stream) {
  int32_t caps[8];
  DCL_CHECK_ARG(counts_dev);
  int rc = dcl_backbone_caps(batch, S, V0_cap, caps);
  if (rc) return rc;
  return point_features(n, points_b4, batch, S, V0_cap, ws, caps, counts_dev, channels_host, level_feats,
                        voxel_extent_host, offset, out, ld, tmp, tmp_bytes, stream);
}

static void fill_readout_levels(const GeoLayout &L, void *ws, const int32_t *channels_host,
                                const float *const *level_feats, const float *voxel_extent_host, DclReadoutLevels *R) {
  int col = 0;
  for (int m = 0; m < kLevels; ++m) {
    const SetLayout &p = L.pool[m];
    R->indices[m] = ws ? at<int32_t>(ws, p.indices) : nullptr;
    R->mask[m] = ws ? at<uint32_t>(ws, p.mask) : nullptr;
    R->wprefix[m] = ws ? at<int32_t>(ws, p.wprefix) : nullptr;
    R->feats[m] = level_feats ? level_feats[m] : nullptr;
    R->S[m] = p.S;
    R->wpc[m] = p.S * p.S * p.S / 32;
    R->c[m] = channels_host ? channels_host[2 * m + 2] : 4;
    R->col[m] = col;
    R->ve[m] = voxel_extent_host ? voxel_extent_host[m] : 1.0f;
    col += R->c[m];
  }
}

static int point_features(int n, const float *points_b4, int batch, int S, int V0, void *ws,
                          const int32_t *counts_host, const int32_t *counts_dev, const int32_t *channels_host,
                          const float *const *level_feats, const float *voxel_extent_host, float offset, float *out,
                          int ld, void *tmp, int64_t tmp_bytes, dclStream_t stream) {
  GeoLayout L;
  DCL_CHECK_ARG(n >= 0 && points_b4 && ws && counts_host && channels_host && level_feats && voxel_extent_host && out &&
                tmp && make_geo_layout(batch, S, V0, &L));
  if (n == 0) return 0;
  // two launches for the whole read-out when the levels qualify (they do for the backbone's own levels): all searches,
  // then all interpolations; dist2 / idx of the 4 levels live in tmp
  DclReadoutLevels R;
  fill_readout_levels(L, ws, channels_host, level_feats, voxel_extent_host, &R);
  const size_t blk = align_up((size_t)n * 48);
  if (tmp_bytes >= (int64_t)(2 * blk) && dcl_internal_readout_fused_ok(R, ld, true)) {
    float *d4 = at<float>(tmp, 0);
    int32_t *i4 = at<int32_t>(tmp, blk);
    if (dcl_internal_readout_one_launch_ok(n)) return dcl_internal_readout_one_launch(n, points_b4, R, batch, offset, d4, i4, out, ld, stream);
    int rc4 = dcl_internal_readout_neighbours(n, points_b4, R, batch, offset, d4, i4, stream);
    if (rc4) return rc4;
    return dcl_internal_readout_interpolate(n, R, i4, d4, out, ld, stream);
  }
  const size_t need = align_up((size_t)n * 12) * 2;               // level by level: one level's dist2 + idx at a time
  DCL_CHECK_ARG(tmp_bytes >= (int64_t)need);
  float *dist2 = at<float>(tmp, 0);
  int32_t *idx = at<int32_t>(tmp, align_up((size_t)n * 12));
  int col = 0, rc;
  for (int m = 0; m < kLevels; ++m) {
    const SetLayout &p = L.pool[m];
    const int np = counts_host[2 * m + 1], c = channels_host[2 * m + 2];
    DCL_CHECK_ARG(col + c <= ld);
    const int wpc = p.S * p.S * p.S / 32;           // mask words per crop (S >= 4 -> >= 2)
    // voxel centres are formed inside the 3-NN kernel while it stages its tiles (no k_voxel_centres launch)
    rc = dcl_three_nn_sp_voxels(n, np, points_b4, at<int32_t>(ws, p.indices), voxel_extent_host[m], offset, dist2, idx,
                                at<int32_t>(ws, p.wprefix), batch, wpc, at<uint32_t>(ws, p.mask), p.S, stream);
    if (rc) return rc;
    rc = dcl_three_interpolate_dist2_sp(c, np, n, level_feats[m], idx, dist2, out + col, ld, stream);
    if (rc) return rc;
    col += c;
  }
  return 0;
}

// ---- the read-out in two halves (searches | interpolation), see include/dclnet_hip.h
static int point_neighbours(int n, const float *points_b4, int batch, int S, int V0, void *ws,
                            const int32_t *counts_host, const float *voxel_extent_host, float offset, float *dist2,
                            int32_t *idx, dclStream_t stream) {
  GeoLayout L;
  DCL_CHECK_ARG(n >= 0 && points_b4 && ws && counts_host && voxel_extent_host && dist2 && idx &&
                make_geo_layout(batch, S, V0, &L));
  if (n == 0) return 0;
  DclReadoutLevels R;
  fill_readout_levels(L, ws, nullptr, nullptr, voxel_extent_host, &R);
  if (dcl_internal_readout_fused_ok(R, 16, true))
    return dcl_internal_readout_neighbours(n, points_b4, R, batch, offset, dist2, idx, stream);
  for (int m = 0; m < kLevels; ++m) {
    const SetLayout &p = L.pool[m];
    const int wpc = p.S * p.S * p.S / 32;
    int rc = dcl_three_nn_sp_voxels(n, counts_host[2 * m + 1], points_b4, at<int32_t>(ws, p.indices), voxel_extent_host[m],
                                    offset, dist2 + (size_t)m * n * 3, idx + (size_t)m * n * 3, at<int32_t>(ws, p.wprefix),
                                    batch, wpc, at<uint32_t>(ws, p.mask), p.S, stream);
    if (rc) return rc;
  }
  return 0;
}

DCL_API int dcl_point_neighbours(int n, const float *points_b4, int batch, int S, int V0, void *ws,
                                 const int32_t *counts_host, const float *voxel_extent_host, float offset, float *dist2,
                                 int32_t *idx, dclStream_t stream) {
  return point_neighbours(n, points_b4, batch, S, V0, ws, counts_host, voxel_extent_host, offset, dist2, idx, stream);
}

DCL_API int dcl_point_neighbours_cap(int n, const float *points_b4, int batch, int S, int V0_cap, void *ws,
                                     const float *voxel_extent_host, float offset, float *dist2, int32_t *idx,
                                     dclStream_t stream) {
  int32_t caps[8];
  int rc = dcl_backbone_caps(batch, S, V0_cap, caps);
  if (rc) return rc;
  return point_neighbours(n, points_b4, batch, S, V0_cap, ws, caps, voxel_extent_host, offset, dist2, idx, stream);
}

DCL_API int dcl_point_interpolate(int n, const int32_t *counts_host, const int32_t *channels_host,
                                  const float *const *level_feats, const float *dist2, const int32_t *idx, float *out,
                                  int ld, dclStream_t stream) {
  DCL_CHECK_ARG(n >= 0 && counts_host && channels_host && level_feats && dist2 && idx && out);
  if (n == 0) return 0;
  {
    DclReadoutLevels R{};
    int c0 = 0;
    for (int m = 0; m < kLevels; ++m) {
      R.feats[m] = level_feats[m];
      R.c[m] = channels_host[2 * m + 2];
      R.col[m] = c0;
      c0 += R.c[m];
    }
    if (dcl_internal_readout_fused_ok(R, ld, false)) return dcl_internal_readout_interpolate(n, R, idx, dist2, out, ld, stream);
  }
  int col = 0;
  for (int m = 0; m < kLevels; ++m) {
    const int c = channels_host[2 * m + 2];
    DCL_CHECK_ARG(col + c <= ld);
    int rc = dcl_three_interpolate_dist2_sp(c, counts_host[2 * m + 1], n, level_feats[m], idx + (size_t)m * n * 3,
                                            dist2 + (size_t)m * n * 3, out + col, ld, stream);
    if (rc) return rc;
    col += c;
  }
  return 0;
}

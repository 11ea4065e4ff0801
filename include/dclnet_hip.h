/*
 * dclnet_hip.h -- C ABI of libdclnet_hip.so: the MI355X (gfx950) kernels of
 * DCL-Net's per-crop RGB-D -> 6-DoF pose forward.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); every
 *     call only enqueues work on it -- no allocation, no synchronisation, no
 *     host read-back (graph-capture safe) unless stated;
 *   - return value: 0 on success, otherwise a hipError_t (>0) or
 *     DCL_EINVAL (-1) for bad arguments; dcl_last_error() gives the text.
 *     (The reference's launchers print to stderr and exit(-1), e.g.
 *     libs/pointnet_lib/src/ball_query_gpu.cu:62-66; a library must not.)
 *   - layouts are the reference's: row-major fp32 / int32.
 *
 * Each entry point cites the reference interface it replaces (paths relative
 * to the upstream repository).  INTEGRATION.md shows the binding a maintainer
 * would add on the reference side.
 */
#ifndef DCLNET_HIP_H_
#define DCLNET_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCL_EINVAL (-1)
typedef void *dclStream_t;

const char *dcl_last_error(void);
/* Version of THIS header's C ABI; a caller built against another version must not call in (dcl-net_amd/_native.py refuses to).
 *   1  rounds 1-4
 *   2  round 5-6: dcl_crop_points gained `int32_t *ws` in front of `stream`; dcl_backbone_features_stage,
 *      dcl_backbone_stage_ws_bytes and DCL_ESTAGE_UNSUPPORTED are gone; dcl_linear_fwd ignores its workspace arguments;
 *      new: dcl_linear_dma_fwd, dcl_linear_pool_fwd, dcl_linear_rowdot_fwd, dcl_conf_softmax, dcl_pool_finish2,
 *      dcl_linear_split_weight(_bytes), dcl_linear_split_fwd / _pool_fwd / _rowdot_fwd / _vpieces_fwd, dcl_cross_attention_ws3
 *      (+ _planes_bytes, _split_crops)                         */
#define DCL_ABI_VERSION 2
int dcl_abi_version(void);

/* ------------------------------------------------------------------ PG_OP ---
 * voxelize_idx: libs/pointgroup_ops/src/pointgroup_ops_api.cpp:6 ->
 * src/voxelize/voxelize.cpp:10-152.  HOST function (the reference runs it in
 * DataLoader workers, YCBV/dataloader_test_YCBV.py:217,223).  Two calls, like
 * the reference's resize-in-place protocol split in two: _count fills
 * input_map and returns sizes, _fill writes the caller-allocated (zeroed)
 * outputs.  All five modes of voxelize.cpp:111-149: 0 = guaranteed unique (an
 * error here, an assert there, if not), 1 = the voxel's first point, 2 = its last
 * point (max_active == 1, rows [1, point]), 3 (sum) / 4 (mean) = every point.
 * dcl_voxelize_idx_fill is _fill_mode with mode 4.                              */
int dcl_voxelize_idx_count(const int64_t *coords_host, int n, int ncol, int batch_size, int mode,
                           int32_t *input_map_host, int32_t *n_active_host, int32_t *max_active_host);
int dcl_voxelize_idx_fill(const int64_t *coords_host, int n, int ncol, const int32_t *input_map_host,
                          int n_active, int max_active, int64_t *output_coords_host,
                          int32_t *output_map_host);
int dcl_voxelize_idx_fill_mode(const int64_t *coords_host, int n, int ncol, const int32_t *input_map_host,
                               int n_active, int max_active, int mode, int64_t *output_coords_host,
                               int32_t *output_map_host);

/* DEVICE voxelize_idx (same results as the host one, coords on the GPU inside a batch x S^3 grid).
 * _count: input_map (n) and info_dev = {n_active, max_active, error flag (coordinate out of range)}; the caller reads
 * info back (one sync) to allocate output_coords (V,4) i64 / output_map (V, max_active+1) i32, then calls _fill.     */
int dcl_voxelize_idx_gpu_ws_bytes(int n, int batch, int S, int64_t *bytes_host);
int dcl_voxelize_idx_gpu_count(const int64_t *coords, int n, int batch, int S, int mode, void *ws, int64_t ws_bytes,
                               int32_t *input_map, int32_t *info_dev, dclStream_t stream);
int dcl_voxelize_idx_gpu_fill(const int64_t *coords, int n, int batch, int S, void *ws, const int32_t *input_map,
                              int n_active, int max_active, int64_t *output_coords, int32_t *output_map,
                              dclStream_t stream);
/* The crop builder's case of the above in ONE launch and without a host read-back (the loader voxelises the
 * b x n sampled points of an image's crops, YCBV/dataloader_test_YCBV.py:213-223): b <= 64 crops of n_per <= 1024 points each
 * (rows c*n_per .. belong to crop c) on 64^3 grids, one workgroup per crop.  Outputs are CAPACITY-pitched: output_map rows of
 * `pitch` ints (1 + the most points a voxel can hold), output_coords (rows,4) int64 (occ32 = 0) or int32 (occ32 = 1), both
 * b*n_per rows of which the ones behind the V live rows are zeroed; info_dev = {V, maxActive, error}, error = a point outside its crop's grid or a voxel with more than
 * pitch-1 points.  Ids in first-encounter order, rows ascending, zero padded -- bit for bit voxelize.cpp:58-152 on the live
 * part.  comm: 2*b ints that persist between calls (zero before the first); gen != 0 must differ from the previous call's. */
int dcl_voxelize_idx_crops(const int64_t *coords, int batch, int n_per, int S, int mode, int pitch, int32_t *comm, int gen,
                           int32_t *input_map, void *output_coords, int occ32, int32_t *output_map, int32_t *info_dev,
                           dclStream_t stream);

/* voxelize_fp: pointgroup_ops_api.cpp:8 -> src/voxelize/voxelize.cu:9-31.
 * feats (N,C), rules (V,1+maxActive) -> out (V,C); out need not be zeroed.     */
int dcl_voxelize_fp(const float *feats, const int32_t *rules, float *out, int n_rows, int max_active,
                    int n_planes, int average, dclStream_t stream);

/* --------------------------------------------------------------- spconv ----
 * Replaces torch.ops.spconv.get_indice_pairs_3d / indice_conv_fp32 /
 * indiceSummaryRF / indice_avgpool_fp32 (libs/spconv/src/spconv/all.cc:19-42;
 * include/spconv/spconv_ops.h:27-137,253-349; pool_ops.h:141-208).
 *
 * An active set on a batch x S^3 grid is described on the device by
 *   mask    u32[batch*S^3/32]   occupancy bits, linear index ((b*S+x)*S+y)*S+z
 *   wprefix i32[batch*S^3/32+1] exclusive popcount prefix (last = #active)
 *   perm    i32[V] or NULL      rank (ascending linear index) -> feature row;
 *                               NULL = rows already in ascending order
 * Output voxels of conv/pool are numbered in ascending linear index, the
 * order the reference's GPU path produces (torch::_unique, spconv_ops.h:126).
 * The rulebook is kept in gather form: nbr[k*cap + o] = input row feeding
 * output row o through kernel offset k (or -1); k = kz + 3*ky + 9*kx as in
 * geometry.h:61-70.  All counts stay on the device.                            */

/* scratch: i32[(nwords+1023)/1024 + 1].  Builds mask/wprefix/perm from an explicit voxel list
 * (rows in any order, e.g. voxelize_idx's first-encounter order).              */
int dcl_grid_from_indices(const int32_t *indices, int n_rows, int batch, int S, uint32_t *mask,
                          int32_t *wprefix, int32_t *perm, int32_t *scratch, dclStream_t stream);

/* Step 1 of a non-submanifold conv / pool rulebook: the OUTPUT active set only
 * (prepareIndicePairsKernel + torch::_unique + assignGridAndIndiceOutKernel, indice.cu.h:24-65,
 * 112-128): out mask/wprefix on the S_out grid, out_indices (cap_out,4) in ascending linear
 * index, n_out_dev[0].  n_in is *n_in_dev if non-NULL (n_in_host then only bounds the launch).
 * scratch as above, sized for the S_out grid.                                   */
int dcl_conv_out_grid(const int32_t *in_indices, const int32_t *n_in_dev, int n_in_host,
                      const uint32_t *in_mask /* optional: enables the bit-parallel path */, int batch,
                      int S_in, int ksize, int stride, int padding, uint32_t *out_mask,
                      int32_t *out_wprefix, int32_t *out_indices, int32_t *n_out_dev, int cap_out,
                      int32_t *scratch, dclStream_t stream);

/* Step 2: the gather-form rulebook nbr i32[kvol*cap] of output rows `out_indices` against the
 * input set (in_mask, in_wprefix, in_perm) on the S_in grid: the input feeding output o through
 * offset k sits at o*stride - padding + k.  Submanifold conv = out set == in set, stride 1,
 * padding ksize/2.                                                              */
int dcl_rulebook_gather(const int32_t *out_indices, const int32_t *n_out_dev, int n_out_host,
                        const uint32_t *in_mask, const int32_t *in_wprefix, const int32_t *in_perm,
                        int batch, int S_in, int ksize, int stride, int padding, int32_t *nbr, int cap,
                        dclStream_t stream);

/* Non-submanifold conv / pool rulebook (ksize^3 offsets, stride, padding, dilation 1):
 * in_indices (n_in rows; n_in read from n_in_dev if non-NULL else n_in_host) ->
 * out mask/wprefix on the S_out grid, out_indices (cap_out,4), n_out_dev[0],
 * nbr i32[27*cap_out].  Rows beyond n_out are untouched.                      */
int dcl_rulebook_conv(const int32_t *in_indices, const int32_t *n_in_dev, int n_in_host,
                      const uint32_t *in_mask, const int32_t *in_wprefix, const int32_t *in_perm,
                      int batch, int S_in, int ksize, int stride, int padding,
                      uint32_t *out_mask, int32_t *out_wprefix, int32_t *out_indices,
                      int32_t *n_out_dev, int32_t *nbr, int cap_out, int32_t *scratch,
                      dclStream_t stream);

/* Submanifold rulebook: out set = in set (same rows).  nbr i32[27*cap].        */
int dcl_rulebook_subm(const int32_t *indices, const int32_t *n_dev, int n_host,
                      const uint32_t *mask, const int32_t *wprefix, const int32_t *perm,
                      int batch, int S, int ksize, int32_t *nbr, int cap, dclStream_t stream);

/* Export a gather-form rulebook in the reference's format: indice_pairs
 * i32[27][2][n_in_cap] (-1 padded) + indice_num i32[27] (spconv_ops.h:55-60).
 * Pair order inside an offset is unspecified, as in the reference.             */
int dcl_rulebook_to_pairs(const int32_t *nbr, int cap, const int32_t *n_out_dev, int n_out_host,
                          int kvol, int32_t *indice_pairs, int n_in_cap, int32_t *indice_num,
                          dclStream_t stream);

/* Importer for callers that hold the REFERENCE's rulebook format (the arguments of torch.ops.spconv.indice_conv_fp32 /
 * indice_avgpool_fp32 / indiceSummaryRF, libs/spconv/src/spconv/all.cc:19-42, as spconv/conv.py:149-166 and
 * pool.py:237-242 pass them): indice_pairs i32 (kvol, 2, pair_stride) [k][0][j] = input row, [k][1][j] = output row, -1
 * padded; indice_num i32 (kvol) ON THE DEVICE (no host read-back -- the reference copies it to the CPU,
 * spconv_ops.h:264).  Fills the gather table nbr (kvol, cap) (cap >= n_out) that dcl_sparse_conv_fwd* /
 * dcl_sparse_avgpool_fwd* consume; pairs that point outside [0,n_in) x [0,n_out) are dropped and counted in
 * *bad_pairs_dev (may be NULL).                                                                                          */
int dcl_rulebook_from_pairs(const int32_t *indice_pairs, int pair_stride, const int32_t *indice_num_dev, int kvol,
                            int n_in, int n_out, int32_t *nbr, int cap, int32_t *bad_pairs_dev, dclStream_t stream);
/* torch.ops.spconv.indiceSummaryRF (all.cc:33; summaryRF.cu:26-68) on the pair format: rf[o] = number of pairs whose
 * output row is o, i32 (n_out).                                                                                          */
int dcl_indice_summary_rf(const int32_t *indice_pairs, int pair_stride, const int32_t *indice_num_dev, int kvol,
                          int n_out, int32_t *rf, dclStream_t stream);

/* indice_conv_fp32 (+ the BatchNorm1d(eval)+ReLU that SparseSequential applies next,
 * models/Modules.py:36-40): out[o] = act( (sum_k feat[nbr[k][o]] * W[k]) * scale + shift ).
 * W (27,Cin,Cout); scale/shift (Cout) or NULL; subm != 0 accumulates the centre offset
 * first (spconv_ops.h:289-299).                                                */
int dcl_sparse_conv_fwd(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev,
                        int n_out_host, const float *W, int cin, int cout, int kvol, int subm,
                        const float *scale, const float *shift, int relu, float *out,
                        dclStream_t stream);

/* Same, with caller scratch (partial-tile slots + tile tickets): the launch's chunk units -- tile-major, 27*Cin/32 chunks
 * per 128 x {32,64,128} output tile -- are dealt evenly to the 2 x 256 resident workgroup slots ("stream-K"), or the tiles
 * are split into aligned K segments, or kept whole, whichever a small cost model prefers; partial tiles are combined
 * INSIDE the launch by the workgroup that draws a tile's last ticket, in ascending chunk order (deterministic), then the
 * epilogue.  Launches of a few rows (cap <= 4096, or <= 65536 capacity rows with n_out_dev: one-image calls are
 * latency-bound on the contraction loop) take 4 chunks per workgroup and combine in a second small launch.
 * The first 8192 int32 of `scratch` are the tickets: the call zeroes them (they are left zero).
 * scratch_floats >= dcl_sparse_conv_scratch_floats(cap, cout) enables every decomposition; NULL = whole tiles only.   */
int dcl_sparse_conv_fwd_ws(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev,
                           int n_out_host, const float *W, int cin, int cout, int kvol, int subm,
                           const float *scale, const float *shift, int relu, float *out, float *scratch,
                           int64_t scratch_floats, dclStream_t stream);

/* Row ordering of a conv layer's output rows + the launch that uses it (csrc/row_order.hip; replaces nothing in the
 * reference -- it re-orders the work of indiceConv, spconv_ops.h:284-344, not its results).  dcl_order_rows: for the output
 * set `out_indices` (rows, 4) [b,x,y,z] of a k3 / s1 / p1 layer whose INPUT set has the occupancy bits `in_mask` on a
 * batch x S_in^3 grid (S_in in {8,16,32,64}), writes
 *   order[i]  = the output row tile slot i computes (rows sorted, stably, by a 9-bit key of their neighbourhood shape),
 *   bal[t]    = number of used kernel-offset steps of the 128-row tiles in front of tile t (bal[ntiles] = their total),
 *   smask[t]  = tile t's step mask (bit s = step s of the visiting order has a neighbour among the tile's rows).
 * order: cap ints; bal: cap/128 + 2 ints; smask: cap/128 + 1 words; ws: dcl_order_rows_ws_bytes(cap).  Live row count from
 * n_out_dev (device-visible) or n_out_host.  dcl_sparse_conv_fwd_ordered = dcl_sparse_conv_fwd_ws with that order: same
 * output rows, same values up to the fp32 summation split points of the decomposition (bal / smask may be NULL: the order
 * alone).                                                                                                             */
int dcl_order_rows_ws_bytes(int cap, int64_t *bytes_host);
int dcl_order_rows(const int32_t *out_indices, const int32_t *n_out_dev, int n_out_host, int cap, const uint32_t *in_mask,
                   int S_in, int subm, void *ws, int64_t ws_bytes, int32_t *order, int32_t *bal, uint32_t *smask,
                   dclStream_t stream);
int dcl_sparse_conv_fwd_ordered(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev, int n_out_host,
                                const float *W, int cin, int cout, int kvol, int subm, const float *scale,
                                const float *shift, int relu, float *out, float *scratch, int64_t scratch_floats,
                                const int32_t *order, const int32_t *bal, const uint32_t *smask, dclStream_t stream);
int dcl_sparse_conv_scratch_floats(int rows_cap, int cout, int64_t *floats_host);

/* indiceSummaryRF + indice_avgpool_fp32 (use_gs=False): rf[o] = #valid offsets,
 * out[o] = sum_k asc feat[nbr[k][o]] / (float)rf[o].  rf may be NULL.          */
int dcl_sparse_avgpool_fwd(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev,
                           int n_out_host, int c, int kvol, float *out, int32_t *rf,
                           dclStream_t stream);

/* torch.ops.spconv.indice_avgpool_fp32 (all.cc:34; pool_ops.h:170-208) with the caller's `summaryrf` (i32 (n_out): the
 * receptive-field counts of indiceSummaryRF when use_gs=False, the kernel volume when use_gs=True, functional.py:146-155)
 * as the divisor: out[o] = sum_k asc feat[nbr[k][o]] / (float)summaryrf[o].                                              */
int dcl_sparse_avgpool_fwd_rf(const float *feat, const int32_t *nbr, int cap, const int32_t *n_out_dev,
                              int n_out_host, int c, int kvol, const int32_t *summaryrf, float *out,
                              dclStream_t stream);

/* Several zero-padded 2-D copies of 4-byte elements in ONE launch (input staging / result hand-over of the whole-forward
 * hipGraph): job: dst (rows_dst x cols_dst; dense, or a column block with row pitch dst_pitch elements) <- src (rows_src x
 * cols_src, row pitch src_pitch elements; int64 elements narrowed to int32 when src_is_i64), zero outside src;
 * src == NULL fills dst with fill_value.                                                                                */
#define DCL_PAD_COPY_MAX_JOBS 12
typedef struct {
  void *dst;
  const void *src;
  int32_t rows_dst, cols_dst, rows_src, cols_src, src_pitch, src_is_i64, fill_value, dst_pitch /* 0 = dense */;
} DclPadCopyJob;
int dcl_pad_copy_many(const DclPadCopyJob *jobs_host, int njobs, dclStream_t stream);

/* ---------------------------------------------------- native backbone runner ---
 * One sparse backbone of DCL-Net (Backbone_SPCONV.forward, models/Modules.py:153-159: 4 x [SparseConv3d k3 s1 p1
 * + BN + ReLU, SubMConv3d k3 + BN + ReLU, SparseAvgPool3d k3 s2 p1]) and its point read-out
 * (Ops_GetPointFeat_spconv.forward, :236-251) as three enqueue-only calls with ONE host read-back of the 8 level
 * sizes in between (the reference: ~2.5k launches, 32 blocking copies).  Workspaces are caller-allocated;
 * `channels_host` = the 9 backbone dims [7,16,32,32,64,64,128,128,256]; counts = [n_conv1, n_pool1, ..., n_pool4].
 * `counts_dev` of the geometry calls is written by kernels: device memory, or device-visible pinned host memory
 * (hipHostMalloc) -- the caller then only waits for the stream and reads the 8 sizes, no copy is enqueued.              */
int dcl_backbone_ws_bytes(int batch, int S, int V0, int64_t *bytes_host);
int dcl_backbone_geometry(const int32_t *occ, int V0, int batch, int S, void *ws, int64_t ws_bytes,
                          int32_t *counts_dev /* i32[8] */, dclStream_t stream);
/* Same over a batch window: only the voxels of crops batch_lo .. batch_lo+batch-1 of `occ` (all V0 rows are scanned)
 * enter the pass, re-based to crop 0.  Sub-batch passes can then share one occupied-voxel / voxel-feature array, which is
 * how Network.forward pipelines the sparse half of chunk c+1 under the dense half of chunk c.  ws sized for (batch, S, V0). */
int dcl_backbone_geometry_window(const int32_t *occ, int V0, int batch_lo, int batch, int S, void *ws, int64_t ws_bytes,
                                 int32_t *counts_dev, dclStream_t stream);
int dcl_backbone_ws2_bytes(const int32_t *counts_host, const int32_t *channels_host, int64_t *bytes_host);
/* weights_host[8]/scales_host[8]/shifts_host[8]: HOST arrays of device pointers ((27,Cin,Cout) / (Cout));
 * level_out_host[4]: HOST array of device pointers, level m = (counts[2m+1], channels[2m+2]) floats.            */
int dcl_backbone_features(const int32_t *occ, int V0, int batch, int S, void *ws, const int32_t *counts_host,
                          const int32_t *channels_host, const float *vox_feats,
                          const float *const *weights_host, const float *const *scales_host,
                          const float *const *shifts_host, void *ws2, int64_t ws2_bytes,
                          float *const *level_out_host, dclStream_t stream);
/* Capacity mode of the three calls (whole-forward hipGraph capture): nothing on the host depends on the data.
 * occ holds V0_cap rows, the first *V0_dev live; buffers (level_out, ws2) are sized by dcl_backbone_caps' row
 * capacities (pass them as counts to dcl_backbone_ws2_bytes); kernels read the live counts from counts_dev.        */
int dcl_backbone_caps(int batch, int S, int V0_cap, int32_t *caps_host /* i32[8] */);
int dcl_backbone_geometry_cap(const int32_t *occ, const int32_t *V0_dev, int V0_cap, int batch, int S, void *ws,
                              int64_t ws_bytes, int32_t *counts_dev, dclStream_t stream);
/* dcl_backbone_geometry_cap + dcl_voxelize_fp(feats, rules, vox_out, rows, max_active, planes, average) of the pass's points:
 * a pass of up to 16 crops carries the voxelisation in its one geometry launch (it depends on the input only). */
int dcl_backbone_geometry_cap_vox(const int32_t *occ, const int32_t *V0_dev, int V0_cap, int batch, int S, void *ws,
                                  int64_t ws_bytes, int32_t *counts_dev, const float *feats, const int32_t *rules,
                                  float *vox_out, int rows, int max_active, int planes, int average, dclStream_t stream);
int dcl_backbone_features_cap(const int32_t *occ, int V0_cap, int batch, int S, void *ws, const int32_t *counts_dev,
                              const int32_t *channels_host, const float *vox_feats, const float *const *weights_host,
                              const float *const *scales_host, const float *const *shifts_host, void *ws2,
                              int64_t ws2_bytes, float *const *level_out_host, dclStream_t stream);
/* BOTH backbones of a forward in one feature stage (DCL-Net runs two of the same shape: observed crops and template clouds,
 * models/DCL_Net.py:93-94,173-186): per-side arrays of 2.  Every layer becomes ONE launch over both sides' tiles -- 8 conv
 * + 4 pool launches per forward instead of 16 + 8 -- with results identical to two dcl_backbone_features calls up to the
 * fp32 summation split points of the stream-K decomposition.  counts_dev != NULL selects capacity mode for both sides
 * (V0 are then capacities and counts_host is ignored); otherwise counts_host[i] are the sides' 8 level sizes.  The sides
 * share batch, S and the channel list; each has its own geometry workspace, parameters, ws2 and level outputs.         */
int dcl_backbone_features_pair(int batch, int S, const int32_t *channels_host, const int32_t *V0, void *const *ws,
                               const int32_t *const *counts_host, const int32_t *const *counts_dev,
                               const float *const *vox_feats, const float *const *const *weights,
                               const float *const *const *scales, const float *const *const *shifts, void *const *ws2,
                               const int64_t *ws2_bytes, float *const *const *level_out, dclStream_t stream);
int dcl_point_features_cap(int n, const float *points_b4, int batch, int S, int V0_cap, void *ws,
                           const int32_t *counts_dev, const int32_t *channels_host,
                           const float *const *level_feats_host, const float *voxel_extent_host, float offset,
                           float *out, int ld, void *tmp, int64_t tmp_bytes, dclStream_t stream);

/* byte offsets (inside ws) of pooled level `level`'s (b,x,y,z) rows and mask-word prefix, and its grid size */
int dcl_backbone_level_info(int batch, int S, int V0, int level, int64_t *indices_off_host,
                            int64_t *wprefix_off_host, int32_t *S_level_host);
/* points_b4 (n,4) [b,x,y,z] -> out (n, ld): levels' channels side by side.  voxel_extent_host[4] = unit*scale per
 * level (fp32), offset = -0.5*unit*64.  tmp >= 2*align256(48n) bytes: then the read-out is two launches (all
 * searches, all interpolations); with only 2*align256(12n) it goes level by level.                                */
int dcl_point_features(int n, const float *points_b4, int batch, int S, int V0, void *ws,
                       const int32_t *counts_host, const int32_t *channels_host,
                       const float *const *level_feats_host, const float *voxel_extent_host, float offset,
                       float *out, int ld, void *tmp, int64_t tmp_bytes, dclStream_t stream);
/* The same read-out in two halves, so that a caller can overlap the searches with the convolutions: the 3-NN searches
 * need the level geometry only (dcl_backbone_geometry), the interpolation needs the level features.
 * dist2 / idx: 4 level blocks of n*3 entries each (level-major).  Results are identical to dcl_point_features.
 * The _cap forms take dcl_backbone_caps' V0_cap (whole-forward hipGraph capture).                                   */
int dcl_point_neighbours(int n, const float *points_b4, int batch, int S, int V0, void *ws, const int32_t *counts_host,
                         const float *voxel_extent_host, float offset, float *dist2, int32_t *idx, dclStream_t stream);
int dcl_point_neighbours_cap(int n, const float *points_b4, int batch, int S, int V0_cap, void *ws,
                             const float *voxel_extent_host, float offset, float *dist2, int32_t *idx,
                             dclStream_t stream);
int dcl_point_interpolate(int n, const int32_t *counts_host, const int32_t *channels_host,
                          const float *const *level_feats_host, const float *dist2, const int32_t *idx, float *out,
                          int ld, dclStream_t stream);

/* ----------------------------------------------------------- pointnet_sp ---
 * three_nn_wrapper(n, m, unknown(N,4), known(M,4), dist2(N,3), idx(N,3)):
 * libs/pointnet_sp/src/pointnet2_api.cpp:7 -> interpolate_gpu.cu:9-77.
 * known_seg (i32[nbatch+1], optional): known rows of batch b are exactly
 * [known_seg[b], known_seg[b+1]) -- lets a query scan only its own crop; NULL
 * scans all m rows comparing the batch column like the reference.              */
int dcl_three_nn_sp(int n, int m, const float *unknown, const float *known, float *dist2,
                    int32_t *idx, const int32_t *known_seg, int nbatch, dclStream_t stream);

/* three_interpolate_wrapper(c, m, n, points(M,C), idx, weight, out(N,C)):
 * pointnet2_api.cpp:8 -> interpolate_gpu.cu:80-122.  out_stride (>= c, in floats) lets the
 * caller write straight into a column block of the 480-channel concat.         */
int dcl_three_interpolate_sp(int c, int m, int n, const float *points, const int32_t *idx,
                             const float *weight, float *out, int out_stride, dclStream_t stream);

/* three_interpolate with the weights of Ops_nearest_neighbor_interpolate computed in-kernel
 * (models/Modules.py:221-224: dist = sqrt(dist2); w = (1/(dist+1e-8)) / sum) -- takes the dist2
 * that dcl_three_nn_sp returns.                                                */
int dcl_three_interpolate_dist2_sp(int c, int m, int n, const float *points, const int32_t *idx,
                                   const float *dist2, float *out, int out_stride,
                                   dclStream_t stream);

/* Ops_tensor2points (models/Modules.py:204-211): voxel rows (V,4) i32 [b,x,y,z] -> centres (V,4)
 * f32 [b, x*ve+off+0.5*ve, ...] (fp32, left to right).  n from n_dev if non-NULL (n_host = bound). */
int dcl_voxel_centres(const int32_t *indices, const int32_t *n_dev, int n_host, float ve, float off,
                      float *centres, dclStream_t stream);

/* ---------------------------------------------------------- pointnet_lib ---
 * libs/pointnet_lib/src/pointnet2_api.cpp:10-25.  Same argument order.         */
int dcl_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int32_t *idx, dclStream_t stream);       /* idx fully written */
int dcl_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                     const int32_t *idx, float *out, dclStream_t stream);
/* Same, writing channel block [0,c) of an output whose batch entries hold out_batch_channels >= c channels (pass `out`
 * already offset to the first channel to fill): lets QueryAndGroup fill its (B, C+3, npoint, nsample) result in place. */
int dcl_group_points_into(int b, int c, int n, int npoints, int nsample, const float *points, const int32_t *idx,
                          float *out, int out_batch_channels, dclStream_t stream);
int dcl_gather_points(int b, int c, int n, int npoints, const float *points, const int32_t *idx,
                      float *out, dclStream_t stream);
/* temp (B,N) must be pre-filled with 1e10 (pointnet2_utils.py:27); updated in place. */
int dcl_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                int32_t *idxs, dclStream_t stream);
int dcl_knn(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2,
            int32_t *idx, dclStream_t stream);                                 /* k <= 200 */
int dcl_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                 int32_t *idx, dclStream_t stream);
int dcl_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                          const float *weight, float *out, dclStream_t stream);

/* ------------------------------------------------------------ dense path ---
 * All dense operands are POINT-major: X[(b*n + p)*ld + c] (row = point, ld = row stride in
 * floats).  That is the layout the 3-NN interpolation emits and 2-D GEMMs consume; the
 * reference's (b,C,n) tensors are transposed views of it.
 *
 * Cross-attention of Aligner.forward + the extra bmm (models/Modules.py:162-169;
 * models/DCL_Net.py:206-215), one direction per call, without materialising A:
 *   S[j,i] = <K[j,:], Q[i,:]> over 64 channels, A = softmax over the KEY axis j,
 *   O1[i,:] = sum_j A[j,i] V1[j,:]  (dv1 ch),  O2 likewise from V2 (dv2 ch; may be NULL/0)
 * Q (b*nq rows), K, V1, V2 (b*nk rows), O1, O2 (b*nq rows).  fp32 MFMA, online softmax.
 * The product library is built for DCL-Net's own channel split, dv1 = 256 and dv2 = 64 (DCL_EINVAL otherwise); the
 * diagnostic library (-DDCL_DIAG) also carries the general-shape predecessors: dv1, dv2 multiples of 32 with
 * (dv1+dv2)/32 in {1,2,4,8,10}.  All ld % 4 == 0, 16-B aligned.   */
int dcl_cross_attention(int b, int nq, int nk, const float *Q, int ldq, const float *K, int ldk,
                        const float *V1, int dv1, int ldv1, float *O1, int ldo1, const float *V2,
                        int dv2, int ldv2, float *O2, int ldo2, dclStream_t stream);
/* Same with caller scratch: launches with few workgroups (small batches) split the keys over up to 8 workgroups per query
 * block; the partial (unnormalised sums, running max, weight sum) records in `scratch` are merged by a second kernel.
 * scratch_floats >= dcl_cross_attention_scratch_floats(b, nq) enables every split; NULL = dcl_cross_attention.         */
int dcl_cross_attention_ws(int b, int nq, int nk, const float *Q, int ldq, const float *K, int ldk,
                           const float *V1, int dv1, int ldv1, float *O1, int ldo1,
                           const float *V2, int dv2, int ldv2, float *O2, int ldo2,
                           float *scratch, int64_t scratch_floats, dclStream_t stream);
/* Same; concurrent_launches = 2 tells the launcher that ANOTHER attention launch of the same size runs side by side on a second
 * stream / graph branch (the two directions of a forward): the workgroup shape is then chosen for the pair (1 = a lone launch =
 * dcl_cross_attention_ws).  A hint for speed only: results do not depend on it beyond the summation order of a key split. */
int dcl_cross_attention_ws2(int b, int nq, int nk, const float *Q, int ldq, const float *K, int ldk,
                            const float *V1, int dv1, int ldv1, float *O1, int ldo1,
                            const float *V2, int dv2, int ldv2, float *O2, int ldo2,
                            float *scratch, int64_t scratch_floats, int concurrent_launches, dclStream_t stream);
/* dcl_cross_attention_ws2 with both products of large calls (S = K Q^T and P.V) on the bf16 matrix pipe at fp32-sized errors
 * (csrc/dense.hip: k_cross_attn_split; the scheme of dcl_linear_split_fwd): `planes` is scratch for the three bf16 pieces of K and
 * V in tile order, dcl_cross_attention_planes_bytes(b, nq, nk, concurrent_launches) bytes (0 = a call of this size keeps the
 * fp32-MFMA kernel; planes = NULL or fewer bytes: likewise), 16-byte aligned.  V must be [256 | 64] channels. */
int64_t dcl_cross_attention_planes_bytes(int b, int nq, int nk, int concurrent_launches);
/* How many of the b crops of such a call take the split kernel: 0, b, or the whole rounds of a pair call.  When it is b, V1 may be
 * NULL in dcl_cross_attention_ws3: its pieces are then expected in `planes` already, written by the GEMM that computes V1
 * (dcl_linear_split_vpieces_fwd) -- the fp32 V1 is never stored and the piece pass skips its 256 channels. */
int dcl_cross_attention_split_crops(int b, int nq, int nk, int concurrent_launches);
int dcl_cross_attention_ws3(int b, int nq, int nk, const float *Q, int ldq, const float *K, int ldk,
                            const float *V1, int dv1, int ldv1, float *O1, int ldo1, const float *V2, int dv2,
                            int ldv2, float *O2, int ldo2, float *scratch, int64_t scratch_floats,
                            int concurrent_launches, void *planes, int64_t planes_bytes, dclStream_t stream);
int dcl_cross_attention_scratch_floats(int b, int nq, int64_t *floats_host);

/* Confidence pooling (models/DCL_Net.py:217-228): conf = sigmoid(cat[logit1 (b,n1), logit2
 * (b,n2)]) -> conf (b,n1+n2); w = softmax(conf) -> w_scratch (b,n1+n2);
 * part1 (b,nslices,C): slice partials of sum_{j<n1} w[j] F1[b,j,:]; part2 likewise over F2 with
 * w[n1+j]; wsum (b,2) = the two partial sums of w.  The caller adds the slices in index order
 * (deterministic).  pooled of the reference = sum(part1)+sum(part2) when F1/F2 already carry their
 * trailing BatchNorm, else s1*P1 + t1*wsum1 + s2*P2 + t2*wsum2.  C % 4 == 0.                    */
int dcl_conf_pool(int b, int c, int n1, int n2, const float *logit1, const float *logit2,
                  const float *F1, int ld1, const float *F2, int ld2, float *conf, float *w_scratch,
                  int nslices, float *part1, float *part2, float *wsum, dclStream_t stream);
/* The second form of that sum in one launch: out (b,C) = ((s1*P1 + t1*wsum1) + s2*P2) + t2*wsum2 with
 * P = the slice partials of dcl_conf_pool added in a fixed order (s, t: the fusers' trailing BatchNorm1d in eval form). */
int dcl_pool_finish(int b, int c, int nslices, const float *part1, const float *part2, const float *wsum,
                    const float *scale1, const float *shift1, const float *scale2, const float *shift2, float *out,
                    dclStream_t stream);

/* out (rows, c) = relu(term + xyz (rows,3) @ W3 (3,c)): the xyz part of the first MLP_share layer of the stage-2 refiner
 * (models/refiner.py:78-80: Conv1d(259 -> 512) over cat[xyz, F_Xo_p]; `term` = F_Xo_p's part + bias, constant over the
 * refine iterations) in ONE pass over the tensor.  c % 4 == 0, 16-B aligned rows.                                       */
int dcl_affine3_relu(int64_t rows, int c, const float *xyz, const float *W3, const float *term, float *out,
                     dclStream_t stream);
/* regressor_rot + regressor_trans (models/DCL_Net.py:139-151,231-235; Head_MultiLayerPerceptron 1024 -> 512 -> 128 -> 9 | 3,
 * ReLU after the first two layers) on the pooled feature (b,1024), both heads in two launches -- meant for a handful of
 * crops (one-image calls); large batches use library GEMMs.  rot_layers / trans_layers: {W1t (1024,512), b1, W2t (512,128),
 * b2, W3t (128,9|3), b3}, matrices stored (in, out) row-major.  h1_scratch: 2*b*512 floats.  R (b,3,3), may be NULL: the
 * rotation matrices ortho9d2matrix (dcl_ortho9d_to_matrix) makes of o9, formed by the second launch itself.              */
/* The confidence regressor (models/DCL_Net.py:115-126, 217-218: Head_MultiLayerPerceptron [128, 128, 128, 1], ReLU, ReLU, none)
 * on point rows, one launch: out[m] = w3 . relu(W2t^T relu(W1t^T x[m] + b1) + b2) + b3.  x (M rows, pitch ldx >= 128, 16-B
 * aligned), W1t / W2t (128 x 128, row = input channel, dense), w3 element k at w3[k * ldw3], b3 one float, out (M).  fp32 MFMA. */
int dcl_mlp128_to1(const float *x, int64_t ldx, int M, const float *W1t, const float *b1, const float *W2t,
                   const float *b2, const float *w3, int64_t ldw3, const float *b3, float *out, dclStream_t stream);
int dcl_pose_heads(int b, const float *pooled, const float *const *rot_layers, const float *const *trans_layers,
                   float *h1_scratch, float *o9, float *trans, float *R, dclStream_t stream);
/* The same heads fed with the confidence pooling's slice partials instead of the finished pooled feature (dcl_conf_pool's part1 /
 * part2 / wsum + the fusers' trailing BatchNorm affines, as dcl_pool_finish takes them): the finish is folded into the first
 * launch -- every workgroup of a crop forms the 1024 inputs itself, same operations, same bits.  For a handful of crops. */
int dcl_pose_heads_parts(int b, int nslices, const float *part1, const float *part2, const float *wsum,
                         const float *scale1, const float *shift1, const float *scale2, const float *shift2,
                         const float *const *rot_layers, const float *const *trans_layers, float *h1_scratch, float *o9,
                         float *trans, float *R, dclStream_t stream);

/* One per-point linear layer -- Conv1d(k=1) / 1x1x1 Conv3d with its BatchNorm folded in (models/Modules.py:58-97, 173-201;
 * the reference runs them as cuDNN pointwise convolutions): y[M x N] = act(x[M x K] Wt[K x N] + bias[N]), row-major, every
 * matrix with its own row pitch (ldx >= K, ldw >= N, ldy >= N floats) so that x, Wt and y can be column blocks of wider
 * buffers.  A library GEMM (hipBLASLt, fp32, bias / ReLU epilogue); bias may be NULL, relu 0/1.  workspace: device scratch
 * the caller may pass (kept for ABI stability) -- NEVER used: only algorithms that ask for no workspace are taken, and the call
 * fails with DCL_EINVAL when the library has none for the shape (two workspace-exchanging stream-K kernels side by side on
 * two streams hang the GPU; csrc/linear.cpp).                                                                            */
int dcl_linear_fwd(const float *x, int64_t ldx, const float *Wt, int64_t ldw, const float *bias, float *y, int64_t ldy,
                   int M, int N, int K, int relu, void *workspace, int64_t workspace_bytes, dclStream_t stream);

/* The same layer on the library's OWN fp32 MFMA GEMM core (csrc/linear_dma.hip: 128 x 128 / 128 x 64 / 64 x 64 tiles, LDS-DMA
 * operand rings, v_mfma_f32_32x32x2_f32; no vendor library, no workspace).  Same argument meaning as dcl_linear_fwd; needs
 * K % 32 == 0, 16-byte aligned x / Wt and ldx % 4 == ldw % 4 == 0, a row of Wt holding N rounded up to 4 floats (DCL_EINVAL
 * otherwise: such layers stay on dcl_linear_fwd).  Per output element the sum is ONE fmaf chain over k.                  */
int dcl_linear_dma_fwd(const float *x, int64_t ldx, const float *Wt, int64_t ldw, const float *bias, float *y, int64_t ldy,
                       int M, int N, int K, int relu, dclStream_t stream);
/* The last fuser layer WITH the confidence-weighted pooling of models/DCL_Net.py:223-228 as its epilogue: instead of storing
 * F = act(x Wt + bias) (M x N) it leaves  part[t][c] = sum over the rows j of row tile t (128 rows) of w_j * F[j][c]  for
 * t = 0 .. ceil(M / 128) - 1 (row pitch ldp >= N), with w_j = roww[(j / rows_per_crop) * w_stride + j % rows_per_crop] (the
 * softmax weights as dcl_conf_softmax leaves them: one row of n1 + n2 weights per crop, a direction's block addressed through
 * the base pointer).  With every crop a whole number of tiles, dcl_pool_finish2 adds a crop's partials in tile order.
 * Same shape constraints as dcl_linear_dma_fwd. */
int dcl_linear_pool_fwd(const float *x, int64_t ldx, const float *Wt, int64_t ldw, const float *bias, const float *roww,
                        int rows_per_crop, int64_t w_stride, float *part, int64_t ldp, int M, int N, int K, int relu,
                        dclStream_t stream);
/* The last TWO layers of a head whose last layer has one output (the confidence regressor, models/DCL_Net.py:115-126: 128 ->
 * 128 -> 1) on the own GEMM core: out[m] = w3 . relu(x[m] Wt + bias) + b3 with the N <= 128 hidden columns never stored (the
 * row dot is the GEMM's epilogue).  w3: N floats with stride ldw3 (a (N, 1) weight with padded rows); b3: one float. */
int dcl_linear_rowdot_fwd(const float *x, int64_t ldx, const float *Wt, int64_t ldw, const float *bias, const float *w3,
                          int64_t ldw3, const float *b3, float *out, int M, int N, int K, dclStream_t stream);
/* The same layers on the bf16 matrix pipe with fp32 results (csrc/linear_split.hip): every fp32 operand is the exact sum of
 * three bf16 pieces and a product is accumulated as its six piece products of weight >= 2^-16, in fp32 accumulators (what is dropped
 * is below the rounding of an fp32 FMA chain).  A layer's weight is prepared ONCE: dcl_linear_split_weight writes its pieces, in the
 * kernel's tile order, into `planes` (dcl_linear_split_weight_bytes(K, N) bytes, 16-byte aligned; 0 = K is no multiple of 16).
 * dcl_linear_split_fwd / _pool_fwd / _rowdot_fwd then mirror dcl_linear_dma_fwd / dcl_linear_pool_fwd / dcl_linear_rowdot_fwd with
 * `planes` in the place of (Wt, ldw); x 16-byte aligned, ldx % 4 == 0, K % 16 == 0; workgroup tiles of 256 rows x 128 columns: for
 * launches of at least a few hundred tiles (the fp32-MFMA core keeps the rest). */
int64_t dcl_linear_split_weight_bytes(int K, int N);
int dcl_linear_split_weight(const float *Wt, int64_t ldw, int K, int N, void *planes, dclStream_t stream);
int dcl_linear_split_fwd(const float *x, int64_t ldx, const void *planes, const float *bias, float *y, int64_t ldy, int M, int N,
                         int K, int relu, dclStream_t stream);
int dcl_linear_split_pool_fwd(const float *x, int64_t ldx, const void *planes, const float *bias, const float *roww,
                              int rows_per_crop, int64_t w_stride, float *part, int64_t ldp, int M, int N, int K, int relu,
                              dclStream_t stream);
/* y = act(x Wt + bias) written AS the attention's V pieces (N <= 320 channels from channel 0, rows = keys of crop row /
 * rows_per_crop, rows_per_crop % 256 == 0 = M / crops): `vplanes` = the `planes` scratch of the dcl_cross_attention_ws3 call that
 * consumes it with V1 = NULL. */
int dcl_linear_split_vpieces_fwd(const float *x, int64_t ldx, const void *planes, const float *bias, void *vplanes,
                                 int rows_per_crop, int M, int N, int K, int relu, dclStream_t stream);
int dcl_linear_split_rowdot_fwd(const float *x, int64_t ldx, const void *planes, const float *bias, const float *w3, int64_t ldw3,
                                const float *b3, float *out, int M, int N, int K, dclStream_t stream);
/* The softmax half of dcl_conf_pool alone: conf (b, n1 + n2) = sigmoid(cat[logit1, logit2]), w (b, n1 + n2) = softmax(conf) per
 * crop, wsum (b, 2) = the weight sums of the two directions (models/DCL_Net.py:217-222). */
int dcl_conf_softmax(int b, int n1, int n2, const float *logit1, const float *logit2, float *conf, float *w, float *wsum,
                     dclStream_t stream);
/* dcl_pool_finish with a slice count per direction: part1 (b, nslices1, c), part2 (b, nslices2, c). */
int dcl_pool_finish2(int b, int c, int nslices1, int nslices2, const float *part1, const float *part2, const float *wsum,
                     const float *scale1, const float *shift1, const float *scale2, const float *shift2, float *out,
                     dclStream_t stream);

/* Several INDEPENDENT per-point linear layers in one launch (csrc/linear_group.hip), for calls of a handful of crops: the
 * reference issues every Conv1d(k=1) / 1x1x1 Conv3d of its MLP stacks as its own launch (models/Modules.py:58-97,173-201; the
 * four disengage second layers of a side, models/DCL_Net.py:188-200, and a regressor_conf layer beside the neck_fuser layer
 * of the same depth, :207-216, do not depend on each other).  Each job is dcl_linear_fwd's problem: y[M x N] = act(x[M x K]
 * Wt[K x N] + bias[N]), row-major with free row pitches; K % 32 == 0, ldx % 4 == 0, ldw % 4 == 0 and ldw >= N rounded up to 4
 * (a layer with N % 4 != 0 passes a zero-padded Wt), x and Wt 16-byte aligned.  Own fp32 MFMA kernel (64 x 64 tiles), the sum
 * of an element runs over k ascending.  jobs: HOST array of njobs <= DCL_LINEAR_MAX_JOBS descriptors holding DEVICE pointers. */
#define DCL_LINEAR_MAX_JOBS 8
typedef struct DclLinearJob {
  const float *x; int64_t ldx;
  const float *Wt; int64_t ldw;
  const float *bias;          /* may be NULL */
  float *y; int64_t ldy;
  int M, N, K, relu;
} DclLinearJob;
int dcl_linear_group_fwd(const DclLinearJob *jobs, int njobs, dclStream_t stream);

/* ortho9d2matrix (models/DCL_Net.py:15-36): o9 (b,9) -> R (b,3,3).             */
int dcl_ortho9d_to_matrix(int b, const float *o9, float *R, dclStream_t stream);

/* ------------------------------------------------------------ training-side kernels (csrc/backward.hip) ---
 * Transposed rulebook: inv[k][i] = o for every nbr[k][o] = i >= 0, -1 elsewhere (inv: i32[kvol][cap_in]).  With it the
 * input gradient of indice_conv (spconv_ops.h:351-438) is dcl_sparse_conv_fwd(dOut, inv, W^T per offset).            */
int dcl_rulebook_transpose(const int32_t *nbr, int cap_out, const int32_t *n_out_dev, int n_out_host, int kvol,
                           int32_t *inv, int cap_in, dclStream_t stream);
/* Filter gradient of indice_conv: dW[k] = sum_o feat[nbr[k][o]]^T dout[o] (spconv_ops.h:413-417).  partial: scratch of
 * splits*kvol*cin*cout floats with splits = dcl_sparse_conv_wgrad_splits(n_out); partial sums are added in split order. */
int dcl_sparse_conv_wgrad_splits(int n_out, int32_t *splits_host);
int dcl_sparse_conv_wgrad(const float *feat, const int32_t *nbr, int cap, int n_out, const float *dout, int cin, int cout,
                          int kvol, float *partial, float *dW, dclStream_t stream);
/* indice_avgpool backward (avgpool.cu:178-206): din[i] = sum_k asc dout[inv[k][i]] / (float)rf[inv[k][i]].  C % 4 == 0. */
int dcl_sparse_avgpool_bwd(const float *dout, const int32_t *inv, int cap_in, int n_in, const int32_t *rf, int c, int kvol,
                           float *din, dclStream_t stream);
/* three_interpolate_grad of libs/pointnet_sp (interpolate_gpu.cu:124-148): atomic scatter into a ZEROED (m,c) buffer.  */
int dcl_three_interpolate_grad_sp(int c, int n, int m, const float *grad_out, const int32_t *idx, const float *weight,
                                  float *grad_points_zeroed, dclStream_t stream);
/* voxelize_bp (voxelize.cu:35-50): d_feats[rules[v][1+i]] += (average ? 1/n_v : 1) * d_out[v]; d_feats ZEROED (N,C).   */
int dcl_voxelize_bp(const float *d_out, const int32_t *rules, float *d_feats_zeroed, int n_rows, int max_active, int c,
                    int average, dclStream_t stream);

/* ------------------------------------------------------------ crop builder ---
 * The per-object crop construction of the data loaders (YCBV/dataloader_test_YCBV.py:124-183), on the device, in the
 * reference's pixel order and float32/float64 arithmetic (results bit-identical to the numpy/torch code).
 *
 * dcl_crop_points (three launches: one workgroup per 4096-pixel chunk of a box -- one workgroup per instance for the
 * sequential centroid sum -- one workgroup per 4096-row chunk): pixels of box [rmin,rmax) x [cmin,cmax) with label == obj_ids[i] and
 * depth != 0, in ascending flat order (:128-133); back-projection pt2 = d/scale, pt0 = (col-cx)*pt2/fx,
 * pt1 = (row-cy)*pt2/fy (:147-154); rgb = float(double(float(v)/255) - mean) (:143-145); centroid = row-order running
 * float32 sum / n (np.mean(axis=0), :156); points centred; those with |x|,|y|,|z| < half_extent kept when more than
 * min_valid (32) of them exist, else all (:160-165).
 *   depth (H,W) u16, label (H,W) i32, rgb (H,W,rgb_channels) u8 -- device; boxes (n,4) i32 rmin,rmax,cmin,cmax (clipped
 *   to the image) and obj_ids (n) i32 -- device; cam_host = {cx, cy, fx, fy, scale, post_div}: the cloud is divided by
 *   post_div after back-projection (1000 for LineMOD, LM/dataloader_test_LM.py:156-160; 1 for YCB-V); always_filter: apply
 *   the grid filter whatever the count (LM eval mode, :197); cap >= every box area.
 *   raw_xyz/raw_rgb: scratch (n,cap,3); out_xyz/out_rgb (n,cap,3); centroid (n,3);
 *   counts (n,3) = {masked pixels, points inside the grid, rows written} (all zero: the reference skips the instance);
 *   ws: dcl_crop_points_ws_ints(n, cap) int32 of device scratch (chunk counts and offsets; zeroed by the call). */
int dcl_crop_points_ws_ints(int n_inst, int cap, int64_t *ints_host);
int dcl_crop_points(const uint16_t *depth, const int32_t *label, const uint8_t *rgb, int H, int W, int rgb_channels,
                    int n_inst, const int32_t *boxes, const int32_t *obj_ids, const float *cam_host,
                    const double *rgb_mean_host, const float *half_extent_host, int min_valid, int always_filter, int cap,
                    float *raw_xyz, float *raw_rgb, float *out_xyz, float *out_rgb, float *centroid,
                    int32_t *counts, int32_t *ws, dclStream_t stream);
/* Sampled points -> feats rows [1,r,g,b,x,y,z] (n*npoint,7) and voxelize_idx input rows [instance,ix,iy,iz] (n*npoint,4)
 * i64 (:166-176,186-190): voxel = trunc((xyz + half_extent0)/unit) in float32, clamped to [0,voxel_limit-1] first for
 * instances with counts[i][1] <= min_valid.  sample_idx (n,npoint) i64 = the caller's np.random.choice draws (NULL:
 * identity, for the template clouds :179-182; then counts may be NULL).  xyz/rgb (n,cap,3).                          */
int dcl_crop_sample(int n_inst, int npoint, int cap, const float *xyz, const float *rgb, const int64_t *sample_idx,
                    const int32_t *counts, int min_valid, float half_extent0, const float *unit_host, int voxel_limit,
                    float *feats, int64_t *coords, dclStream_t stream);

/* ------------------------------------------------------------ eval metric ---
 * ADD-S per object (tools/test_YCBV_stage1.py:186-189): out[o] = mean_i min_j |R_pred x_i + t_pred - (R_gt x_j + t_gt)|
 * over the P points of the object's class cloud.  cld (n_clouds, P, 3); cls i32[b] selects the cloud of object o
 * (NULL: cloud o).  partial_scratch: b * ceil(P/256) floats.  No (b,P,P,3) intermediate.                           */
int dcl_add_s(int b, int P, const float *cld, const int32_t *cls, const float *R_pred, const float *t_pred,
              const float *R_gt, const float *t_gt, float *partial_scratch, float *out, dclStream_t stream);

/* LineMOD metric (tools/test_LM.py:123-135).  dcl_add: ADD = mean_i |R_pred x_i + t_pred - (R_gt x_i + t_gt)| (the
 * reference's `l2_dis`, non-symmetric objects).  dcl_add_by_symmetry: per object, sym_flag[o] == 0 -> ADD, != 0 -> ADD-S
 * (`cd_dis`), one launch.  Same buffers as dcl_add_s.                                                                    */
int dcl_add(int b, int P, const float *cld, const int32_t *cls, const float *R_pred, const float *t_pred,
            const float *R_gt, const float *t_gt, float *partial_scratch, float *out, dclStream_t stream);
int dcl_add_by_symmetry(int b, int P, const float *cld, const int32_t *cls, const int32_t *sym_flag,
                        const float *R_pred, const float *t_pred, const float *R_gt, const float *t_gt,
                        float *partial_scratch, float *out, dclStream_t stream);

/* The loaders' point-sampling draws, bit for bit (host code, no GPU call): for each of k objects out[o*n .. o*n+n) =
 * np.random.permutation(m[o])[:n] = what np.random.choice(m[o], n, replace=False) returns (YCBV/dataloader_test_YCBV.py:166-169,
 * LM/dataloader_test_LM.py:176-181) -- numpy's legacy Fisher-Yates walk (mtrand.pyx: _shuffle_raw, random_interval: MT19937
 * words masked and rejected) on the CALLER's generator state: key624 = np.random.get_state()[1] (updated in place), *pos_io =
 * [2] (in / out); put them back with np.random.set_state() and the global stream has advanced exactly as under numpy.
 * Requires n <= m[o] < 2^31; scratch = max(m) int32.  ~3x numpy's speed (a tight 32-bit loop instead of 3 memcpy per swap). */
int dcl_legacy_permutation_heads(uint32_t *key624, int32_t *pos_io, const int32_t *m, int k, int n, int64_t *out,
                                 int32_t *scratch);

/* The ONE measurement facility of the product library (it selects nothing and changes no result; every tuning / A-B switch
 * lives in the diagnostic library below): between _begin and _end every sparse-conv call (runner or op API) is bracketed by
 * HIP events on its launch stream; _end waits for them and returns the summed device milliseconds and the number of calls.
 * bench.py's `roofline_sparse_conv` is measured with it on the product library itself.  Thread-safe (a mutex around the
 * event list); do not use under stream capture.                                                                          */
int dcl_profile_conv_begin(void);
int dcl_profile_conv_end(double *ms_total_host, int32_t *calls_host);
/* The same, plus the calls one by one in call order: ms_per_call_host[i] = device milliseconds of call i and
 * what_per_call_host[4 i ..] = its (Cin, Cout, subm, problems in the launch), for the first `cap` calls (either array may be NULL).
 * bench.py's `roofline_sparse_conv.layers[]` is taken with it. */
int dcl_profile_conv_end_calls(double *ms_total_host, int32_t *calls_host, float *ms_per_call_host, int32_t *what_per_call_host,
                               int32_t cap);

/* ---- DIAGNOSTIC library only (csrc/Makefile `make diag` -> tests/_diag/libdclnet_hip_diag.so, built with -DDCL_DIAG) ----
 * The dcl_debug_* entry points are TEST / TUNING hooks, not part of the operator API and NOT exported by the product library
 * libdclnet_hip.so (where every switch is a compile-time constant and the superseded kernel variants are not compiled):
 * each sets one process-wide atomic switch that later calls from any thread read (A/B kernel variants in tests, tuning
 * sweeps in tools/).  They never change results beyond the tolerance of the op.                                          */
#ifdef DCL_DIAG
/* Test hook, sparse-conv kernel variant: 0 = automatic (LDS-DMA implicit GEMM where Cout % 64 == 0), 1 = plain VALU
 * kernel for every layer (A/B check of the MFMA ones), 2 = MFMA without LDS staging (the general fallback),
 * 4 = register-staged tile kernel instead of the LDS-DMA one, 5 = 4-wave 128x64 tiles for Cout = 64 (the default has 8 waves). */
void dcl_debug_force_valu_conv(int on);
/* Test hook, attention kernel: 0 = automatic, 1 = shared-tile 8-wave variant, 2 = double-buffered 4-wave register
 * staging, 3 / 4 = LDS-DMA pipeline with 8 / 4 waves. */
void dcl_debug_attention_variant(int v);
/* Tuning hook: 0 = automatic key split of small attention launches (dcl_cross_attention_ws), n = force n splits. */
void dcl_debug_attention_split(int n);
/* Tuning hook: 1 (default) = the LDS-DMA attention kernel renumbers its workgroups so that the query blocks of one crop share
 * an XCD (one L2 fetch of the crop's K/V per XCD group), 0 = plain blockIdx order (traffic A/B). */
void dcl_debug_attention_xcd_remap(int on);
/* Test hook, geometry stage: 1 (default) = the 8 occupancy masks of a pass come from one launch (one workgroup per crop walks
 * the conv/pool chain in LDS; 64^3 grids), 0 = 8 chained launches.  Both produce identical masks.  Process-wide atomic. */
int dcl_debug_geometry_chain(int mode);
/* Tuning hook: smallest batch whose deep conv layers get a row order (default 12); a huge value switches the ordering off. */
void dcl_debug_order_min_batch(int n);
/* Tuning hook: first backbone level (0..3) whose conv layers get a row order (default 2: the two deep levels). */
void dcl_debug_order_min_level(int m);
/* Test hook, 3-NN of the point read-out: 1 (default) = grid-pruned search on the 32^3 / 16^3 levels, 0 = per-crop scan on
 * every level, 2 = grid kernel with its scan fallback forced for every query, 3 / 4 / 5 = grid kernel with one / four /
 * eight lanes per query whatever the number of points (automatic: eight up to 40960 points, four up to 131072, else
 * one).  All give identical results. */
void dcl_debug_three_nn_grid(int mode);
/* Test hook, batched three_nn / knn (k = 1): 0 (default) = bucketed exact search, 1 = the plain scan (A/B). */
void dcl_debug_nn_batched_mode(int mode);
/* Tuning hook: queries per thread of the bucketed batched search (1..4; 0 = automatic). */
void dcl_debug_nn_qpt(int q);
/* Diagnostic: (query, point) distance evaluations the bucketed three_nn / knn searches have executed since the last reset
 * (synchronises the device; reset != 0 zeroes the counter) -- what bench.py prices those kernels against. */
unsigned long long dcl_debug_nn_tests_executed(int reset);
/* Tuning hook: 0 = automatic split-K choice in dcl_sparse_conv_fwd_ws, n = force n splits (when the scratch allows),
 * -1 = at most 8 splits even for few-row launches, -2 = never split. */
void dcl_debug_conv_split(int n);
/* Tuning hook: least number of chunks a workgroup of a few-row conv launch walks (default 4). */
void dcl_debug_conv_few_chunks(int n);
/* Tuning hook: 1 (default) = few-row conv launches use 64-row tiles, 0 = 128-row tiles for every launch. */
void dcl_debug_conv_few_tiles(int on);
/* Tuning hook: 1 (default) = the Cin 16 / 32 -> 32 conv layers of many rows run the filter-resident kernel, 0 = LDS-DMA kernel. */
void dcl_debug_conv_wlds(int on);
/* Tuning hooks of the own GEMM core: tile shape (0 = automatic, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 / 5 = 64x64 with
 * the two halves / four quarters of K on two / four wave groups), XCD-aware
 * workgroup renumbering (default 1). */
void dcl_debug_linear_tile(int t);
void dcl_debug_linear_xcd_remap(int on);
void dcl_debug_linear_persist(int rounds);   /* rounds of resident workgroups from which a GEMM launch is persistent (default: never) */
/* Diagnostic: times the GEMM library's first ncand heuristic candidates (32 MiB of workspace on offer) for an (M, N, K)
 * linear layer, each alone on the GPU; ms_out[i] = mean ms, ws_out[i] (may be NULL) = the workspace candidate i asks for.
 * dcl_linear_fwd itself only ever takes an algorithm that asks for none (it queries with a maximum of 0 bytes and refuses
 * the call when nothing is left), so candidates with ws_out[i] > 0 are the ones it can never run. */
int dcl_debug_linear_candidates(int M, int N, int K, int ncand, float *ms_out, long long *ws_out, int *found_out);
/* Diagnostic: bytes of workspace the algorithm dcl_linear_fwd takes for (M, N, K) asks for: 0, or -1 = the library has no
 * zero-workspace algorithm for the shape (dcl_linear_fwd returns DCL_EINVAL then -- never a workspace-exchanging kernel). */
long long dcl_debug_linear_plan_workspace(int M, int N, int K);
/* Tuning hook: 1 (default) = a pair of attention launches (dcl_cross_attention_ws2, concurrent = 2) that makes whole rounds of
 * 8-wave workgroups plus a rest is issued as two launches (rounds, rest); 0 = one launch. */
void dcl_debug_attention_pair_split(int on);
/* Tuning hook: 0 = large attention calls keep the fp32-MFMA kernel even when `planes` are handed in (dcl_cross_attention_ws3);
 * dcl_cross_attention_planes_bytes then returns 0.  1 (default) = P.V on the bf16 matrix pipe (k_cross_attn_split). */
void dcl_debug_attention_bf16(int on);
/* Diagnostic: 0 = plain tile numbering in k_linear_split (1 = XCD-aware, default); what-if runs of k_linear_split (WRONG results;
 * timing only) -- bit 0: no LDS-DMA after a tile's first chunk, bit 1: no operand split, bit 2: no store epilogue. */
void dcl_debug_linear_split_xcd_remap(int on);
void dcl_debug_linear_split_whatif(int bits);
/* Diagnostic: what-if runs of k_cross_attn_split (WRONG results; timing only) -- bit 0: no P.V phase, bit 1: no S / softmax phase,
 * bit 2: no LDS-DMA after the first tile. */
void dcl_debug_attention_whatif(int bits);
/* Tuning hook: most crops of a pass whose geometry stage runs as one launch (k_geometry_small; default 16, at most 64;
 * passes of more than 8 crops also need at most 32768 voxel rows). */
void dcl_debug_geometry_small_batch(int n);
int dcl_debug_geometry_small_stamps(unsigned long long *host32);   /* s_memrealtime (100 MHz) at the phase boundaries of workgroup 0 of the last k_geometry_small (0..9), after each mask-chain stage (16..23) */
/* Diagnostic: a one-thread launch that writes the 100 MHz wall clock into *slot_dev (a time stamp inside a stream or a
 * captured graph: tools/graph_timeline.py). */
int dcl_debug_stamp(unsigned long long *slot_dev, dclStream_t stream);
/* Tuning hook: row CAPACITY up to which a capacity-mode conv launch (whole-forward graph) counts as a few-row launch. */
void dcl_debug_conv_few_cap(int rows);
/* Tuning hook: EXPECTED rows (the backbone runner's hint) up to which a capacity-mode conv launch counts as few-row. */
void dcl_debug_conv_few_hint(int rows);
/* Tuning hook: number of workgroups the stream-K / split-K decompositions of a sparse-conv launch are dealt over (default
 * 512 = the 2 x 256 resident slots; 256 leaves one slot per CU to a concurrent launch of the other backbone).  64..512. */
void dcl_debug_conv_slots(int n);
/* Test hook, row order of the LDS-DMA conv launches (dcl_sparse_conv_fwd_ordered, the backbone runner): 0 = as given,
 * 1 = ignore the order (natural rows, nominal units), 2 = keep the order but deal nominal chunk units. */
void dcl_debug_conv_order_mode(int mode);
/* Diagnostic: s_memrealtime stamps (100 MHz) of the phases of the last row-order launch's first workgroup. */
int dcl_debug_order_stamps(unsigned long long *host16);
/* Tuning hook: 1 (default) = the LDS-DMA conv kernel renumbers its workgroups XCD-aware (column tiles of a row tile and
 * neighbouring row tiles share an L2), 0 = plain blockIdx order. */
void dcl_debug_conv_xcd_remap(int on);
/* Tuning hook for dcl_group_points' LDS-staged kernel: channel rows per workgroup, x-blocks, threads per workgroup,
 * store kind (2 = plain instead of nontemporal); 0 = built-in choice for each. */
void dcl_debug_group_points_cfg(int cc, int xb, int threads, int nontemporal);
/* Launch census: every kernel launch of the diagnostic library is counted by kernel (demangled name incl. template arguments).
 * _census writes "name<TAB>launches<NEWLINE>" lines into buf (NUL-terminated, cut at cap) and returns the bytes the whole text
 * needs; _reset empties the table.  tests/test_kernel_census.py: every kernel a committed profile names must have been
 * launched by a test that compares with the oracle. */
void dcl_debug_launch_census_reset(void);
long long dcl_debug_launch_census(char *buf, long long cap);
#endif /* DCL_DIAG */

#ifdef __cplusplus
}
#endif
#endif /* DCLNET_HIP_H_ */

"""DCL-Net stage-1 network: drop-in for the reference's models/DCL_Net.py.

Same `Network(cfg, mode)` constructor, `forward(data) -> dict` contract (reference
models/DCL_Net.py:155-259, return dict :238-255, side effect on data["labels"] :257-258) and the same
270-key state_dict, so `tools/test_YCBV_stage1.py` can import this module instead.

Two execution paths over the same parameters:
  fused=True  (default) -- the MI355X pipeline:
      * geometry pass: all 8 sparse-layer active sets of BOTH backbones from bitmask rulebooks, then ONE
        host read-back of the 16 row counts (the reference does 32 blocking indiceNum.to(CPU));
      * feature pass: voxel mean-pool -> [sparse conv + folded BN + ReLU] x2 -> sparse avg-pool, one
        launch per layer; 3-NN search restricted to the point's own crop; interpolation written
        straight into the 480-channel point-major buffer;
      * dense pass on POINT-major activations (b*n, C): the 1x1x1-conv/BN/ReLU stacks become 2-D GEMMs
        with fused bias+ReLU epilogues (4 stacks' first layers share one GEMM; BatchNorms folded),
        the correspondence attention is the fused MFMA kernel (no (b,M,N) map in memory),
        confidence pooling and the 9-D -> SO(3) projection are single kernels.
  fused=False -- composes the mirrored modules exactly like the reference graph (materialised attention
      map, per-layer rulebooks); kept as an executable specification and cross-check.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .. import spconv
from ..libs.pointgroup_ops.functions import pointgroup_ops
from .losses import losses  # noqa: F401  (reference: models/DCL_Net.py::losses)
from .Modules import (Aligner, Backbone_SPCONV, BasicBlock_3DCONV, Head_MultiLayerPerceptron,
                      Ops_GetPointFeat_spconv)

HOST_TIMES = None                   # tools/host_timeline.py sets this to a list: (label, perf_counter) marks of _forward_fused
_QUEUE_SHIMS = []            # streams made only to shift the runtime's round-robin deal of streams onto hardware queues (_select_capture)
SIDE_STREAM_PRIORITY = -1    # the side streams carry the sparse half -- chains of short, latency-bound launches -- and get dispatch
                            # priority over whatever GEMM / attention launch they meet (the previous call's tail under
                            # async_inputs, the other direction's); read when a side stream is first made (0: A/B)
_SIDE_STREAMS = {}          # (device index, which) -> torch.cuda.Stream, shared by every Network of the process
SCALE_LISTS = [2, 4, 6, 8]          # sic -- reference models/DCL_Net.py:54 (true strides are 2,4,8,16)
VOXEL_NUM_LIMIT = [64, 64, 64]


def normalize_vector(v):
    """utils/transform3D.py:6-30 (torch branch): v / (|v| + 1e-8)."""
    return v / (torch.sqrt(v.pow(2).sum(1, keepdim=True)) + 1e-8)


def ortho9d2matrix(x_raw, y_raw, z_raw):
    """Rotation from three raw axes (reference models/DCL_Net.py:15-36): on-device 3x3 SVD kernel.  When a gradient is
    needed (training) the same projection is composed from differentiable torch ops -- normalised axes as columns,
    U diag(1, 1, det(U V^T)) V^T -- with the batched 3x3 SVD on the host (b tiny matrices)."""
    if not (torch.is_grad_enabled() and (x_raw.requires_grad or y_raw.requires_grad or z_raw.requires_grad)):
        return ops.ortho9d_to_matrix(torch.cat([x_raw, y_raw, z_raw], dim=1))
    dev = x_raw.device
    m = torch.stack([normalize_vector(x_raw), normalize_vector(y_raw), normalize_vector(z_raw)], dim=2).cpu()
    U, _, Vh = torch.linalg.svd(m)
    ones = torch.ones(m.shape[0], dtype=m.dtype)
    sigma = torch.diag_embed(torch.stack([ones, ones, torch.det(U @ Vh)], dim=1))
    return (U @ sigma @ Vh).to(dev)


def _mlp3(dims):
    return Head_MultiLayerPerceptron(dims, ["relu", "relu", "none"], [False] * 3, [0.0] * 3)


def _fuser():
    return Head_MultiLayerPerceptron([512, 512, 512, 1024], ["relu"] * 3, [True] * 3, [0.0] * 3)


class Network(nn.Module):
    def __init__(self, cfg, mode="train", fused=True, graph_max_batch=8, async_inputs=False, graph_max_points=98304,
                 single_stream=False, pipeline_chunks=1, capture_graph=True, pair_features=None):
        """graph_max_batch > 0 (default 8): eval-mode calls with at most that many crops go through forward_graphed (one
        whole-forward hipGraph per batch size, captured on first use) -- the one-image-at-a-time eval loops of the
        reference (tools/test_LM.py:104-112: one object per call) are launch-bound otherwise: 0.53 instead of 1.1 ms
        per one-crop call.  Larger batches replay a graph too while the call is small in points, b * (N + M) <=
        graph_max_points (measured, tools/graph_crossover.py: N = M = 1024 wins 3-9 % at every batch size tried, up to 96
        crops; N = 12288 / M = 2048 loses 1-2 % from 8 crops on) -- async_inputs instances too: a replay has no cross-call overlap, but
        since the split-bf16 kernels a replayed reference-shape call (2.95-3.04 ms) beats the pipelined launch-by-launch one (3.06-3.16).
        graph_max_batch = 0 switches all of it off (every call launch by launch).
        async_inputs=True: the caller guarantees that `data`'s CUDA tensors are complete when forward() is called (or hands
        over data["ready_event"]) and are not overwritten until the results have been consumed.  The sparse half of a call
        (side streams) then does not wait for the dense half of the previous call still running on the current stream, so
        back-to-back calls pipeline: backbones of batch k+1 underneath the GEMMs / attention of batch k.
        single_stream=True: the whole call on the caller's stream (no side streams; what bench.py's per-kernel conv timing
        uses).  pipeline_chunks=K > 1: the sparse half in K passes over b/K crops (measured slower, kept runnable).
        capture_graph=False: forward_graphed runs its capacity-mode body launch by launch (debugging aid).  These are
        pair_features: True = the feature stage of both backbones as ONE launch sequence (every layer one launch over both
        sides' tiles, ops.backbone_features_pair: 8 conv + 4 pool launches per forward instead of 16 + 8), False = each side's
        own launches on its own stream, None (default) = automatic: paired when everything runs on one stream anyway
        (single_stream: measured 1.63 -> 1.53 ms of conv time per bs-32 forward), separate otherwise -- with two side streams
        the sides' stages overlap each other and the first dense GEMMs, which a common feature stage would serialise (measured
        at N = M = 1024: bs 32 4.28 vs 4.40 ms per step, one crop 0.61 vs 0.65 ms).
        These are constructor arguments on purpose: nothing on the call path reads the environment."""
        super().__init__()
        self.single_stream = bool(single_stream)
        self._pair_features = pair_features              # both backbones' layers as ONE launch each; None = automatic (see property)
        self.pipeline_chunks = int(pipeline_chunks)
        self.capture_graph = bool(capture_graph)
        self.graph_max_batch = int(graph_max_batch)
        self.graph_max_points = int(graph_max_points)
        self.async_inputs = bool(async_inputs)
        self.voxelization_mode = cfg.voxelization_mode
        self.unit_voxel_extent = np.array(cfg.unit_voxel_extent)
        self.mode = mode
        self.fused = fused
        self.n_inp = cfg.n_inp
        self.n_tmp = cfg.n_tmp
        self.backbone_dims = [7, 16, 32, 32, 64, 64, 128, 128, 256]
        self.backbone_stride_layers = [1, 3, 5]
        self.backbone_inp = Backbone_SPCONV(self.backbone_dims, self.backbone_stride_layers, cfg.backbone)
        self.backbone_tmp = Backbone_SPCONV(self.backbone_dims, self.backbone_stride_layers, cfg.backbone)
        self.stage1_get_point_feats = Ops_GetPointFeat_spconv(SCALE_LISTS, self.unit_voxel_extent, VOXEL_NUM_LIMIT)
        block = partial(BasicBlock_3DCONV, size=1, bias=False, stride=1, padding=0, norm=True, act="relu", drop=0.0)
        for grp in ("1", "2"):                       # registration order = the reference's (state_dict order)
            for side in ("Xc", "Yo"):
                for kind, d_out in (("p", 256), ("m", 64)):
                    setattr(self, "disengage_%s_%s%s" % (side, kind, grp),
                            nn.Sequential(block(dim_in=480, dim_out=256), block(dim_in=256, dim_out=d_out)))
        self.neck_cross_att = Aligner()
        self.regressor_Xo = _mlp3([256, 256, 128, 3])
        self.regressor_Yc = _mlp3([256, 256, 128, 3])
        self.regressor_conf = _mlp3([128, 128, 128, 1])
        self.regressor_conf_bi = _mlp3([128, 128, 128, 1])
        self.neck_fuser = _fuser()
        self.neck_fuser_bi = _fuser()
        self.regressor_rot = _mlp3([1024, 512, 128, 9])
        self.regressor_trans = _mlp3([1024, 512, 128, 3])
        self._folded = None

    @property
    def pair_features(self):
        return self.single_stream if self._pair_features is None else bool(self._pair_features)

    # ------------------------------------------------------------------ parameter folding (eval mode)
    @staticmethod
    def _drop_graph(ent):
        """release a captured forward NOW: the entry's `body` closure refers back to the entry, and a cycle is only freed by
        the cyclic collector at a moment of its choosing -- possibly in the middle of another capture, where destroying a
        hipGraph (and freeing its pool) aborts the process.  Breaking the cycle here frees it by reference count."""
        if ent is not None:
            if ent.get("graph") is not None and torch.cuda.is_available():
                torch.cuda.synchronize()                   # (evictions are rare: a replay of the graph may still be in flight)
            ent["body"] = None
            ent.clear()

    def _invalidate(self):
        """folded weights and captured graphs (which have the folded tensors' addresses baked in) follow the parameters"""
        self._folded = None
        for ent in (self.__dict__.pop("_graphs", None) or {}).values():
            self._drop_graph(ent)
        self.__dict__.pop("_graph_seen", None)
        self.__dict__.pop("_vlist", None)

    def train(self, mode=True):
        self._invalidate()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._invalidate()
        return super().load_state_dict(*a, **k)

    def __getstate__(self):
        """copy.deepcopy / pickling: streams, graphs and folded tensors are per-instance runtime state"""
        state = dict(self.__dict__)
        for k in ("_side", "_graphs", "_graph_seen", "_crop_id_cache", "_vlist", "_counts_host"):
            state.pop(k, None)
        state["_folded"] = None
        return state

    @staticmethod
    def _bn_affine(bn):
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        return s, bn.bias - bn.running_mean * s

    def _fold(self):
        """BatchNorm(eval) folded into scale/shift or neighbouring weights; cached until weights change -- also in place
        (the cache and the captured graphs are keyed on the parameters' version counters: optimizer.step() under frozen
        BatchNorm, an EMA swap through p.data.copy_ or manual surgery all drop them)."""
        from .refiner import _param_version
        ver = _param_version(self)
        if self._folded is not None and self.__dict__.get("_fold_version") != ver:
            self._invalidate()
        if self._folded is not None:
            return self._folded
        self.__dict__["_fold_version"] = ver
        f = {}
        with torch.no_grad():
            for bb in ("backbone_inp", "backbone_tmp"):
                layers = []
                for m in range(1, 5):
                    for blk in getattr(getattr(self, bb), "module%d" % m):
                        conv, bn = blk.layers[0], blk.layers[1]
                        s, t = self._bn_affine(bn)
                        W = conv.weight.reshape(-1, conv.in_channels, conv.out_channels).contiguous()
                        layers.append((W, s.contiguous(), t.contiguous(), conv.subm))
                f[bb] = layers
                f[bb + "_ptrs"] = (ops._ptr_array([l[0] for l in layers]), ops._ptr_array([l[1] for l in layers]),
                                   ops._ptr_array([l[2] for l in layers]))
            for side in ("Xc", "Yo"):
                W1, t1, second = [], [], []
                for tag in ("p1", "m1", "p2", "m2"):
                    seq = getattr(self, "disengage_%s_%s" % (side, tag))
                    c0, b0 = seq[0].layers[0], seq[0].layers[1]
                    s, t = self._bn_affine(b0)
                    W1.append(c0.weight.reshape(c0.out_channels, c0.in_channels) * s[:, None])
                    t1.append(t)
                    c1, b1 = seq[1].layers[0], seq[1].layers[1]
                    s, t = self._bn_affine(b1)
                    second.append(((c1.weight.reshape(c1.out_channels, c1.in_channels) * s[:, None]).t().contiguous(),
                                   t.contiguous()))
                f["dis_" + side] = (torch.cat(W1, 0).t().contiguous(), torch.cat(t1, 0).contiguous(), second)
            for name in ("regressor_conf", "regressor_conf_bi", "regressor_rot", "regressor_trans", "regressor_Xo",
                         "regressor_Yc"):
                L = getattr(self, name).layers
                # (a last layer of one / three / nine output columns sits in rows padded to 4 floats: what the 16-byte DMA
                # pieces of ops.linear_group want
                # and the own GEMM core's: every head that runs through _mlp keeps its last layer that way).  The two-launch
                # pose heads (ops.pose_heads) take dense (128, 9) / (128, 3) matrices; the padded twins serve batches beyond them
                dense = name in ("regressor_rot", "regressor_trans")
                pad = (lambda w: w.contiguous()) if dense else ops.pad_linear_weight
                f[name] = [(pad(L[i].weight[:, :, 0].t()), L[i].bias.contiguous()) for i in (0, 2, 4)]
                if dense:
                    f[name + "_padded"] = [(ops.pad_linear_weight(Wt), bv) for Wt, bv in f[name]]
            for name in ("neck_fuser", "neck_fuser_bi"):
                L = getattr(self, name).layers      # Conv,ReLU,BN, Conv,ReLU,BN, Conv,ReLU,BN
                out, s_prev, t_prev = [], None, None
                for ci, bi in ((0, 2), (3, 5), (6, 8)):
                    W, bvec = L[ci].weight[:, :, 0], L[ci].bias
                    if s_prev is not None:          # fold the previous layer's trailing BN into this conv
                        bvec = bvec + W @ t_prev
                        W = W * s_prev[None, :]
                    out.append((W.t().contiguous(), bvec.contiguous()))
                    s_prev, t_prev = self._bn_affine(L[bi])
                f[name] = (out, s_prev.contiguous(), t_prev.contiguous())   # last BN is applied after pooling
            # the dense layers' weights prepared for the split-bf16 GEMM core (ops.prepare_linear: three bf16 pieces per weight, in
            # the kernel's tile order): their launches of many crops run there, at fp32-sized errors (csrc/linear_split.hip)
            for side in ("Xc", "Yo"):
                W1t, _, second = f["dis_" + side]
                ops.prepare_linear(W1t)
                for Wt, _ in second:
                    ops.prepare_linear(Wt)
            for name in ("regressor_conf", "regressor_conf_bi", "regressor_Xo", "regressor_Yc"):
                for Wt, _ in f[name][:2]:
                    ops.prepare_linear(Wt)
            for name in ("neck_fuser", "neck_fuser_bi"):
                for Wt, _ in f[name][0]:
                    ops.prepare_linear(Wt)
        self._folded = f
        return f

    # ------------------------------------------------------------------ fused pipeline
    @staticmethod
    def _lin_relu(x, Wt, bias):
        # relu(x @ Wt + bias) on the own fp32 MFMA GEMM core (ops.linear -> csrc/linear_dma.hip); never torch._addmm_activation:
        # torch takes the vendor library's first-choice algorithm -- for row counts that do not tile the chip evenly (25, 33, ...
        # crops of 1024 points) a stream-K kernel that spins on flags in a workspace, and two of those side by side on the
        # forward's two branches hang the GPU
        return ops.linear(x, Wt, bias, True)

    def _mlp(self, x, layers):
        x = self._lin_relu(x, *layers[0])
        x = self._lin_relu(x, *layers[1])
        # last layer: bias, no activation (torch.addmm would first broadcast the bias into the output with an elementwise
        # kernel and then accumulate onto it)
        return ops.linear(x, layers[2][0], layers[2][1], False)

    def _side_stream(self, dev, which=0):
        """per-side streams of the sparse half (single_stream: everything on the current stream)"""
        if self.single_stream:
            return torch.cuda.current_stream(dev)
        # one set of side streams per device for the whole process: the runtime deals streams round-robin onto a few
        # hardware queues, so per-instance streams of a second Network can land on a queue its first one (or the other
        # side) already uses and the two backbones serialise (measured: +0.3 ms per reference-shape step for the
        # second instance)
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), which)
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(dev, priority=SIDE_STREAM_PRIORITY)
        return _SIDE_STREAMS[key]

    def _pipeline_chunks(self, b):
        """number of crop chunks of the sparse/disengage software pipeline.  Measured on MI355X (bs 32): K = 1/2/4 ->
        25.2/26.0/27.8 ms at N=12288 and 5.3/6.3/8.0 ms at N=1024 -- the sparse passes are latency-bound, so K passes over
        b/K crops cost almost K times one pass over b crops and the overlap cannot pay for that.  Default 1 (the two
        backbones still overlap each other and the first disengage GEMMs); Network(pipeline_chunks=k) keeps the experiment
        runnable."""
        k = self.pipeline_chunks
        while k > 1 and (b % k != 0 or b // k < 4):
            k -= 1
        return max(k, 1)

    def _forward_fused(self, data):
        import time as _t
        mark = (lambda lbl: HOST_TIMES.append((lbl, _t.perf_counter()))) if HOST_TIMES is not None else (lambda lbl: None)
        mark("start")
        f = self._fold()
        dev = self.regressor_rot.layers[0].weight.device
        if dev.type != "cuda":
            raise RuntimeError("dcl-net_amd.Network runs on the GPU only: call .cuda() first (no CPU fallback)")
        b = int(data["batch_offsets"].size(0)) - 1
        S = int(np.asarray(data["voxel_num_limit"]).astype(np.int64)[0])
        # Schedule.  (1) The two backbones are independent until the correspondence attention: each side has its own HIP
        # stream (input conversion, geometry, features, point read-out); the dense half runs on the caller's stream and
        # starts a side's disengage GEMMs as soon as that side is done.  (2) With async_inputs the side streams do not wait
        # for the caller's stream at all, so the sparse half of this call overlaps the dense half of the previous one.
        # (3) Optional chunking of the sparse half (pipeline_chunks, measured slower).  One host read-back of the level sizes,
        # made on a side stream.  single_stream=True switches all overlap off.
        main = torch.cuda.current_stream(dev)
        sstream = {"inp": self._side_stream(dev, 0), "tmp": self._side_stream(dev, 1)}
        single = sstream["inp"] is main
        K = self._pipeline_chunks(b)
        bc = b // K
        on_device = all(data[s][k].is_cuda for s in ("inp", "tmp") for k in ("feats", "v2p_maps", "occupied_voxels"))
        decoupled = self.async_inputs and on_device and not single
        for st in sstream.values():
            if not decoupled:
                st.wait_stream(main)
            elif data.get("ready_event") is not None:
                st.wait_event(data["ready_event"])
        # (4) What does not depend on the level sizes -- voxel means, the points' (crop, xyz) rows, output buffers -- is
        # issued before the read-back so that it runs underneath it; the read-back itself waits for the two geometry
        # stages only (on a helper stream).  Running the 3-NN searches of the read-out beside the convolutions (they need
        # the geometry only) and each level's interpolation as soon as that level is pooled was built and measured: the
        # same step time within 0.1 % -- these kernels fill the GPU, so overlap only re-divides it -- and dropped.
        npts = {"inp": self.n_inp, "tmp": self.n_tmp}
        side_in, runs, geo, vox, pb4, pf, pts = {}, {}, {}, {}, {}, {}, {}
        # The host is the slow party at the head of a call (a geometry stage is a dozen dependent small launches: ~50 us to
        # issue, done ~40 us later).  One side at a time -- geometry, then what the feature pass needs besides the level
        # sizes (issued while the geometry runs), then the sizes, then features + read-out -- puts the observed side's
        # convolutions on the GPU ~150 us after the call starts, and the template side's whole head is issued underneath
        # them.  The level sizes land in pinned host memory, written by the geometry kernels themselves (device-visible host
        # allocation): the host waits for that side's geometry event and reads them -- no copy kernel, no D2H enqueue.
        occ = {}
        counts_host = self._counts_pinned(K)
        def geometry(s):
            with torch.cuda.stream(sstream[s]):
                occ[s] = data[s]["occupied_voxels"].to(dev, non_blocking=True).int().contiguous()
                for c in range(K):
                    runs[s, c] = ops.BackboneRun(occ[s], bc, S, batch_lo=c * bc, counts_dev=counts_host[s][8 * c:8 * c + 8])
                geo[s] = torch.cuda.Event()
                geo[s].record(sstream[s])
        def stage(s):
            """what a side's feature pass needs besides the level sizes: voxel means, the points' (crop, xyz) rows, buffers"""
            with torch.cuda.stream(sstream[s]):
                d = data[s]
                side_in[s] = (d["feats"].to(dev, non_blocking=True).float().contiguous(),
                              d["v2p_maps"].to(dev, non_blocking=True).int().contiguous(), occ[s])
                pb4[s] = torch.cat([self._crop_ids(dev, b, bc, npts[s]), side_in[s][0][:, 4:7]], 1)   # (crop in its chunk, xyz)
                vox[s] = ops.voxelize_fp(side_in[s][0], side_in[s][1], self.voxelization_mode)   # all crops at once
                pf[s] = torch.empty((b * npts[s], 480), dtype=torch.float32, device=dev)   # read on `main`
            if not single:
                pf[s].record_stream(main)
                side_in[s][0].record_stream(main)                      # `pts` below is handed to the caller
            pts[s] = side_in[s][0][:, 4:7].reshape(b, npts[s], 3)
        mark("geometry issued")
        unit = self.unit_voxel_extent
        assert unit[0] == unit[1] == unit[2], "anisotropic voxels: use fused=False"
        off = float(np.float32(-0.5 * unit[0] * VOXEL_NUM_LIMIT[0]))
        extents = [float(np.float32(unit[0] * sc)) for sc in SCALE_LISTS]
        done = {}
        if self.pair_features and K == 1:
            # Both sides' geometry first (each on its stream, staging underneath), ONE wait for the 16 level sizes, then the
            # feature stage of BOTH backbones as one launch sequence -- every layer one launch over both sides' tiles -- on the
            # observed side's stream; the two read-outs run side by side again.
            staged = {}
            for side in ("inp", "tmp"):
                geometry(side)
                stage(side)
                staged[side] = torch.cuda.Event()
                staged[side].record(sstream[side])
            for side in ("inp", "tmp"):
                geo[side].synchronize()
                runs[side, 0].set_counts(counts_host[side].tolist()[:8])
            mark("counts read back")
            s_inp, s_tmp = sstream["inp"], sstream["tmp"]
            with torch.cuda.stream(s_inp):
                if not single:
                    s_inp.wait_event(staged["tmp"])
                    for t in (vox["tmp"], runs["tmp", 0].ws):
                        t.record_stream(s_inp)
                ops.backbone_features_pair(runs["inp", 0], vox["inp"], f["backbone_inp_ptrs"],
                                           runs["tmp", 0], vox["tmp"], f["backbone_tmp_ptrs"])
                feat_done = torch.cuda.Event()
                feat_done.record(s_inp)
                if not single:
                    for t in runs["tmp", 0].levels:
                        t.record_stream(s_tmp)
                runs["inp", 0].point_features(pb4["inp"], extents, off, out=pf["inp"])
                done["inp", 0] = torch.cuda.Event()
                done["inp", 0].record(s_inp)
            with torch.cuda.stream(s_tmp):
                if not single:
                    s_tmp.wait_event(feat_done)
                runs["tmp", 0].point_features(pb4["tmp"], extents, off, out=pf["tmp"])
                done["tmp", 0] = torch.cuda.Event()
                done["tmp", 0].record(s_tmp)
        paired = bool(done)
        # Small calls issue BOTH sides' geometry stages before either side's level sizes are waited for: the second side's
        # feature stage then reaches the GPU ~0.25 ms after the first side's and the two run side by side (same-job A/B at
        # N = M = 1024, 32 crops: 4.005 -> 3.928 ms).  Large point counts keep one side after the other: their conv launches fill
        # the chip alone (N = 12288 / M = 2048, 32 crops: both first 23.47 against 23.46 ms; measured the same within 0.1 %: the
        # second side's convs chained behind the first side's by an event instead of by the host; a side's disengage GEMMs
        # issued right behind its read-out, under the other side's feature stage -- the conv launch that meets a GEMM takes
        # 3.2 ms instead of 0.13, the sparse prefix of the stress step is GPU time, not host time).
        both_first = (self.HEAD_ORDER == 1 if self.HEAD_ORDER is not None else b * max(self.n_inp, self.n_tmp) <= 65536) and not paired
        if both_first:
            for side in ("inp", "tmp"):
                geometry(side)
                stage(side)
        for side, bb in (() if paired else (("inp", "backbone_inp"), ("tmp", "backbone_tmp"))):
            n = npts[side]
            if not both_first:
                geometry(side)
                stage(side)                                            # runs underneath the geometry
            geo[side].synchronize()                                    # the host waits for THIS side's geometry only
            counts = counts_host[side].tolist()
            mark("counts read back " + side)
            for c in range(K):
                runs[side, c].set_counts(counts[8 * c:8 * c + 8])
            with torch.cuda.stream(sstream[side]):
                for c in range(K):
                    rows = slice(c * bc * n, (c + 1) * bc * n)
                    runs[side, c].features(vox[side], *f[bb + "_ptrs"])
                    runs[side, c].point_features(pb4[side][rows], extents, off, out=pf[side][rows])
                    done[side, c] = torch.cuda.Event()
                    done[side, c].record(sstream[side])
        act = {}
        tail_par = (not self.single_stream and not self.async_inputs and self._tail_parallel(b, launch_by_launch=True))
        for side, key in (("Xc", "inp"), ("Yo", "tmp")):
            act.update(self._disengage_buffers(side, b * npts[key], dev, fuse=(b, 2 if tail_par else 1) if K == 1 else None))
        mark("sparse issued")
        for c in range(K):                                                     # dense stage 1, chunk by chunk on main
            for side, key in (("Xc", "inp"), ("Yo", "tmp")):
                main.wait_event(done[key, c])
                rows = slice(c * bc * npts[key], (c + 1) * bc * npts[key])
                self._disengage(f, side, pf[key][rows], act, rows)
        for st in sstream.values():
            main.wait_stream(st)
        # (launch by launch too, the tail's two directions run side by side -- on the observed side's stream, idle by now --
        # unless the instance is single-stream or a pipelining one: a partial last round of one direction's attention / GEMM workgroups then
        # overlaps the other direction instead of idling the chip)
        # (not for a pipelining instance: its side streams already belong to the next call's sparse half -- measured 3.92 ->
        # 4.36 ms per back-to-back reference-shape call with the tail on one of them)
        tail_side = sstream["inp"] if tail_par else None
        prediction = self._dense_tail(f, act, b, dev, side=tail_side)
        mark("dense issued")
        if self.mode != "test":
            prediction["sym_flag"] = data["flags"].to(dev)
        data["labels"]["points_tmp"] = pts["tmp"]
        data["labels"]["points_inp"] = pts["inp"]
        return prediction

    def _counts_pinned(self, K):
        """per side, 8 K int32 of pinned (device-visible) host memory for the level sizes of a call's K passes; reused by
        every call (the host has consumed a call's sizes before it issues the next call's geometry)"""
        cache = self.__dict__.setdefault("_counts_host", {})
        if K not in cache:
            cache[K] = {s: torch.zeros(8 * K, dtype=torch.int32).pin_memory() for s in ("inp", "tmp")}
        assert all(t.is_pinned() for t in cache[K].values()), "level-size buffers must be pinned host memory"
        return cache[K]

    def _crop_ids(self, dev, b, bc, n):
        """(b*n, 1) float column: crop id (inside its chunk of bc crops) of every point row; constant per shape, cached"""
        cache = self.__dict__.setdefault("_crop_id_cache", {})
        key = (str(dev), b, bc, n)
        if key not in cache:
            if len(cache) > 16:
                cache.clear()
            cache[key] = (torch.arange(b, device=dev) % bc).float().repeat_interleave(n).unsqueeze(1)
            torch.cuda.current_stream(dev).synchronize()          # built once; read from either side stream afterwards
        return cache[key]

    _DIS_TAGS = (("p1", 256), ("m1", 64), ("p2", 256), ("m2", 64))
    POSE_HEADS_MAX = 128      # crops up to which the two pose heads run as dcl_pose_heads (two launches) instead of six library
                              # GEMMs + glue (same-job A/B: 8 crops -3.4 %, 12: -2.8 %, 16: -2 %, 32: -0.9 %, 40: -1.2 % of the forward)
    CONF_MLP_ROWS = 16384     # point rows up to which the confidence regressor runs as ONE launch (ops.mlp128_to1: both filters in LDS,
                              # the hidden rows never leave the CU -- one workgroup per CU, 0.4 of the MFMA rate); beyond, as two GEMMs
                              # of the own core, the second with the 128 -> 1 row dot as its epilogue (ops.linear_rowdot): stand-alone
                              # 393216 rows 421 -> 259 us, 65536: 72 -> 49, 32768: 37 -> 32, 6144: 12 against 23 (hence the bound)
    POSE_PARTS_MAX = 8        # crops up to which the pooling's finish is folded into the heads' first launch (a launch less)
    PAR_TAIL = None           # None = by shape (_tail_parallel); True / False force the dense tail's two directions onto two
                              # streams / one (A/B runs)
    GROUP_ROWS = 2 * 1024     # calls of at most this many points per side issue independent MLP layers as ONE launch each
                              # (ops.linear_group; same-job A/B: -2 % at one crop, +1.3 % at four, 0 at eight); larger ones
                              # keep one library GEMM per layer
    HEAD_ORDER = None         # launch by launch: 0 = one side after the other (geometry, level sizes, features), 1 = both geometry
                              # stages first, None = by size (A/B switch: tools/ab_attr.py)
    GRAPH_TRIES = 5           # two-branch captures tried for a new whole-forward graph before its one-stream capture is kept instead
                              # (_select_capture: a capture whose second branch landed on another hardware queue than the launch
                              # stream replays 1.3-4x slower for as long as it lives).  1 = first capture or the one-stream one.
    MAX_GRAPHS = 8            # captured whole-forward graphs kept per instance (one per batch size; least recently used goes)
    GRAPH_ADMIT = 3           # with a full cache, a new batch size is captured (evicting the LRU one) on its 3rd call

    def _disengage_buffers(self, side, rows, dev, fuse=None):
        """outputs of a side's four disengage stacks.  Two of them are one half of a later concatenation -- cat[F_Xc_p1,
        F_Xo_p] / cat[F_Xc_m1, F_Xo_m] feed neck_fuser / regressor_conf (models/DCL_Net.py:207-211), cat[F_Yc_p, F_Yo_p2] /
        cat[F_Yc_m, F_Yo_m2] the *_bi twins -- so they are column blocks of those wider buffers from the start (the GEMM
        writes them in place, the attention fills the other block later: no copies)."""
        new = lambda c: torch.empty((rows, c), dtype=torch.float32, device=dev)               # noqa: E731
        fuse_buf, conf_in = new(512), new(128)
        # fuse = (b, par): the whole batch's rows of this side, and how the tail's attention launches will run.  This side's points
        # are the KEYS of one attention direction; when that call takes the split-bf16 kernel for all of its crops (ops.
        # attention_planes), its scratch is allocated here and this side's V stack ("p2" of Xc, "p1" of Yo) is written straight
        # into it as bf16 pieces by the GEMM that computes it (_disengage: ops.linear_split_vpieces) -- no fp32 V1, no piece pass
        # over its 256 channels.  Inference only (the training graph keeps every activation).
        planes, fused = None, False
        if fuse is not None and self.mode == "test":
            b, par = fuse
            nk = self.n_inp if side == "Xc" else self.n_tmp
            nq = self.n_tmp if side == "Xc" else self.n_inp
            if rows == b * nk and nk % 256 == 0:
                planes, whole = ops.attention_planes(b, nq, nk, par, dev)
                fused = planes is not None and whole
        v1 = None if fused else new(256)
        if side == "Xc":
            return {"fuse1": fuse_buf, "conf_in1": conf_in, "Xcp1": fuse_buf[:, :256], "Xcm1": conf_in[:, :64],
                    "Xcp2": v1, "Xcm2": new(64), "planes2": planes}
        return {"fuse2": fuse_buf, "conf_in2": conf_in, "Yop2": fuse_buf[:, 256:], "Yom2": conf_in[:, 64:],
                "Yop1": v1, "Yom1": new(64), "planes1": planes}

    def _disengage(self, f, side, pf_rows, act, rows=slice(None)):
        """the four disengage stacks of one side on a block of points: one shared 480->1024 GEMM (BN folded), then the four
        second layers, written into rows `rows` of the activation buffers in `act`"""
        W1t, t1, second = f["dis_" + side]
        H = self._lin_relu(pf_rows, W1t, t1)                                # (rows, 1024): 4 stacks at once
        if pf_rows.shape[0] <= self.GROUP_ROWS:                             # a handful of crops: the four layers in ONE launch
            ops.linear_group([(H[:, 256 * j:256 * (j + 1)], second[j][0], second[j][1], True, act[side + tag][rows])
                              for j, (tag, _) in enumerate(self._DIS_TAGS)])
            return
        for j, (tag, _) in enumerate(self._DIS_TAGS):
            Wt, bias = second[j]
            Hj = H[:, 256 * j:256 * (j + 1)]
            if act[side + tag] is None:                                     # this side's V stack, fused into the attention's scratch
                sw = ops.prepared_linear(Wt, Hj)
                planes = act["planes2" if side == "Xc" else "planes1"]
                all_rows = act[side + ("m2" if side == "Xc" else "m1")].shape[0]
                if sw is not None and pf_rows.shape[0] == all_rows:             # (the whole batch in this call)
                    ops.linear_split_vpieces(Hj, sw, bias, planes, self.n_inp if side == "Xc" else self.n_tmp, relu=True)
                    continue
                act[side + tag] = torch.empty((all_rows, 256), dtype=torch.float32, device=H.device)   # (not after all: the plain form)
            ops.linear(Hj, Wt, bias, True, out=act[side + tag][rows])

    def _dense(self, f, pf_inp, pf_tmp, b, dev):
        """dense half of the fused pipeline on point-major activations: disengage stacks, correspondence attention,
        confidence + fuser heads, pooling, pose heads.  Static shapes only (graph-capturable)."""
        act = {}
        for side, pfs in (("Xc", pf_inp), ("Yo", pf_tmp)):
            act.update(self._disengage_buffers(side, pfs.shape[0], dev, fuse=(b, 1)))
            self._disengage(f, side, pfs, act)
        return self._dense_tail(f, act, b, dev)

    def _dense_tail(self, f, act, b, dev, side=None):
        """everything behind the disengage stacks (needs the whole batch: the attention launches want >= 256 workgroups).
        side: a second stream (whole-forward graph of a few crops): the two directions of the attention, the two confidence /
        fuser chains and the two pose heads are independent pairs -- with a handful of crops every kernel leaves the GPU
        mostly idle, so the halves run as parallel graph branches (fork from / join into the current stream each time)."""
        main = torch.cuda.current_stream(dev)

        class _Both(object):                           # `with both: ...` runs its body on `side` between a fork and a join
            def __enter__(self_inner):
                if side is not None:
                    side.wait_stream(main)
                    self_inner.ctx = torch.cuda.stream(side)
                    self_inner.ctx.__enter__()

            def __exit__(self_inner, *a):
                if side is not None:
                    self_inner.ctx.__exit__(*a)
        second = _Both()
        par = 2 if side is not None else 1             # the two attention launches run side by side: told to the launcher

        def join():
            if side is not None:
                main.wait_stream(side)
        # cat[F_Xc_p1, F_Xo_p] / cat[F_Xc_m1, F_Xo_m] / cat[F_Yc_p, F_Yo_p2] / cat[F_Yc_m, F_Yo_m2]: the disengage GEMMs have
        # already written their column block (_disengage_buffers), the attention fills the other one
        fuse1, conf_in1, fuse2, conf_in2 = act["fuse1"], act["conf_in1"], act["fuse2"], act["conf_in2"]
        (l1, sA, tA), (l2, sB, tB) = f["neck_fuser"], f["neck_fuser_bi"]
        def conf_and_fuser(conf_in, fuse, conf_layers, fuser_layers):
            """regressor_conf* (128 -> 128 -> 128 -> 1) and neck_fuser* (512 -> 512 -> 512 -> 1024) of one direction.  (Layer d
            of both stacks as one ops.linear_group launch was measured for a handful of crops: 56 against 54 us at one crop,
            156 against 142 at six -- the library's small-tile kernels are faster per layer than the grouped kernel's 64 x 64
            tiles at K = 512, and the chain is three layers deep either way.)"""
            F = fuse
            for Wt, bias in fuser_layers:
                F = self._lin_relu(F, Wt, bias)
            return conf_logits(conf_in, conf_layers), F
        def conf_logits(conf_in, conf_layers):
            if conf_in.shape[0] <= self.CONF_MLP_ROWS:
                return ops.mlp128_to1(conf_in, conf_layers)                 # (three K = 128 layers in one launch)
            (W1t, b1), (W2t, b2), (W3t, b3) = conf_layers
            return ops.linear_rowdot(self._lin_relu(conf_in, W1t, b1), W2t, b2, W3t, b3)
        fused_pool = (b > self.POSE_PARTS_MAX and self.n_inp % ops.LINEAR_POOL_TILE == 0 and
                      self.n_tmp % ops.LINEAR_POOL_TILE == 0)
        if fused_pool:
            # The last fuser layer carries the pooling as its epilogue (ops.linear_pool): F_p -- b * (N + M) rows of 1024 floats,
            # 1.9 GB at the stress shape -- is never stored and the weighted column sums are no kernel of their own.  The
            # weights are the softmax over BOTH directions' confidences, so the two confidence regressors run first and the
            # directions meet once more in the middle: attention + logits | softmax | fuser layers 1, 2, 3 + pool | finish.
            with second:
                ops.cross_attention(b, act["Yom2"], act["Xcm2"], act["Xcp2"], fuse2[:, :256], act["Xcm2"], conf_in2[:, :64], concurrent=par,
                                    planes=act.get("planes2"))
                logit2 = conf_logits(conf_in2, f["regressor_conf_bi"])
                if side is not None:
                    logit2.record_stream(main)
            ops.cross_attention(b, act["Xcm1"], act["Yom1"], act["Yop1"], fuse1[:, 256:], act["Yom1"], conf_in1[:, 64:], concurrent=par,
                                planes=act.get("planes1"))
            logit1 = conf_logits(conf_in1, f["regressor_conf"])
            join()
            conf, w, wsum = ops.conf_softmax(b, logit1.reshape(-1), logit2.reshape(-1))
            L = self.n_inp + self.n_tmp
            with second:                                   # (forks again behind the softmax)
                if side is not None:
                    w.record_stream(side)
                H = self._lin_relu(self._lin_relu(fuse2, *l2[0]), *l2[1])
                part2 = ops.linear_pool(H, l2[2][0], l2[2][1], w.view(-1)[self.n_inp:], relu=True, rows_per_crop=self.n_tmp, w_stride=L)
                if side is not None:
                    part2.record_stream(main)
            H = self._lin_relu(self._lin_relu(fuse1, *l1[0]), *l1[1])
            part1 = ops.linear_pool(H, l1[2][0], l1[2][1], w.view(-1), relu=True, rows_per_crop=self.n_inp, w_stride=L)
            join()
            F_p_wei = ops.pool_finish2(part1, part2, wsum, (sA, tA, sB, tB))
        else:
            with second:
                ops.cross_attention(b, act["Yom2"], act["Xcm2"], act["Xcp2"], fuse2[:, :256], act["Xcm2"], conf_in2[:, :64], concurrent=par,
                                    planes=act.get("planes2"))
                logit2, Fp2 = conf_and_fuser(conf_in2, fuse2, f["regressor_conf_bi"], l2)
                if side is not None:
                    logit2.record_stream(main); Fp2.record_stream(main)          # allocated on `side`, read on `main` below
            ops.cross_attention(b, act["Xcm1"], act["Yom1"], act["Yop1"], fuse1[:, 256:], act["Yom1"], conf_in1[:, 64:], concurrent=par,
                                planes=act.get("planes1"))
            logit1, Fp1 = conf_and_fuser(conf_in1, fuse1, f["regressor_conf"], l1)   # (b*N, 1), (b*N, 1024)
            join()
            # trailing BNs after pooling: F_p_wei = sA*P1 + tA*sum(w1) + sB*P2 + tB*sum(w2), finished inside the pooling op
            if b <= self.POSE_PARTS_MAX:                   # a handful of crops: the pooling's finish inside the heads' first launch
                conf, parts = ops.conf_pool(b, logit1.reshape(-1), logit2.reshape(-1), Fp1, Fp2, affine=(sA, tA, sB, tB), finish=False)
                o9, trans_pred, rot_pred = ops.pose_heads_parts(parts, (sA, tA, sB, tB), f["regressor_rot"], f["regressor_trans"],
                                                                with_rotation=True)
                F_p_wei = None
            else:
                conf, F_p_wei = ops.conf_pool(b, logit1.reshape(-1), logit2.reshape(-1), Fp1, Fp2, affine=(sA, tA, sB, tB))
        if F_p_wei is None:
            pass
        elif b <= self.POSE_HEADS_MAX:                 # up to this many crops: both heads in two launches (csrc/dense.hip)
            o9, trans_pred, rot_pred = ops.pose_heads(F_p_wei, f["regressor_rot"], f["regressor_trans"], with_rotation=True)
        else:
            with second:
                trans_pred = self._mlp(F_p_wei, f["regressor_trans_padded"])
                if side is not None:
                    trans_pred.record_stream(main)
            o9 = self._mlp(F_p_wei, f["regressor_rot_padded"])
            rot_pred = ops.ortho9d_to_matrix(o9)
            join()
        F_Xo_p = fuse1[:, 256:].reshape(b, self.n_inp, 256).transpose(1, 2)  # (b,256,N) view
        prediction = {"trans_pred": trans_pred, "rot_pred": rot_pred, "conf": conf, "F_Xo_p": F_Xo_p}
        if self.mode != "test":
            F_Yc_p = fuse2[:, :256]
            prediction["Xo_pred"] = self._mlp(fuse1[:, 256:], f["regressor_Xo"]).reshape(b, self.n_inp, 3)
            prediction["Yc_pred"] = self._mlp(F_Yc_p, f["regressor_Yc"]).reshape(b, self.n_tmp, 3)
        return prediction


    def _tail_parallel(self, b, launch_by_launch=False):
        """Do the dense tail's two directions (attention + conf / fuser chains each) run side by side -- as parallel graph
        branches, or on two streams launch by launch?  Every kernel of the tail fills the GPU, so side by side only pays
        where a launch leaves CUs idle: while both attention launches are small (4-wave path: fewer than 256 eight-wave
        workgroups per direction -- every N = M = 1024 call: same-job A/B 6 crops -5.6 %, 12: -4.1 %, 32: -1.3 %, 40: -4 %), or
        when the larger direction's last round of 256 workgroups is mostly empty (N = 12288: 8 crops = 1.5 rounds -17 %,
        24 crops = 4.5 rounds -9 %; 16 and 32 crops = whole rounds: +2.8 % / +0.5 % side by side, so those stay serial)."""
        if self.PAR_TAIL is not None:
            return bool(self.PAR_TAIL)
        if launch_by_launch and b < 3:                 # host-bound calls: the extra stream hand-overs cost more (one crop +5 %)
            return False
        blocks = max(b * -(-self.n_inp // 256), b * -(-self.n_tmp // 256))        # 8-wave workgroups of the larger direction
        if blocks < 256:
            return True
        rounds = -(-blocks // 256)
        return (rounds - blocks / 256.0) / rounds >= 0.1

    def _stage_done(self, name, stream):
        """called behind every stage of the two branches while the whole-forward graph is laid out (name, the stage's stream);
        does nothing -- a measuring tool may replace it on an instance to enqueue a marker there"""

    # ------------------------------------------------------------------ whole-forward hipGraph (small-batch latency)
    def forward_graphed(self, data):
        """Same results as forward() (eval mode), but the whole forward -- sparse backbones in capacity mode (device-side
        row counts, no host read-back), dense part, heads -- is captured once per (b, N, M) into a hipGraph and replayed.
        Meant for the reference's actual eval regime (one image = a handful of crops per call), where the eager path is
        bound by ~350 launches + Python, not by the GPU.  Buffers are sized for the worst case of the shape, so keep it
        to small batches."""
        assert not self.training and self.fused
        f = self._fold()
        dev = self.regressor_rot.layers[0].weight.device
        b = int(data["batch_offsets"].size(0)) - 1
        S = int(np.asarray(data["voxel_num_limit"]).astype(np.int64)[0])
        need_ma = {s: int(data[s]["v2p_maps"].shape[1]) - 1 for s in ("inp", "tmp")}
        key = (b, self.n_inp, self.n_tmp, S)
        cache = self.__dict__.setdefault("_graphs", {})
        ent = cache.pop(key, None)                                           # re-inserted below: dict order = recency
        if ent is None or any(need_ma[s] > ent["ma"][s] for s in ("inp", "tmp")):
            self._drop_graph(ent)                                            # an outgrown capture is released first
            ent = None
            while len(cache) >= self.MAX_GRAPHS:                             # capacity-sized buffers per batch size: bounded
                self._drop_graph(cache.pop(next(iter(cache))))               # least recently used
            ent = self._capture(f, dev, b, S, {s: max(32, 2 * need_ma[s]) for s in ("inp", "tmp")})
        cache[key] = ent
        # resident inputs go through ops.pad_copy_many, which wants dense 2-D rows of 4-byte elements (int64 voxel rows are
        # narrowed); anything else -- host tensors, column slices, other dtypes -- takes the copy_() staging below
        def stageable(t, dtypes):
            return t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and t.dtype in dtypes
        on_dev = all(stageable(data[s]["feats"], (torch.float32,)) and stageable(data[s]["v2p_maps"], (torch.int32,)) and
                     stageable(data[s]["occupied_voxels"], (torch.int32, torch.int64)) for s in ("inp", "tmp"))
        if on_dev:                                     # resident inputs: all eight staging copies / fills in one launch
            jobs = []
            for s in ("inp", "tmp"):
                st, d = ent[s], data[s]
                # (capacity-form crops carry their live row count on the device: copied, not read)
                v0 = d["v0_dev"].view(1, 1) if "v0_dev" in d else int(d["occupied_voxels"].shape[0])
                jobs += [(st["feats"], d["feats"]), (st["occ"], d["occupied_voxels"]), (st["v2p"], d["v2p_maps"]),
                         (st["v0"].view(1, 1) if torch.is_tensor(v0) else st["v0"], v0),
                         (st["pb4"][:, 1:4], d["feats"][:, 4:7])]                  # (crop, xyz) rows of the read-out: no cat node in the graph
            ops.pad_copy_many(jobs)
        else:
            assert "v0_dev" not in data["inp"], "capacity-form crops must be resident, dense CUDA tensors"
            for s, n in (("inp", self.n_inp), ("tmp", self.n_tmp)):
                st, d = ent[s], data[s]
                v0 = int(d["occupied_voxels"].shape[0])
                st["feats"].copy_(d["feats"], non_blocking=True)
                st["pb4"][:, 1:4].copy_(st["feats"][:, 4:7])
                st["occ"][:v0].copy_(d["occupied_voxels"], non_blocking=True)
                st["v2p"].zero_()
                st["v2p"][:v0, :need_ma[s] + 1].copy_(d["v2p_maps"], non_blocking=True)
                st["v0"].fill_(v0)
        if ent.pop("fresh", False) and ent["graph"] is not None and self.GRAPH_TRIES > 0 and not self.single_stream:
            ent = cache[key] = self._select_capture(ent, f, dev, b, S)      # (first call of a new capture, inputs staged: see GRAPH_TRIES)
        if ent["graph"] is None:                       # capture_graph=False: the capacity-mode body, launch by launch
            with torch.no_grad():
                ent["out"] = ent["body"]()
        else:
            ent["graph"].replay()
        # hand-over: the graph's output buffers are overwritten by the next replay -> copies, one launch for all six
        o = ent["out"]
        F = o["F_Xo_p"].transpose(1, 2)                                      # (b, N, 256) rows of the 512-wide fuser input
        res = {"trans_pred": torch.empty_like(o["trans_pred"]), "rot_pred": torch.empty_like(o["rot_pred"]),
               "conf": torch.empty_like(o["conf"]), "F": torch.empty((b, self.n_inp, 256), dtype=torch.float32, device=dev),
               "pts_tmp": torch.empty((b, self.n_tmp, 3), dtype=torch.float32, device=dev),
               "pts_inp": torch.empty((b, self.n_inp, 3), dtype=torch.float32, device=dev)}
        ops.pad_copy_many([(res["trans_pred"], o["trans_pred"]), (res["rot_pred"], o["rot_pred"].reshape(b, 9)),
                           (res["conf"], o["conf"]), (res["F"], F.reshape(b * self.n_inp, 256)),
                           (res["pts_tmp"].view(b * self.n_tmp, 3), ent["tmp"]["feats"][:, 4:7]),
                           (res["pts_inp"].view(b * self.n_inp, 3), ent["inp"]["feats"][:, 4:7])])
        out = {"trans_pred": res["trans_pred"], "rot_pred": res["rot_pred"], "conf": res["conf"],
               "F_Xo_p": res["F"].transpose(1, 2)}
        for k in o:                                                          # train-mode extras (Xo_pred, Yc_pred)
            if k not in out:
                out[k] = o[k].clone()
        if self.mode != "test":
            out["sym_flag"] = data["flags"].to(dev)                          # as in forward() (models/DCL_Net.py:249)
        data["labels"]["points_tmp"] = res["pts_tmp"]
        data["labels"]["points_inp"] = res["pts_inp"]
        if "vi_info" in data["inp"]:
            # capacity-form crops (crops.CropBuilder(capacity=True)) are never read back on the way in: their device-side error
            # flag -- a voxel with more points than the row pitch holds (its row is truncated), or a point outside the grid --
            # travels WITH the predictions (a 1-element int32 device tensor, no synchronisation here); a caller that tabulates
            # these poses checks it once after its own synchronisation
            out["crop_overflow"] = data["inp"]["vi_info"][2:3]
        return out

    def _select_capture(self, ent, f, dev, b, S):
        """Which capture of a new whole-forward graph is kept (one-time cost per captured batch size).  A graph's second branch runs
        on a stream the runtime makes at instantiation and deals onto one of its few hardware queues; when that is not the launch
        stream's queue every fork / join edge becomes a cross-queue dependency and the replay is 1.3x (32 crops) to 4x (one crop)
        slower, for as long as the graph lives (tools/recapture_probe.py; with GPU_MAX_HW_QUEUES = 2 every capture is fast, with 16
        none).  A ONE-stream capture of the same body has no such edge and is the yardstick: a two-branch capture within 3 % of it
        (or faster) is in the fast mode and kept; otherwise up to GRAPH_TRIES - 1 more two-branch captures are tried, and if none
        gets there the one-stream capture itself is kept -- never the slow mode."""
        import time

        def timed(e):
            for _ in range(3):                                     # (the first replays of an instantiated graph pay its upload)
                e["graph"].replay()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(6):
                e["graph"].replay()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / 6

        def capture(one_stream, src):
            # (the one-stream capture issues the SAME launches -- each backbone's own, the tail's pair hints -- on one stream: same
            #  bits as the two-branch capture, only the placement differs)
            keep, keep_pf = self.single_stream, self._pair_features
            self._pair_features = self.pair_features
            self.single_stream = bool(one_stream) or keep
            try:
                cand = self._capture(f, dev, b, S, ma)
            finally:
                self.single_stream, self._pair_features = keep, keep_pf
            cand.pop("fresh", None)
            for sd in ("inp", "tmp"):
                for k in ("feats", "occ", "v2p", "v0", "pb4"):
                    cand[sd][k].copy_(src[sd][k])
            return cand
        ma = dict(ent["ma"])
        best, t_best = ent, timed(ent)
        seen = [("two branches", t_best)]
        ref = capture(True, best)
        t_ref = timed(ref)
        seen.append(("one stream", t_ref))
        tries = 1
        while t_best > 1.03 * t_ref and tries < self.GRAPH_TRIES:
            # (the runtime deals new streams onto its queues round-robin: a stream made here before every second try makes the tries
            #  walk through all queue positions whatever a capture itself advances the deal by)
            if tries % 2 == 1 and len(_QUEUE_SHIMS) < 16:
                _QUEUE_SHIMS.append(torch.cuda.Stream(dev))
            cand = capture(False, best)
            t = timed(cand)
            seen.append(("two branches", t))
            tries += 1
            if t < t_best:
                self._drop_graph(best)
                best, t_best = cand, t
            else:
                self._drop_graph(cand)
        if t_ref < t_best:
            self._drop_graph(best)
            best, t_best = ref, t_ref
        else:
            self._drop_graph(ref)
        best["capture_ms"] = [(k, round(x * 1e3, 3)) for k, x in seen]
        return best

    def _capture(self, f, dev, b, S, ma):
        unit = self.unit_voxel_extent
        assert unit[0] == unit[1] == unit[2]
        off = float(np.float32(-0.5 * unit[0] * VOXEL_NUM_LIMIT[0]))
        extents = [float(np.float32(unit[0] * sc)) for sc in SCALE_LISTS]
        ent = {"ma": ma}
        for s, n in (("inp", self.n_inp), ("tmp", self.n_tmp)):
            st = {"feats": torch.zeros((b * n, 7), dtype=torch.float32, device=dev),
                  "occ": torch.zeros((b * n, 4), dtype=torch.int32, device=dev),
                  "v2p": torch.zeros((b * n, ma[s] + 1), dtype=torch.int32, device=dev),
                  "v0": torch.zeros(1, dtype=torch.int32, device=dev)}
            st["run"] = ops.BackboneRunCap(st["occ"], st["v0"], b, S)
            st["pf"] = torch.zeros((b * n, 480), dtype=torch.float32, device=dev)
            st["tmpbuf"] = torch.empty(st["run"].tmp_bytes(b * n), dtype=torch.uint8, device=dev)
            # (crop index, x, y, z) per point for the level read-out: column 0 is constant, the coordinates arrive with the inputs
            st["pb4"] = torch.zeros((b * n, 4), dtype=torch.float32, device=dev)
            st["pb4"][:, 0] = torch.arange(b, device=dev, dtype=torch.float32).repeat_interleave(n)
            ent[s] = st

        def body():
            # Two parallel branches in the graph, one per side.  The eager path also runs each side's 3-NN searches on a
            # helper stream beside its convolutions; in a graph that does not pay (bare replay at b=1: 1.04 ms with two
            # branches, 1.22 ms with four -- every cross-branch edge costs a barrier packet), and ROCm 7.2 crashes in
            # hipStreamEndCapture when a stream joins the capture through an event of an already forked stream.
            main, side_stream = torch.cuda.current_stream(dev), self._side_stream(dev)
            act = {}
            par_dense = self._tail_parallel(b)
            stamp = self._stage_done                               # a no-op; tools/graph_timeline.py hangs time stamps on it
            stamp("start", main)
            side_stream.wait_stream(main)
            # The two branches are issued stage by stage, alternating: a graph launch hands its nodes to the queues in
            # creation order, so a branch captured as a whole after the other one starts ~50 nodes late on replay.
            sides = (("inp", "backbone_inp", main, "Xc"), ("tmp", "backbone_tmp", side_stream, "Yo"))
            xs, pb4s = {}, {}
            for stage in range(5):
                if stage == 2 and self.pair_features:
                    # the feature stage of both backbones as ONE launch sequence on the main branch (join, run, fork again)
                    main.wait_stream(side_stream)
                    ops.backbone_features_pair(ent["inp"]["run"], xs["inp"], f["backbone_inp_ptrs"],
                                               ent["tmp"]["run"], xs["tmp"], f["backbone_tmp_ptrs"])
                    side_stream.wait_stream(main)
                    continue
                for s, bb, stream, dside in sides:
                    with torch.cuda.stream(stream):
                        st = ent[s]
                        if stage == 0:                   # (the voxelisation rides on the geometry stage's launch up to 16 crops)
                            xs[s] = st["run"].geometry(voxelize=(st["feats"], st["v2p"], self.voxelization_mode))
                        elif stage == 1:
                            pb4s[s] = st["pb4"]
                        elif stage == 2:
                            st["run"].features(xs[s], *f[bb + "_ptrs"])
                        elif stage == 3:
                            st["run"].point_features(pb4s[s], extents, off, st["pf"], st["tmpbuf"])
                            st["keep"] = (xs[s], pb4s[s])
                        else:                                              # each side's disengage stacks stay on its branch
                            act.update(self._disengage_buffers(dside, st["pf"].shape[0], dev, fuse=(b, 2 if par_dense else 1)))
                            self._disengage(f, dside, st["pf"], act)
                    stamp("%s stage %d done" % (s, stage), stream)
            main.wait_stream(side_stream)                                  # join
            out_ = self._dense_tail(f, act, b, dev, side=side_stream if par_dense else None)
            stamp("end", main)
            return out_

        ent["body"] = body
        if not self.capture_graph:
            ent["graph"], ent["out"] = None, None
            return ent
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                                         # warm-up: lazy inits, allocator
                    body()
            torch.cuda.current_stream().wait_stream(side)
            try:
                g = torch.cuda.CUDAGraph(keep_graph=True)                  # keeps the hipGraph_t: its nodes can be counted
            except TypeError:
                g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = body()
        ent["graph"], ent["out"] = g, out
        ent["nodes"] = self._graph_nodes(g)
        ent["fresh"] = True
        return ent

    @staticmethod
    def _graph_nodes(g):
        """number of nodes of a captured hipGraph (None if the runtime does not let us ask)"""
        import ctypes
        try:
            raw = g.raw_cuda_graph()
            hip = ctypes.CDLL("libamdhip64.so")
            n = ctypes.c_size_t(0)
            rc = hip.hipGraphGetNodes(ctypes.c_void_p(int(raw)), None, ctypes.byref(n))
            return int(n.value) if rc == 0 else None
        except Exception:
            return None

    # ------------------------------------------------------------------ compatibility path
    def _forward_compat(self, data):
        dev = self.regressor_rot.layers[0].weight.device
        if dev.type != "cuda":
            raise RuntimeError("dcl-net_amd.Network runs on the GPU only: call .cuda() first (no CPU fallback)")
        vlim = np.asarray(data["voxel_num_limit"]).astype(np.int64)
        b = int(data["batch_offsets"].size(0)) - 1
        out = {}
        for side, bb, n in (("inp", self.backbone_inp, self.n_inp), ("tmp", self.backbone_tmp, self.n_tmp)):
            feats = data[side]["feats"].to(dev).float().contiguous()
            v2p = data[side]["v2p_maps"].to(dev).int().contiguous()
            occ = data[side]["occupied_voxels"].to(dev).int().contiguous()
            vox = pointgroup_ops.voxelization(feats, v2p, self.voxelization_mode)
            levels = bb(spconv.SparseConvTensor(vox, occ, vlim, b))
            points = feats[:, 4:].reshape(b, n, 3).reshape(-1, 3)
            bids = torch.arange(b, device=dev).unsqueeze(1).repeat(1, n).view(-1, 1)
            F = self.stage1_get_point_feats(points, bids, *levels)
            out[side] = (F.view(b, n, -1).transpose(1, 2)[:, :, :, None, None], points.view(b, n, 3))
        F_Xc, points_inp = out["inp"]
        F_Yo, points_tmp = out["tmp"]

        def dis(name, x):
            return getattr(self, name)(x).squeeze(-1).squeeze(-1)
        Xc = {t: dis("disengage_Xc_" + t, F_Xc) for t in ("p1", "m1", "p2", "m2")}
        Yo = {t: dis("disengage_Yo_" + t, F_Yo) for t in ("p1", "m1", "p2", "m2")}
        F_Xo_p, A1 = self.neck_cross_att(Xc["m1"], Yo["m1"], Yo["p1"])
        F_Yc_p, A2 = self.neck_cross_att(Yo["m2"], Xc["m2"], Xc["p2"])
        conf_1 = self.regressor_conf(torch.cat([Xc["m1"], torch.bmm(Yo["m1"], A1)], dim=1))
        conf_2 = self.regressor_conf_bi(torch.cat([torch.bmm(Xc["m2"], A2), Yo["m2"]], dim=1))
        conf = torch.sigmoid(torch.cat([conf_1, conf_2], dim=2))
        conf_softmax = torch.softmax(conf, dim=2)
        F_p = torch.cat([self.neck_fuser(torch.cat([Xc["p1"], F_Xo_p], dim=1)),
                         self.neck_fuser_bi(torch.cat([F_Yc_p, Yo["p2"]], dim=1))], dim=2)
        F_p_wei = torch.sum(F_p * conf_softmax, dim=2, keepdim=True)
        o9 = self.regressor_rot(F_p_wei).squeeze(-1)
        rot_pred = ortho9d2matrix(o9[:, :3], o9[:, 3:6], o9[:, 6:])
        trans_pred = self.regressor_trans(F_p_wei).squeeze(-1)
        prediction = {"trans_pred": trans_pred, "rot_pred": rot_pred, "conf": conf.squeeze(1), "F_Xo_p": F_Xo_p}
        if self.mode != "test":
            prediction["sym_flag"] = data["flags"].to(dev)
            prediction["Xo_pred"] = self.regressor_Xo(F_Xo_p).transpose(1, 2)
            prediction["Yc_pred"] = self.regressor_Yc(F_Yc_p).transpose(1, 2)
        data["labels"]["points_tmp"] = points_tmp
        data["labels"]["points_inp"] = points_inp
        return prediction

    def _admit_graph(self, b, data):
        """Eval loops whose crop count varies per image (YCB-V test: 1 to 9+ objects) can cycle through more batch sizes than
        the cache holds; recapturing on every call (two warm-up forwards + a capture each) would be far slower than launch
        by launch.  So with a FULL cache a new size is only admitted (evicting the least recently used one) once it has been
        seen GRAPH_ADMIT times; until then its calls go launch by launch."""
        S = int(np.asarray(data["voxel_num_limit"]).astype(np.int64)[0])
        key = (b, self.n_inp, self.n_tmp, S)
        cache = self.__dict__.setdefault("_graphs", {})
        if key in cache or len(cache) < self.MAX_GRAPHS:
            return True
        seen = self.__dict__.setdefault("_graph_seen", {})
        if len(seen) > 64:
            seen.clear()
        seen[key] = seen.get(key, 0) + 1
        if seen[key] >= self.GRAPH_ADMIT:
            del seen[key]
            return True
        return False

    def replays_graph(self, b):
        """routing rule of eval-mode calls (see __init__): True = a call of b crops replays its whole-forward hipGraph"""
        if self.graph_max_batch <= 0 or b <= 0:
            return False
        if b <= self.graph_max_batch:
            return True
        return b * (self.n_inp + self.n_tmp) <= self.graph_max_points

    def forward(self, data):
        """eval(): the fused inference pipeline -- outputs carry no autograd graph, whether or not the caller wrapped the call
        in torch.no_grad() (tools/test_LM.py:110 does not).  train() (or fused=False): the module path, differentiable."""
        if self.fused and not self.training:
            b = int(data["batch_offsets"].size(0)) - 1
            if self.replays_graph(b) and self._admit_graph(b, data):
                return self.forward_graphed(data)
            if "v0_dev" in data["inp"]:                 # capacity-form crops (CropBuilder(capacity=True)) off the graph path:
                from ..crops import exact_form          # the launch-by-launch path sizes its buffers on the host
                exact = exact_form(data)
                with torch.no_grad():
                    pred = self._forward_fused(exact)
                data["labels"] = exact["labels"]
                return pred
            with torch.no_grad():
                return self._forward_fused(data)
        return self._forward_compat(data)

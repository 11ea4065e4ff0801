"""Stage-2 pose refiner: drop-in for the reference's models/refiner.py:57-95 (same 18-key state_dict),
plus the iterative refinement loop of tools/test_YCBV_stage2.py:204-225 as a reusable function.

fused path: point-major GEMMs with bias+ReLU epilogues for MLP_share (259->512->512->1024), the
un-renormalised softmax slice (refiner.py:81) and the weighted sum as one GEMV-shaped reduction,
on-device 3x3 SVD.  Shapes are static, so `refine_loop(..., graph=True)` captures the loop body in a
hipGraph (torch.cuda.CUDAGraph) and replays it.
"""
import torch
import torch.nn as nn

from .. import ops
from .Modules import Head_MultiLayerPerceptron
from .losses import losses_refiner  # noqa: F401  (reference: models/refiner.py::losses_refiner)


def ortho9d2matrix(x_raw, y_raw, z_raw):
    """models/refiner.py:34-56 == models/DCL_Net.py:15-36; differentiable when a gradient is needed (DCL_Net.ortho9d2matrix)"""
    from .DCL_Net import ortho9d2matrix as _o
    return _o(x_raw, y_raw, z_raw)


_VERSION_OF = __import__("operator").attrgetter("_version")


def _param_version(mod):
    """sum of the in-place version counters of every parameter and buffer: changes whenever a weight is rewritten in
    place (optimizer.step(), EMA swaps via p.data.copy_, manual surgery), so caches keyed on it cannot go stale"""
    ts = mod.__dict__.get("_vlist")
    if ts is None:
        ts = mod.__dict__["_vlist"] = list(mod.parameters()) + list(mod.buffers())
    return sum(map(_VERSION_OF, ts))


class Refiner(nn.Module):
    POSE_HEADS_MAX = 128      # crops up to which the two heads run as dcl_pose_heads (two launches; Network.POSE_HEADS_MAX)

    def __init__(self, cfg=None):
        super().__init__()
        self.MLP_share = Head_MultiLayerPerceptron([256 + 3, 512, 512, 1024], ["relu"] * 3, [False] * 3, [0.0] * 3)
        self.regressor_rot2 = Head_MultiLayerPerceptron([1024, 512, 128, 9], ["relu", "relu", "none"], [False] * 3,
                                                        [0.0] * 3)
        self.regressor_trans2 = Head_MultiLayerPerceptron([1024, 512, 128, 3], ["relu", "relu", "none"], [False] * 3,
                                                          [0.0] * 3)
        self._folded = None

    def _invalidate(self):
        """folded weights and captured refine-loop graphs (folded tensors' addresses baked in) follow the parameters"""
        self._folded = None
        self.__dict__.pop("_graphs", None)
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
        state = dict(self.__dict__)
        state.pop("_graphs", None)
        state.pop("_vlist", None)
        state["_folded"] = None
        return state

    def _fold(self):
        ver = _param_version(self)
        if self._folded is not None and self.__dict__.get("_fold_version") != ver:
            self._invalidate()                       # weights were rewritten in place since the fold / the graph capture
        if self._folded is None:
            self.__dict__["_fold_version"] = ver
            with torch.no_grad():
                f = {}
                for name in ("MLP_share", "regressor_rot2", "regressor_trans2"):
                    L = getattr(self, name).layers
                    f[name] = [(L[i].weight[:, :, 0].t().contiguous(), L[i].bias.contiguous()) for i in (0, 2, 4)]
                # split the 259-row first weight: rows 0..2 act on xyz, rows 3.. on F_Xo_p
                Wt, bias = f["MLP_share"][0]
                f["share0_xyz"], f["share0_feat"] = Wt[:3].contiguous(), Wt[3:].contiguous()
                # the big layers prepared for the split-bf16 GEMM core (ops.prepare_linear; models/DCL_Net.py: _fold)
                for Wt in (f["share0_feat"], f["MLP_share"][1][0], f["MLP_share"][2][0]):
                    ops.prepare_linear(Wt)
            self._folded = f
        return self._folded

    def forward_pm(self, xyz_pm, feat_term, conf_w):
        """Point-major core: xyz_pm (b*n,3) canonicalised points, feat_term (b*n,512) = F_Xo_p @ W0[3:] + b0
        (constant over refinement iterations), conf_w (b,n) softmax slice.  -> (delta_t (b,3), delta_R (b,3,3))."""
        f = self._fold()
        b, n = conf_w.shape
        # first shared layer: the K = 3 product + the constant term + ReLU in one pass (ops.affine3_relu; a library GEMM and a
        # ReLU sweep before: 49 -> ~27 us per iteration at 32 768 rows)
        h = ops.affine3_relu(xyz_pm.contiguous(), f["share0_xyz"], feat_term)
        # (the own GEMM core, ops.linear -> csrc/linear_dma.hip: no vendor kernel that a caller overlapping the refiner with
        #  another forward's GEMMs could hang the GPU with; see models/DCL_Net.py: _lin_relu)
        h = ops.linear(h, f["MLP_share"][1][0], f["MLP_share"][1][1], True)
        if n % ops.LINEAR_POOL_TILE == 0:
            # the last shared layer with the confidence-weighted sum over the points as its epilogue: the (b*n, 1024) activation
            # is never stored; a crop's n / 128 tile partials are added in tile order
            part = ops.linear_pool(h, f["MLP_share"][2][0], f["MLP_share"][2][1], conf_w.reshape(-1), relu=True)
            shared = part.view(b, n // ops.LINEAR_POOL_TILE, -1).sum(dim=1)                  # (b, 1024)
        else:
            h = ops.linear(h, f["MLP_share"][2][0], f["MLP_share"][2][1], True)              # (b*n, 1024)
            shared = torch.bmm(conf_w.unsqueeze(1), h.view(b, n, -1)).squeeze(1)             # (b, 1024)
        # both pose heads (1024 -> 512 -> 128 -> 9 | 3) and the rotation in two launches (ops.pose_heads, as in stage 1) instead of
        # six library GEMMs of 5-14 us each on 32 rows + the ortho kernel; larger batches keep the library
        if b <= self.POSE_HEADS_MAX:
            _, dt, dR = ops.pose_heads(shared, f["regressor_rot2"], f["regressor_trans2"], with_rotation=True)
            return dt, dR

        def head(x, layers):
            x = ops.linear(x, layers[0][0], layers[0][1], True)
            x = ops.linear(x, layers[1][0], layers[1][1], True)
            return ops.linear(x, layers[2][0], layers[2][1], False)
        o9 = head(shared, f["regressor_rot2"])
        return head(shared, f["regressor_trans2"]), ops.ortho9d_to_matrix(o9)

    def _forward_modules(self, x, conf):
        """training path (tools/train_YCBV_stage2.py:169,243-262 calls refiner.train(); outputs = refiner(inp);
        loss.backward()): the registered modules composed as the reference composes them (models/refiner.py:78-95), every
        step differentiable, no cached weights."""
        conf_softmax = torch.softmax(conf.unsqueeze(1), dim=2)[:, :, :1024]
        shared = (self.MLP_share(x) * conf_softmax).sum(dim=2, keepdim=True)
        o9 = self.regressor_rot2(shared).squeeze(-1)
        delta_t = self.regressor_trans2(shared).squeeze(-1)
        return {"trans_pred": delta_t, "rot_pred": ortho9d2matrix(o9[:, :3], o9[:, 3:6], o9[:, 6:])}

    def forward(self, input_dict):
        """reference contract: {"input_features" (b,259,n), "conf" (b,n+m), "obj_idx"} ->
        {"trans_pred" (b,3), "rot_pred" (b,3,3)}.  train() mode: differentiable module path; eval(): fused inference
        path (folded weights, no autograd graph)."""
        x = input_dict["input_features"]
        conf = input_dict["conf"]
        if not x.is_cuda:
            raise RuntimeError("dcl-net_amd.Refiner runs on the GPU only (no CPU fallback)")
        if self.training:
            return self._forward_modules(x, conf)
        with torch.no_grad():
            f = self._fold()
            b, _, n = x.shape
            conf_w = torch.softmax(conf.unsqueeze(1), dim=2)[:, 0, :1024].contiguous()       # refiner.py:81
            pm = x.transpose(1, 2).reshape(b * n, -1)
            feat_term = ops.linear(pm[:, 3:], f["share0_feat"], f["MLP_share"][0][1], False)
            dt, dR = self.forward_pm(pm[:, :3].contiguous(), feat_term, conf_w)
        return {"trans_pred": dt, "rot_pred": dR}


def refine_loop(refiner, pred, points_inp, iteration=2, graph=False):
    """Iterative refinement of tools/test_YCBV_stage2.py:204-225 on the outputs of Network.forward:
    pred {"rot_pred","trans_pred","F_Xo_p" (b,256,n),"conf"}, points_inp (b,n,3) -> (rot (b,3,3), trans (b,3)).
    With graph=True the whole loop is captured once per shape into a hipGraph and replayed."""
    f = refiner._fold()
    rot0, trans0, conf = pred["rot_pred"], pred["trans_pred"], pred["conf"]
    F_pm = pred["F_Xo_p"].transpose(1, 2)                                 # (b,n,256) (a view when fused)
    b, n, _ = F_pm.shape

    def feature_term(F_pm, out=None):
        # the feature half of the first shared layer: the same in every iteration, computed once per call
        bias, W = f["MLP_share"][0][1], f["share0_feat"]
        # point-major rows (what Network.forward hands over: dense from a graph replay, a 256-column block of the 512-wide fuser
        # input -- row pitch 512 -- launch by launch): one GEMM of the own core on the (b*n, 256) rows as they lie
        if F_pm.stride(2) == 1 and F_pm.stride(0) == n * F_pm.stride(1):
            rows = F_pm.as_strided((b * n, F_pm.shape[2]), (F_pm.stride(1), 1))
            return ops.linear(rows, W, bias, False, out=out)
        # channel-first features (the reference's own layout, (b, 256, n) storage): one transposing copy (33 MB at 32 crops, ~15 us)
        # and the same GEMM -- not torch.baddbmm on the transposed view: torch takes the vendor library's first-choice algorithm,
        # for some row counts a workspace-exchanging stream-K kernel that must never meet a second one on the GPU (csrc/linear.cpp)
        return ops.linear(F_pm.contiguous().view(b * n, -1), W, bias, False, out=out)

    def body(rot, trans, feat_term, conf, pts):
        conf_w = torch.softmax(conf.unsqueeze(1), dim=2)[:, 0, :1024].contiguous()
        for _ in range(iteration):
            cur = torch.bmm(pts - trans.unsqueeze(1), rot).reshape(b * n, 3)
            dt, dR = refiner.forward_pm(cur, feat_term, conf_w)
            trans = (rot @ dt.unsqueeze(2)).squeeze(2) + trans
            rot = rot @ dR
        return rot, trans

    with torch.no_grad():
        if not graph:
            return body(rot0, trans0, feature_term(F_pm), conf, points_inp)
        key = (b, n, iteration, conf.shape[1])
        cache = refiner.__dict__.setdefault("_graphs", {})
        if key not in cache:
            cc = lambda t: t.clone(memory_format=torch.contiguous_format)                     # noqa: E731
            static = [cc(rot0), cc(trans0), feature_term(F_pm), cc(conf), cc(points_inp)]
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):                                        # warm-up (allocator, lazy inits)
                    body(*static)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = body(*static)
            cache[key] = (g, static, out)
        g, static, out = cache[key]
        # the 256-channel features are not copied into the graph (33 MB at 32 crops): their GEMM runs in front of the replay and
        # writes the graph's static feature term; the small inputs are copied
        feature_term(F_pm, out=static[2])
        ops.pad_copy_many([(static[0].view(b, 9), rot0.reshape(b, 9)), (static[1], trans0.reshape(b, 3)),
                           (static[3], conf.reshape(b, -1)), (static[4].view(b * n, 3), points_inp.reshape(b * n, 3))])
        g.replay()
        rot, trans = torch.empty_like(out[0]), torch.empty_like(out[1])     # (the graph's own outputs are overwritten by the next replay)
        ops.pad_copy_many([(rot.view(b, 9), out[0].reshape(b, 9)), (trans, out[1].reshape(b, 3))])
        return rot, trans


def stage2_chain(model, refiner, data, iteration=2, graph=True):
    """BASELINE configs[4] as one call: the body of the reference's stage-2 eval loop (tools/test_YCBV_stage2.py:204-225):
    `model(data)` -> canonicalised points (P - t) R -> cat[points, F_Xo_p] -> `iteration` x refiner with pose composition
    t <- R dt + t, R <- R dR.  -> (rot (b,3,3), trans (b,3), stage-1 prediction dict).  With graph=True the refine loop
    replays its hipGraph (static shapes); the stage-1 forward keeps its own schedule (eager or forward_graphed)."""
    with torch.no_grad():
        pred = model(data)
        points_inp = data["labels"]["points_inp"]
        rot, trans = refine_loop(refiner, pred, points_inp, iteration, graph=graph)
    return rot, trans, pred

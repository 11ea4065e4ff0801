"""Generates tests/golden/*.npz by running the REFERENCE's own Python graph.

Runs only in the build container (needs /root/reference; nothing from it is copied): imports the
reference's models/DCL_Net.py, models/Modules.py and models/refiner.py unmodified, with
  * its CUDA extension modules (spconv, PG_OP wrappers, pointnet_sp/pointnet_lib) replaced by CPU stand-ins
    built on oracle/native.py (the C restatement of those kernels, pinned separately), and
  * `.cuda()` made a no-op,
feeds it the seeded synthetic crops + seeded weights of dcl-net_amd/synth.py, and stores inputs and outputs.
tests/test_oracle_golden.py then checks oracle/graph.py (the restatement that travels to the GPU box)
against these vectors; the GPU parity tests check the HIP path against both.

    python tests/golden/make_golden.py        # rewrites tests/golden/dclnet_b2_n256.npz, refiner_b2.npz,
                                              # dclnet_s0_train.npz, dclnet_nm384_train.npz,
                                              # dclnet_b4_n1024_chain.npz, dclnet_stress_b1.npz
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import native as K  # noqa: E402

dcl = importlib.import_module("dcl-net_amd")


# ----------------------------------------------------------------------------- CPU stand-ins for natives
class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features, self.indices = features, indices.int()
        self.spatial_shape, self.batch_size = [int(s) for s in spatial_shape], int(batch_size)
        self.indice_dict, self.grid = {}, grid


class _SpModule(nn.Module):
    pass


class _Conv(_SpModule):
    def __init__(self, cin, cout, ksize, stride=1, padding=0, dilation=1, groups=1, bias=True, indice_key=None,
                 subm=False):
        super().__init__()
        assert not bias
        self.k = ksize
        self.stride = stride[0] if isinstance(stride, (tuple, list)) else stride
        self.padding, self.subm = padding, subm
        self.weight = nn.Parameter(torch.zeros(ksize, ksize, ksize, cin, cout))

    def forward(self, x):
        if self.subm:
            outids, pairs, num, oshape = K.get_indice_pairs(x.indices.numpy(), x.batch_size, x.spatial_shape, self.k,
                                                            subm=True)
        else:
            outids, pairs, num, oshape = K.get_indice_pairs(x.indices.numpy(), x.batch_size, x.spatial_shape, self.k,
                                                            self.stride, self.padding, 1)
        f = K.indice_conv(x.features.detach().numpy(), self.weight.detach().numpy(), pairs, num, outids.shape[0],
                          subm=self.subm)
        out = SparseConvTensor(torch.from_numpy(f), torch.from_numpy(outids), oshape, x.batch_size)
        out.indice_dict = x.indice_dict
        return out


class SparseConv3d(_Conv):
    pass


class SubMConv3d(_Conv):
    def __init__(self, *a, **k):
        super().__init__(*a, subm=True, **k)


class SparseAvgPool3d(_SpModule):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1, use_gs=True):
        super().__init__()
        assert not use_gs
        self.k, self.s, self.p = kernel_size, stride, padding

    def forward(self, x):
        outids, pairs, num, oshape = K.get_indice_pairs(x.indices.numpy(), x.batch_size, x.spatial_shape, self.k,
                                                        self.s, self.p, 1)
        f, _ = K.indice_avgpool(x.features.detach().numpy(), pairs, num, outids.shape[0])
        return SparseConvTensor(torch.from_numpy(f), torch.from_numpy(outids), oshape, x.batch_size)


class SparseSequential(_SpModule):
    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def forward(self, x):
        for m in self._modules.values():
            if isinstance(m, _SpModule):
                x = m(x)
            elif x.indices.shape[0] != 0:
                x.features = m(x.features)
        return x


def install_stubs():
    if not hasattr(np, "int"):
        np.int, np.float = int, float
    torch.Tensor.cuda = lambda self, *a, **k: self
    sp = types.ModuleType("spconv")
    for c in (SparseConvTensor, SparseConv3d, SubMConv3d, SparseAvgPool3d, SparseSequential):
        setattr(sp, c.__name__, c)
    sys.modules["spconv"] = sp
    sys.modules["ipdb"] = types.ModuleType("ipdb")
    tbx = types.ModuleType("tensorboardX")          # utils/tools_train.py imports it at module scope
    tbx.SummaryWriter = object
    sys.modules["tensorboardX"] = tbx
    sys.dont_write_bytecode = True                  # never write into /root/reference

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    for pkg in ("libs", "libs.pointnet_lib", "libs.pointnet_sp", "libs.pointgroup_ops"):
        mod(pkg).__path__ = []

    def three_nn(unknown, known):
        d2, idx = K.three_nn_sp(unknown.numpy(), known.numpy())
        return torch.sqrt(torch.from_numpy(d2)), torch.from_numpy(idx)

    def three_interpolate(feats, idx, weight):
        return torch.from_numpy(K.three_interpolate_sp(feats.detach().numpy(), idx.numpy(), weight.numpy()))
    mod("libs.pointnet_sp.pointnet2_utils", three_nn=three_nn, three_interpolate=three_interpolate)
    mod("libs.pointnet_lib.pointnet2_utils", knn=None)
    pg = mod("libs.pointgroup_ops.functions.pointgroup_ops",
             voxelization=lambda feats, rule, mode=4: torch.from_numpy(K.voxelize_fp(feats.numpy(), rule.numpy(), mode)))
    mod("libs.pointgroup_ops.functions", pointgroup_ops=pg).__path__ = []
    sys.path.insert(0, REF)


def attr(d):
    class A(dict):
        __getattr__ = dict.__getitem__
    return A({k: attr(v) if isinstance(v, dict) else v for k, v in d.items()})


def clone_data(data):
    out = {}
    for k, v in data.items():
        out[k] = clone_data(v) if isinstance(v, dict) else (v.clone() if torch.is_tensor(v) else v)
    return out


def main():
    install_stubs()
    ref_net_mod = importlib.import_module("models.DCL_Net")
    ref_refiner_mod = importlib.import_module("models.refiner")
    b, n = 2, 256
    cfg = dcl.synth.default_cfg(n, n)
    torch.manual_seed(0)
    ref_net = ref_net_mod.Network(attr(dict(cfg)), mode="test")
    sd = dcl.synth.synth_state_dict(ref_net, seed=1)            # same key set as our Network (checked below)
    ours = dcl.DCL_Net.Network(cfg, mode="test")
    assert list(ours.state_dict().keys()) == list(ref_net.state_dict().keys())
    assert all(ours.state_dict()[k].shape == v.shape for k, v in ref_net.state_dict().items())
    ref_net.load_state_dict(sd)
    ref_net.eval()
    data = dcl.synth.make_batch(b, n, n, voxelize_idx=lambda c, bs, mode: tuple(
        torch.from_numpy(a) for a in K.voxelize_idx(c.numpy(), bs, mode)))
    inputs = clone_data(data)
    with torch.no_grad():
        pred = ref_net(data)
    out = {"trans_pred": pred["trans_pred"].numpy(), "rot_pred": pred["rot_pred"].numpy(), "conf": pred["conf"].numpy(),
           "F_Xo_p_sub": pred["F_Xo_p"][:, ::8, ::8].contiguous().numpy(),
           "F_Xo_p_sum": pred["F_Xo_p"].double().sum(dim=2).numpy()}
    for side in ("inp", "tmp"):
        for k in ("feats", "occupied_voxels", "p2v_maps", "v2p_maps"):
            out["%s_%s" % (side, k)] = inputs[side][k].numpy()
    out["meta"] = np.array([b, n, n, 1], np.int64)               # b, n_inp, n_tmp, weight seed
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dclnet_b2_n256.npz"), **out)

    # ---- refiner (reference models/refiner.py) on the stage-1 outputs, 2 iterations like the eval script
    ref_ref = ref_refiner_mod.Refiner(None)
    sdr = dcl.synth.synth_state_dict(ref_ref, seed=2)
    assert list(dcl.refiner.Refiner().state_dict().keys()) == list(ref_ref.state_dict().keys())
    ref_ref.load_state_dict(sdr)
    ref_ref.eval()
    bb, nn_ = 2, 1024                                            # the reference slices conf[:, :1024] (refiner.py:81)
    g = torch.Generator().manual_seed(5)
    F = torch.randn(bb, 256, nn_, generator=g)
    pts = torch.randn(bb, nn_, 3, generator=g) * 0.05
    conf = torch.rand(bb, 2 * nn_, generator=g)
    o9 = torch.randn(bb, 9, generator=g)
    rot = ref_refiner_mod.ortho9d2matrix(o9[:, :3], o9[:, 3:6], o9[:, 6:])
    trans = torch.randn(bb, 3, generator=g) * 0.02
    rot0, trans0 = rot.clone(), trans.clone()
    with torch.no_grad():
        cur = torch.bmm(pts - trans.unsqueeze(1), rot)
        inp = torch.cat([cur.transpose(1, 2), F], dim=1)
        first = None
        for _ in range(2):                                       # tools/test_YCBV_stage2.py:214-225
            o = ref_ref({"input_features": inp, "conf": conf, "obj_idx": None})
            first = first or {k: v.clone() for k, v in o.items()}
            trans = (rot @ o["trans_pred"].unsqueeze(2)).squeeze(2) + trans
            rot = rot @ o["rot_pred"]
            cur = torch.bmm(pts - trans.unsqueeze(1), rot)
            inp = torch.cat([cur.transpose(1, 2), F], dim=1)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "refiner_b2.npz"), gen_seed=np.array([5]),
                        o9=o9.numpy(), rot0=rot0.numpy(), trans0=trans0.numpy(),
                        dt_first=first["trans_pred"].numpy(), dR_first=first["rot_pred"].numpy(),
                        rot_final=rot.numpy(), trans_final=trans.numpy())
    print("golden written:", {k: v.shape for k, v in out.items() if k != "meta"})

    # ---- mode='train' outputs on (a) BASELINE config 0's shape (LineMOD-style crop: N=2048 observed, M=500 template,
    # 5 mm voxels) and (b) an N=M crop, where the reference's own objectives are defined (`losses`,
    # models/DCL_Net.py:261-304 needs N == M in CD_Dis; `losses_refiner`, models/refiner.py:98-125)
    for tag, b2, n_inp, n_tmp, first, with_losses in (("s0", 2, 2048, 500, 9, False), ("nm384", 2, 384, 384, 4, True)):
        cfg2 = dcl.synth.default_cfg(n_inp, n_tmp, unit=0.005)
        ref2 = ref_net_mod.Network(attr(dict(cfg2)), mode="train")
        ref2.load_state_dict(dcl.synth.synth_state_dict(ref2, seed=3))
        ref2.eval()
        data2 = dcl.synth.make_batch(b2, n_inp, n_tmp, unit=0.005, first=first, voxelize_idx=lambda c, bs, mode: tuple(
            torch.from_numpy(a) for a in K.voxelize_idx(c.numpy(), bs, mode)))
        data2["flags"] = torch.tensor([0.0, 1.0])
        in2 = clone_data(data2)
        with torch.no_grad():
            p2 = ref2(data2)
        out2 = {"trans_pred": p2["trans_pred"].numpy(), "rot_pred": p2["rot_pred"].numpy(), "conf": p2["conf"].numpy(),
                "Xo_pred": p2["Xo_pred"].numpy(), "Yc_pred": p2["Yc_pred"].numpy(),
                "F_Xo_p_sum": p2["F_Xo_p"].double().sum(dim=2).numpy(),
                "rot_gt": in2["labels"]["rot_gt"].numpy(), "trans_gt": in2["labels"]["trans_gt"].numpy(),
                "flags": in2["flags"].numpy(), "meta": np.array([b2, n_inp, n_tmp, 3], np.int64)}
        if with_losses:
            with torch.no_grad():
                lo = ref_net_mod.losses(None)(p2, data2["labels"])
                pr = {"rot_pred": p2["rot_pred"].transpose(1, 2).contiguous(), "trans_pred": -p2["trans_pred"] * 0.5}
                lr = ref_refiner_mod.losses_refiner(None)(pr, p2["trans_pred"], p2["rot_pred"],
                                                          data2["labels"]["points_tmp"], p2["sym_flag"], data2["labels"])
            out2["losses"] = np.array([float(lo[k]) for k in ("loss_pose", "loss_Xo", "loss_Yc", "loss_conf", "loss_all")])
            out2["loss_refiner"] = np.array([float(lr["loss_all"])])
        for side in ("inp", "tmp"):
            for k in ("feats", "occupied_voxels", "p2v_maps", "v2p_maps"):
                out2["%s_%s" % (side, k)] = in2[side][k].numpy()
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dclnet_%s_train.npz" % tag), **out2)
        print("golden written: dclnet_%s_train.npz" % tag, out2.get("losses"), out2.get("loss_refiner"))
    chain_golden(ref_net_mod, ref_refiner_mod)
    stress_golden(ref_net_mod)


def chain_golden(ref_net_mod, ref_refiner_mod):
    """The shape the shipped yaml defines (N = M = 1024, configs/config_YCBV_bs32.yaml:24-25), b = 4, test mode, and
    BASELINE configs[4]: the reference Network's outputs chained into 2 refiner iterations exactly as
    tools/test_YCBV_stage2.py:204-225 does it, plus the ADD-S distance of the final pose (:233-235, with the template
    cloud standing in for the class's CAD points)."""
    b, n, iteration = 4, 1024, 2
    cfg = dcl.synth.default_cfg(n, n)
    net = ref_net_mod.Network(attr(dict(cfg)), mode="test")
    net.load_state_dict(dcl.synth.synth_state_dict(net, seed=1))
    net.eval()
    refiner = ref_refiner_mod.Refiner(None)
    refiner.load_state_dict(dcl.synth.synth_state_dict(refiner, seed=2))
    refiner.eval()
    data = dcl.synth.make_batch(b, n, n, first=50, voxelize_idx=lambda c, bs, mode: tuple(
        torch.from_numpy(a) for a in K.voxelize_idx(c.numpy(), bs, mode)))
    inputs = clone_data(data)
    cls_label = data["obj_idx"]
    with torch.no_grad():
        outputs_main = net(data)
        rot_cur, trans_cur = outputs_main["rot_pred"], outputs_main["trans_pred"]
        points_inp = data["labels"]["points_inp"]
        points_inp_cur = torch.bmm(points_inp - trans_cur.unsqueeze(1), rot_cur)
        F_Xo_p = outputs_main["F_Xo_p"]
        inp_refiner = torch.cat([points_inp_cur.transpose(1, 2), F_Xo_p], dim=1).detach()
        conf = outputs_main["conf"]
        per_iter = []
        for _ in range(iteration):
            o = refiner({"input_features": inp_refiner, "conf": conf, "obj_idx": cls_label})
            trans_cur = (rot_cur @ o["trans_pred"].unsqueeze(2)).squeeze(2) + trans_cur
            rot_cur = rot_cur @ o["rot_pred"]
            per_iter.append((rot_cur.clone(), trans_cur.clone()))
            points_inp_cur = torch.bmm(points_inp - trans_cur.unsqueeze(1), rot_cur)
            inp_refiner = torch.cat([points_inp_cur.transpose(1, 2), F_Xo_p], dim=1).detach()
        cld = data["labels"]["points_tmp"]
        rot_gt, trans_gt = inputs["labels"]["rot_gt"], inputs["labels"]["trans_gt"]
        adds = []
        for R, t in ((outputs_main["rot_pred"], outputs_main["trans_pred"]), (rot_cur, trans_cur)):
            pp = torch.bmm(cld, R.transpose(1, 2)) + t.unsqueeze(1)
            pg = torch.bmm(cld, rot_gt.transpose(1, 2)) + trans_gt.unsqueeze(1)
            adds.append(torch.mean(torch.min(torch.norm(pp.unsqueeze(2) - pg.unsqueeze(1), dim=3), 2)[0], dim=1))
    out = {"trans_pred": outputs_main["trans_pred"].numpy(), "rot_pred": outputs_main["rot_pred"].numpy(),
           "conf": conf.numpy(), "F_Xo_p_sub": F_Xo_p[:, ::8, ::8].contiguous().numpy(),
           "F_Xo_p_sum": F_Xo_p.double().sum(dim=2).numpy(),
           "rot_iter1": per_iter[0][0].numpy(), "trans_iter1": per_iter[0][1].numpy(),
           "rot_final": rot_cur.numpy(), "trans_final": trans_cur.numpy(),
           "adds_stage1": adds[0].numpy(), "adds_final": adds[1].numpy(),
           "rot_gt": rot_gt.numpy(), "trans_gt": trans_gt.numpy(), "obj_idx": cls_label.numpy(),
           "meta": np.array([b, n, n, 1], np.int64), "refiner_seed": np.array([2]), "first_crop": np.array([50])}
    for side in ("inp", "tmp"):
        for k in ("feats", "occupied_voxels", "p2v_maps", "v2p_maps"):
            out["%s_%s" % (side, k)] = inputs[side][k].numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dclnet_b4_n1024_chain.npz"), **out)
    print("golden written: dclnet_b4_n1024_chain.npz  ADD-S stage1", adds[0].numpy(), "final", adds[1].numpy())


def stress_golden(ref_net_mod):
    """ONE crop of BASELINE configs[1]'s shape (N = 12288 observed / M = 2048 template points, 6 mm voxels) through the
    reference's own test-mode Network (the stress shape was only checked against oracle/graph.py).  The
    inputs are the procedural crop `synth.make_batch(1, 12288, 2048, first=3)` -- regenerated by the tests, pinned here by
    a checksum instead of 0.5 MB of points -- so the fixture holds outputs only."""
    b, n_inp, n_tmp, first = 1, 12288, 2048, 3
    cfg = dcl.synth.default_cfg(n_inp, n_tmp)
    ref = ref_net_mod.Network(attr(dict(cfg)), mode="test")
    ref.load_state_dict(dcl.synth.synth_state_dict(ref, seed=1))
    ref.eval()
    data = dcl.synth.make_batch(b, n_inp, n_tmp, first=first, voxelize_idx=lambda c, bs, mode: tuple(
        torch.from_numpy(a) for a in K.voxelize_idx(c.numpy(), bs, mode)))
    inputs = clone_data(data)
    with torch.no_grad():
        pred = ref(data)
    out = {"trans_pred": pred["trans_pred"].numpy(), "rot_pred": pred["rot_pred"].numpy(), "conf": pred["conf"].numpy(),
           "F_Xo_p_sub": pred["F_Xo_p"][:, ::8, ::64].contiguous().numpy(),
           "F_Xo_p_sum": pred["F_Xo_p"].double().sum(dim=2).numpy(),
           "meta": np.array([b, n_inp, n_tmp, 1, first], np.int64)}
    for side in ("inp", "tmp"):                                # input checksums (float64 sums / integer sums) + sizes
        out["%s_check" % side] = np.array([float(inputs[side]["feats"].double().sum()),
                                           float(inputs[side]["occupied_voxels"].double().sum()),
                                           float(inputs[side]["v2p_maps"].double().sum()),
                                           inputs[side]["occupied_voxels"].shape[0], inputs[side]["v2p_maps"].shape[1]])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dclnet_stress_b1.npz"), **out)
    print("golden written: dclnet_stress_b1.npz", out["trans_pred"], out["inp_check"], out["tmp_check"])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "stress":          # only the one-crop stress-shape fixture
        install_stubs()
        stress_golden(importlib.import_module("models.DCL_Net"))
    elif len(sys.argv) > 1 and sys.argv[1] == "chain":           # only the N=M=1024 / stage-2 chain fixture
        install_stubs()
        chain_golden(importlib.import_module("models.DCL_Net"), importlib.import_module("models.refiner"))
    else:
        main()

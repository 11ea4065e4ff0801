"""The oracle's graph restatement (oracle/graph.py) against vectors produced by the REFERENCE's own
models/*.py (tests/golden/make_golden.py, run in the build container)."""
import os

import numpy as np
import torch

from conftest import load_golden_data, load_stress_golden
from oracle import graph as G


def _state(dcl, cfg, seed, cls):
    net = cls(cfg, mode="test") if cfg is not None else cls()
    return dcl.synth.synth_state_dict(net, seed)


def test_forward_matches_reference_graph(dcl, oracle, golden_dir):
    data, exp, (b, n_inp, n_tmp, wseed) = load_golden_data(os.path.join(golden_dir, "dclnet_b2_n256.npz"))
    cfg = dcl.synth.default_cfg(n_inp, n_tmp)
    sd = _state(dcl, cfg, wseed, dcl.DCL_Net.Network)
    pred = G.forward(sd, dict(cfg), data, mode="test")
    # same natives, same torch build: the two graphs differ only in dense-op call shapes (conv3d vs conv1d)
    assert np.abs(pred["rot_pred"].numpy() - exp["rot_pred"]).max() <= 1e-4
    assert np.abs(pred["trans_pred"].numpy() - exp["trans_pred"]).max() <= 1e-5
    assert np.abs(pred["conf"].numpy() - exp["conf"]).max() <= 1e-5
    ref_sub = exp["F_Xo_p_sub"]
    got_sub = pred["F_Xo_p"][:, ::8, ::8].numpy()
    assert np.abs(got_sub - ref_sub).max() <= 1e-4 * max(1.0, np.abs(ref_sub).max())
    assert data["labels"]["points_inp"].shape == (b, n_inp, 3) and data["labels"]["points_tmp"].shape == (b, n_tmp, 3)


def test_forward_matches_reference_graph_at_the_stress_shape(dcl, oracle, golden_dir):
    """one crop of BASELINE configs[1] (N = 12288, M = 2048): oracle/graph.py against the reference's own Network"""
    vox = lambda c, bs, mode: tuple(torch.from_numpy(a) for a in oracle.voxelize_idx(c.numpy(), bs, mode))   # noqa: E731
    data, exp, (b, n_inp, n_tmp, wseed) = load_stress_golden(dcl, os.path.join(golden_dir, "dclnet_stress_b1.npz"), vox)
    cfg = dcl.synth.default_cfg(n_inp, n_tmp)
    sd = _state(dcl, cfg, wseed, dcl.DCL_Net.Network)
    pred = G.forward(sd, dict(cfg), data, mode="test")
    assert np.abs(pred["rot_pred"].numpy() - exp["rot_pred"]).max() <= 1e-4
    assert np.abs(pred["trans_pred"].numpy() - exp["trans_pred"]).max() <= 1e-5
    assert np.abs(pred["conf"].numpy() - exp["conf"]).max() <= 1e-5
    sub = pred["F_Xo_p"][:, ::8, ::64].numpy()
    assert np.abs(sub - exp["F_Xo_p_sub"]).max() <= 1e-4 * max(1.0, np.abs(exp["F_Xo_p_sub"]).max())


def test_synth_inputs_are_reproducible(dcl, oracle, golden_dir):
    """the procedural crops regenerate bit-identically (fixture inputs == synth.make_batch)."""
    data, _, (b, n_inp, n_tmp, _) = load_golden_data(os.path.join(golden_dir, "dclnet_b2_n256.npz"))
    again = dcl.synth.make_batch(b, n_inp, n_tmp, voxelize_idx=lambda c, bs, mode: tuple(
        torch.from_numpy(a) for a in oracle.voxelize_idx(c.numpy(), bs, mode)))
    for side in ("inp", "tmp"):
        for k in ("feats", "occupied_voxels", "p2v_maps", "v2p_maps"):
            assert torch.equal(again[side][k], data[side][k]), (side, k)


def test_refiner_matches_reference(dcl, golden_dir):
    z = np.load(os.path.join(golden_dir, "refiner_b2.npz"))
    sd = _state(dcl, None, 2, dcl.refiner.Refiner)
    g = torch.Generator().manual_seed(int(z["gen_seed"][0]))
    bb, n = 2, 1024
    F = torch.randn(bb, 256, n, generator=g)
    pts = torch.randn(bb, n, 3, generator=g) * 0.05
    conf = torch.rand(bb, 2 * n, generator=g)
    o9 = torch.randn(bb, 9, generator=g)
    assert np.allclose(o9.numpy(), z["o9"])
    rot0 = G.ortho9d2matrix(o9[:, :3], o9[:, 3:6], o9[:, 6:])
    assert np.abs(rot0.numpy() - z["rot0"]).max() <= 1e-6
    trans0 = torch.from_numpy(z["trans0"])
    cur = torch.bmm(pts - trans0.unsqueeze(1), rot0)
    first = G.refiner_forward(sd, torch.cat([cur.transpose(1, 2), F], 1), conf)
    assert np.abs(first["trans_pred"].numpy() - z["dt_first"]).max() <= 1e-5
    assert np.abs(first["rot_pred"].numpy() - z["dR_first"]).max() <= 1e-4
    rot, trans = G.refine_loop(sd, {"rot_pred": rot0, "trans_pred": trans0, "F_Xo_p": F, "conf": conf}, pts, 2)
    assert np.abs(rot.numpy() - z["rot_final"]).max() <= 1e-4
    assert np.abs(trans.numpy() - z["trans_final"]).max() <= 1e-5


def test_reference_shape_and_stage2_chain_match_reference(dcl, oracle, golden_dir):
    """N = M = 1024 (what config_YCBV_bs32.yaml defines), b = 4: the oracle graph against the reference Network's
    outputs, then BASELINE configs[4] -- stage-1 outputs chained into 2 refiner iterations with pose composition
    (tools/test_YCBV_stage2.py:204-225) -- and the ADD-S distance of both poses"""
    data, exp, (b, n_inp, n_tmp, wseed) = load_golden_data(os.path.join(golden_dir, "dclnet_b4_n1024_chain.npz"))
    cfg = dcl.synth.default_cfg(n_inp, n_tmp)
    sd = _state(dcl, cfg, wseed, dcl.DCL_Net.Network)
    pred = G.forward(sd, dict(cfg), data, mode="test")
    assert np.abs(pred["rot_pred"].numpy() - exp["rot_pred"]).max() <= 1e-4
    assert np.abs(pred["trans_pred"].numpy() - exp["trans_pred"]).max() <= 1e-5
    assert np.abs(pred["conf"].numpy() - exp["conf"]).max() <= 1e-5
    got_sub = pred["F_Xo_p"][:, ::8, ::8].numpy()
    assert np.abs(got_sub - exp["F_Xo_p_sub"]).max() <= 1e-4 * max(1.0, np.abs(exp["F_Xo_p_sub"]).max())
    sdr = _state(dcl, None, int(exp["refiner_seed"][0]), dcl.refiner.Refiner)
    rot1, trans1 = G.refine_loop(sdr, pred, data["labels"]["points_inp"], 1)
    assert np.abs(rot1.numpy() - exp["rot_iter1"]).max() <= 1e-4
    assert np.abs(trans1.numpy() - exp["trans_iter1"]).max() <= 1e-5
    rot, trans = G.refine_loop(sdr, pred, data["labels"]["points_inp"], 2)
    assert np.abs(rot.numpy() - exp["rot_final"]).max() <= 1e-4
    assert np.abs(trans.numpy() - exp["trans_final"]).max() <= 1e-5
    cld = data["labels"]["points_tmp"]
    Rg, tg = torch.from_numpy(exp["rot_gt"]), torch.from_numpy(exp["trans_gt"])
    for (R, t), key in (((pred["rot_pred"], pred["trans_pred"]), "adds_stage1"), ((rot, trans), "adds_final")):
        d = dcl.sharding.add_s(cld, R, t, Rg, tg)
        assert np.abs(d.numpy() - exp[key]).max() <= 2e-5, key


def _train_cases(golden_dir):
    return [os.path.join(golden_dir, f) for f in ("dclnet_s0_train.npz", "dclnet_nm384_train.npz")]


def test_train_mode_forward_matches_reference_graph(dcl, oracle, golden_dir):
    """mode='train' outputs (Xo_pred, Yc_pred too) at BASELINE config 0's shape (N=2048, M=500, 5 mm) and at N=M=384"""
    for path in _train_cases(golden_dir):
        data, exp, (b, n_inp, n_tmp, wseed) = load_golden_data(path)
        cfg = dcl.synth.default_cfg(n_inp, n_tmp, unit=0.005)
        sd = dcl.synth.synth_state_dict(dcl.DCL_Net.Network(cfg, mode="train"), wseed)
        pred = G.forward(sd, dict(cfg), data, mode="train")
        assert np.abs(pred["rot_pred"].numpy() - exp["rot_pred"]).max() <= 1e-4
        assert np.abs(pred["trans_pred"].numpy() - exp["trans_pred"]).max() <= 1e-5
        for k in ("Xo_pred", "Yc_pred", "conf"):
            assert np.abs(pred[k].numpy() - exp[k]).max() <= 1e-4 * max(1.0, np.abs(exp[k]).max()), k


def test_objectives_match_the_reference_values(dcl, golden_dir):
    """dcl losses / losses_refiner on the reference's outputs == the reference's own loss values"""
    data, exp, (b, n_inp, n_tmp, _) = load_golden_data(os.path.join(golden_dir, "dclnet_nm384_train.npz"))
    pred = {k: torch.from_numpy(exp[k]) for k in ("rot_pred", "trans_pred", "conf", "Xo_pred", "Yc_pred")}
    pred["sym_flag"] = data["flags"]
    gt = dict(data["labels"])
    gt["points_inp"] = data["inp"]["feats"][:, 4:7].reshape(b, n_inp, 3)
    gt["points_tmp"] = data["tmp"]["feats"][:, 4:7].reshape(b, n_tmp, 3)
    lo = dcl.DCL_Net.losses(None)(pred, gt)
    got = np.array([float(lo[k]) for k in ("loss_pose", "loss_Xo", "loss_Yc", "loss_conf", "loss_all")])
    assert np.abs(got - exp["losses"]).max() <= 1e-5 * max(1.0, np.abs(exp["losses"]).max())
    pr = {"rot_pred": pred["rot_pred"].transpose(1, 2).contiguous(), "trans_pred": -pred["trans_pred"] * 0.5}
    lr = dcl.refiner.losses_refiner(None)(pr, pred["trans_pred"], pred["rot_pred"], gt["points_tmp"], pred["sym_flag"], gt)
    assert abs(float(lr["loss_all"]) - float(exp["loss_refiner"][0])) <= 1e-6

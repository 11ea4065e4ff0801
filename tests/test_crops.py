"""Crop builder: oracle / host logic on CPU, device builder vs oracle bit-exact on the GPU."""
import numpy as np
import pytest
import torch

from crop_scene import make_scene

CFG = dict(input_size=96, tmp_size=64, unit_voxel_extent=[0.006] * 3, voxel_num_limit=[64] * 3, voxelization_mode=4)


def _oracle_build(sc, cfg, seed):
    from oracle import crops as oc
    np.random.seed(seed)
    return oc.build_image(sc["img"], sc["depth"], sc["label"], sc["rois"], sc["gt_obj"], sc["cad_pts"], sc["cad_col"], cfg,
                          poses=sc["poses"])


def test_numpy_mean_axis0_is_a_row_order_running_sum():
    """the device centroid relies on it (oracle/crops.py, csrc/crops.hip step 2)"""
    rng = np.random.default_rng(0)
    for n in (1, 7, 1000, 33333):
        a = (rng.normal(size=(n, 3)) * 0.05 + np.array([0.1, -0.2, 0.9])).astype(np.float32)
        s = np.zeros(3, np.float32)
        for i in range(n):
            s = s + a[i]
        assert np.array_equal(np.mean(a, axis=0), (s / np.float32(n)).astype(np.float32))


def test_snap_box_matches_oracle(dcl):
    from oracle import crops as oc
    rng = np.random.default_rng(1)
    for _ in range(3000):
        x1, y1 = rng.integers(-30, 640), rng.integers(-30, 480)
        roi = np.array([[0, 1, x1, y1, x1 + rng.integers(1, 700), y1 + rng.integers(1, 600)]], np.float64)
        assert dcl.crops.snap_box(roi, 0) == oc.get_bbox(roi, 0)
    for side in (0, 1, 39, 40, 41, 80, 679, 680, 681):                       # the border values themselves
        roi = np.array([[0, 1, 99, 99, 101 + side, 101 + side]], np.float64)
        assert dcl.crops.snap_box(roi, 0) == oc.get_bbox(roi, 0)


def test_oracle_crops_are_well_formed():
    sc = make_scene(3, n_obj=4, tmp_size=CFG["tmp_size"], tiny=1, empty=2, undetected=3)
    d = _oracle_build(sc, CFG, 5)
    assert d["all_flags"].tolist() == [1, 1, 0, 0]
    b = 2
    assert d["inp"]["feats"].shape == (b * CFG["input_size"], 7) and d["tmp"]["feats"].shape == (b * CFG["tmp_size"], 7)
    assert d["counts"][1][1] <= 32 and d["counts"][1][2] == d["counts"][1][0]      # tiny instance: nothing filtered
    assert d["counts"][0][1] > 32 and d["counts"][0][2] == d["counts"][0][1]
    c = d["inp"]["coords"]
    assert int(c[:, 1:].min()) >= 0 and int(c[:, 1:].max()) <= 63
    assert torch.equal(d["inp"]["feats"][:, 0], torch.ones(b * CFG["input_size"]))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,kw", [(11, {}), (12, dict(tiny=0)), (13, dict(empty=1, undetected=2, rgba=True)),
                                      (14, dict(n_obj=6, tiny=5))])
def test_device_crop_builder_bit_exact(dcl, seed, kw):
    cfg = dict(CFG)
    if seed == 14:
        cfg["input_size"] = 4000                                             # more samples than points: replace=True path
    sc = make_scene(seed, tmp_size=cfg["tmp_size"], **kw)
    want = _oracle_build(sc, cfg, 100 + seed)
    builder = dcl.crops.CropBuilder(cfg, sc["cad_pts"], sc["cad_col"])
    np.random.seed(100 + seed)
    got = builder.build(sc["img"], sc["depth"], sc["label"], sc["rois"], sc["gt_obj"], poses=sc["poses"])
    after_device = np.random.random()                      # both builders must leave the global RNG stream at the same place
    _oracle_build(sc, cfg, 100 + seed)
    assert after_device == np.random.random()
    assert np.array_equal(got["counts"], want["counts"])
    assert got["all_flags"].tolist() == want["all_flags"].tolist()
    assert torch.equal(got["all_centroids"].cpu(), want["all_centroids"])
    for side in ("inp", "tmp"):
        for k in ("feats", "coords", "occupied_voxels", "p2v_maps", "v2p_maps"):
            assert torch.equal(got[side][k].cpu(), want[side][k]), (side, k)
    assert torch.equal(got["labels"]["rot_gt"].cpu(), want["labels"]["rot_gt"])        # (device tensors: no centroid read-back)
    assert torch.equal(got["labels"]["trans_gt"].cpu(), want["labels"]["trans_gt"])


def test_lm_box_matches_oracle(dcl):
    from oracle import crops as oc
    rng = np.random.default_rng(2)
    for _ in range(3000):
        bb = [int(rng.integers(-20, 640)), int(rng.integers(-20, 480)), int(rng.integers(1, 700)), int(rng.integers(1, 600))]
        assert dcl.crops.lm_box(bb) == oc.lm_get_bbox(bb)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,eval_mode,small", [(31, False, False), (32, True, False), (33, False, True), (34, True, True)])
def test_device_lm_sample_bit_exact(dcl, seed, eval_mode, small):
    """LineMOD loader arithmetic (millimetre depth, /1000 after back-projection, 128-point rule, eval-mode filtering)"""
    from oracle import crops as oc
    cfg = dict(CFG, unit_voxel_extent=[0.005] * 3, input_size=128)
    sc = make_scene(seed, n_obj=1, tmp_size=cfg["tmp_size"], tiny=0 if small else None)
    cls = int(sc["gt_obj"][0])
    mask_label = sc["label"] == cls
    depth = (sc["depth"].astype(np.float64) / 10).astype(np.uint16)              # ~0.6-1.4 m in millimetres
    ys, xs = np.nonzero(mask_label)
    bb = [int(xs.min()) - 3, int(ys.min()) - 2, int(xs.max() - xs.min()) + 7, int(ys.max() - ys.min()) + 5]
    builder = dcl.crops.CropBuilder(cfg, sc["cad_pts"], sc["cad_col"], camera=dcl.crops.LM_CAMERA)
    np.random.seed(seed)
    want = oc.build_lm_sample(sc["img"], depth, mask_label, bb, cls, sc["cad_pts"], sc["cad_col"], cfg, eval_mode)
    np.random.seed(seed)
    got = builder.build_lm(sc["img"], depth, mask_label, bb, cls, eval_mode)
    if want is None:
        assert got is None and small and not eval_mode
        return
    for g, w in zip(got, want):
        assert torch.equal(g.cpu(), w)


@pytest.mark.gpu
def test_built_crops_run_through_the_network(dcl):
    """image -> device crop builder -> Network.forward, everything resident on the GPU"""
    cfg = dict(CFG, input_size=256, tmp_size=256)
    sc = make_scene(21, n_obj=3, tmp_size=256)
    for c in sc["cad_pts"]:
        sc["cad_pts"][c] = sc["cad_pts"][c] * 0.8
    builder = dcl.crops.CropBuilder(cfg, sc["cad_pts"], sc["cad_col"])
    np.random.seed(7)
    data = builder.build(sc["img"], sc["depth"], sc["label"], sc["rois"], sc["gt_obj"])
    net = dcl.DCL_Net.Network(dcl.synth.default_cfg(256, 256), mode="test", graph_max_batch=0)
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
    net = net.cuda().eval()
    with torch.no_grad():
        out = net(data)
    b = int(data["all_flags"].sum())
    assert out["rot_pred"].shape == (b, 3, 3) and torch.isfinite(out["rot_pred"]).all()
    # the same crops through a pipelining network: its side streams wait on the builder's ready_event only
    anet = dcl.DCL_Net.Network(dcl.synth.default_cfg(256, 256), mode="test", async_inputs=True, graph_max_batch=0)
    anet.load_state_dict(dcl.synth.synth_state_dict(anet, 1))
    anet = anet.cuda().eval()
    for _ in range(3):
        np.random.seed(7)
        d2 = builder.build(sc["img"], sc["depth"], sc["label"], sc["rois"], sc["gt_obj"])
        assert d2["ready_event"] is not None
        with torch.no_grad():
            out2 = anet(d2)
        assert torch.equal(out2["rot_pred"], out["rot_pred"]) and torch.equal(out2["trans_pred"], out["trans_pred"])
    want = _oracle_build(sc, cfg, 7)
    with torch.no_grad():
        ref = net({k: v for k, v in want.items()})
    assert float((out["rot_pred"] - ref["rot_pred"]).abs().max()) <= 1e-6
    assert float((out["trans_pred"] - ref["trans_pred"]).abs().max()) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("b,n_per,span", [(1, 1024, 64), (3, 1024, 20), (6, 1024, 40), (17, 500, 12), (64, 257, 64), (5, 1, 64)])
def test_one_launch_crop_voxelisation_equals_the_general_device_op(dcl, b, n_per, span):
    """ops.voxelize_idx_crops (one workgroup per crop, capacity-pitched outputs, no read-back) == ops.voxelize_idx_gpu on
    the builder's layout: V, maxActive, voxel rows, point -> voxel map and the ascending, zero-padded point lists bit for
    bit; the capacity rows behind the V live ones are zero; a voxel beyond the pitch is flagged, not silently dropped"""
    ops = dcl.ops
    rng = np.random.default_rng(b * 1000 + n_per)
    xyz = rng.integers(0, span, size=(b, n_per, 3))                      # small spans: many points per voxel, long rows
    coords = torch.from_numpy(np.concatenate([np.repeat(np.arange(b), n_per)[:, None], xyz.reshape(-1, 3)], 1).astype(np.int64)).cuda()
    occ_w, p2v_w, v2p_w = ops.voxelize_idx_gpu(coords, b, 64, 4)
    V, ma = occ_w.shape[0], v2p_w.shape[1] - 1
    pitch = ma + 3
    for dtype in (torch.int64, torch.int32):
        occ, p2v, v2p, info = ops.voxelize_idx_crops(coords, b, n_per, 64, 4, pitch=pitch, occ_dtype=dtype)
        assert info.cpu().tolist() == [V, ma, 0]
        assert torch.equal(occ[:V].long(), occ_w) and torch.equal(p2v, p2v_w)
        assert torch.equal(v2p[:V, :ma + 1], v2p_w) and int(v2p[:V, ma + 1:].abs().sum()) == 0
        assert int(v2p[V:].abs().sum()) == 0 and int(occ[V:].abs().sum()) == 0
        occ2, p2v2, v2p2, info2 = ops.voxelize_idx_crops(coords, b, n_per, 64, 4, pitch=pitch, occ_dtype=dtype)   # next generation
        assert torch.equal(v2p2[:V], v2p[:V]) and torch.equal(p2v2, p2v) and info2.cpu().tolist() == [V, ma, 0]
    if ma >= 2:
        _, _, _, info = ops.voxelize_idx_crops(coords, b, n_per, 64, 4, pitch=ma)           # one column short
        assert info.cpu().tolist()[2] == 1
    bad = coords.clone()
    bad[0, 1] = 64                                                                         # outside the grid
    _, _, _, info = ops.voxelize_idx_crops(bad, b, n_per, 64, 4, pitch=pitch)
    assert info.cpu().tolist()[2] == 1


@pytest.mark.gpu
def test_capacity_form_crops_give_the_same_poses(dcl):
    """CropBuilder(capacity=True): one host read-back per frame instead of two -- the observed side's voxel rows stay
    capacity-shaped with their count on the device.  Through the network's graph path (which takes that form as it is) and
    through the launch-by-launch path (which converts: exact_form) the poses equal those of the exact-form crops."""
    cfg = dict(CFG, input_size=256, tmp_size=256)
    sc = make_scene(23, n_obj=4, tmp_size=256)
    for c in sc["cad_pts"]:
        sc["cad_pts"][c] = sc["cad_pts"][c] * 0.8
    exact_b = dcl.crops.CropBuilder(cfg, sc["cad_pts"], sc["cad_col"])
    cap_b = dcl.crops.CropBuilder(cfg, sc["cad_pts"], sc["cad_col"], capacity=True)
    np.random.seed(3)
    want_d = exact_b.build(sc["img"], sc["depth"], sc["label"], sc["rois"], sc["gt_obj"])
    np.random.seed(3)
    cap_d = cap_b.build(sc["img"], sc["depth"], sc["label"], sc["rois"], sc["gt_obj"])
    assert "v0_dev" in cap_d["inp"] and cap_d["inp"]["occupied_voxels"].shape[0] == cap_d["inp"]["feats"].shape[0]
    back = dcl.crops.exact_form(cap_d)
    for k in ("feats", "occupied_voxels", "p2v_maps", "v2p_maps"):
        assert torch.equal(back["inp"][k], want_d["inp"][k]), k
    for graph in (8, 0):
        net = dcl.DCL_Net.Network(dcl.synth.default_cfg(256, 256), mode="test", graph_max_batch=graph)
        net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
        net = net.cuda().eval()
        with torch.no_grad():
            want = net(dict(want_d, labels={}))
            got = net(dict(cap_d, labels={}))
            again = net(dict(cap_d, labels={}))
        for k in ("rot_pred", "trans_pred", "conf"):
            assert torch.equal(want[k], got[k]) and torch.equal(want[k], again[k]), (graph, k)
        if graph:       # the graph path never reads the crops' error flag on the way in: it hands it over with the poses
            assert "crop_overflow" in got and got["crop_overflow"].is_cuda and int(got["crop_overflow"][0]) == 0
            assert "crop_overflow" not in want


@pytest.mark.gpu
def test_prefetcher_yields_the_serial_loops_crops(dcl):
    """crops.CropPrefetcher (the loader workers' role, tools/test_YCBV_stage1.py:133-137): a builder thread two frames ahead of
    the caller yields, frame by frame, exactly what a serial loop over CropBuilder.build yields under the same seed; an error
    in the builder thread surfaces in the caller"""
    cfg = dict(CFG, input_size=256, tmp_size=256)
    scenes = [make_scene(30 + i, n_obj=3, tmp_size=256) for i in range(3)]
    builder = dcl.crops.CropBuilder(cfg, scenes[0]["cad_pts"], scenes[0]["cad_col"])
    order = [0, 1, 2, 1, 0, 2, 2]
    args = lambda i: (scenes[i]["img"], scenes[i]["depth"], scenes[i]["label"], scenes[i]["rois"], scenes[i]["gt_obj"])  # noqa: E731
    np.random.seed(11)
    want = [builder.build(*args(i)) for i in order]
    torch.cuda.synchronize()
    np.random.seed(11)
    n = 0
    with dcl.crops.CropPrefetcher(builder, (args(i) for i in order), depth=2) as feed:
        for got, ref in zip(feed, want):
            for side in ("inp", "tmp"):
                for k in ("feats", "occupied_voxels", "p2v_maps", "v2p_maps"):
                    assert torch.equal(got[side][k], ref[side][k]), (n, side, k)
            assert torch.equal(got["all_flags"], ref["all_flags"])
            n += 1
    assert n == len(order)
    bad = (args(0), (None,) * 5)
    with dcl.crops.CropPrefetcher(builder, bad, depth=2) as feed:
        next(feed)
        with pytest.raises(Exception):
            next(feed)


@pytest.mark.gpu
def test_device_crop_builder_timing(dcl, capsys):
    """reported number: device builder vs the CPU restatement of the loader on one 6-object frame"""
    import time
    cfg = dict(CFG, input_size=1024, tmp_size=1024)
    sc = make_scene(5, n_obj=6, tmp_size=1024)
    builder = dcl.crops.CropBuilder(cfg, sc["cad_pts"], sc["cad_col"])
    res = {}
    for name, fn, reps in (("device", lambda: builder.build(sc["img"], sc["depth"], sc["label"], sc["rois"], sc["gt_obj"]), 20),
                           ("cpu", lambda: _oracle_build(sc, cfg, 1), 3)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / reps * 1e3
    with capsys.disabled():
        print("\ncrop builder: device %.2f ms, CPU restatement %.2f ms per frame" % (res["device"], res["cpu"]))
    assert res["device"] < res["cpu"]


# ---- pins against the REFERENCE's own loader code (tests/golden/make_crops_golden.py ran YCBDataset.__getitem__ and the
# LineMOD Dataset.__getitem__ unmodified, file I/O stubbed, on these same seeded scenes)
def _ref_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "crops_ref.npz"))


YCBV_REF = [(11, {}), (12, dict(tiny=0)), (13, dict(empty=1, undetected=2))]
LM_REF = [(31, False, False), (32, True, False), (33, False, True), (34, True, True)]


def _check_ycbv(got, z, tag):
    for side in ("inp", "tmp"):
        for k in ("feats", "occupied_voxels", "p2v_maps", "v2p_maps"):
            assert np.array_equal(got[side][k].cpu().numpy(), z[tag + side + "_" + k]), (tag, side, k)
    assert np.array_equal(got["all_centroids"].cpu().numpy(), z[tag + "centroids"])
    assert np.array_equal(got["all_flags"].numpy(), z[tag + "flags"])
    assert np.array_equal(got["labels"]["rot_gt"].cpu().numpy(), z[tag + "rot_gt"])
    assert np.array_equal(got["labels"]["trans_gt"].cpu().numpy(), z[tag + "trans_gt"])


@pytest.mark.parametrize("seed,kw", YCBV_REF)
def test_oracle_ycbv_crops_equal_the_reference_loader(seed, kw):
    sc = make_scene(seed, tmp_size=CFG["tmp_size"], **kw)
    _check_ycbv(_oracle_build(sc, CFG, 100 + seed), _ref_golden(), "ycbv%d_" % seed)


def _lm_case(seed, small):
    cfg = dict(CFG, unit_voxel_extent=[0.005] * 3, input_size=128)
    sc = make_scene(seed, n_obj=1, tmp_size=cfg["tmp_size"], tiny=0 if small else None)
    cls = int(sc["gt_obj"][0])
    mask_label = sc["label"] == cls
    depth = (sc["depth"].astype(np.float64) / 10).astype(np.uint16)
    ys, xs = np.nonzero(mask_label)
    bb = [int(xs.min()) - 3, int(ys.min()) - 2, int(xs.max() - xs.min()) + 7, int(ys.max() - ys.min()) + 5]
    return cfg, sc, cls, mask_label, depth, bb


def _check_lm(got, z, tag):
    if float(z[tag + "flag"][0]) == -1:
        assert got is None
        return
    for g, k in zip(got, ("feat_inp", "vox_inp", "feat_tmp", "vox_tmp", "centroid")):
        assert np.array_equal(g.cpu().numpy(), z[tag + k]), (tag, k)


@pytest.mark.parametrize("seed,eval_mode,small", LM_REF)
def test_oracle_lm_sample_equals_the_reference_loader(seed, eval_mode, small):
    from oracle import crops as oc
    cfg, sc, cls, mask_label, depth, bb = _lm_case(seed, small)
    np.random.seed(seed)
    got = oc.build_lm_sample(sc["img"], depth, mask_label, bb, cls, sc["cad_pts"], sc["cad_col"], cfg, eval_mode)
    _check_lm(got, _ref_golden(), "lm%d_" % seed)


@pytest.mark.gpu
def test_device_builders_equal_the_reference_loaders(dcl):
    z = _ref_golden()
    for seed, kw in YCBV_REF:
        sc = make_scene(seed, tmp_size=CFG["tmp_size"], **kw)
        builder = dcl.crops.CropBuilder(CFG, sc["cad_pts"], sc["cad_col"])
        np.random.seed(100 + seed)
        _check_ycbv(builder.build(sc["img"], sc["depth"], sc["label"], sc["rois"], sc["gt_obj"], poses=sc["poses"]), z,
                    "ycbv%d_" % seed)
    for seed, eval_mode, small in LM_REF:
        cfg, sc, cls, mask_label, depth, bb = _lm_case(seed, small)
        builder = dcl.crops.CropBuilder(cfg, sc["cad_pts"], sc["cad_col"], camera=dcl.crops.LM_CAMERA)
        np.random.seed(seed)
        _check_lm(builder.build_lm(sc["img"], depth, mask_label, bb, cls, eval_mode), z, "lm%d_" % seed)

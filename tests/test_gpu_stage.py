"""The feature stage of both backbones as ONE launch (csrc/feature_stage.hip, ops.backbone_features_stage,
Network(feature_stage=True)) against the per-layer launches it can replace (csrc/sparse_conv.hip through
dcl_backbone_features).  Its work items are the per-layer launches' workgroups -- same segments, same summation order -- so
with the same number of work items per layer (slots = 512, what a one-sided per-layer launch uses) and the same kernel family
per layer the two paths must agree BIT FOR BIT: at one-image sizes (few-row tiles + deferred combines), mid sizes, and at 16+
crops (ordered rows, used-chunk dealing).  One layer differs by design in the product build -- the one-sided per-layer launch
of the 32 -> 32 layer takes the filter-resident kernel (256 registers, 108 KiB of LDS: no room in the stage), the stage the
LDS-DMA kernel -- so the word-for-word comparison at sizes where that layer has many rows runs in the diagnostic library with
the filter-resident kernels switched off on both sides; the product build is compared at BASELINE's pose gates.  The bounded
waits are exercised with a budget of one poll: the launch must end, the status word must say so when a wait gave up, and the
instance must fall back to per-layer launches with the right bits.
Reference: Backbone_SPCONV.forward (models/Modules.py:153-159), spconv_ops.h:284-344, pool_ops.h:170-208."""
import copy

import pytest
import torch

from test_gpu_ops import enter_diag

pytestmark = pytest.mark.gpu


def _sides(dcl, b, n, seed=1, first=0):
    ops = dcl.ops
    cfg = dcl.synth.default_cfg(n, n)
    net = dcl.DCL_Net.Network(cfg, mode="test")
    net.load_state_dict(dcl.synth.synth_state_dict(net, seed))
    net = net.cuda().eval()
    f = net._fold()
    data = dcl.synth.make_batch(b, n, n, first=first)
    runs, xs, ptrs = [], [], []
    for s in ("inp", "tmp"):
        occ = data[s]["occupied_voxels"].int().cuda().contiguous()
        xs.append(ops.voxelize_fp(data[s]["feats"].cuda(), data[s]["v2p_maps"].cuda(), 4))
        run = ops.BackboneRun(occ, b, 64)
        run.set_counts(run.counts_dev.cpu().tolist())
        runs.append(run)
        ptrs.append(f["backbone_%s_ptrs" % s])
    return net, runs, xs, ptrs


def _levels(runs):
    torch.cuda.synchronize()
    return [[t.clone() for t in r.levels] for r in runs]


@pytest.mark.parametrize("b,n,wlds_off", [(1, 1024, False), (2, 512, False), (1, 1024, True), (6, 1024, True), (16, 1024, True),
                                          (33, 512, True)])
def test_stage_equals_per_layer_launches_bit_for_bit(dcl, request, b, n, wlds_off):
    ops = dcl.ops
    if wlds_off:
        lib = enter_diag(dcl, request)
        lib.dcl_debug_conv_wlds(0)
        request.addfinalizer(lambda: lib.dcl_debug_conv_wlds(1))
    net, runs, xs, ptrs = _sides(dcl, b, n)
    status = ops.stage_status_buffer()
    for r, x, p in zip(runs, xs, ptrs):
        r.features(x, *p)                                             # one launch per layer and side
    want = _levels(runs)
    for rep in range(3):                                              # both sides in one launch; then a launch per side
        assert ops.backbone_features_stage(runs, xs, ptrs, status, slots=512)
        got = _levels(runs)
        assert int(status[0]) == 0
        for side in range(2):
            for m in range(4):
                assert torch.equal(got[side][m], want[side][m]), (rep, side, m)
    for side in range(2):
        assert ops.backbone_features_stage(runs[side:side + 1], xs[side:side + 1], ptrs[side:side + 1], status, slots=512)
    got = _levels(runs)
    assert int(status[0]) == 0
    for side in range(2):
        for m in range(4):
            assert torch.equal(got[side][m], want[side][m]), ("per side", side, m)


@pytest.mark.parametrize("b,graph", [(2, 0), (2, 8), (16, 0), (20, 64)])
def test_network_with_staged_features_within_pose_gates(dcl, b, graph):
    """whole forward, product library, default work items per side: only the split points of the fp32 sums (and the 32 -> 32
    layer's kernel family) differ from the per-layer launches"""
    n = 1024
    cfg = dcl.synth.default_cfg(n, n)
    ref = dcl.DCL_Net.Network(cfg, mode="test", graph_max_batch=graph)
    sd = dcl.synth.synth_state_dict(ref, 5)
    ref.load_state_dict(sd)
    stg = dcl.DCL_Net.Network(cfg, mode="test", graph_max_batch=graph, feature_stage=True)
    stg.load_state_dict(sd)
    ref, stg = ref.cuda().eval(), stg.cuda().eval()
    for it in range(2):
        data = dcl.synth.make_batch(b, n, n, first=3 * it)
        with torch.no_grad():
            want, got = ref(copy.deepcopy(data)), stg(copy.deepcopy(data))
        torch.cuda.synchronize()
        assert stg.check_feature_stage()
        assert float((want["rot_pred"] - got["rot_pred"]).abs().max()) <= 1e-4
        assert float((want["trans_pred"] - got["trans_pred"]).abs().max()) <= 1e-5
        assert float((want["conf"] - got["conf"]).abs().max()) <= 1e-4


def test_stage_soak_under_uneven_load(dcl, request):
    """many stages of changing size back to back while a second stream keeps the GPU unevenly busy: every word of the levels
    equal to the per-layer path (the hand-offs must hold under load, consumer caches warm).  Diagnostic library with the
    filter-resident kernels off, so that every size is bit-comparable (see the module docstring)."""
    ops = dcl.ops
    lib = enter_diag(dcl, request)
    lib.dcl_debug_conv_wlds(0)
    request.addfinalizer(lambda: lib.dcl_debug_conv_wlds(1))
    noise_stream = torch.cuda.Stream()
    big = torch.randn(4096, 4096, device="cuda")
    status = ops.stage_status_buffer()
    for it in range(18):
        b = (1, 5, 18, 2, 9, 32)[it % 6]
        net, runs, xs, ptrs = _sides(dcl, b, 512, seed=11, first=it)
        for r, x, p in zip(runs, xs, ptrs):
            r.features(x, *p)
        want = _levels(runs)
        for rep in range(3):
            with torch.cuda.stream(noise_stream):
                for _ in range((it + rep) % 3):
                    big = (big @ big).clamp_(-1, 1)
            assert ops.backbone_features_stage(runs, xs, ptrs, status, slots=512)
            got = _levels(runs)
            assert int(status[0]) == 0
            for side in range(2):
                for m in range(4):
                    assert torch.equal(got[side][m], want[side][m]), (it, rep, side, m)


def test_bounded_wait_ends_the_launch_and_the_network_falls_back(dcl):
    """a budget of ONE poll per wait: the launch ends all the same; if a wait gave up the status word says so, the instance
    switches to per-layer launches, and the repeated call has the per-layer bits"""
    b, n = 16, 1024
    cfg = dcl.synth.default_cfg(n, n)
    ref = dcl.DCL_Net.Network(cfg, mode="test", graph_max_batch=0)      # what an instance without the stage runs
    sd = dcl.synth.synth_state_dict(ref, 3)
    ref.load_state_dict(sd)
    stg = dcl.DCL_Net.Network(cfg, mode="test", graph_max_batch=0, feature_stage=True, stage_spin_limit=1)
    stg.load_state_dict(sd)
    ref, stg = ref.cuda().eval(), stg.cuda().eval()
    data = dcl.synth.make_batch(b, n, n)
    with torch.no_grad():
        want = ref(copy.deepcopy(data))
        stg(copy.deepcopy(data))
    torch.cuda.synchronize()                                     # ... returns: no hang
    if stg.check_feature_stage():                                # every wait happened to be satisfied at its first poll
        return
    assert not stg.feature_stage                                 # fell back for good: per-layer launches from here on
    with torch.no_grad():
        again = stg(copy.deepcopy(data))
    torch.cuda.synchronize()
    for k in ("rot_pred", "trans_pred", "conf", "F_Xo_p"):
        assert torch.equal(want[k], again[k]), k


def test_forward_raises_after_an_unnoticed_timeout(dcl):
    b, n = 2, 256
    cfg = dcl.synth.default_cfg(n, n)
    stg = dcl.DCL_Net.Network(cfg, mode="test", graph_max_batch=0, feature_stage=True)
    stg.load_state_dict(dcl.synth.synth_state_dict(stg, 3))
    stg = stg.cuda().eval()
    data = dcl.synth.make_batch(b, n, n)
    with torch.no_grad():
        stg(copy.deepcopy(data))
    torch.cuda.synchronize()
    stg._stage_status[0] = 1                                     # what a timed-out wait writes
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            stg(copy.deepcopy(data))
    assert not stg.feature_stage
    with torch.no_grad():
        stg(copy.deepcopy(data))                                 # per-layer launches from here on


def test_stage_refuses_layers_it_has_no_form_for(dcl):
    """channel lists outside the backbone's family: DCL_ESTAGE_UNSUPPORTED, nothing launched (the caller takes the per-layer path)"""
    import ctypes as C
    ops = dcl.ops
    net, runs, xs, ptrs = _sides(dcl, 2, 256)
    status = ops.stage_status_buffer()
    old = runs[0].chan
    try:
        for r in runs:
            r.chan = (C.c_int32 * 9)(7, 16, 32, 32, 64, 64, 128, 128, 48)       # pool over 48 channels: 12 quads do not divide 512
        nb = C.c_int64(0)
        assert ops.backbone_features_stage(runs, xs, ptrs, status) is False
    finally:
        for r in runs:
            r.chan = old
    assert int(status[0]) == 0

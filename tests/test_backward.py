"""Training-side kernels (csrc/backward.hip, dcl-net_amd/autograd.py) vs the numpy restatement of the reference's
backward code (oracle/backward.py)."""
import numpy as np
import pytest
import torch

from test_gpu_ops import cuda, rand_voxels


def test_oracle_conv_backward_is_the_gradient_of_the_oracle_forward(oracle):
    """ties oracle/backward.py to the pinned forward oracle: <dOut, conv(X+eps*dX)> directional derivative in float64-ish"""
    from oracle import backward as ob
    rng = np.random.default_rng(0)
    b, S, cin, cout = 1, 6, 5, 4
    idx = rand_voxels(rng, b, S, 40)
    for subm in (False, True):
        out_idx, pairs, num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, 3, 1, 1, 1, subm=subm)
        n_out = idx.shape[0] if subm else out_idx.shape[0]
        X = rng.normal(size=(idx.shape[0], cin)).astype(np.float32)
        W = rng.normal(size=(3, 3, 3, cin, cout)).astype(np.float32)
        G = rng.normal(size=(n_out, cout)).astype(np.float32)
        dX, dW = ob.indice_conv_backward(X, W, G, pairs, num, subm)
        # conv is bilinear: <G, conv(X, W)> = <dX, X> = <dW, W>
        val = float((G.astype(np.float64) * oracle.indice_conv(X, W, pairs, num, n_out, subm=subm)).sum())
        assert abs(float((dX.astype(np.float64) * X).sum()) - val) <= 1e-3 * max(1.0, abs(val))
        assert abs(float((dW.astype(np.float64) * W).sum()) - val) <= 1e-3 * max(1.0, abs(val))


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,subm", [(7, 16, False), (16, 32, True), (32, 32, False), (32, 64, True),
                                           (64, 128, True), (128, 128, False), (128, 256, True)])
def test_sparse_conv_backward_matches_oracle(dcl, oracle, cin, cout, subm):
    from oracle import backward as ob
    rng = np.random.default_rng(cin * 3 + cout)
    b, S = 2, 8
    idx = rand_voxels(rng, b, S, 150)
    feat = rng.normal(size=(idx.shape[0], cin)).astype(np.float32)
    W = (rng.normal(size=(3, 3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    aset = dcl.ops.grid_from_indices(cuda(idx), b, S)
    out, nbr = dcl.spconv.ops.build_rulebook(aset, 3, 1, 1, subm)
    n_out = idx.shape[0] if subm else out.n
    _, r_pairs, r_num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, 3, 1, 1, 1, subm=subm)
    G = rng.normal(size=(n_out, cout)).astype(np.float32)
    want_dx, want_dw = ob.indice_conv_backward(feat, W, G, r_pairs, r_num, subm)
    Wd = cuda(W).reshape(27, cin, cout).contiguous()
    dx, dw = dcl.ops.sparse_conv_backward(cuda(feat), Wd, cuda(G), nbr, n_out, subm)
    assert np.abs(dx.cpu().numpy() - want_dx).max() <= 2e-5 * max(1.0, np.abs(want_dx).max())
    assert np.abs(dw.cpu().numpy().reshape(W.shape) - want_dw).max() <= 5e-5 * max(1.0, np.abs(want_dw).max())
    # through autograd, as the module mirror calls it
    f = cuda(feat).requires_grad_(True)
    w = Wd.clone().requires_grad_(True)
    y = dcl.autograd.SparseConvFn.apply(f, w, nbr, n_out, subm)
    y.backward(cuda(G))
    assert torch.equal(f.grad, dx) and torch.equal(w.grad, dw)


@pytest.mark.gpu
def test_avgpool_backward_bit_exact(dcl, oracle):
    from oracle import backward as ob
    rng = np.random.default_rng(3)
    b, S, c = 2, 16, 32
    idx = rand_voxels(rng, b, S, 300)
    aset = dcl.ops.grid_from_indices(cuda(idx), b, S)
    out, nbr = dcl.spconv.ops.build_rulebook(aset, 3, 2, 1, False)
    o_idx, pairs, num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, 3, 2, 1, 1, subm=False)
    feat = rng.normal(size=(idx.shape[0], c)).astype(np.float32)
    _, rf = oracle.indice_avgpool(feat, pairs, num, o_idx.shape[0])
    G = rng.normal(size=(o_idx.shape[0], c)).astype(np.float32)
    want = ob.indice_avgpool_backward(idx.shape[0], G, pairs, num, rf)
    f = cuda(feat).requires_grad_(True)
    y = dcl.autograd.SparseAvgPoolFn.apply(f, nbr, out.n)
    y.backward(cuda(G))
    assert np.array_equal(f.grad.cpu().numpy(), want)


@pytest.mark.gpu
def test_interpolate_and_voxelize_backward(dcl, oracle):
    from oracle import backward as ob
    rng = np.random.default_rng(4)
    n, m, c = 700, 90, 64
    idx = rng.integers(0, m, (n, 3)).astype(np.int32)
    w = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    feats = rng.normal(size=(m, c)).astype(np.float32)
    G = rng.normal(size=(n, c)).astype(np.float32)
    f = cuda(feats).requires_grad_(True)
    y = dcl.autograd.ThreeInterpolateFn.apply(f, cuda(idx), cuda(w))
    y.backward(cuda(G))
    want = ob.three_interpolate_grad(G, idx, w, m)
    assert np.abs(f.grad.cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())      # atomic order
    # voxelization (mean): every point belongs to exactly one voxel -> exact
    coords = np.concatenate([np.zeros((500, 1), np.int64), rng.integers(0, 6, (500, 3))], 1)
    occ, p2v, v2p = oracle.voxelize_idx(coords, 1, 4)
    pf = rng.normal(size=(500, 7)).astype(np.float32)
    Gv = rng.normal(size=(occ.shape[0], 7)).astype(np.float32)
    p = cuda(pf).requires_grad_(True)
    dcl.autograd.VoxelizationFn.apply(p, cuda(v2p), 4).backward(cuda(Gv))
    assert np.array_equal(p.grad.cpu().numpy(), ob.voxelize_bp(Gv, v2p, 500, True))


@pytest.mark.gpu
def test_network_trains_on_the_gpu(dcl):
    """mode='train' composes the module mirrors (autograd through every custom op): all parameters receive finite
    gradients and a few SGD steps on a fixed batch reduce a pose loss"""
    n = 256
    net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n, n), mode="train")
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
    net = net.cuda().train()
    data = dcl.synth.make_batch(3, n, n)
    data["flags"] = torch.IntTensor([0, 0, 0])
    rot_gt, t_gt = data["labels"]["rot_gt"].cuda().float(), data["labels"]["trans_gt"].cuda().float()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    losses = []
    for step in range(4):
        opt.zero_grad()
        out = net(data)
        loss = (out["rot_pred"] - rot_gt).abs().mean() + (out["trans_pred"] - t_gt).abs().mean()
        loss.backward()
        if step == 0:
            missing = [k for k, p in net.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
            dead = ("regressor_Xo", "regressor_Yc")          # heads that do not feed rot/trans
            assert [k for k in missing if not k.startswith(dead)] == []
            assert float(net.backbone_inp.module1[0].layers[0].weight.grad.abs().sum()) > 0
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]

"""Training objectives (dcl-net_amd/models/losses.py) against a literal restatement of the reference formulas
(models/DCL_Net.py:264-318, models/refiner.py:101-139) with the (b,N,M,3) difference tensor written out."""
import numpy as np
import pytest
import torch


def _cd_literal(pred, target):
    dis = torch.norm(pred.unsqueeze(2) - target.unsqueeze(1), dim=3)
    return 0.5 * (torch.min(dis, 2)[0] + torch.min(dis, 1)[0])


def _rand_rot(g, b):
    q, _ = torch.linalg.qr(torch.randn(b, 3, 3, generator=g))
    return q


def test_losses_match_literal_formulas(dcl):
    g = torch.Generator().manual_seed(0)
    b, n = 3, 40
    tmp, inp = torch.randn(b, n, 3, generator=g) * 0.05, torch.randn(b, n, 3, generator=g) * 0.05
    pred = {"rot_pred": _rand_rot(g, b), "trans_pred": torch.randn(b, 3, generator=g) * 0.01,
            "sym_flag": torch.tensor([0.0, 1.0, 0.0]), "conf": torch.rand(b, 2 * n, generator=g) * 0.8 + 0.1,
            "Xo_pred": torch.randn(b, n, 3, generator=g) * 0.05, "Yc_pred": torch.randn(b, n, 3, generator=g) * 0.05}
    gt = {"rot_gt": _rand_rot(g, b), "trans_gt": torch.randn(b, 3, generator=g) * 0.01, "points_tmp": tmp, "points_inp": inp}
    out = dcl.DCL_Net.losses(None)(pred, gt)
    # literal
    R, t, s = pred["rot_pred"], pred["trans_pred"], pred["sym_flag"]
    Rg, tg = gt["rot_gt"], gt["trans_gt"]
    L2 = lambda a, c: torch.norm(a - c, dim=2)                                               # noqa: E731
    pp = torch.bmm(tmp, R.transpose(1, 2)) + t.unsqueeze(1)
    pg = torch.bmm(tmp, Rg.transpose(1, 2)) + tg.unsqueeze(1)
    loss_pose = ((1 - s).unsqueeze(1) * L2(pp, pg) + s.unsqueeze(1) * _cd_literal(pp, pg)).mean(dim=1).mean()
    ip = torch.bmm(inp - t.unsqueeze(1), R)
    ig = torch.bmm(inp - tg.unsqueeze(1), Rg)
    lXo = (1 - s).unsqueeze(1) * L2(pred["Xo_pred"], ig) + 0.5 * s.unsqueeze(1) * (_cd_literal(pred["Xo_pred"], tmp) + L2(pred["Xo_pred"], ip))
    lYc = (1 - s).unsqueeze(1) * L2(pred["Yc_pred"], pg) + 0.5 * s.unsqueeze(1) * (_cd_literal(pred["Yc_pred"], pg) + L2(pred["Yc_pred"], pp))
    lconf = torch.mean(torch.cat([lXo, lYc], dim=1) * pred["conf"] - 0.01 * torch.log(pred["conf"]))
    want = loss_pose + 5 * lXo.mean() + lYc.mean() + lconf
    assert abs(float(out["loss_all"]) - float(want)) <= 1e-6
    assert abs(float(out["loss_pose"]) - float(loss_pose)) <= 1e-7 and abs(float(out["loss_conf"]) - float(lconf)) <= 1e-7
    # refiner loss
    pr = {"rot_pred": _rand_rot(g, b), "trans_pred": torch.randn(b, 3, generator=g) * 0.01}
    lo = dcl.refiner.losses_refiner(None)(pr, t, R, tmp, s, gt)
    pd = torch.bmm(tmp, pr["rot_pred"].transpose(1, 2)) + pr["trans_pred"].unsqueeze(1)
    rf = torch.bmm(pd, R.transpose(1, 2)) + t.unsqueeze(1)
    want_r = ((1 - s).unsqueeze(1) * L2(rf, pg) + s.unsqueeze(1) * _cd_literal(rf, pg)).mean(dim=1).mean()
    assert abs(float(lo["loss_all"]) - float(want_r)) <= 1e-7


@pytest.mark.gpu
def test_full_training_step_with_the_reference_objective(dcl):
    """Network(mode='train') + losses: forward, loss_all.backward(), optimiser step -- all on the GPU"""
    n = 256
    net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n, n), mode="train")
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
    net = net.cuda().train()
    crit = dcl.DCL_Net.losses(None)
    data = dcl.synth.make_batch(4, n, n)
    data["flags"] = torch.tensor([0.0, 1.0, 0.0, 1.0])
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    vals = []
    for _ in range(3):
        opt.zero_grad()
        pred = net(data)
        loss = crit(pred, data["labels"])
        loss["loss_all"].backward()
        bad = [k for k, p in net.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
        assert bad == []
        opt.step()
        vals.append(float(loss["loss_all"].detach()))
    assert np.isfinite(vals).all()
    lab = dcl.DCL_Net.losses.get_cano_label(data["labels"]["points_tmp"], data["labels"]["points_inp"],
                                            pred["rot_pred"].detach(), data["labels"]["trans_gt"].cuda().float().unsqueeze(1))
    assert lab.shape == (4, n, 3)


@pytest.mark.gpu
def test_chamfer_term_at_the_shipped_batch_shape(dcl):
    """bs 32 x 1024 points (config_YCBV_bs32.yaml) -- torch.cdist's exact mode refuses this launch shape on ROCm"""
    g = torch.Generator().manual_seed(1)
    a, c = torch.randn(32, 1024, 3, generator=g).cuda(), torch.randn(32, 1024, 3, generator=g).cuda()
    got = dcl.DCL_Net.losses.CD_Dis(a, c)
    want = _cd_literal(a[:2].cpu(), c[:2].cpu())
    assert got.shape == (32, 1024) and float((got[:2].cpu() - want).abs().max()) <= 1e-6


def _refiner_case(dcl, b=3, n=1024, seed=4):
    g = torch.Generator().manual_seed(seed)
    x = torch.cat([torch.randn(b, 3, n, generator=g) * 0.05, torch.randn(b, 256, n, generator=g)], 1)
    conf = torch.rand(b, 2 * n, generator=g)
    R = _rand_rot(g, b)
    gt = {"rot_gt": _rand_rot(g, b), "trans_gt": torch.randn(b, 3, generator=g) * 0.01}
    tmp = torch.randn(b, 500, 3, generator=g) * 0.05
    return x, conf, R, torch.randn(b, 3, generator=g) * 0.01, tmp, torch.tensor([0.0, 1.0, 0.0][:b]), gt


def test_refiner_training_path_is_the_reference_graph_and_differentiable(dcl):
    """Refiner.train(): forward composes the registered modules like models/refiner.py:78-95 (== the oracle restatement that
    is pinned to the reference's outputs) and losses_refiner(...).backward() reaches all 18 parameters -- the call
    sequence of tools/train_YCBV_stage2.py:243-262.  (Pure torch modules: this half runs on the CPU.)"""
    from oracle import graph as G
    ref = dcl.refiner.Refiner()
    sd = dcl.synth.synth_state_dict(ref, 2)
    ref.load_state_dict(sd)
    ref.train()
    x, conf, R, t, tmp, sym, gt = _refiner_case(dcl)
    out = ref._forward_modules(x, conf)
    want = G.refiner_forward(sd, x, conf)
    assert float((out["trans_pred"] - want["trans_pred"]).abs().max()) <= 1e-6
    assert float((out["rot_pred"] - want["rot_pred"]).abs().max()) <= 1e-5
    loss = dcl.refiner.losses_refiner(None)(out, t, R, tmp, sym, gt)["loss_all"]
    loss.backward()
    grads = {k: p.grad for k, p in ref.named_parameters()}
    assert len(grads) == 18 and all(g is not None and bool(torch.isfinite(g).all()) for g in grads.values())
    assert all(float(g.abs().max()) > 0 for g in grads.values())


@pytest.mark.gpu
def test_refiner_trains_on_the_gpu(dcl):
    """train() mode on the GPU: same outputs as the fused eval path, gradients equal to an fp64 CPU evaluation of the same
    graph, an optimiser step changes the weights and the eval path (folded weights, captured graphs) follows them"""
    import copy
    ref = dcl.refiner.Refiner()
    ref.load_state_dict(dcl.synth.synth_state_dict(ref, 2))
    ref = ref.cuda()
    x, conf, R, t, tmp, sym, gt = _refiner_case(dcl)
    cu = lambda v: v.cuda()                                                     # noqa: E731
    inp = {"input_features": cu(x), "conf": cu(conf), "obj_idx": None}
    ev = ref.eval()(inp)
    assert not ev["rot_pred"].requires_grad
    out = ref.train()(inp)
    assert out["rot_pred"].requires_grad and out["trans_pred"].requires_grad
    assert float((out["rot_pred"] - ev["rot_pred"]).abs().max()) <= 1e-4
    assert float((out["trans_pred"] - ev["trans_pred"]).abs().max()) <= 1e-5
    crit = dcl.refiner.losses_refiner(None)
    gtc = {k: cu(v) for k, v in gt.items()}
    loss = crit(out, cu(t), cu(R), cu(tmp), cu(sym), gtc)["loss_all"]
    loss.backward()
    twin = copy.deepcopy(ref).cpu().double().train()
    for p in twin.parameters():
        p.grad = None
    o64 = twin._forward_modules(x.double(), conf.double())
    l64 = crit(o64, t.double(), R.double(), tmp.double(), sym.double(), {k: v.double() for k, v in gt.items()})["loss_all"]
    l64.backward()
    assert abs(float(loss) - float(l64)) <= 1e-5 * max(1.0, abs(float(l64)))
    for (k, p), (_, q) in zip(ref.named_parameters(), twin.named_parameters()):
        scale = max(float(q.grad.abs().max()), 1e-8)
        assert float((p.grad.cpu().double() - q.grad).abs().max()) <= 2e-3 * scale, k
    before = ref.eval()(inp)["trans_pred"].clone()
    opt = torch.optim.SGD(ref.parameters(), lr=1e-2)
    opt.step()                                                                 # in-place update, no load_state_dict
    after = ref.eval()(inp)["trans_pred"]
    assert float((after - before).abs().max()) > 1e-7                          # the fold cache followed the weights
    want = ref.train()(inp)["trans_pred"].detach()
    assert float((after - want).abs().max()) <= 1e-5

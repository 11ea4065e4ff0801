#!/usr/bin/env python3
"""Runs Network.forward on a list of (b, N, M) shapes and prints ms/step: quick coverage of batch sizes (config 3 uses
b=40 per GPU) and point counts outside the benchmarked ones."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
shapes = [(1, 1024, 1024), (2, 1024, 1024), (40, 1024, 1024), (40, 12288, 2048), (7, 500, 1000), (64, 1024, 1024)]
for b, n, m in shapes:
    cfg = dcl.synth.default_cfg(n, m)
    net = dcl.DCL_Net.Network(cfg, mode="test")
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
    net = net.cuda().eval()
    data = dcl.synth.make_batch(b, n, m)
    dev = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) and k in ("inp", "tmp") else v) for k, v in data.items()}
    with torch.no_grad():
        for _ in range(2):
            p = net(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            p = net(dev)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    R = p["rot_pred"].double()
    ok = float((R @ R.transpose(1, 2) - torch.eye(3, dtype=torch.float64, device="cuda")).abs().max()) < 1e-5
    print("b=%3d N=%5d M=%5d  %8.3f ms/step  %9.1f frames/s  finite=%s orthonormal=%s" % (
        b, n, m, ms, b / ms * 1e3, bool(torch.isfinite(p["trans_pred"]).all()), ok), flush=True)
    del net, dev, p
    torch.cuda.empty_cache()

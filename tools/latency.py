#!/usr/bin/env python3
"""Small-batch latency of Network.forward: eager vs whole-forward hipGraph (the reference's eval loop feeds one image =
a few crops per call)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
n = 1024
cfg = dcl.synth.default_cfg(n, n)
net = dcl.DCL_Net.Network(cfg, mode="test")
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
for b in (1, 4, 8):
    data = dcl.synth.make_batch(b, n, n)
    res = {}
    for name, fn in (("eager", lambda: net(data)), ("hipgraph", lambda: net.forward_graphed(data))):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 20 * 1e3
    print("b=%d N=M=%d: eager %.3f ms, hipgraph %.3f ms per forward (host data dict -> pose)" % (b, n, res["eager"], res["hipgraph"]), flush=True)

#!/usr/bin/env python3
"""Same-job A/B of a diagnostic hook on the whole forward (graph replay of b crops at the reference shape, two-branch and
one-stream layouts), alternating, three rounds -- box-to-box spread (1-3 %) hides anything smaller across jobs.
usage: ab_hook.py <dcl_debug_* function> <value A> <value B> [b]      e.g.  ab_hook.py dcl_debug_conv_wlds 0 1 32"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
L = use_diag(dcl)
fn, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
b = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = torch.device("cuda:0")
data = bench.to_device(dcl.synth.make_batch(b, 1024, 1024), dev)
nets = {}
for kw in ({}, {"single_stream": True}):
    net = dcl.DCL_Net.Network(dcl.synth.default_cfg(1024, 1024), mode="test", **kw)
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
    nets["one stream" if kw else "two branches"] = net.to(dev).eval()
res = {}
for rep in range(3):
    for val in (va, vb):
        getattr(L, fn)(val)
        for name, net in nets.items():
            net._invalidate()                                   # recapture under the new setting
            with torch.no_grad():
                for _ in range(3):
                    net.forward_graphed(data)
                ent = next(iter(net._graphs.values()))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(60):
                    ent["graph"].replay()
                torch.cuda.synchronize()
            res.setdefault((val, name), []).append((time.perf_counter() - t0) / 60 * 1e3)
for (val, name), v in sorted(res.items()):
    print("%s(%d) %-13s b=%d: %s ms" % (fn, val, name, b, " ".join("%.3f" % x for x in v)))

#!/usr/bin/env python3
"""Same-job A/B of a diagnostic hook on whole forwards, launch by launch and (where the shape replays one) as a graph, alternating,
three rounds -- box-to-box spread (1-3 %) hides anything smaller across jobs.
usage: ab_forward.py <dcl_debug_* function> <value A> <value B> [b] [n_inp] [n_tmp]     e.g.  ab_forward.py dcl_debug_linear_persist 4 1000000 32 12288 2048"""
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
n_inp = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
n_tmp = int(sys.argv[6]) if len(sys.argv) > 6 else 1024
dev = torch.device("cuda:0")
data = bench.to_device(dcl.synth.make_batch(b, n_inp, n_tmp), dev)
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n_inp, n_tmp), mode="test", graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.to(dev).eval()
steps = 10 if n_inp > 4096 else 40
res = {}
for rep in range(3):
    for val in (va, vb):
        getattr(L, fn)(val)
        with torch.no_grad():
            for _ in range(3):
                net(data)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                net(data)
            torch.cuda.synchronize()
        res.setdefault(val, []).append((time.perf_counter() - t0) / steps * 1e3)
for val, v in sorted(res.items()):
    print("%s(%d) b=%d N=%d M=%d launch by launch: %s ms" % (fn, val, b, n_inp, n_tmp, " ".join("%.3f" % x for x in v)))

#!/usr/bin/env python3
"""High-priority side streams for the sparse half (and the disengage GEMMs of a side right behind its read-out) -- one process per
setting, same box.  usage: ab_prio.py <priority> <HEAD_ORDER> <persist rounds> [b] [n_inp] [n_tmp] [async]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
L = use_diag(dcl)
prio, order, persist = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
b = int(sys.argv[4]) if len(sys.argv) > 4 else 32
n_inp = int(sys.argv[5]) if len(sys.argv) > 5 else 12288
n_tmp = int(sys.argv[6]) if len(sys.argv) > 6 else 2048
asyn = len(sys.argv) > 7 and sys.argv[7] == "1"
dcl.DCL_Net.SIDE_STREAM_PRIORITY = prio
L.dcl_debug_linear_persist(persist)
dev = torch.device("cuda:0")
data = bench.to_device(dcl.synth.make_batch(b, n_inp, n_tmp), dev)
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n_inp, n_tmp), mode="test", graph_max_batch=0, async_inputs=asyn)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.to(dev).eval()
net.HEAD_ORDER = None if order < 0 else order
steps = 10 if n_inp > 4096 else 40
res = []
for rep in range(4):
    with torch.no_grad():
        for _ in range(3):
            net(data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net(data)
        torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / steps * 1e3)
print("priority %d HEAD_ORDER %d persist %d async %d b=%d N=%d M=%d: %s ms" % (prio, order, persist, asyn, b, n_inp, n_tmp, " ".join("%.3f" % x for x in res)))

#!/usr/bin/env python3
"""Same-job A/B of a Network class attribute on whole launch-by-launch forwards, alternating, three rounds.
usage: ab_attr.py <ATTRIBUTE> <b> <n_inp> <n_tmp> <value> [<value> ...]     e.g.  ab_attr.py HEAD_ORDER 32 12288 2048 0 1 2"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
attr, b, n_inp, n_tmp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
vals = [int(v) for v in sys.argv[5:]]
dev = torch.device("cuda:0")
data = bench.to_device(dcl.synth.make_batch(b, n_inp, n_tmp), dev)
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n_inp, n_tmp), mode="test", graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.to(dev).eval()
steps = 10 if n_inp > 4096 else 40
res = {}
for rep in range(4):
    for val in vals:
        setattr(net, attr, val)
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
    print("%s=%d b=%d N=%d M=%d launch by launch: %s ms" % (attr, val, b, n_inp, n_tmp, " ".join("%.3f" % x for x in v)))

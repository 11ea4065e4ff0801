#!/usr/bin/env python3
"""Host-side timeline of Network._forward_fused (serial calls, inputs resident): where the CPU is when it issues the
geometry, reads the level sizes back, has issued the sparse half and the dense half, vs the step time.
usage: tools/host_timeline.py [ref|stress] [batch]"""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
shape = sys.argv[1] if len(sys.argv) > 1 else "ref"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n_inp, n_tmp = bench.SHAPES[shape]
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n_inp, n_tmp), mode="test", graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
data = bench.to_device(dcl.synth.make_batch(b, n_inp, n_tmp), torch.device("cuda"))
with torch.no_grad():
    for _ in range(5):
        net(data)
    torch.cuda.synchronize()
    M = dcl.DCL_Net
    rows = []
    for _ in range(20):
        M.HOST_TIMES = []
        t0 = time.perf_counter()
        net(data)
        t_ret = time.perf_counter()
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        rows.append([t - t0 for _, t in M.HOST_TIMES] + [t_ret - t0, t_end - t0])
        labels = [l for l, _ in M.HOST_TIMES] + ["forward returned", "GPU done"]
    M.HOST_TIMES = None
a = np.array(rows) * 1e3
for l, v in zip(labels, np.median(a, axis=0)):
    print("%-20s %7.3f ms" % (l, v))

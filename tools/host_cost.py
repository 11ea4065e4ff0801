#!/usr/bin/env python3
"""Host time of one Network.forward_graphed call (one crop, resident inputs) against the GPU time of its replay: which of the
two bounds a stream of one-crop calls.  Per piece: cProfile of 200 calls.  usage: tools/host_cost.py [b]"""
import cProfile, importlib, os, pstats, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(1024, 1024, unit=0.005), mode="test", graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.to(dev).eval()
data = bench.to_device(dcl.synth.make_batch(b, 1024, 1024, unit=0.005), dev)
with torch.no_grad():
    for _ in range(5):
        net.forward_graphed(data)
    torch.cuda.synchronize()
    ent = next(iter(net._graphs.values()))
    # GPU time of a replay alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        ent["graph"].replay()
    e1.record()
    torch.cuda.synchronize()
    print("graph replay alone: %.3f ms per replay (GPU, back to back)" % (e0.elapsed_time(e1) / 100))
    # host time of a call while the GPU is idle enough not to push back: one call, then wait
    host = []
    for _ in range(100):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net.forward_graphed(data)
        host.append(time.perf_counter() - t0)
    print("forward_graphed host time per call (GPU drained before each): median %.3f ms, p10 %.3f, p90 %.3f" % (
        np.median(host) * 1e3, np.percentile(host, 10) * 1e3, np.percentile(host, 90) * 1e3))
    t = []
    for _ in range(100):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ent["graph"].replay()
        t.append(time.perf_counter() - t0)
    print("  of which graph.replay(): median %.3f ms" % (np.median(t) * 1e3))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        net.forward_graphed(data)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

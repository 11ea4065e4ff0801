#!/usr/bin/env python3
"""What a caller of the reference API gets without touching anything: ms per `model(data)` call of a default-constructed
Network for a few batch sizes (<= 8 crops: whole-forward hipGraph replay, above: launch by launch)."""
import importlib, sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dcl = importlib.import_module("dcl-net_amd")
cfg = dcl.synth.default_cfg(1024, 1024, unit=0.005)
net = dcl.DCL_Net.Network(cfg, mode="test")
net.load_state_dict(dcl.synth.synth_state_dict(net, 1)); net = net.cuda().eval()
for b in (1, 3, 8, 9):
    data = bench.to_device(dcl.synth.make_batch(b, 1024, 1024, unit=0.005), torch.device("cuda"))
    for _ in range(3): net(data)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): out = net(data)
    torch.cuda.synchronize()
    print("default Network, b=%d: %.3f ms per model(data) call" % (b, (time.perf_counter() - t0) / 50 * 1e3))

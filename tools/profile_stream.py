#!/usr/bin/env python3
"""one-crop stream (LineMOD regime): 50 whole-forward graph replays, for rocprofv3 --kernel-trace --stats"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(1024, 1024, unit=0.005), mode="test", graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
data = bench.to_device(dcl.synth.make_batch(b, 1024, 1024, unit=0.005), torch.device("cuda"))
for _ in range(50):
    net.forward_graphed(data)
torch.cuda.synchronize()

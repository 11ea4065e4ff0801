#!/usr/bin/env python3
"""forward() vs forward_graphed() with inputs resident in HBM: tools/graph_vs_eager.py [b] [N] [M]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
m = int(sys.argv[3]) if len(sys.argv) > 3 else n
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n, m), mode="test", graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
data = bench.to_device(dcl.synth.make_batch(b, n, m), torch.device("cuda"))
for name, fn in (("eager", lambda: net(data)), ("hipgraph", lambda: net.forward_graphed(data))):
    with torch.no_grad():
        for _ in range(4): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): fn()
        torch.cuda.synchronize()
    print("b=%d N=%d M=%d %s: %.3f ms" % (b, n, m, name, (time.perf_counter() - t0) / 30 * 1e3), flush=True)
print("peak memory GB", torch.cuda.max_memory_allocated() / 1e9)

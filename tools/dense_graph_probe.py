#!/usr/bin/env python3
"""dense half of the fused forward (disengage stacks .. pose heads) on random point features: eager launches vs one
hipGraph replay -- what a captured dense stage could save at a given shape.  usage: dense_graph_probe.py [b] [N] [M]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n, m), mode="test")
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
f = net._fold()
dev = torch.device("cuda")
pf_i = torch.randn(b * n, 480, device=dev).relu_(); pf_t = torch.randn(b * m, 480, device=dev).relu_()
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
with torch.no_grad():
    eager = timeit(lambda: net._dense(f, pf_i, pf_t, b, dev))
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        net._dense(f, pf_i, pf_t, b, dev)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = net._dense(f, pf_i, pf_t, b, dev)
    graph = timeit(g.replay)
print("b=%d N=%d M=%d dense stage: eager %.3f ms, hipGraph replay %.3f ms" % (b, n, m, eager, graph))

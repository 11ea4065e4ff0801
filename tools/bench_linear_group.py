#!/usr/bin/env python3
"""ops.linear_group (one launch) against the same layers as separate library GEMMs, on the model's small-batch shapes;
both captured into a hipGraph (what a call of a handful of crops replays), us per replay.  usage: bench_linear_group.py [b ...]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
ops = dcl.ops
dev = torch.device("cuda:0")

def graph_us(fn, reps=200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3

for b in [int(x) for x in sys.argv[1:]] or [1, 2, 6]:
    M = 1024 * b
    H = torch.randn(M, 1024, device=dev)
    second = [(torch.randn(256, n, device=dev) * 0.05, torch.randn(n, device=dev)) for n in (256, 64, 256, 64)]
    outs = [torch.empty(M, n, device=dev) for n in (256, 64, 256, 64)]
    sep = lambda: [ops.linear(H[:, 256 * j:256 * (j + 1)], second[j][0], second[j][1], True, out=outs[j]) for j in range(4)]
    grp = lambda: ops.linear_group([(H[:, 256 * j:256 * (j + 1)], second[j][0], second[j][1], True, outs[j]) for j in range(4)])
    print("b=%d disengage second layers (4 x K=256): separate %.1f us, grouped %.1f us" % (b, graph_us(sep), graph_us(grp)))
    conf_in, fuse = torch.randn(M, 128, device=dev), torch.randn(M, 512, device=dev)
    cl = [(torch.randn(128, 128, device=dev) * 0.05, torch.randn(128, device=dev)) for _ in range(2)]
    cl.append((ops.pad_linear_weight(torch.randn(128, 1, device=dev) * 0.05), torch.randn(1, device=dev)))
    fl = [(torch.randn(512, n, device=dev) * 0.03, torch.randn(n, device=dev)) for n in (512, 512, 1024)]
    def sep2():
        h, F = conf_in, fuse
        for d in range(3):
            h = ops.linear(h, cl[d][0], cl[d][1], d < 2)
        for d in range(3):
            F = ops.linear(F, fl[d][0], fl[d][1], True)
        return h, F
    def grp2():
        h, F = conf_in, fuse
        for d in range(3):
            h, F = ops.linear_group([(h, cl[d][0], cl[d][1], d < 2, None), (F, fl[d][0], fl[d][1], True, None)])
        return h, F
    print("b=%d conf + fuser stacks (3 + 3 layers): separate %.1f us, grouped %.1f us" % (b, graph_us(sep2), graph_us(grp2)))
    for d, (K, n) in enumerate(((512, 512), (512, 1024), (256, 256), (128, 128), (480, 1024))):
        x = torch.randn(M, K, device=dev); W = torch.randn(K, n, device=dev) * 0.03; bb = torch.randn(n, device=dev); o = torch.empty(M, n, device=dev)
        print("   b=%d single %dx%dx%d: library %.1f us, own %.1f us" % (b, M, n, K, graph_us(lambda: ops.linear_lt(x, W, bb, True, out=o)),
                                                                   graph_us(lambda: ops.linear_group([(x, W, bb, True, o)]))))

#!/usr/bin/env python3
"""Same-job A/B of the split-bf16 switches (ops.GEMM_SPLIT, ops.ATTENTION_SPLIT) on the PRODUCT library: whole forwards of b crops at
the reference shape as graph replays (two-branch default) and launch by launch, alternating, three rounds.
usage: ab_split.py [b] [n_inp] [n_tmp]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_inp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n_tmp = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda:0")
data = bench.to_device(dcl.synth.make_batch(b, n_inp, n_tmp), dev)
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n_inp, n_tmp), mode="test")
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.to(dev).eval()
graphed = b * max(n_inp, n_tmp) <= 98304
res = {}
for rep in range(3):
    for gemm, att in ((False, False), (True, False), (True, True)):
        dcl.ops.GEMM_SPLIT, dcl.ops.ATTENTION_SPLIT = gemm, att
        net._invalidate()
        for name, fn in (("graph replay", lambda: net.forward_graphed(data)), ("launch by launch", lambda: net(data))):
            if name == "graph replay" and not graphed:
                continue
            with torch.no_grad():
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 40 if n_inp <= 4096 else 10
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
            res.setdefault((gemm, att, name), []).append((time.perf_counter() - t0) / n * 1e3)
dcl.ops.GEMM_SPLIT, dcl.ops.ATTENTION_SPLIT = True, True
for (gemm, att, name), v in sorted(res.items()):
    print("GEMM_SPLIT=%d ATTENTION_SPLIT=%d b=%d N=%d M=%d %-17s %s ms" % (gemm, att, b, n_inp, n_tmp, name + ":", " ".join("%.3f" % x for x in v)))

#!/usr/bin/env python3
"""Soak run at the stress shape (N = 12288 / M = 2048): 120 pipelined forwards of 32 / 33 / 17 / 40 crops -- the split-bf16 GEMMs and
the split attention with odd crop counts and partial last rounds -- every recurrence of a batch must reproduce its first result bit
for bit, all outputs finite.  usage: tools/soak_stress.py"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
dev = torch.device("cuda")
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(12288, 2048), mode="test", async_inputs=True, graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1)); net = net.cuda().eval()
batches = [bench.to_device(dcl.synth.make_batch(b, 12288, 2048, first=f), dev) for b, f in ((32, 0), (33, 3), (17, 9), (40, 1))]
first = {}
t0 = time.perf_counter()
for it in range(120):
    k = it % len(batches)
    with torch.no_grad():
        out = net(batches[k])
    got = (out["rot_pred"].clone(), out["trans_pred"].clone(), out["conf"].clone())
    if k not in first: first[k] = got
    else:
        for a, c in zip(first[k], got):
            assert torch.equal(a, c), ("not reproducible", it, k)
    assert all(bool(torch.isfinite(t).all()) for t in got)
torch.cuda.synchronize()
print("stress-shape soak: 120 forwards of 32 / 33 / 17 / 40 crops, bit-reproducible, finite; %.1f s; peak memory %.1f GB" % (time.perf_counter() - t0, torch.cuda.max_memory_allocated() / 2**30))

#!/usr/bin/env python3
"""Soak run: many forwards (eager and graphed, varying crops) -- memory must stay flat, outputs finite."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
n = 1024
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n, n), mode="test", async_inputs=True)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
dev = torch.device("cuda")
batches = [bench.to_device(dcl.synth.make_batch(b, n, n, first=f), dev) for b, f in ((32, 0), (8, 5), (3, 40), (1, 77), (32, 64))]
marks = []
t0 = time.perf_counter()
for it in range(400):
    d = batches[it % len(batches)]
    with torch.no_grad():
        out = net(d) if it % 2 == 0 or d["batch_offsets"].numel() - 1 > 8 else net.forward_graphed(d)
    if it % 50 == 49:
        torch.cuda.synchronize()
        assert torch.isfinite(out["rot_pred"]).all()
        marks.append((it + 1, torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
print("elapsed %.1f s" % (time.perf_counter() - t0))
for m in marks:
    print("iter %4d allocated %6d MiB reserved %6d MiB" % m)
assert marks[-1][1] <= marks[1][1] * 1.05 + 8, "allocated memory keeps growing"
print("OK")

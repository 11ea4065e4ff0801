#!/usr/bin/env python3
"""Soak run: many pipelined forwards (eager and graphed, varying crops) -- memory must stay flat, outputs finite, and every
recurrence of a (batch, path) pair must reproduce its first result bit for bit (level sizes travel through one reused pinned
buffer per side, the side streams run ahead of the caller's stream)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
n = 1024
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n, n), mode="test", async_inputs=True, graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
dev = torch.device("cuda")
batches = [bench.to_device(dcl.synth.make_batch(b, n, n, first=f), dev) for b, f in ((32, 0), (8, 5), (3, 40), (1, 77), (32, 64))]
marks = []
first, pending = {}, []
t0 = time.perf_counter()
for it in range(400):
    d = batches[it % len(batches)]
    with torch.no_grad():
        out = net(d) if it % 2 == 0 or d["batch_offsets"].numel() - 1 > 8 else net.forward_graphed(d)
    keyed = (it % len(batches), it % 2 == 0 or d["batch_offsets"].numel() - 1 > 8)
    got = (out["rot_pred"].clone(), out["trans_pred"].clone(), out["conf"].clone())
    if keyed not in first:
        first[keyed] = got
    else:
        pending.append((it, first[keyed], got))
    if it % 50 == 49:
        torch.cuda.synchronize()
        assert torch.isfinite(out["rot_pred"]).all()
        for i, a, g in pending:
            assert all(torch.equal(x, y) for x, y in zip(a, g)), "iteration %d differs from the first run of its batch" % i
        pending = []
        marks.append((it + 1, torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
print("elapsed %.1f s" % (time.perf_counter() - t0))
for m in marks:
    print("iter %4d allocated %6d MiB reserved %6d MiB" % m)
assert marks[-1][1] <= marks[1][1] * 1.05 + 8, "allocated memory keeps growing"
print("OK")

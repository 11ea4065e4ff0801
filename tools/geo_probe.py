#!/usr/bin/env python3
"""Geometry stage of both sides on two streams, as the forward issues it: when does each side finish (event times from a
common start, no profiler attached) and how long does the host take to issue it."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
n, b = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 32
data = dcl.synth.make_batch(b, n, 1024 if n == 1024 else 2048)
occ = {s: data[s]["occupied_voxels"].int().cuda().contiguous() for s in ("inp", "tmp")}
st = {s: torch.cuda.Stream() for s in occ}
pinned = {s: torch.zeros(8, dtype=torch.int32).pin_memory() for s in occ}
for rep in range(6):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    ev, host = {}, {}
    t0 = time.perf_counter()
    for s in ("inp", "tmp"):
        st[s].wait_event(e0)
        with torch.cuda.stream(st[s]):
            run = dcl.ops.BackboneRun(occ[s], b, 64, counts_dev=pinned[s])
            ev[s] = torch.cuda.Event(enable_timing=True); ev[s].record(st[s])
        host[s] = (time.perf_counter() - t0) * 1e6
    ev["inp"].synchronize(); t_inp = (time.perf_counter() - t0) * 1e6
    ev["tmp"].synchronize(); t_tmp = (time.perf_counter() - t0) * 1e6
    print("rep %d: host issued inp by %.0f us, tmp by %.0f us; GPU done inp %.0f us, tmp %.0f us after start; host saw inp at %.0f, tmp at %.0f; counts %s" %
          (rep, host["inp"], host["tmp"], e0.elapsed_time(ev["inp"]) * 1e3, e0.elapsed_time(ev["tmp"]) * 1e3, t_inp, t_tmp, pinned["inp"].tolist()[:3]))

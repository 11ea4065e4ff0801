#!/usr/bin/env python3
"""Feature stage of ONE backbone pass stand-alone (no second side, no dense work): ms per pass at bs=32, for comparing
runner-level changes (implicit vs tabulated neighbours, decompositions) without the concurrency of the full forward."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
DIAG = use_diag(dcl)          # kernel-variant hooks exist in the diagnostic library only
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = dcl.synth.default_cfg(n, 64)
net = dcl.DCL_Net.Network(cfg, mode="test")
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
f = net._fold()
data = dcl.synth.make_batch(b, n, 64)
occ = data["inp"]["occupied_voxels"].int().cuda().contiguous()
x = dcl.ops.voxelize_fp(data["inp"]["feats"].cuda(), data["inp"]["v2p_maps"].cuda(), 4)
def geo():
    run = dcl.ops.BackboneRun(occ, b, 64)
    run.set_counts(run.counts_dev.cpu().tolist())
    return run
run = geo()
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
print("b=%d " % b, end="")
print("N=%d: features stage %.1f us per pass; geometry + read-back %.1f us" %
      (n, timeit(lambda: run.features(x, *f["backbone_inp_ptrs"])), timeit(geo)))
lib = dcl._native.lib()
for mode in (0, 1):
    lib.dcl_debug_geometry_chain(mode)
    def geo_only():
        dcl.ops.BackboneRun(occ, b, 64)
    print("geometry stage without read-back, mask chain %s: %.1f us per pass (launch-bound: the host issues back to back)" %
          ("in one launch" if mode else "as 8 launches", timeit(geo_only, 50)))
lib.dcl_debug_geometry_chain(1)

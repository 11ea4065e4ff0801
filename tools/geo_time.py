#!/usr/bin/env python3
"""Time of the geometry stage of one backbone pass (ops.BackboneRun: active sets of the four levels + row orders), b crops of
1024 points, by HIP events.  DCL_USE_DIAG=1 DCL_GEO_SMALL=<n> switches the one-launch stage on for up to n crops.
usage: tools/geo_time.py [b ...]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
if os.environ.get("DCL_USE_DIAG"):
    from _diag import use_diag
    use_diag(dcl)
ops = dcl.ops
dev = torch.device("cuda:0")
for b in [int(a) for a in sys.argv[1:]] or [1, 6, 32]:
    for npts in (1024, 12288):
        data = dcl.synth.make_batch(b, npts, 1024)
        occ = data["inp"]["occupied_voxels"].to(dev).int().contiguous()
        for _ in range(5):
            run = ops.BackboneRun(occ, b, 64)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            run = ops.BackboneRun(occ, b, 64)
        e1.record()
        torch.cuda.synchronize()
        print("b=%d, %d points per crop (%d voxel rows): %.1f us per geometry stage" % (b, npts, occ.shape[0], e0.elapsed_time(e1) / reps * 1e3))

#!/usr/bin/env python3
"""The point read-out of one backbone side alone (grid-pruned 3-NN of the 4 levels + interpolation into the 480-channel
rows) on the real active sets: usage bench_readout.py [N] [batch]"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = dcl.synth.default_cfg(n, 64)
net = dcl.DCL_Net.Network(cfg, mode="test")
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
f = net.cuda().eval()._fold()
data = dcl.synth.make_batch(b, n, 64)
occ = data["inp"]["occupied_voxels"].int().cuda().contiguous()
run = dcl.ops.BackboneRun(occ, b, 64)
run.set_counts(run.counts_dev.cpu().tolist())
x = dcl.ops.voxelize_fp(data["inp"]["feats"].cuda(), data["inp"]["v2p_maps"].cuda(), 4)
run.features(x, *f["backbone_inp_ptrs"])
feats = data["inp"]["feats"].cuda()
pb4 = torch.cat([torch.arange(b, device="cuda").float().repeat_interleave(n).unsqueeze(1), feats[:, 4:7]], 1).contiguous()
unit = 0.006
off = float(np.float32(-0.5 * unit * 64))
ext = [float(np.float32(unit * s)) for s in (2, 4, 6, 8)]
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
d2, idx = run.point_neighbours(pb4, ext, off)
print("N=%d b=%d: 3-NN of 4 levels %.1f us, interpolation %.1f us, both %.1f us" % (
    n, b, timeit(lambda: run.point_neighbours(pb4, ext, off)), timeit(lambda: run.point_interpolate(d2, idx)),
    timeit(lambda: run.point_features(pb4, ext, off))))

#!/usr/bin/env python3
"""Host-side profile of CropBuilder.build on a resident 6-object frame (cProfile, top functions by own time)."""
import cProfile, importlib, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
cfg = dict(input_size=1024, tmp_size=1024, unit_voxel_extent=[0.006] * 3, voxel_num_limit=[64] * 3, voxelization_mode=4)
f = dcl.synth.make_frame(500, n_obj=6, tmp_size=1024)
builder = dcl.crops.CropBuilder(cfg, f["cad_pts"], f["cad_col"])
res = dcl.crops.CropBuilder.resident(f["img"], f["depth"], f["label"])
def build():
    return builder.build(res[0], res[1], res[2], f["rois"], f["gt_obj"], poses=f["poses"])
for _ in range(5): build()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): build()
torch.cuda.synchronize()
print("build: %.3f ms per frame (draws %.3f ms)" % ((time.perf_counter() - t0) / 50 * 1e3, builder.draw_seconds / 55 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(50): build()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)

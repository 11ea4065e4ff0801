#!/usr/bin/env python3
"""Crop builder timing: device builder (dcl-net_amd/crops.py) on a synthetic 480x640 frame with 6 object instances,
1024 points per crop.  (The CPU restatement of the reference loader is timed next to it in
tests/test_crops.py::test_device_crop_builder_timing -- only tests may run the oracle.)"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from crop_scene import make_scene
dcl = importlib.import_module("dcl-net_amd")
cfg = dict(input_size=1024, tmp_size=1024, unit_voxel_extent=[0.006] * 3, voxel_num_limit=[64] * 3, voxelization_mode=4)
sc = make_scene(5, n_obj=6, tmp_size=1024)
builder = dcl.crops.CropBuilder(cfg, sc["cad_pts"], sc["cad_col"])
def dev():
    return builder.build(sc["img"], sc["depth"], sc["label"], sc["rois"], sc["gt_obj"], poses=sc["poses"])
for name, fn, reps in (("device", dev, 50),):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    print("%s: %.3f ms per image (%d crops)" % (name, (time.perf_counter() - t0) / reps * 1e3, int(fn()["all_flags"].sum())), flush=True)

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
res = dcl.crops.CropBuilder.resident(sc["img"], sc["depth"], sc["label"])          # the frame decoded ahead, in HBM
def dev():
    return builder.build(res[0], res[1], res[2], sc["rois"], sc["gt_obj"], poses=sc["poses"])
def dev_host():
    return builder.build(sc["img"], sc["depth"], sc["label"], sc["rois"], sc["gt_obj"], poses=sc["poses"])
for name, fn, reps in (("device builder, frame resident in HBM", dev, 50), ("device builder, numpy frame uploaded per call", dev_host, 20)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    builder.draw_seconds = 0.0
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    d = fn()
    bytes_frame = sum(int(t.numel()) * t.element_size() for t in res) + sum(
        int(d[s_][k].numel()) * d[s_][k].element_size() for s_ in ("inp", "tmp") for k in ("feats", "occupied_voxels", "p2v_maps", "v2p_maps"))
    print("%s: %.3f ms per image (%d crops; of which %.3f ms are the loader's np.random.choice draws on the host); "
          "bytes per frame (image + depth + label read, crops written): %.2f MB" % (
              name, dt, int(d["all_flags"].sum()), builder.draw_seconds / reps * 1e3, bytes_frame / 1e6), flush=True)

"""CropBuilder.build alone on resident frames: ms per frame, the host time between its marks, and (under rocprofv3
--kernel-trace) which kernels a frame issues.  usage: python tools/builder_trace.py [frames]"""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dcl = importlib.import_module("dcl-net_amd")
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg_b = dict(input_size=1024, tmp_size=1024, unit_voxel_extent=[0.006] * 3, voxel_num_limit=[64] * 3, voxelization_mode=4)
frames = [dcl.synth.make_frame(500 + i, n_obj=6, tmp_size=1024) for i in range(4)]
for cap in (False, True):
    builder = dcl.crops.CropBuilder(cfg_b, frames[0]["cad_pts"], frames[0]["cad_col"], device=dev, capacity=cap)
    res = [dcl.crops.CropBuilder.resident(f["img"], f["depth"], f["label"], dev) for f in frames]
    def build(i):
        f, (im, de, la) = frames[i % 4], res[i % 4]
        return builder.build(im, de, la, f["rois"], f["gt_obj"], poses=f["poses"])
    np.random.seed(1)
    for i in range(6):
        build(i)
    torch.cuda.synchronize()
    builder.draw_seconds = 0.0
    dcl.crops.MARKS = []
    t = time.perf_counter()
    for i in range(n):
        build(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("capacity=%s: %.3f ms per frame, of which draws %.3f ms" % (cap, dt * 1e3, builder.draw_seconds / n * 1e3))
    marks = dcl.crops.MARKS
    dcl.crops.MARKS = None
    if marks:
        per = {}
        for frame in range(n):
            seg = marks[frame * (len(marks) // n):(frame + 1) * (len(marks) // n)]
            for (a, ta), (b, tb) in zip(seg[:-1], seg[1:]):
                per.setdefault("%s -> %s" % (a, b), []).append(tb - ta)
        for k, v in per.items():
            print("   %-46s %.1f us" % (k, 1e6 * sum(v) / len(v)))

#!/usr/bin/env python3
"""Where the one-launch geometry stage of a one-image call (rulebook.hip: k_geometry_small) spends its time: s_memrealtime stamps
of workgroup 0 at its phase boundaries (diagnostic library).  usage: tools/geo_stamps.py [crops, default 1]"""
import ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
use_diag(dcl)
ops = dcl.ops
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
data = dcl.synth.make_batch(b, 1024, 1024)
occ = data["inp"]["occupied_voxels"].to(dev).int().contiguous()
names = ["n, zero LDS, occupancy scatter", "set 0 out", "8-stage mask chain", "counts: packed wave scans", "wave totals scanned",
         "counts out / bases in", "word prefixes + decoded rows", "totals", "level-0 permutation"]
lib = ops.N.lib()
acc = np.zeros(9)
chain = np.zeros(8)
reps = 20
for i in range(reps + 3):
    run = ops.BackboneRun(occ, b, 64)
    torch.cuda.synchronize()
    st = np.zeros(32, np.uint64)
    lib.dcl_debug_geometry_small_stamps(st.ctypes.data_as(ctypes.c_void_p))
    if i >= 3:
        acc += np.diff(st[:10].astype(np.int64)) * 0.01
        chain += np.diff(np.concatenate([st[2:3], st[16:24]]).astype(np.int64)) * 0.01
print("k_geometry_small, %d crop(s), workgroup 0: %.1f us inside the kernel" % (b, acc.sum() / reps))
for n_, t in zip(names, acc / reps):
    print("  %-34s %6.2f us" % (n_, t))
print("  mask-chain stages (conv set, pool set per level): " + "  ".join("%.2f" % t for t in chain / reps))

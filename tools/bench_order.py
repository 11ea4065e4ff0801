#!/usr/bin/env python3
"""Row-order launch (csrc/row_order.hip) stand-alone: us per call for one layer's rows at several sizes."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
ops, sp = dcl.ops, dcl.spconv.ops
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
rng = np.random.default_rng(0)
for b, S, per in ((1, 32, 2000), (4, 32, 2000), (8, 32, 3000), (32, 32, 700), (32, 32, 1900), (32, 16, 700), (32, 8, 300)):
    rows = []
    for bi in range(b):
        lin = rng.choice(S ** 3, size=per, replace=False)
        rows.append(np.stack([np.full(per, bi), lin // (S * S), (lin // S) % S, lin % S], 1))
    idx = torch.from_numpy(np.concatenate(rows).astype(np.int32)).cuda()
    aset = ops.grid_from_indices(idx, b, S)
    for subm in (False, True):
        out, nbr = sp.build_rulebook(aset, 3, 1, 1, subm)
        us = timeit(lambda: ops.order_rows(out, aset.mask, subm))
        print("b=%2d S=%2d subm=%d rows %7d (%3d windows): %.1f us per call (incl. the wrapper's allocations)" % (b, S, subm, out.n, (out.n + 8191) // 8192, us), flush=True)

# phase stamps of the window kernel's first workgroup (diagnostic library only)
import ctypes
with dcl._native.diagnostic_library() as L:
    out, nbr = sp.build_rulebook(aset, 3, 1, 1, False)
    ops.order_rows(out, aset.mask, False)
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 16)()
    L.dcl_debug_order_stamps(st)
    t = [int(x) for x in st]
    names = ["load+keys", "->radix", "p0 count", "p0 scan", "p0 scatter", "p1 count", "p1 scan", "p1 scatter", "p2 count", "p2 scan", "p2 scatter(+sync)", "order+tiles", "drain+barrier", "ticket", "prefix"]
    print("window kernel phases of workgroup (0,0), us:", ", ".join("%s %.2f" % (n, (t[i + 1] - t[i]) * 0.01) for i, n in enumerate(names)))

#!/usr/bin/env python3
"""Per-layer timing of the sparse-conv kernels on the backbone's real active sets (synthetic crops, bs=32):
kernel variant 0 = default, 2 = MFMA without LDS staging, 3 = LDS-weights kernel.  usage: bench_conv.py [N] [modes]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
ops, sp = dcl.ops, dcl.spconv.ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
modes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
DIAG = None
if modes != [0] or len(sys.argv) > 3:             # kernel-variant hooks exist in the diagnostic library only; mode 0 alone runs the PRODUCT library
    from _diag import use_diag
    DIAG = use_diag(dcl)
if len(sys.argv) > 3:
    ops.N.lib().dcl_debug_conv_split(int(sys.argv[3]))
b, S = int(os.environ.get('DCL_BENCH_B', '32')), 64
data = dcl.synth.make_batch(b, n, 64)
occ = data["inp"]["occupied_voxels"].int().cuda().contiguous()
aset = ops.grid_from_indices(occ, b, S)
chans = [7, 16, 32, 32, 64, 64, 128, 128, 256]
feat = torch.randn(occ.shape[0], 7, device="cuda")
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
tot = {m: 0.0 for m in modes}
for lvl in range(4):
    c0, c1, c2 = chans[2 * lvl], chans[2 * lvl + 1], chans[2 * lvl + 2]
    out, nbr1 = sp.build_rulebook(aset, 3, 1, 1, False)
    _, nbr2 = sp.build_rulebook(out, 3, 1, 1, True)
    pool, nbr3 = sp.build_rulebook(out, 3, 2, 1, False)
    W1 = torch.randn(27, c0, c1, device="cuda") * 0.05
    W2 = torch.randn(27, c1, c2, device="cuda") * 0.05
    x1 = None
    for name, f_in, nbr, W, subm in (("conv", feat, nbr1, W1, False), ("subm", None, nbr2, W2, True)):
        if f_in is None: f_in = x1
        # the runner's row order (csrc/row_order.hip) for the layers it orders (levels >= 2), as the product path uses it
        order = ops.order_rows(out, (out if subm else aset).mask, subm) if lvl >= 2 and os.environ.get("DCL_BENCH_NO_ORDER") != "1" else None
        pairs = int((nbr[:, :out.n] >= 0).sum())
        flop = 2.0 * pairs * W.shape[1] * W.shape[2]
        line = "L%d %s %3d->%3d rows %6d pairs %8d density %.2f :" % (lvl, name, W.shape[1], W.shape[2], out.n, pairs, pairs / (27.0 * out.n))
        for m in modes:
            if DIAG is not None:
                ops.N.lib().dcl_debug_force_valu_conv(m)
            us = timeit(lambda: ops.sparse_conv(f_in, nbr, out.n, W, subm, order=order))
            tot[m] += us
            line += "  mode%d %7.1f us (%5.1f TF useful)" % (m, us, flop / us / 1e6)
        if DIAG is not None:
            ops.N.lib().dcl_debug_force_valu_conv(0)
        print(line, flush=True)
        if x1 is None: x1 = ops.sparse_conv(f_in, nbr, out.n, W, subm)
    x2 = ops.sparse_conv(x1, nbr2, out.n, W2, True)
    us = timeit(lambda: ops.sparse_avgpool(x2, nbr3, pool.n))
    print("L%d pool rows %d: %.1f us" % (lvl, pool.n, us))
    feat = ops.sparse_avgpool(x2, nbr3, pool.n)
    aset = pool
print("totals (conv only):", {m: round(v, 1) for m, v in tot.items()})

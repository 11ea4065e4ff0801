import importlib, os, sys
import torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/tools") else ".")
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
DIAG = use_diag(dcl)          # kernel-variant hooks exist in the diagnostic library only
ops, sp = dcl.ops, dcl.spconv.ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b, S = 32, 64
data = dcl.synth.make_batch(b, n, 64)
aset = ops.grid_from_indices(data["inp"]["occupied_voxels"].int().cuda().contiguous(), b, S)
chans = [7, 16, 32, 32, 64, 64, 128, 128, 256]
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
feat_rows = aset.n
for lvl in range(4):
    c0, c1 = chans[2 * lvl], chans[2 * lvl + 1]
    out, nbr = sp.build_rulebook(aset, 3, 1, 1, False)
    if lvl > 0:
        feat = torch.randn(feat_rows, c0, device="cuda")
        W = torch.randn(27, c0, c1, device="cuda") * 0.05
        valid = (nbr[:, :out.n] >= 0)
        v3 = valid.view(3, 3, 3, out.n)
        key = torch.zeros(out.n, dtype=torch.int64, device="cuda")
        for ax in range(3):
            for q in range(3):
                anyp = v3.select(ax, q).reshape(9, out.n).any(0)
                key = key * 2 + anyp.long()
        perm = torch.argsort(key, stable=True)
        nbr_s = nbr[:, :out.n][:, perm].contiguous()
        t0 = timeit(lambda: ops.sparse_conv(feat, nbr, out.n, W, False))
        t1 = timeit(lambda: ops.sparse_conv(feat, nbr_s, out.n, W, False))
        # work dealt in USED chunks (dcl_debug_conv_balance): per 128-row tile of the sorted order, the mask of offsets any
        # of its rows uses and the prefix of their counts
        import ctypes
        nblk = (out.n + 127) // 128
        vs = (nbr_s >= 0)
        pad = nblk * 128 - out.n
        if pad:
            vs = torch.cat([vs, torch.zeros(27, pad, dtype=torch.bool, device="cuda")], 1)
        used = vs.view(27, nblk, 128).any(2)                                        # (27, nblk)
        smask = (used.long() << torch.arange(27, device="cuda").view(27, 1)).sum(0)  # bit = offset (visiting order of a conv)
        cnt = used.sum(0)
        bal = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), cnt.cumsum(0), smask]).int().contiguous()
        lib = ops.N.lib()
        want = ops.sparse_conv(feat, nbr_s, out.n, W, False)
        lib.dcl_debug_conv_balance(ctypes.c_void_p(bal.data_ptr()))
        try:
            got = ops.sparse_conv(feat, nbr_s, out.n, W, False)
            t3 = timeit(lambda: ops.sparse_conv(feat, nbr_s, out.n, W, False))
        finally:
            lib.dcl_debug_conv_balance(None)
        err = float((got - want).abs().max())
        print("   sorted + work dealt in used chunks: %.1f us (max |diff| to the plain decomposition %.2e; used steps per tile %.1f of 27)"
              % (t3, err, float(cnt.float().mean())))
        # random permutation: locality loss alone
        rp = torch.randperm(out.n, device="cuda")
        nbr_r = nbr[:, :out.n][:, rp].contiguous()
        t2 = timeit(lambda: ops.sparse_conv(feat, nbr_r, out.n, W, False))
        print("L%d conv %d->%d rows %d: natural %.1f us, key9-sorted %.1f us, random order %.1f us" % (lvl, c0, c1, out.n, t0, t1, t2))
    pool, _ = sp.build_rulebook(out, 3, 2, 1, False)
    aset = pool
    feat_rows = pool.n

import sys, importlib, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
dcl = importlib.import_module("dcl-net_amd")
from oracle import native as oracle
from test_gpu_ops import rand_voxels, cuda
for (cin, cout, subm) in [(128, 256, True), (128, 128, False), (64, 128, True)]:
    rng = np.random.default_rng(cin + cout)
    b, S = 2, 8
    idx = rand_voxels(rng, b, S, 150)
    feat = rng.normal(size=(idx.shape[0], cin)).astype(np.float32)
    W = (rng.normal(size=(3, 3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    aset = dcl.ops.grid_from_indices(cuda(idx), b, S)
    out, nbr = dcl.spconv.ops.build_rulebook(aset, 3, 1, 1, subm)
    n_out = idx.shape[0] if subm else out.n
    r_out, r_pairs, r_num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, 3, 1, 1, 1, subm=subm)
    want = oracle.indice_conv(feat, W, r_pairs, r_num, n_out, subm=subm)
    Wd = cuda(W).reshape(27, cin, cout).contiguous()
    fd = cuda(feat)
    for rep in range(3):
        got = dcl.ops.sparse_conv(fd, nbr, n_out, Wd, subm).cpu().numpy()
        err = np.abs(got - want)
        bad = err > 1e-3
        print(cin, cout, subm, "n_out", n_out, "maxerr", err.max(), "bad frac", bad.mean(),
              "bad rows", np.unique(np.where(bad)[0])[:20], "bad cols", np.unique(np.where(bad)[1])[:40])
    # single-offset weights: find which offsets are broken
    for k in range(27):
        Wk = np.zeros_like(W).reshape(27, cin, cout); Wk[k] = W.reshape(27, cin, cout)[k]
        wantk = oracle.indice_conv(feat, Wk.reshape(3,3,3,cin,cout), r_pairs, r_num, n_out, subm=subm)
        gotk = dcl.ops.sparse_conv(fd, nbr, n_out, cuda(Wk), subm).cpu().numpy()
        e = np.abs(gotk - wantk).max()
        if e > 1e-4: print("  offset", k, "err", e)
    # single input channel
    for ci in range(0, cin, 1):
        Wc = np.zeros_like(W).reshape(27, cin, cout); Wc[:, ci] = W.reshape(27, cin, cout)[:, ci]
        wantc = oracle.indice_conv(feat, Wc.reshape(3,3,3,cin,cout), r_pairs, r_num, n_out, subm=subm)
        gotc = dcl.ops.sparse_conv(fd, nbr, n_out, cuda(Wc), subm).cpu().numpy()
        e = np.abs(gotc - wantc).max()
        if e > 1e-4: print("  channel", ci, "err", e)

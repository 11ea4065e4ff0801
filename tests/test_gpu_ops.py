"""GPU parity of every C-ABI kernel against the CPU oracle (oracle/native.py), same seeded inputs.
Index / integer outputs must be bit-exact; float outputs bit-exact where the arithmetic order is pinned
(voxelize_fp, avg-pool, interpolation, distances), else within the stated fp32 tolerance (GEMM-shaped sums)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cuda(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def enter_diag(dcl, request):
    """switch the package to the DIAGNOSTIC library (tests/_diag/libdclnet_hip_diag.so, built with -DDCL_DIAG: the product
    library exports no dcl_debug_* hook and carries no superseded kernel variant) until the test ends; returns its handle"""
    ctx = dcl._native.diagnostic_library()
    lib = ctx.__enter__()
    request.addfinalizer(lambda: ctx.__exit__(None, None, None))
    return lib


def rand_voxels(rng, b, S, n_per):
    """unsorted unique voxel rows per crop, batch-sorted like the loaders produce"""
    rows = []
    for bi in range(b):
        lin = rng.choice(S ** 3, size=n_per, replace=False)
        xyz = np.stack([lin // (S * S), (lin // S) % S, lin % S], 1)
        rows.append(np.concatenate([np.full((n_per, 1), bi), xyz], 1))
    return np.concatenate(rows, 0).astype(np.int32)


def pair_sets(pairs, num):
    return [set(zip(pairs[k, 0, :num[k]].tolist(), pairs[k, 1, :num[k]].tolist())) for k in range(pairs.shape[0])]


# ------------------------------------------------------------------------------------------- voxelize
def test_voxelize_idx_host_matches_oracle(dcl, oracle):
    rng = np.random.default_rng(0)
    coords = np.concatenate([np.repeat(np.arange(3), 500)[:, None], rng.integers(0, 12, (1500, 3))], 1).astype(np.int64)
    oc, im, om = dcl.ops.voxelize_idx(torch.from_numpy(coords), 3, 4)
    rc, rm, rom = oracle.voxelize_idx(coords, 3, 4)
    assert np.array_equal(oc.numpy(), rc) and np.array_equal(im.numpy(), rm) and np.array_equal(om.numpy(), rom)


def test_voxelize_idx_gpu_bit_exact(dcl, oracle):
    """device voxelize_idx == host/oracle hashing: first-encounter ids, ascending point lists, first-point coords"""
    rng = np.random.default_rng(5)
    for b, n, S, spread in ((3, 500, 64, 12), (2, 4000, 64, 20), (4, 257, 16, 16), (1, 1, 8, 1)):
        coords = np.concatenate([np.repeat(np.arange(b), n)[:, None], rng.integers(0, spread, (b * n, 3))], 1).astype(np.int64)
        oc, im, om = dcl.ops.voxelize_idx_gpu(torch.from_numpy(coords).cuda(), b, S, 4)
        rc, rm, rom = oracle.voxelize_idx(coords, b, 4)
        assert np.array_equal(im.cpu().numpy(), rm)
        assert np.array_equal(oc.cpu().numpy(), rc)
        assert np.array_equal(om.cpu().numpy(), rom)
    d = dcl.synth.make_batch(3, 1024, 1024)                    # and on real synthetic crops, through the loader contract
    coords = torch.cat([torch.repeat_interleave(torch.arange(3), 1024)[:, None],
                        ((d["inp"]["feats"][:, 4:7] + 0.192) / 0.006).long()], 1).contiguous()
    oc, im, om = dcl.ops.voxelize_idx_gpu(coords.cuda(), 3, 64, 4)
    assert torch.equal(oc.cpu(), d["inp"]["occupied_voxels"]) and torch.equal(im.cpu(), d["inp"]["p2v_maps"])
    assert torch.equal(om.cpu(), d["inp"]["v2p_maps"])
    with pytest.raises(RuntimeError):
        bad = torch.tensor([[0, 1, 2, 99]], dtype=torch.int64).cuda()
        dcl.ops.voxelize_idx_gpu(bad, 1, 64, 4)


def test_voxelize_fp_bit_exact(dcl, oracle):
    rng = np.random.default_rng(1)
    coords = np.concatenate([np.repeat(np.arange(2), 700)[:, None], rng.integers(0, 9, (1400, 3))], 1).astype(np.int64)
    _, _, rules = oracle.voxelize_idx(coords, 2, 4)
    feats = rng.normal(size=(1400, 7)).astype(np.float32)
    for mode in (4, 3):
        got = dcl.ops.voxelize_fp(cuda(feats), cuda(rules), mode).cpu().numpy()
        assert np.array_equal(got, oracle.voxelize_fp(feats, rules, mode))


# ------------------------------------------------------------------------------------------- rulebooks
@pytest.mark.parametrize("S,ks,st,pad,subm", [(16, 3, 1, 1, False), (16, 3, 2, 1, False), (8, 3, 1, 1, True),
                                               (64, 3, 1, 1, False), (64, 3, 2, 1, False), (4, 3, 2, 1, False)])
def test_rulebook_matches_oracle(dcl, oracle, S, ks, st, pad, subm):
    rng = np.random.default_rng(S * 10 + st)
    b = 3
    idx = rand_voxels(rng, b, S, min(S ** 3 // 3, 300))
    aset = dcl.ops.grid_from_indices(cuda(idx), b, S)
    out, nbr = dcl.spconv.ops.build_rulebook(aset, ks, st, pad, subm)
    r_out, r_pairs, r_num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, ks, st, pad, 1, subm=subm)
    n_out = idx.shape[0] if subm else out.n
    assert n_out == r_out.shape[0]
    got_idx = (aset.indices if subm else out.indices).cpu().numpy()
    assert np.array_equal(got_idx, r_out)                      # ascending linear index == sort+unique order
    pairs, num = dcl.ops.rulebook_to_pairs(nbr, n_out, idx.shape[0])
    pairs, num = pairs.cpu().numpy(), num.cpu().numpy()
    assert np.array_equal(num, r_num)
    assert pair_sets(pairs, num) == pair_sets(r_pairs, r_num)
    # gather table <-> pair list consistency
    nb = nbr.cpu().numpy()
    for k in range(ks ** 3):
        assert (nb[k, :n_out] >= 0).sum() == r_num[k]


def test_rulebook_empty_and_single(dcl, oracle):
    one = np.array([[0, 5, 5, 5]], np.int32)
    aset = dcl.ops.grid_from_indices(cuda(one), 1, 16)
    out, nbr = dcl.spconv.ops.build_rulebook(aset, 3, 1, 1, False)
    assert out.n == 27
    out2, _ = dcl.spconv.ops.build_rulebook(aset, 3, 2, 1, False)
    assert out2.n == 8                                        # odd coordinates reach 2 outputs per axis
    corner = np.array([[0, 0, 0, 0]], np.int32)
    o3, _ = dcl.spconv.ops.build_rulebook(dcl.ops.grid_from_indices(cuda(corner), 1, 16), 3, 1, 1, False)
    assert o3.n == 8


# ------------------------------------------------------------------------------------------- sparse conv / pool
@pytest.mark.parametrize("cin,cout,subm", [(7, 16, False), (16, 32, True), (32, 32, False), (64, 128, True),
                                           (128, 256, True), (128, 128, False)])
def test_sparse_conv_matches_oracle(request, dcl, oracle, cin, cout, subm):
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
    got = dcl.ops.sparse_conv(cuda(feat), nbr, n_out, Wd, subm).cpu().numpy()
    tol = 2e-5 * max(1.0, np.abs(want).max())                  # fp32, different summation association
    assert np.abs(got - want).max() <= tol
    # MFMA kernel vs plain VALU kernel on the device (A/B)
    lib = enter_diag(dcl, request)      # kernel variants live in the diagnostic library only (tests/_diag)
    for mode in (1, 2, 4, 5):                                  # 1: VALU, 2: MFMA no LDS, 4: reg-staged tiles, 5: 4-wave 128x64
        lib.dcl_debug_force_valu_conv(mode)
        try:
            alt = dcl.ops.sparse_conv(cuda(feat), nbr, n_out, Wd, subm).cpu().numpy()
        finally:
            lib.dcl_debug_force_valu_conv(0)
        assert np.abs(alt - want).max() <= tol
    # folded BN + ReLU epilogue
    s = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    t = rng.normal(size=cout).astype(np.float32)
    got2 = dcl.ops.sparse_conv(cuda(feat), nbr, n_out, Wd, subm, cuda(s), cuda(t), True).cpu().numpy()
    assert np.abs(got2 - np.maximum(want * s + t, 0)).max() <= 2 * tol


def test_stem_conv_forms_match_the_oracle_on_either_side_of_the_switch(dcl, oracle):
    """The 7 -> 16 stem takes a lane per output row for launches of many rows and four lanes per row for a handful of crops (two
    summation orders).  An op-level call picks by its row count; inside the backbone runner the choice goes by crops x 3600 --
    the same number launch by launch and under graph capture (round-5 review).  Both forms against the oracle here: 4 crops
    (four lanes) and 32 crops (one lane) of the same synthetic scene."""
    ops, sp = dcl.ops, dcl.spconv.ops
    S = 64
    rng = np.random.default_rng(9)
    W = (rng.normal(size=(3, 3, 3, 7, 16)) / np.sqrt(9 * 7)).astype(np.float32)
    for b in (4, 32):
        occ = dcl.synth.make_batch(b, 1024, 64)["inp"]["occupied_voxels"].int()
        aset = ops.grid_from_indices(occ.cuda().contiguous(), b, S)
        out, nbr = sp.build_rulebook(aset, 3, 1, 1, False)
        assert (out.n <= 49152) == (b == 4)                    # the two launches sit on either side of the form switch
        feat = np.random.default_rng(1).normal(size=(occ.shape[0], 7)).astype(np.float32)
        got = ops.sparse_conv(cuda(feat), nbr, out.n, cuda(W).reshape(27, 7, 16).contiguous(), False).cpu().numpy()
        o_ids, o_pairs, o_num, _ = oracle.get_indice_pairs(occ.numpy(), b, [S] * 3, 3, 1, 1, 1, subm=False)
        want = oracle.indice_conv(feat, W, o_pairs, o_num, o_ids.shape[0], subm=False)
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), b


def test_conv_layers_of_a_32_crop_batch_op_by_op_match_the_oracle(dcl, oracle):
    """the conv / pool ops called LAYER BY LAYER on the real active sets of 32 crops (explicit gather tables, the op-level
    path tools/bench_conv.py profiles: profiles/*_conv_layers_kernel_stats.csv) -- every layer's output against the oracle's
    indice_conv / indice_avgpool on the same inputs; the deep levels also in the runner's row order"""
    ops, sp = dcl.ops, dcl.spconv.ops
    b, S = 32, 64
    data = dcl.synth.make_batch(b, 1024, 64)
    occ = data["inp"]["occupied_voxels"].int()
    aset = ops.grid_from_indices(occ.cuda().contiguous(), b, S)
    idx_in = occ.numpy()
    chans = [7, 16, 32, 32, 64, 64, 128, 128, 256]
    rng = np.random.default_rng(5)
    feat = rng.normal(size=(occ.shape[0], 7)).astype(np.float32)
    for lvl in range(4):
        c0, c1, c2 = chans[2 * lvl], chans[2 * lvl + 1], chans[2 * lvl + 2]
        out, nbr1 = sp.build_rulebook(aset, 3, 1, 1, False)
        _, nbr2 = sp.build_rulebook(out, 3, 1, 1, True)
        pool, nbr3 = sp.build_rulebook(out, 3, 2, 1, False)
        o_ids, o_pairs, o_num, oshape = oracle.get_indice_pairs(idx_in, b, [S] * 3, 3, 1, 1, 1, subm=False)
        assert np.array_equal(out.indices[:out.n].cpu().numpy(), o_ids)
        W1 = (rng.normal(size=(3, 3, 3, c0, c1)) / np.sqrt(9 * c0)).astype(np.float32)
        W2 = (rng.normal(size=(3, 3, 3, c1, c2)) / np.sqrt(9 * c1)).astype(np.float32)
        want1 = np.maximum(oracle.indice_conv(feat, W1, o_pairs, o_num, o_ids.shape[0], subm=False), 0)
        got1 = ops.sparse_conv(cuda(feat), nbr1, out.n, cuda(W1).reshape(27, c0, c1).contiguous(), False, relu=True,
                               scale=torch.ones(c1, device="cuda"), shift=torch.zeros(c1, device="cuda"))
        tol = 2e-5 * max(1.0, np.abs(want1).max())
        assert np.abs(got1.cpu().numpy() - want1).max() <= tol, (lvl, "conv")
        s_ids, s_pairs, s_num, _ = oracle.get_indice_pairs(o_ids, b, oshape, 3, 1, 1, 1, subm=True)
        want2 = oracle.indice_conv(want1, W2, s_pairs, s_num, o_ids.shape[0], subm=True)
        W2d = cuda(W2).reshape(27, c1, c2).contiguous()
        x1 = cuda(want1)
        got2 = ops.sparse_conv(x1, nbr2, out.n, W2d, True)
        tol2 = 2e-5 * max(1.0, np.abs(want2).max())
        assert np.abs(got2.cpu().numpy() - want2).max() <= tol2, (lvl, "subm")
        if lvl >= 2:                                            # the runner's row order for the deep levels
            order = ops.order_rows(out, out.mask, True)
            got2o = ops.sparse_conv(x1, nbr2, out.n, W2d, True, order=order)
            assert np.abs(got2o.cpu().numpy() - want2).max() <= tol2, (lvl, "subm ordered")
            order1 = ops.order_rows(out, aset.mask, False)
            got1o = ops.sparse_conv(cuda(feat), nbr1, out.n, cuda(W1).reshape(27, c0, c1).contiguous(), False, relu=True,
                                    scale=torch.ones(c1, device="cuda"), shift=torch.zeros(c1, device="cuda"), order=order1)
            assert np.abs(got1o.cpu().numpy() - want1).max() <= tol, (lvl, "conv ordered")
        p_ids, p_pairs, p_num, pshape = oracle.get_indice_pairs(o_ids, b, oshape, 3, 2, 1, 1, subm=False)
        want3, _ = oracle.indice_avgpool(want2, p_pairs, p_num, p_ids.shape[0])
        got3 = ops.sparse_avgpool(cuda(want2), nbr3, pool.n)
        assert np.array_equal(pool.indices[:pool.n].cpu().numpy(), p_ids)
        assert np.array_equal(got3.cpu().numpy(), want3), (lvl, "pool")
        feat, idx_in, aset, S = want3, p_ids, pool, pshape[0]


@pytest.mark.parametrize("cin,cout,subm", [(128, 256, True), (128, 128, False), (64, 128, True), (32, 64, True)])
def test_sparse_conv_split_k_in_launch_combine(request, dcl, oracle, cin, cout, subm):
    """split-K launches combine their partial tiles inside the launch (last-arriver ticket per tile, partials added in split
    order): a mid-size layer (several hundred workgroups, like the backbone's deep levels at bs 32) with forced split counts
    equals the unsplit launch within the fp32 association tolerance, equals the oracle, leaves its ticket counters at zero
    (back-to-back launches on the same scratch) and is bit-reproducible over many launches racing each other on the GPU"""
    rng = np.random.default_rng(cin + 7 * cout)
    b, S = 8, 16
    idx = rand_voxels(rng, b, S, 700)
    feat = rng.normal(size=(idx.shape[0], cin)).astype(np.float32)
    W = (rng.normal(size=(3, 3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    aset = dcl.ops.grid_from_indices(cuda(idx), b, S)
    out, nbr = dcl.spconv.ops.build_rulebook(aset, 3, 1, 1, subm)
    n_out = idx.shape[0] if subm else out.n
    r_out, r_pairs, r_num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, 3, 1, 1, 1, subm=subm)
    want = oracle.indice_conv(feat, W, r_pairs, r_num, n_out, subm=subm)
    tol = 2e-5 * max(1.0, np.abs(want).max())
    f, Wd = cuda(feat), cuda(W).reshape(27, cin, cout).contiguous()
    s_, t_ = cuda(rng.uniform(0.5, 1.5, cout).astype(np.float32)), cuda(rng.normal(size=cout).astype(np.float32))
    lib = enter_diag(dcl, request)      # kernel variants live in the diagnostic library only (tests/_diag)
    res = {}
    try:
        for ns in (-2, 2, 5, 8, 0):                      # -2: never split ... 0: the automatic choice
            lib.dcl_debug_conv_split(ns)
            res[ns] = dcl.ops.sparse_conv(f, nbr, n_out, Wd, subm)
            assert np.abs(res[ns].cpu().numpy() - want).max() <= tol, ns
            if ns > 0:
                again = [dcl.ops.sparse_conv(f, nbr, n_out, Wd, subm) for _ in range(30)]      # 30 launches in flight
                assert all(torch.equal(a, res[ns]) for a in again), ns
                e1 = dcl.ops.sparse_conv(f, nbr, n_out, Wd, subm, s_, t_, True)
                ref = torch.relu(res[ns] * s_ + t_)
                assert float((e1 - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    finally:
        lib.dcl_debug_conv_split(0)
    # uneven load: the same layer racing a different, smaller one on a second stream (their workgroups interleave on the
    # CUs and finish at unrelated times): every launch must still reproduce its own first result bit for bit
    idx2 = rand_voxels(rng, 3, S, 300)
    f2 = cuda(rng.normal(size=(idx2.shape[0], cin)).astype(np.float32))
    out2, nbr2 = dcl.spconv.ops.build_rulebook(dcl.ops.grid_from_indices(cuda(idx2), 3, S), 3, 1, 1, subm)
    n2 = idx2.shape[0] if subm else out2.n
    ref_a, ref_b = dcl.ops.sparse_conv(f, nbr, n_out, Wd, subm), dcl.ops.sparse_conv(f2, nbr2, n2, Wd, subm)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    got_a, got_b = [], []
    for _ in range(12):
        with torch.cuda.stream(sa):
            got_a.append(dcl.ops.sparse_conv(f, nbr, n_out, Wd, subm))
        with torch.cuda.stream(sb):
            got_b.append(dcl.ops.sparse_conv(f2, nbr2, n2, Wd, subm))
            got_b.append(dcl.ops.sparse_conv(f2, nbr2, n2, Wd, subm))
    torch.cuda.synchronize()
    assert all(torch.equal(a, ref_a) for a in got_a) and all(torch.equal(x, ref_b) for x in got_b)


@pytest.mark.parametrize("cin,cout", [(16, 32), (32, 32), (32, 64), (64, 128)])
def test_sparse_conv_decompositions_agree_across_sizes(request, dcl, cin, cout):
    """row counts from a handful of tiles to tens of thousands of rows drive the launcher through its decompositions
    (deferred-combine few-row mode on 64-row tiles, aligned split-K, stream-K, whole tiles; the filter-resident kernel for
    Cin 16 / 32 -> 32 channels once a launch is not a few-row one); every one must agree with the plain VALU kernel on the
    same rulebook, with and without the BN+ReLU epilogue -- and the filter-resident kernel with the LDS-DMA kernel it replaces"""
    rng = np.random.default_rng(cin)
    lib = enter_diag(dcl, request)      # kernel variants live in the diagnostic library only (tests/_diag)
    W = cuda((rng.normal(size=(27, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32))
    s_, t_ = cuda(rng.uniform(0.5, 1.5, cout).astype(np.float32)), cuda(rng.normal(size=cout).astype(np.float32))
    for b, S, per in ((1, 8, 130), (2, 16, 900), (6, 16, 1500), (8, 32, 5000)):
        idx = rand_voxels(rng, b, S, per)
        aset = dcl.ops.grid_from_indices(cuda(idx), b, S)
        for subm in (True, False):
            out, nbr = dcl.spconv.ops.build_rulebook(aset, 3, 1, 1, subm)
            n_out = idx.shape[0] if subm else out.n
            feat = cuda(rng.normal(size=(idx.shape[0], cin)).astype(np.float32))
            got = dcl.ops.sparse_conv(feat, nbr, n_out, W, subm)
            got2 = dcl.ops.sparse_conv(feat, nbr, n_out, W, subm, s_, t_, True)
            lib.dcl_debug_force_valu_conv(1)
            try:
                ref = dcl.ops.sparse_conv(feat, nbr, n_out, W, subm)
            finally:
                lib.dcl_debug_force_valu_conv(0)
            tol = 2e-5 * max(1.0, float(ref.abs().max()))
            assert float((got - ref).abs().max()) <= tol, (b, S, per, subm)
            assert float((got2 - torch.relu(ref * s_ + t_)).abs().max()) <= 2 * tol, (b, S, per, subm)
            assert torch.equal(got, dcl.ops.sparse_conv(feat, nbr, n_out, W, subm))          # reproducible
            if cout == 32:                                                                   # A/B: LDS-DMA kernel for the same launch
                lib.dcl_debug_conv_wlds(0)
                try:
                    dma = dcl.ops.sparse_conv(feat, nbr, n_out, W, subm)
                finally:
                    lib.dcl_debug_conv_wlds(1)
                assert float((dma - ref).abs().max()) <= tol, (b, S, per, subm)
                assert torch.equal(got, dma) == (n_out <= 24576), (b, S, per, subm)           # few-row launches never take it


def _plane_key(valid27):
    """9-bit key of csrc/row_order.hip::plane_key from a (27, n) bool table of present neighbours (k = kz + 3 ky + 9 kx)"""
    v3 = valid27.view(3, 3, 3, -1)                                             # [kx][ky][kz]
    key = torch.zeros(valid27.shape[1], dtype=torch.int64, device=valid27.device)
    bit = 0
    for ax in range(3):                                                        # x planes: bits 0-2, y: 3-5, z: 6-8
        for q in range(3):
            key |= v3.select(ax, q).reshape(9, -1).any(0).long() << bit
            bit += 1
    return key


@pytest.mark.parametrize("S,per,subm", [(32, 3000, False), (16, 700, True), (8, 90, False), (64, 1500, True), (32, 9000, True)])
def test_row_order_is_the_stable_sort_by_neighbourhood_shape(dcl, S, per, subm):
    """dcl_order_rows: inside every window of 8192 consecutive rows, order == the STABLE argsort of the 9-bit plane key
    computed from the layer's own gather table; the tiles' step masks / used-step prefix follow from it, and two runs give
    the same bits"""
    rng = np.random.default_rng(S + per)
    b = 5
    idx = rand_voxels(rng, b, S, per)
    aset = dcl.ops.grid_from_indices(cuda(idx), b, S)
    out, nbr = dcl.spconv.ops.build_rulebook(aset, 3, 1, 1, subm)
    n = out.n
    order, bal, smask = dcl.ops.order_rows(out, aset.mask, subm)
    valid = nbr[:, :n] >= 0
    key = _plane_key(valid)
    want = torch.cat([w0 + torch.argsort(key[w0:w0 + 8192], stable=True) for w0 in range(0, n, 8192)]).int()
    assert torch.equal(order, want)
    assert torch.equal(torch.sort(order.long())[0], torch.arange(n, device="cuda"))       # a permutation of the rows
    nt = (n + 127) // 128
    vs = torch.cat([valid[:, want.long()], torch.zeros(27, nt * 128 - n, dtype=torch.bool, device="cuda")], 1)
    used = vs.view(27, nt, 128).any(2)                                          # [offset k][tile]
    steps = list(range(27)) if not subm else [13] + list(range(13)) + list(range(14, 27))    # visiting order (centre first)
    want_mask = sum(used[k].long() << s for s, k in enumerate(steps))
    assert torch.equal(smask.long() & 0xFFFFFFFF, want_mask)
    cnt = used.sum(0)
    assert torch.equal(bal.long(), torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), cnt.cumsum(0)]))
    again = dcl.ops.order_rows(out, aset.mask, subm)
    assert all(torch.equal(x, y) for x, y in zip((order, bal, smask), again))


@pytest.mark.parametrize("cin,cout,subm", [(64, 64, False), (128, 128, False), (32, 64, True), (128, 256, True), (32, 32, False),
                                           (16, 32, True)])
def test_sparse_conv_in_row_order_matches_the_plain_launch(request, dcl, oracle, cin, cout, subm):
    """the product path of the big launches: rows ordered on the device (dcl_order_rows), work dealt in USED chunks -- same
    values as the oracle / the natural-order launch within the fp32 summation tolerance, every output row written exactly
    once, reproducible bits; A/B modes of the diagnostic library (order ignored / order with nominal units) agree too"""
    rng = np.random.default_rng(cin + cout)
    b, S = 6, 32
    idx = rand_voxels(rng, b, S, 4000 if not subm else 6000)
    aset = dcl.ops.grid_from_indices(cuda(idx), b, S)
    out, nbr = dcl.spconv.ops.build_rulebook(aset, 3, 1, 1, subm)
    n = out.n
    feat = cuda(rng.normal(size=(idx.shape[0], cin)).astype(np.float32))
    W = cuda((rng.normal(size=(27, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32))
    order = dcl.ops.order_rows(out, aset.mask, subm)
    plain = dcl.ops.sparse_conv(feat, nbr, n, W, subm)
    got = dcl.ops.sparse_conv(feat, nbr, n, W, subm, order=order)
    tol = 2e-5 * max(1.0, float(plain.abs().max()))
    assert float((got - plain).abs().max()) <= tol
    assert torch.equal(got, dcl.ops.sparse_conv(feat, nbr, n, W, subm, order=order))
    s_ = cuda(rng.uniform(0.5, 1.5, cout).astype(np.float32))
    t_ = cuda(rng.normal(size=cout).astype(np.float32))
    got2 = dcl.ops.sparse_conv(feat, nbr, n, W, subm, s_, t_, True, order=order)
    assert float((got2 - torch.relu(plain * s_ + t_)).abs().max()) <= 2 * tol
    # a poisoned output buffer: every row must be written (the order is a permutation of the rows)
    assert bool(torch.isfinite(got).all())
    r_out, r_pairs, r_num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, 3, 1, 1, 1, subm=subm)
    sub = slice(0, 20000)
    want = oracle.indice_conv(feat.cpu().numpy(), W.cpu().numpy().reshape(3, 3, 3, cin, cout), r_pairs, r_num, n, subm=subm)
    assert np.abs(got.cpu().numpy()[sub] - want[sub]).max() <= 2e-5 * max(1.0, np.abs(want).max())
    lib = enter_diag(dcl, request)
    for mode in (1, 2):
        lib.dcl_debug_conv_order_mode(mode)
        try:
            alt = dcl.ops.sparse_conv(feat, nbr, n, W, subm, order=order)
        finally:
            lib.dcl_debug_conv_order_mode(0)
        assert float((alt - plain).abs().max()) <= tol, mode


@pytest.mark.parametrize("c", [32, 7])
def test_sparse_avgpool_bit_exact(dcl, oracle, c):
    rng = np.random.default_rng(c)
    b, S = 3, 16
    idx = rand_voxels(rng, b, S, 400)
    feat = rng.normal(size=(idx.shape[0], c)).astype(np.float32)
    aset = dcl.ops.grid_from_indices(cuda(idx), b, S)
    out, nbr = dcl.spconv.ops.build_rulebook(aset, 3, 2, 1, False)
    r_out, r_pairs, r_num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, 3, 2, 1, 1)
    want, want_rf = oracle.indice_avgpool(feat, r_pairs, r_num, r_out.shape[0])
    got, rf = dcl.ops.sparse_avgpool(cuda(feat), nbr, out.n, want_rf=True)
    assert np.array_equal(rf.cpu().numpy(), want_rf)
    assert np.array_equal(got.cpu().numpy(), want)             # same divide-then-add order, k ascending


def test_sparse_avgpool_quotients_outside_the_fast_range_bit_exact(dcl, oracle):
    """the pool's f / rf is computed as RN(1/rf) and two FMAs where that is provably the IEEE quotient (2^-100 <= |f| <= 2^100,
    tests/test_pool_division.py) and by the division itself for any window that holds something else: subnormals, values
    whose quotient would be subnormal, huge values, infinities, NaNs, signed zeros -- sprinkled so that some windows stay on the
    fast path and some leave it; every finite result bit for bit, non-finite ones in kind"""
    rng = np.random.default_rng(77)
    b, S, c = 3, 16, 32
    idx = rand_voxels(rng, b, S, 500)
    feat = rng.normal(size=(idx.shape[0], c)).astype(np.float32)
    special = np.array([1e-42, -3e-44, 1e-35, -2.5e-33, 7.0e-31, 1.3e30, -4e35, 3e38, np.inf, -np.inf, np.nan, -0.0, 0.0,
                        np.float32(2.0) ** -100, np.float32(2.0) ** 100, -np.float32(2.0) ** -101], np.float32)
    rows = rng.choice(idx.shape[0], size=idx.shape[0] // 6, replace=False)           # most windows keep ordinary values only
    for r in rows:
        cols = rng.choice(c, size=int(rng.integers(1, 5)), replace=False)
        feat[r, cols] = rng.choice(special, size=cols.size)
    aset = dcl.ops.grid_from_indices(cuda(idx), b, S)
    out, nbr = dcl.spconv.ops.build_rulebook(aset, 3, 2, 1, False)
    r_out, r_pairs, r_num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, 3, 2, 1, 1)
    with np.errstate(all="ignore"):
        want, want_rf = oracle.indice_avgpool(feat, r_pairs, r_num, r_out.shape[0])
    got = dcl.ops.sparse_avgpool(cuda(feat), nbr, out.n).cpu().numpy()
    fin = np.isfinite(want)
    assert fin.any() and (~fin).any() and np.any((np.abs(want) < 1.2e-38) & (want != 0))        # the cases are really there
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))


# ------------------------------------------------------------------------------------------- spconv op-level boundary
def _shuffled_pairs(rng, pairs, num):
    """the reference fills each offset's pair list by atomicAdd (indice.cu.h:57): any order inside an offset is legal"""
    out = pairs.copy()
    for k in range(pairs.shape[0]):
        perm = rng.permutation(int(num[k]))
        out[k, :, :num[k]] = pairs[k][:, perm]
    return out


@pytest.mark.parametrize("cin,cout,subm", [(7, 16, False), (32, 64, True), (64, 64, False), (128, 128, True)])
def test_op_level_indice_conv_takes_reference_pairs(dcl, oracle, cin, cout, subm):
    """spconv.ops.indice_conv / spconv.functional.indice_conv / indice_subm_conv with the argument lists of the reference
    (libs/spconv/spconv/ops.py:102-118, functional.py:22-43,69-91), fed with the ORACLE's rulebook in the reference's pair
    format (pinned to geometry.h): equals oracle.indice_conv; indice_pair_num on the host (as after spconv_ops.h:264) or
    on the device; backward through the Function == the gather-table backward"""
    rng = np.random.default_rng(cin * 3 + cout)
    b, S = 2, 8
    idx = rand_voxels(rng, b, S, 150)
    feat = rng.normal(size=(idx.shape[0], cin)).astype(np.float32)
    W = (rng.normal(size=(3, 3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
    r_out, r_pairs, r_num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, 3, 1, 1, 1, subm=subm)
    n_out = r_out.shape[0]
    want = oracle.indice_conv(feat, W, r_pairs, r_num, n_out, subm=subm)
    tol = 2e-5 * max(1.0, np.abs(want).max())
    pairs = cuda(_shuffled_pairs(rng, r_pairs, r_num))
    sp = dcl.spconv
    for num in (torch.from_numpy(r_num), cuda(r_num)):
        got = sp.ops.indice_conv(cuda(feat), cuda(W), pairs, num, n_out, False, subm).cpu().numpy()
        assert got.shape == want.shape and np.abs(got - want).max() <= tol
    fn = sp.functional.indice_subm_conv if subm else sp.functional.indice_conv
    f, w = cuda(feat).requires_grad_(), cuda(W).requires_grad_()
    out = fn(f, w, pairs, torch.from_numpy(r_num), n_out)
    assert np.abs(out.detach().cpu().numpy() - want).max() <= tol
    g = cuda(rng.normal(size=want.shape).astype(np.float32))
    out.backward(g)
    nbr = dcl.ops.rulebook_from_pairs(pairs, cuda(r_num), idx.shape[0], n_out, check=True)
    dx, dW = dcl.ops.sparse_conv_backward(cuda(feat), cuda(W).reshape(27, cin, cout).contiguous(), g, nbr, n_out, subm)
    assert torch.equal(f.grad, dx) and torch.equal(w.grad, dW.view_as(w))
    # our own exporter and the importer are inverses
    aset = dcl.ops.grid_from_indices(cuda(idx), b, S)
    _, nbr0 = sp.ops.build_rulebook(aset, 3, 1, 1, subm)
    p2, n2 = dcl.ops.rulebook_to_pairs(nbr0, n_out, idx.shape[0])
    assert torch.equal(dcl.ops.rulebook_from_pairs(p2, n2, idx.shape[0], n_out)[:, :n_out], nbr0[:, :n_out])
    with pytest.raises(ValueError):
        bad = r_pairs.copy()
        bad[3, 1, 0] = n_out + 5
        dcl.ops.rulebook_from_pairs(cuda(bad), cuda(r_num), idx.shape[0], n_out, check=True)
    with pytest.raises(NotImplementedError):
        sp.ops.indice_conv(cuda(feat), cuda(W), pairs, cuda(r_num), n_out, True, False)


@pytest.mark.parametrize("c", [32, 7])
def test_op_level_avgpool_and_summaryrf_take_reference_pairs(dcl, oracle, c):
    """spconv.ops.get_indice_summaryrf / indice_avgpool (ops.py:168-179) and spconv.functional.indice_avgpool
    (functional.py:137-166, use_gs False and True) on the oracle's pair-format rulebook: rf and values bit-exact"""
    rng = np.random.default_rng(40 + c)
    b, S = 3, 16
    idx = rand_voxels(rng, b, S, 400)
    feat = rng.normal(size=(idx.shape[0], c)).astype(np.float32)
    r_out, r_pairs, r_num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, 3, 2, 1, 1)
    n_out = r_out.shape[0]
    want, want_rf = oracle.indice_avgpool(feat, r_pairs, r_num, n_out)
    pairs, num = cuda(_shuffled_pairs(rng, r_pairs, r_num)), cuda(r_num)
    sp = dcl.spconv
    rf = sp.ops.get_indice_summaryrf(pairs, num, n_out)
    assert rf.dtype == torch.int32 and np.array_equal(rf.cpu().numpy(), want_rf)
    got = sp.ops.indice_avgpool(cuda(feat), pairs, num, n_out, rf)
    assert np.array_equal(got.cpu().numpy(), want)
    f = cuda(feat).requires_grad_()
    out = sp.functional.indice_avgpool(f, pairs, num, n_out, False)
    assert np.array_equal(out.detach().cpu().numpy(), want)
    if c % 4 == 0:                                             # the backward kernel handles the backbone's widths (32..256)
        g = cuda(rng.normal(size=want.shape).astype(np.float32))
        out.backward(g)
        nbr = dcl.ops.rulebook_from_pairs(pairs, num, idx.shape[0], n_out)
        assert torch.equal(f.grad, dcl.ops.sparse_avgpool_backward(g, nbr, n_out, idx.shape[0], rf))
    # use_gs=True: every term divided by the kernel volume (functional.py:150-152)
    gs = sp.functional.indice_avgpool(cuda(feat), pairs, num, n_out, True).cpu().numpy()
    ref = np.zeros_like(want)
    for k in range(27):
        for j in range(int(r_num[k])):
            i, o = r_pairs[k, 0, j], r_pairs[k, 1, j]
            ref[o] = ref[o] + feat[i] / np.float32(27)
    assert np.array_equal(gs, ref)


# ------------------------------------------------------------------------------------------- pointnet_sp
def _sp_case(rng, b, n, m, dup=True):
    unk = np.concatenate([np.repeat(np.arange(b), n)[:, None], rng.uniform(-0.2, 0.2, (b * n, 3))], 1).astype(np.float32)
    grid = rng.integers(0, 8, (b * m, 3)).astype(np.float32) * 0.05 - 0.2      # lattice => many exact ties
    kn = np.concatenate([np.repeat(np.arange(b), m)[:, None], grid], 1).astype(np.float32)
    return unk, kn


def test_three_nn_sp_bit_exact(dcl, oracle):
    rng = np.random.default_rng(3)
    b, n, m = 4, 300, 90
    unk, kn = _sp_case(rng, b, n, m)
    want_d2, want_idx = oracle.three_nn_sp(unk, kn)
    d2, idx = dcl.ops.three_nn_sp(cuda(unk), cuda(kn))
    assert np.array_equal(idx.cpu().numpy(), want_idx) and np.array_equal(d2.cpu().numpy(), want_d2)
    seg = torch.arange(b + 1, dtype=torch.int32).cuda() * m
    d2s, idxs = dcl.ops.three_nn_sp(cuda(unk), cuda(kn), seg)
    assert np.array_equal(idxs.cpu().numpy(), want_idx) and np.array_equal(d2s.cpu().numpy(), want_d2)


def test_three_nn_sp_edge_cases(dcl, oracle):
    rng = np.random.default_rng(4)
    unk, kn = _sp_case(rng, 3, 70, 5)
    kn = kn[kn[:, 0] != 1]                    # crop 1 has no voxels, crops 0/2 have 5
    kn = np.concatenate([kn[:2], kn[5:]])      # crop 0 has only 2 (< 3 neighbours)
    want_d2, want_idx = oracle.three_nn_sp(unk, kn)
    d2, idx = dcl.ops.three_nn_sp(cuda(unk), cuda(kn))
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    assert np.array_equal(d2.cpu().numpy(), want_d2)           # unmatched slots: idx 0, dist2 +inf
    assert np.isinf(want_d2).any()
    seg = cuda(np.array([0, 2, 2, 7], np.int32))
    d2s, idxs = dcl.ops.three_nn_sp(cuda(unk), cuda(kn), seg)
    assert np.array_equal(idxs.cpu().numpy(), want_idx) and np.array_equal(d2s.cpu().numpy(), want_d2)


def test_three_interpolate_sp_bit_exact(dcl, oracle):
    rng = np.random.default_rng(5)
    b, n, m = 2, 257, 64
    unk, kn = _sp_case(rng, b, n, m)
    d2, idx = oracle.three_nn_sp(unk, kn)
    dist = np.sqrt(d2)
    r = (1.0 / (dist + np.float32(1e-8))).astype(np.float32)
    w = (r / ((r[:, 0] + r[:, 1]) + r[:, 2])[:, None]).astype(np.float32)
    for c in (32, 7):
        feats = rng.normal(size=(b * m, c)).astype(np.float32)
        want = oracle.three_interpolate_sp(feats, idx, w)
        got = dcl.ops.three_interpolate_sp(cuda(feats), cuda(idx), cuda(w)).cpu().numpy()
        assert np.array_equal(got, want)
        fused = dcl.ops.three_interpolate_sp(cuda(feats), cuda(idx), cuda(d2), from_dist2=True).cpu().numpy()
        assert np.array_equal(fused, want)
    big = torch.zeros((b * n, 96), device="cuda")
    feats = rng.normal(size=(b * m, 32)).astype(np.float32)
    dcl.ops.three_interpolate_sp(cuda(feats), cuda(idx), cuda(w), out=big[:, 32:64])
    assert np.array_equal(big[:, 32:64].cpu().numpy(), oracle.three_interpolate_sp(feats, idx, w))
    assert float(big[:, :32].abs().sum()) == 0.0 and float(big[:, 64:].abs().sum()) == 0.0


def test_voxel_centres_match_torch_expression(dcl):
    rng = np.random.default_rng(6)
    idx = np.concatenate([rng.integers(0, 4, (500, 1)), rng.integers(0, 32, (500, 3))], 1).astype(np.int32)
    unit = 0.006
    for scale in (2, 4, 6, 8):
        ve, off = np.float32(unit * scale), np.float32(-0.5 * unit * 64)
        ind = torch.from_numpy(idx).float()
        ind[:, 1:] = ind[:, 1:] * torch.tensor(ve) + torch.tensor(off) + .5 * torch.tensor(ve)   # Modules.py:204-211
        got = dcl.ops.voxel_centres(cuda(idx), float(ve), float(off)).cpu()
        assert torch.equal(got, ind)


# ------------------------------------------------------------------------------------------- pointnet_lib
def _cloud(rng, b, n, dup=0.2):
    x = rng.uniform(-0.15, 0.15, (b, n, 3)).astype(np.float32)
    k = int(n * dup)
    if k:
        x[:, rng.choice(n, k, replace=False)] = x[:, rng.choice(n, k, replace=True)]     # duplicates => exact ties
    return x


@pytest.mark.parametrize("n,m,ns,r", [(1000, 300, 16, 0.05), (4096, 512, 64, 0.03), (333, 77, 5, 0.2), (500, 64, 200, 0.05),
                                       (3001, 50, 64, 0.02)])
def test_ball_query_bit_exact(dcl, oracle, n, m, ns, r):
    rng = np.random.default_rng(n + ns)
    xyz = _cloud(rng, 3, n)
    new = xyz[:, rng.choice(n, m, replace=False)] + rng.normal(0, 1e-3, (3, m, 3)).astype(np.float32)
    new[:, 0] = 9.0                                            # a centre with no neighbour at all -> zeros
    want = oracle.ball_query(r, ns, xyz, new)
    got = dcl.ops.ball_query(r, ns, cuda(xyz), cuda(new)).cpu().numpy()
    assert np.array_equal(got, want)


def test_group_and_gather_bit_exact(dcl, oracle):
    rng = np.random.default_rng(8)
    b, c, n, npo, ns = 3, 20, 700, 50, 16
    feats = rng.normal(size=(b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (b, npo, ns)).astype(np.int32)
    assert np.array_equal(dcl.ops.group_points(cuda(feats), cuda(idx)).cpu().numpy(), oracle.group_points(feats, idx))
    idx3 = rng.integers(0, n, (b, npo, 3)).astype(np.int32)     # nsample not a multiple of 4 -> scalar kernel
    assert np.array_equal(dcl.ops.group_points(cuda(feats), cuda(idx3)).cpu().numpy(), oracle.group_points(feats, idx3))
    gi = rng.integers(0, n, (b, 123)).astype(np.int32)
    assert np.array_equal(dcl.ops.gather_points(cuda(feats), cuda(gi)).cpu().numpy(), oracle.gather_points(feats, gi))


def test_ball_query_and_group_points_at_the_benchmarked_shape(dcl, oracle):
    """BASELINE's primitive shape, the one bench.py's `ball_query+group_points` roofline is quoted on: B = 32 clouds of the
    synthetic crops, N = 12288, npoint = 2048 (FPS centres), r = 0.03, nsample = 64, C = 64.  ball_query is compared bit for
    bit on every cloud; group_points -- whose LDS-staged kernel (k_group_points_lds, rows of 12288 floats) only runs from
    npoint*nsample >= 4096 -- on clouds 0 / 13 / 31 of the full launch (the op is independent per cloud; 1 GiB of output is
    not worth hauling to the host).  Reference semantics: libs/pointnet_lib/src/ball_query_gpu.cu:23-44,
    group_points_gpu.cu:47-66."""
    B, N, NP, NS, C, r = 32, 12288, 2048, 64, 64, 0.03
    data = dcl.synth.make_batch(B, N, 64)
    xyz = data["inp"]["feats"][:, 4:7].reshape(B, N, 3).contiguous()
    fps = dcl.ops.furthest_point_sampling(xyz.cuda(), NP)
    assert np.array_equal(fps[:2].cpu().numpy(), oracle.furthest_point_sample(xyz[:2].numpy(), NP))
    new_xyz = torch.gather(xyz, 1, fps.cpu().long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = dcl.ops.ball_query(r, NS, xyz.cuda(), new_xyz.cuda())
    want_idx = oracle.ball_query(r, NS, xyz.numpy(), new_xyz.numpy())
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    feats = torch.randn(B, C, N, generator=torch.Generator().manual_seed(5))
    out = dcl.ops.group_points(feats.cuda(), idx)
    assert tuple(out.shape) == (B, C, NP, NS)
    for bi in (0, 13, 31):
        want = oracle.group_points(feats[bi:bi + 1].numpy(), want_idx[bi:bi + 1])
        assert np.array_equal(out[bi:bi + 1].cpu().numpy(), want), bi


@pytest.mark.parametrize("n,c,npoint,ns", [
    (12288, 5, 256, 16),       # 1 channel row per workgroup (rows of 6145..12288 floats), ragged channel count
    (20000, 3, 128, 32),       # 1 row, longer than the 48 KiB target (80 KB of LDS)
    (5000, 7, 64, 64),         # 2 rows per workgroup, last chunk holds one
    (4000, 7, 512, 8),         # 3 rows, last chunk holds one
    (3000, 10, 100, 44),       # 4 rows, last chunk holds two; npoint*nsample a multiple of 4 only
    (700, 9, 1024, 4),         # 4 short rows
])
def test_group_points_lds_staged_rows_bit_exact(dcl, oracle, n, c, npoint, ns):
    """every rows-per-workgroup choice of the LDS-staged grouping kernel (csrc/pointnet.hip, dcl_group_points_into: 1..4
    channel rows by row length), with channel counts that leave a partial last chunk, and the fill-a-channel-block form
    `out=` used by QueryAndGroup (the block's batch stride is the full tensor's)"""
    rng = np.random.default_rng(n + c)
    b = 2
    assert npoint * ns >= 4096 and (npoint * ns) % 4 == 0 and n <= 36 * 1024
    feats = rng.normal(size=(b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (b, npoint, ns)).astype(np.int32)
    idx[0, 0, :4] = [0, n - 1, 0, n - 1]
    want = oracle.group_points(feats, idx)
    assert np.array_equal(dcl.ops.group_points(cuda(feats), cuda(idx)).cpu().numpy(), want)
    full = torch.full((b, c + 5, npoint, ns), -7.0, device="cuda")
    dcl.ops.group_points(cuda(feats), cuda(idx), out=full[:, 3:3 + c])
    got = full.cpu().numpy()
    assert np.array_equal(got[:, 3:3 + c], want)
    assert (got[:, :3] == -7.0).all() and (got[:, 3 + c:] == -7.0).all()          # nothing written outside the block


def test_grouping_modules_match_the_composed_oracle(dcl, oracle):
    """QueryAndGroup / GroupAll / KNNAndGroup of the pointnet_lib mirror (forward composition of the primitives)"""
    import importlib
    pu = importlib.import_module("dcl-net_amd.libs.pointnet_lib.pointnet2_utils")
    rng = np.random.default_rng(21)
    B, N, NP, C, ns, r = 2, 900, 64, 12, 16, 0.06
    xyz = _cloud(rng, B, N)
    new = xyz[:, rng.choice(N, NP, replace=False)].copy()
    feats = rng.normal(size=(B, C, N)).astype(np.float32)
    idx = oracle.ball_query(r, ns, xyz, new)
    gx = oracle.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - new.transpose(0, 2, 1)[..., None]
    want = np.concatenate([oracle.group_points(feats, idx), gx], 1)
    got = pu.QueryAndGroup(r, ns)(cuda(xyz), cuda(new), cuda(feats)).cpu().numpy()
    assert np.array_equal(got, want)
    assert np.array_equal(pu.QueryAndGroup(r, ns)(cuda(xyz), cuda(new)).cpu().numpy(), gx)
    ga = pu.GroupAll()(cuda(xyz), None, cuda(feats)).cpu().numpy()
    assert ga.shape == (B, 3 + C, 1, N) and np.array_equal(ga[:, :3, 0], xyz.transpose(0, 2, 1)) and np.array_equal(ga[:, 3:, 0], feats)
    _, kidx = oracle.knn(ns, new, xyz)
    kx = oracle.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), kidx) - new.transpose(0, 2, 1)[..., None]
    wantk = np.concatenate([kx, oracle.group_points(feats, kidx)], 1)
    assert np.array_equal(pu.KNNAndGroup(r, ns)(cuda(xyz), cuda(new), None, cuda(feats)).cpu().numpy(), wantk)


@pytest.mark.parametrize("n,m", [(1024, 128), (12288, 64), (777, 100), (100, 20), (40, 10), (2048, 33), (5000, 50), (16384, 20),
                                 (16385, 9)])
def test_fps_bit_exact_with_ties(dcl, oracle, n, m):
    rng = np.random.default_rng(n)
    xyz = _cloud(rng, 2, n, dup=0.3)
    want = oracle.furthest_point_sample(xyz, m)
    got = dcl.ops.furthest_point_sampling(cuda(xyz), m).cpu().numpy()
    assert np.array_equal(got, want)


def test_fps_lattice_ties(dcl, oracle):
    """integer lattice: massive exact distance ties exercise the reduction-tree tie rule"""
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(0)
    xyz = np.stack([g[rng.permutation(512)], g[rng.permutation(512)]]).astype(np.float32)
    want = oracle.furthest_point_sample(xyz, 64)
    got = dcl.ops.furthest_point_sampling(cuda(xyz), 64).cpu().numpy()
    assert np.array_equal(got, want)


def test_knn_three_nn_three_interpolate_batched(dcl, oracle):
    rng = np.random.default_rng(9)
    unk, kn = _cloud(rng, 2, 150), _cloud(rng, 2, 260)
    for k in (1, 2, 3, 5, 200):                                # k <= 3 runs on the tiled three_nn kernel
        wd, wi = oracle.knn(k, unk, kn)
        d2, idx = dcl.ops.knn(k, cuda(unk), cuda(kn))
        assert np.array_equal(idx.cpu().numpy(), wi) and np.array_equal(d2.cpu().numpy(), wd)
    wd, wi = oracle.knn(3, unk, kn[:, :2].copy())                # fewer known points than k: (inf, 0) fillers
    d2, idx = dcl.ops.knn(3, cuda(unk), cuda(kn[:, :2].copy()))
    assert np.array_equal(idx.cpu().numpy(), wi) and np.array_equal(d2.cpu().numpy(), wd)
    wd, wi = oracle.three_nn(unk, kn)
    d2, idx = dcl.ops.three_nn(cuda(unk), cuda(kn))
    assert np.array_equal(idx.cpu().numpy(), wi) and np.array_equal(d2.cpu().numpy(), wd)
    feats = rng.normal(size=(2, 9, 260)).astype(np.float32)
    w = rng.uniform(size=(2, 150, 3)).astype(np.float32)
    assert np.array_equal(dcl.ops.three_interpolate(cuda(feats), cuda(wi), cuda(w)).cpu().numpy(),
                          oracle.three_interpolate(feats, wi, w))


@pytest.mark.parametrize("case", ["surface", "lattice", "flat", "identical", "outliers", "big"])
def test_batched_three_nn_and_knn1_bucketed_search_is_exact(request, dcl, oracle, case):
    """the pruned search of dcl_three_nn / dcl_knn(k=1) -- known points bucketed along their widest axis, walk outwards
    until the axis gap exceeds the worst kept distance -- returns the scan's (dist2, idx) bit for bit: random clouds, an
    integer lattice (exact ties: the lowest index must win), a flat cloud (axis choice), all points identical (one
    bucket), far outlier queries (clamped start bucket) and BASELINE's primitive shape"""
    rng = np.random.default_rng(len(case))
    B, n, m = 2, 700, 900
    if case == "surface":
        kn = _cloud(rng, B, m); unk = _cloud(rng, B, n)
    elif case == "lattice":
        kn = rng.integers(-6, 7, size=(B, m, 3)).astype(np.float32) * np.float32(0.01)
        unk = rng.integers(-6, 7, size=(B, n, 3)).astype(np.float32) * np.float32(0.01)
    elif case == "flat":
        kn = _cloud(rng, B, m); kn[:, :, 0] *= 1e-4; kn[:, :, 2] *= 1e-3
        unk = _cloud(rng, B, n); unk[:, :, 0] *= 1e-4
    elif case == "identical":
        kn = np.tile(np.float32([[0.01, -0.02, 0.03]]), (B, m, 1)); unk = _cloud(rng, B, n)
    elif case == "outliers":
        kn = _cloud(rng, B, m); unk = _cloud(rng, B, n) * np.float32(40.0)
    else:
        B, n, m = 2, 12288, 2048
        kn = _cloud(rng, B, m); unk = _cloud(rng, B, n)
    wd, wi = oracle.three_nn(unk, kn)
    d2, idx = dcl.ops.three_nn(cuda(unk), cuda(kn))
    assert np.array_equal(idx.cpu().numpy(), wi) and np.array_equal(d2.cpu().numpy(), wd)
    wd1, wi1 = oracle.knn(1, unk, kn)
    d1, i1 = dcl.ops.knn(1, cuda(unk), cuda(kn))
    assert np.array_equal(i1.cpu().numpy(), wi1) and np.array_equal(d1.cpu().numpy(), wd1)
    lib = enter_diag(dcl, request)                              # A/B: the plain scans of the diagnostic library
    lib.dcl_debug_nn_batched_mode(1)
    try:
        d2s, idxs = dcl.ops.three_nn(cuda(unk), cuda(kn))
    finally:
        lib.dcl_debug_nn_batched_mode(0)
    assert torch.equal(idxs, idx) and torch.equal(d2s, d2)


@pytest.mark.parametrize("B,n,m", [(1, 1, 64), (3, 513, 64), (2, 5000, 8192), (40, 300, 333), (32, 3100, 1024), (1, 20000, 4096),
                                   (32, 12288, 2048)])                  # the last: BASELINE's primitive shape, what bench.py times
def test_batched_nn_search_shapes(dcl, oracle, B, n, m):
    """the wave-coherent bucketed search at the edges of its launch plan: one query, a ragged last workgroup, the smallest
    and the largest staged cloud (64 / 8192 known points), every queries-per-thread choice (launches of few and of many
    workgroups) -- (dist2, idx) of three_nn and knn(k = 1) bit for bit against the oracle's scans, duplicates included"""
    rng = np.random.default_rng(B * 1000 + n + m)
    kn, unk = _cloud(rng, B, m, dup=0.1), _cloud(rng, B, n, dup=0.0)
    unk[:, : min(n, 7)] = kn[:, : min(n, 7)]                     # queries ON known points: zero distances and exact ties
    wd, wi = oracle.three_nn(unk, kn)
    d2, idx = dcl.ops.three_nn(cuda(unk), cuda(kn))
    assert np.array_equal(idx.cpu().numpy(), wi) and np.array_equal(d2.cpu().numpy(), wd)
    wd1, wi1 = oracle.knn(1, unk, kn)
    d1, i1 = dcl.ops.knn(1, cuda(unk), cuda(kn))
    assert np.array_equal(i1.cpu().numpy(), wi1) and np.array_equal(d1.cpu().numpy(), wd1)


# ------------------------------------------------------------------------------------------- dense kernels
def _attn_ref(Q, K, V):
    """torch fp64 reference of Aligner (models/Modules.py:166-169) on point-major operands"""
    S = torch.einsum("bjc,bic->bji", K.double(), Q.double())          # (b, nk, nq)
    A = torch.softmax(S, dim=1)
    return torch.einsum("bji,bjc->bic", A, V.double())


@pytest.mark.parametrize("b,nq,nk,scale", [(2, 256, 256, 1.0), (1, 200, 500, 1.0), (3, 64, 96, 6.0), (1, 1000, 132, 0.3)])
def test_cross_attention_matches_fp64(request, dcl, b, nq, nk, scale):
    g = torch.Generator().manual_seed(nq + nk)
    Q = (torch.randn(b, nq, 64, generator=g) * scale).cuda()
    K = torch.randn(b, nk, 64, generator=g).cuda()
    V1 = torch.randn(b, nk, 256, generator=g).cuda()
    V2 = torch.randn(b, nk, 64, generator=g).cuda()
    want = _attn_ref(Q, K, torch.cat([V1, V2], 2))
    lib = enter_diag(dcl, request)      # kernel variants live in the diagnostic library only (tests/_diag)
    for variant in (0, 1, 2, 3, 4):            # auto, shared-tile 8-wave, register-staged 4-wave, LDS-DMA 8 / 4 waves
        O1 = torch.empty(b * nq, 256, device="cuda")
        O2 = torch.empty(b * nq, 64, device="cuda")
        lib.dcl_debug_attention_variant(variant)
        try:
            dcl.ops.cross_attention(b, Q.reshape(-1, 64), K.reshape(-1, 64), V1.reshape(-1, 256), O1,
                                    V2.reshape(-1, 64), O2)
        finally:
            lib.dcl_debug_attention_variant(0)
        got = torch.cat([O1.view(b, nq, 256), O2.view(b, nq, 64)], 2).double()
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), variant


def test_cross_attention_forced_rescale(request, dcl):
    """spike one key against the queries late in the key axis so the running max jumps past the lazy-rescale
    threshold at a chosen tile (guide rule 26)."""
    b, nq, nk = 1, 96, 320
    g = torch.Generator().manual_seed(0)
    Q = torch.randn(b, nq, 64, generator=g)
    K = torch.randn(b, nk, 64, generator=g) * 0.1
    K[0, 200] = Q[0, 5] * 3.0                      # huge score for query 5 in tile 6
    K[0, 300] = Q[0, 40] * 5.0
    V = torch.randn(b, nk, 64, generator=g)
    want = _attn_ref(Q, K, V)[0]
    lib = enter_diag(dcl, request)      # kernel variants live in the diagnostic library only (tests/_diag)
    for variant in (1, 2):
        O = torch.empty(b * nq, 64, device="cuda")
        lib.dcl_debug_attention_variant(variant)
        try:
            dcl.ops.cross_attention(b, Q.cuda().reshape(-1, 64), K.cuda().reshape(-1, 64), V.cuda().reshape(-1, 64), O)
        finally:
            lib.dcl_debug_attention_variant(0)
        assert float((O.cpu().double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), variant


def test_cross_attention_key_split(request, dcl):
    """small launches split the keys over several workgroups and merge (max, sum, partial output) records: forced split
    counts, ragged key / query counts, a score spike inside one split's range"""
    b, nq, nk = 2, 200, 1000 + 13
    g = torch.Generator().manual_seed(5)
    Q = torch.randn(b, nq, 64, generator=g)
    K = torch.randn(b, nk, 64, generator=g) * 0.3
    K[0, 700] = Q[0, 3] * 5.0
    K[1, 1010] = Q[1, 199] * 6.0
    V1, V2 = torch.randn(b, nk, 256, generator=g), torch.randn(b, nk, 64, generator=g)
    want = _attn_ref(Q, K, torch.cat([V1, V2], 2))
    lib = enter_diag(dcl, request)      # kernel variants live in the diagnostic library only (tests/_diag)
    Kd, V1d, V2d = K.cuda().reshape(-1, 64), V1.cuda().reshape(-1, 256), V2.cuda().reshape(-1, 64)
    try:
        for variant in (0, 3):                     # 4-wave kernel (small launch) and the 8-wave one (badly quantised grids)
            lib.dcl_debug_attention_variant(variant)
            for split in (0, 1, 2, 3, 5, 8):
                lib.dcl_debug_attention_split(split)
                O1 = torch.empty(b * nq, 256, device="cuda")
                O2 = torch.empty(b * nq, 64, device="cuda")
                # the wrapper sizes the scratch for the automatic choice; forced splits get the full amount
                scratch = torch.empty(8 * b * nq * 324, device="cuda")
                N = dcl.ops.N
                N.check(lib.dcl_cross_attention_ws(b, nq, nk, N.ptr(Q.cuda()), 64, N.ptr(Kd), 64, N.ptr(V1d), 256, 256,
                                                   N.ptr(O1), 256, N.ptr(V2d), 64, 64, N.ptr(O2), 64, N.ptr(scratch),
                                                   dcl.ops.C.c_int64(scratch.numel()), N.stream()), "attn")
                got = torch.cat([O1.view(b, nq, 256), O2.view(b, nq, 64)], 2).double().cpu()
                assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), (variant, split)
    finally:
        lib.dcl_debug_attention_split(0)
        lib.dcl_debug_attention_variant(0)
    # automatic choice on a badly quantised large launch: 5 query blocks x 64 crops = 320 workgroups -> split
    b2, nq2, nk2 = 64, 1280, 96
    g2 = torch.Generator().manual_seed(6)
    Q2, K2 = torch.randn(b2, nq2, 64, generator=g2), torch.randn(b2, nk2, 64, generator=g2) * 0.3
    V12, V22 = torch.randn(b2, nk2, 256, generator=g2), torch.randn(b2, nk2, 64, generator=g2)
    O1 = torch.empty(b2 * nq2, 256, device="cuda")
    O2 = torch.empty(b2 * nq2, 64, device="cuda")
    dcl.ops.cross_attention(b2, Q2.cuda().reshape(-1, 64), K2.cuda().reshape(-1, 64), V12.cuda().reshape(-1, 256), O1,
                            V22.cuda().reshape(-1, 64), O2)
    want2 = _attn_ref(Q2[:4], K2[:4], torch.cat([V12, V22], 2)[:4])
    got2 = torch.cat([O1.view(b2, nq2, 256), O2.view(b2, nq2, 64)], 2)[:4].double().cpu()
    assert float((got2 - want2).abs().max()) <= 2e-5 * max(1.0, float(want2.abs().max()))


def test_cross_attention_dma_variant_ragged_and_rescale(request, dcl):
    """the LDS-DMA pipeline (variant 3, DCL-Net's 256+64 channel split): key count not a multiple of 32, query count not
    a multiple of 256, and a forced late rescale"""
    b, nq, nk = 2, 300, 1000 + 13
    g = torch.Generator().manual_seed(11)
    Q = torch.randn(b, nq, 64, generator=g)
    K = torch.randn(b, nk, 64, generator=g) * 0.2
    K[0, 900] = Q[0, 7] * 4.0
    K[1, 1012] = Q[1, 299] * 6.0                      # spike in the very last (partial) tile
    V1, V2 = torch.randn(b, nk, 256, generator=g), torch.randn(b, nk, 64, generator=g)
    want = _attn_ref(Q, K, torch.cat([V1, V2], 2))
    lib = enter_diag(dcl, request)      # kernel variants live in the diagnostic library only (tests/_diag)
    O1 = torch.empty(b * nq, 256, device="cuda")
    O2 = torch.empty(b * nq, 64, device="cuda")
    lib.dcl_debug_attention_variant(3)
    try:
        for _ in range(3):                              # repeated launches: races in the tile pipeline would show as flakiness
            dcl.ops.cross_attention(b, Q.cuda().reshape(-1, 64), K.cuda().reshape(-1, 64), V1.cuda().reshape(-1, 256), O1,
                                    V2.cuda().reshape(-1, 64), O2)
            got = torch.cat([O1.view(b, nq, 256), O2.view(b, nq, 64)], 2).cpu().double()
            assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    finally:
        lib.dcl_debug_attention_variant(0)


def test_library_gemms_never_take_a_workspace_exchanging_algorithm(request, dcl):
    """dcl_linear_fwd must not pick a GEMM algorithm that asks for workspace: the library's stream-K kernels hand leftover tiles
    around between workgroups through it, spinning on flags, and two of them side by side on the forward's two streams hang the
    GPU (seen at 33 crops of 1024 points, where M = 33792 does not tile into whole rounds of the chip)"""
    import ctypes
    lib = enter_diag(dcl, request)
    lib.dcl_debug_linear_plan_workspace.restype = ctypes.c_longlong
    torch.zeros(1, device="cuda")
    shapes = ((480, 1024), (256, 256), (256, 64), (512, 512), (512, 1024), (1024, 512), (512, 128))
    rows = set(b * 1024 for b in (1, 6, 25, 32, 33, 35, 37, 40, 41, 47))
    rows |= set(b * n for b in range(1, 41, 3) for n in (12288, 2048))       # the stress shape's two sides
    rows |= set(b * n for b in (1, 2, 3, 7, 25, 33) for n in (333, 517, 500))  # odd point counts (the oracle-graph tests feed them)
    for M in sorted(rows):
        for K, n in shapes:
            # 0: an algorithm without workspace was found and is the one taken; -1 would mean the library has none and
            # dcl_linear_fwd FAILS for the shape (it never falls through to a workspace-exchanging kernel)
            assert lib.dcl_debug_linear_plan_workspace(M, n, K) == 0, (M, K, n)


@pytest.mark.timeout(120)
def test_vendor_gemm_refuses_instead_of_taking_a_workspace_algorithm(dcl):
    """ops.linear_lt never hands the library a workspace (csrc/linear.cpp queries with a maximum of 0 bytes and calls with
    (NULL, 0)): the awkward row counts -- 25 / 33 crops of 1024 points -- run, two side by side on two streams, and finish"""
    g = torch.Generator(device="cuda").manual_seed(2)
    s2 = torch.cuda.Stream()
    for M in (25 * 1024, 33 * 1024, 33 * 333):
        x = torch.randn(M, 512, device="cuda", generator=g)
        Wt = torch.randn(512, 512, device="cuda", generator=g) * 0.05
        bias = torch.randn(512, device="cuda", generator=g)
        y1, y2 = torch.empty(M, 512, device="cuda"), torch.empty(M, 512, device="cuda")
        torch.cuda.synchronize()
        for _ in range(4):
            dcl.ops.linear_lt(x, Wt, bias, True, out=y1)
            with torch.cuda.stream(s2):
                dcl.ops.linear_lt(x, Wt, bias, True, out=y2)
        torch.cuda.synchronize()
        want = torch.relu(x[:64].double() @ Wt.double() + bias.double())
        assert float((y1[:64].double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
        assert torch.equal(y1, y2)


@pytest.mark.timeout(300)
def test_attention_pair_of_whole_rounds_plus_rest_matches_the_lone_launch(dcl):
    """dcl_cross_attention_ws3(concurrent = 2) issues 40 crops of 1024 x 1024 as 32 (one round of 8-wave workgroups with the
    other direction: the split-bf16 kernel) + 8 (4-wave fp32 kernel, keys split); the lone launch runs all 40 on the 4-wave fp32
    kernel.  Unscaled normal Q and K make logits of magnitude ~40, where one ulp of a float32 logit is 4e-6: two float32
    evaluations of S in different summation orders differ by a few of those, i.e. by a few 1e-5 of a softmax weight -- the bound
    between the two paths; both stay within 5e-5 of the float64 definition (Aligner.forward + the extra bmm,
    models/Modules.py:162-169), checked on a crop of the 32 and a crop of the 8, and the split kernel is no further from it than
    twice the fp32 kernel's distance"""
    g = torch.Generator().manual_seed(40)
    b, nq, nk = 40, 1024, 1024
    Q, K = torch.randn((b * nq, 64), generator=g).cuda(), torch.randn((b * nk, 64), generator=g).cuda()
    V1, V2 = torch.randn((b * nk, 256), generator=g).cuda(), torch.randn((b * nk, 64), generator=g).cuda()
    outs = []
    for conc in (1, 2):
        O1, O2 = torch.full((b * nq, 256), float("nan"), device="cuda"), torch.full((b * nq, 64), float("nan"), device="cuda")
        dcl.ops.cross_attention(b, Q, K, V1, O1, V2, O2, concurrent=conc)
        outs.append((O1, O2))
    for a, c in zip(outs[0], outs[1]):
        assert bool(torch.isfinite(c).all()) and float((a - c).abs().max()) <= 5e-5
    for i in (5, 37):
        S = (K[i * nk:(i + 1) * nk].double() @ Q[i * nq:(i + 1) * nq].double().t())
        A = torch.softmax(S, dim=0)
        want = A.t() @ V1[i * nk:(i + 1) * nk].double()
        e_lone = float((outs[0][0][i * nq:(i + 1) * nq].double() - want).abs().max())
        e_pair = float((outs[1][0][i * nq:(i + 1) * nq].double() - want).abs().max())
        assert e_lone <= 5e-5 and e_pair <= 5e-5 and e_pair <= 2.0 * e_lone + 1e-6, (i, e_lone, e_pair)


@pytest.mark.parametrize("b", [1, 6, 20])
def test_voxelisation_riding_on_the_geometry_launch_is_bit_identical(dcl, b):
    """BackboneRunCap.geometry(voxelize=...) -- PG_OP.voxelize_fp carried by the one-launch geometry stage (up to 16 crops), a
    launch of its own otherwise -- against ops.voxelize_fp, and the geometry itself against a run without the rider"""
    d = dcl.synth.make_batch(b, 1024, 64, first=31 + b)["inp"]
    occ = d["occupied_voxels"].int().cuda().contiguous()
    feats, v2p = d["feats"].float().cuda().contiguous(), d["v2p_maps"].int().cuda().contiguous()
    v0 = torch.tensor([occ.shape[0]], dtype=torch.int32, device="cuda")
    cap = torch.zeros((b * 1024, 4), dtype=torch.int32, device="cuda")
    cap[:occ.shape[0]] = occ
    rules = torch.zeros((b * 1024, v2p.shape[1]), dtype=torch.int32, device="cuda")
    rules[:v2p.shape[0]] = v2p
    plain, ride = dcl.ops.BackboneRunCap(cap, v0, b, 64), dcl.ops.BackboneRunCap(cap, v0, b, 64)
    plain.geometry()
    got = ride.geometry(voxelize=(feats, rules, 4))
    want = dcl.ops.voxelize_fp(feats, rules, 4)
    assert torch.equal(got, want) and float(got[:v2p.shape[0]].abs().sum()) > 0
    assert torch.equal(plain.counts_dev, ride.counts_dev) and int(plain.counts_dev[0]) > 0
    n = int(plain.counts_dev[0])
    assert torch.equal(plain.ws[:1 << 16], ride.ws[:1 << 16])      # the level-0 mask (first bytes of the workspace)
    del n


@pytest.mark.parametrize("b", [1, 3, 8])
def test_pose_heads_fed_with_pooling_parts_are_bit_identical(dcl, b):
    """dcl_pose_heads_parts (the pooling's finish folded into the heads' first launch) == dcl_pool_finish + dcl_pose_heads"""
    g = torch.Generator().manual_seed(b)
    n1 = n2 = 256
    F1, F2 = torch.randn((b * n1, 1024), generator=g).cuda(), torch.randn((b * n2, 1024), generator=g).cuda()
    l1, l2 = torch.randn(b * n1, generator=g).cuda(), torch.randn(b * n2, generator=g).cuda()
    aff = tuple(torch.randn(1024, generator=g).cuda() for _ in range(4))
    mk = lambda i, o: ((torch.randn((i, o), generator=g) / i ** 0.5).cuda(), (torch.randn(o, generator=g) * 0.1).cuda())     # noqa: E731
    rot, tr = [mk(1024, 512), mk(512, 128), mk(128, 9)], [mk(1024, 512), mk(512, 128), mk(128, 3)]
    conf_a, pooled = dcl.ops.conf_pool(b, l1, l2, F1, F2, affine=aff)
    want = dcl.ops.pose_heads(pooled, rot, tr, with_rotation=True)
    conf_b, parts = dcl.ops.conf_pool(b, l1, l2, F1, F2, affine=aff, finish=False)
    got = dcl.ops.pose_heads_parts(parts, aff, rot, tr, with_rotation=True)
    assert torch.equal(conf_a, conf_b)
    for a, c in zip(got, want):
        assert torch.equal(a, c)


@pytest.mark.parametrize("M,wide", [(1024, False), (1000, True), (33, False), (32 * 1024, True), (1, False)])
def test_confidence_regressor_in_one_launch_matches_its_three_layers(dcl, M, wide):
    """dcl_mlp128_to1 (the regressor_conf stack, models/DCL_Net.py:115-126, in one launch) against the three layers in float64
    and against the three library GEMMs it replaces; x as a column block of a wider buffer too"""
    g = torch.Generator().manual_seed(M)
    buf = torch.randn((M, 192 if wide else 128), generator=g).cuda()
    x = buf[:, 64:192] if wide else buf
    W1, W2 = (torch.randn((128, 128), generator=g) / 11.3).cuda(), (torch.randn((128, 128), generator=g) / 11.3).cuda()
    W3 = dcl.ops.pad_linear_weight((torch.randn((128, 1), generator=g) / 11.3).cuda())
    b1, b2, b3 = (torch.randn(128, generator=g) * 0.1).cuda(), (torch.randn(128, generator=g) * 0.1).cuda(), torch.randn(1, generator=g).cuda()
    got = dcl.ops.mlp128_to1(x, [(W1, b1), (W2, b2), (W3, b3)])
    ref = (((x.double() @ W1.double() + b1.double()).relu() @ W2.double() + b2.double()).relu() @ W3.double() + b3.double())
    assert got.shape == (M, 1)
    assert float((got.double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    lib3 = dcl.ops.linear_lt(dcl.ops.linear_lt(dcl.ops.linear_lt(x, W1, b1, True), W2, b2, True), W3, b3, False)
    assert float((got - lib3).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("M", [1024, 1000, 33, 32 * 1024 + 37, 393216])
def test_confidence_regressor_as_two_gemms_with_a_row_dot_epilogue(dcl, M):
    """ops.linear (128 -> 128, ReLU) + ops.linear_rowdot (128 -> 128 -> 1: the row dot is the second GEMM's epilogue, csrc/
    linear_dma.hip EPI = 2) -- what large batches run instead of dcl_mlp128_to1 -- against the three layers in float64 and
    against dcl_mlp128_to1; x a column block of a wider buffer; two launches give the same bits"""
    g = torch.Generator().manual_seed(M % 1000)
    buf = torch.randn((M, 192), generator=g).cuda()
    x = buf[:, 64:192]
    W1, W2 = (torch.randn((128, 128), generator=g) / 11.3).cuda(), (torch.randn((128, 128), generator=g) / 11.3).cuda()
    W3 = dcl.ops.pad_linear_weight((torch.randn((128, 1), generator=g) / 11.3).cuda())
    b1, b2, b3 = (torch.randn(128, generator=g) * 0.1).cuda(), (torch.randn(128, generator=g) * 0.1).cuda(), torch.randn(1, generator=g).cuda()
    h1 = dcl.ops.linear(x, W1, b1, True)
    got = dcl.ops.linear_rowdot(h1, W2, b2, W3, b3)
    assert got.shape == (M, 1) and torch.equal(got, dcl.ops.linear_rowdot(h1, W2, b2, W3, b3))
    rows = torch.arange(M) if M <= 40000 else torch.cat([torch.randint(0, M, (2000,), generator=g), torch.tensor([0, M - 1])])
    xr = x[rows.cuda()].double()
    ref = (((xr @ W1.double() + b1.double()).relu() @ W2.double() + b2.double()).relu() @ W3.double() + b3.double())
    assert float((got[rows.cuda()].double() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    one = dcl.ops.mlp128_to1(x, [(W1, b1), (W2, b2), (W3, b3)])
    assert float((got - one).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_conf_pool_matches_torch(dcl):
    g = torch.Generator().manual_seed(1)
    b, n1, n2, c = 3, 300, 170, 1024
    l1, l2 = torch.randn(b, n1, generator=g) * 3, torch.randn(b, n2, generator=g) * 3
    F1, F2 = torch.randn(b, n1, c, generator=g), torch.randn(b, n2, c, generator=g)
    conf, p1, p2, ws = dcl.ops.conf_pool(b, l1.cuda().reshape(-1), l2.cuda().reshape(-1), F1.cuda().reshape(-1, c),
                                         F2.cuda().reshape(-1, c))
    cw = torch.sigmoid(torch.cat([l1, l2], 1).double())
    w = torch.softmax(cw, dim=1)
    assert float((conf.cpu().double() - cw).abs().max()) <= 1e-6
    want = torch.einsum("bjc,bj->bc", torch.cat([F1, F2], 1).double(), w)
    assert float(((p1 + p2).cpu().double() - want).abs().max()) <= 1e-5
    assert float((ws.cpu().double().sum(1) - 1).abs().max()) <= 1e-6
    assert float((ws[:, 0].cpu().double() - w[:, :n1].sum(1)).abs().max()) <= 1e-6
    # one-launch finish behind trailing BatchNorms: sum_i w_i (s x_i + t) over both sides
    s1, t1, s2, t2 = [torch.randn(c, generator=g) for _ in range(4)]
    conf2, pooled = dcl.ops.conf_pool(b, l1.cuda().reshape(-1), l2.cuda().reshape(-1), F1.cuda().reshape(-1, c),
                                      F2.cuda().reshape(-1, c), affine=(s1.cuda(), t1.cuda(), s2.cuda(), t2.cuda()))
    assert torch.equal(conf2, conf)
    want2 = (torch.einsum("bjc,bj->bc", F1.double() * s1.double() + t1.double(), w[:, :n1]) +
             torch.einsum("bjc,bj->bc", F2.double() * s2.double() + t2.double(), w[:, n1:]))
    assert float((pooled.cpu().double() - want2).abs().max()) <= 2e-5


def test_conf_pool_of_a_handful_of_crops_in_one_launch_equals_the_two_launches(dcl):
    """up to 8 crops of at most 2048 logits: dcl_conf_pool is ONE launch (every workgroup forms its crop's softmax itself);
    the same crops inside a larger batch take the two launches: conf and the weight sums bit for bit, the pooled rows to
    rounding (the slice count, hence the order of the partial sums, depends on the batch)"""
    g = torch.Generator().manual_seed(4)
    b, n1, n2, c = 12, 1024, 1024, 1024
    l1, l2 = (torch.randn(b, n1, generator=g) * 3).cuda(), (torch.randn(b, n2, generator=g) * 3).cuda()
    F1, F2 = torch.randn(b, n1, c, generator=g).cuda(), torch.randn(b, n2, c, generator=g).cuda()
    big = dcl.ops.conf_pool(b, l1.reshape(-1), l2.reshape(-1), F1.reshape(-1, c), F2.reshape(-1, c))
    k = 5
    small = dcl.ops.conf_pool(k, l1[:k].reshape(-1), l2[:k].reshape(-1), F1[:k].reshape(-1, c), F2[:k].reshape(-1, c))
    assert torch.equal(small[0], big[0][:k]) and torch.equal(small[3], big[3][:k])
    for a, w in ((small[1], big[1][:k]), (small[2], big[2][:k])):
        assert float((a - w).abs().max()) <= 1e-5 * max(1.0, float(w.abs().max()))


def test_ortho9d_matches_torch_svd(dcl):
    g = torch.Generator().manual_seed(2)
    o9 = torch.randn(64, 9, generator=g)
    o9[0] = torch.tensor([1., 0, 0, 0, 1, 0, 0, 0, -1])        # reflection: det fix path
    o9[1] = torch.tensor([1., 0, 0, 0, 1, 0, 0, 0, 1])         # identity: repeated singular values
    o9[2, 6:] = o9[2, :3] * 0.999 + 1e-3 * o9[2, 3:6]          # nearly rank-2
    from oracle import graph as G
    want = G.ortho9d2matrix(o9[:, :3], o9[:, 3:6], o9[:, 6:])
    got = dcl.ops.ortho9d_to_matrix(o9.cuda()).cpu()
    sel = [i for i in range(64) if i != 2]
    assert float((got[sel] - want[sel]).abs().max()) <= 1e-5
    assert float((torch.linalg.det(got.double()) - 1).abs().max()) <= 1e-5
    assert float((got.double() @ got.double().transpose(1, 2) - torch.eye(3, dtype=torch.float64)).abs().max()) <= 1e-5


def test_add_s_matches_reference_expression(dcl):
    """fused ADD-S vs the harness expression (tools/test_YCBV_stage1.py:186-189) evaluated in fp64"""
    g = torch.Generator().manual_seed(3)
    b, P = 5, 2620
    clouds = torch.randn(7, P, 3, generator=g) * 0.05
    cls = torch.tensor([3, 0, 6, 6, 1], dtype=torch.int32)
    from oracle import graph as G
    Rp = G.ortho9d2matrix(*torch.randn(3, b, 3, generator=g))
    Rg = G.ortho9d2matrix(*torch.randn(3, b, 3, generator=g))
    tp, tg = torch.randn(b, 3, generator=g) * 0.02, torch.randn(b, 3, generator=g) * 0.02
    cur = clouds[cls.long()].double()
    pred = torch.bmm(cur, Rp.double().transpose(1, 2)) + tp.double().unsqueeze(1)
    gt = torch.bmm(cur, Rg.double().transpose(1, 2)) + tg.double().unsqueeze(1)
    want = torch.mean(torch.min(torch.norm(pred.unsqueeze(2) - gt.unsqueeze(1), dim=3), 2)[0], dim=1)
    got = dcl.ops.add_s(clouds.cuda(), Rp.cuda(), tp.cuda(), Rg.cuda(), tg.cuda(), cls.cuda()).cpu().double()
    assert float((got - want).abs().max()) <= 1e-6
    same = dcl.ops.add_s(clouds[:b].cuda(), Rp.cuda(), tp.cuda(), Rp.cuda(), tp.cuda()).cpu()
    assert float(same.abs().max()) == 0.0                      # identical poses: every point matches itself exactly


def test_linemod_add_and_adds_kernel_matches_reference_loop_golden(dcl, golden_dir):
    """fused ADD / ADD-S selection kernel (dcl_add_by_symmetry, dcl_add) vs the distances the reference's own eval-loop body
    computed (tools/test_LM.py:118-124 via tests/golden/make_lm_metric_golden.py), and the resulting 13-object success
    table"""
    import os
    z = np.load(os.path.join(golden_dir, "lm_metric_ref.npz"))
    clouds = cuda(z["clouds"])
    table = dcl.sharding.LmTable(z["diameter"])
    for f in range(int(z["n_frames"][0])):
        g = {k: z["f%d_%s" % (f, k)] for k in ("flags", "idx", "Rp", "tp", "Rg", "tg", "l2", "cd")}
        sym = g["flags"][g["flags"] != -1].astype(np.int32)
        args = [cuda(g[k]) for k in ("Rp", "tp", "Rg", "tg")]
        d = dcl.sharding.add_lm(clouds[cuda(g["idx"]).long()].contiguous(), *args, cuda(sym)).cpu().numpy()
        assert np.abs(d - np.where(sym != 0, g["cd"], g["l2"])).max() <= 1e-6
        via_cls = dcl.ops.add_s(clouds, *args, cls=cuda(g["idx"]), sym_flag=cuda(sym)).cpu().numpy()
        assert np.array_equal(via_cls, d)
        assert np.abs(dcl.ops.add(clouds, *args, cls=cuda(g["idx"])).cpu().numpy() - g["l2"]).max() <= 1e-6
        assert np.abs(dcl.ops.add_s(clouds, *args, cls=cuda(g["idx"])).cpu().numpy() - g["cd"]).max() <= 1e-6
        table.add_batch(g["idx"], d.tolist(), g["flags"])
    assert np.array_equal(table.counts[:, 0], z["num_count"]) and np.array_equal(table.counts[:, 1], z["success_count"])


def test_pose_heads_match_the_module_heads(dcl):
    """dcl_pose_heads (both 1024 -> 512 -> 128 -> 9 | 3 heads in two launches, small batches) vs the registered
    Head_MultiLayerPerceptron modules (reference models/DCL_Net.py:139-151) evaluated in float64"""
    cfg = dcl.synth.default_cfg(64, 64)
    net = dcl.DCL_Net.Network(cfg, mode="test")
    net.load_state_dict(dcl.synth.synth_state_dict(net, 9))
    net = net.cuda().eval()
    f = net._fold()
    g = torch.Generator().manual_seed(1)
    for b in (1, 5, 8):
        x = torch.randn(b, 1024, generator=g).cuda()
        o9, t3 = dcl.ops.pose_heads(x, f["regressor_rot"], f["regressor_trans"])
        with torch.no_grad():
            r64 = net.regressor_rot.double()(x.double().unsqueeze(2)).squeeze(2)
            t64 = net.regressor_trans.double()(x.double().unsqueeze(2)).squeeze(2)
        net.regressor_rot.float(); net.regressor_trans.float()
        assert float((o9.double() - r64).abs().max()) <= 2e-5 * max(1.0, float(r64.abs().max()))
        assert float((t3.double() - t64).abs().max()) <= 2e-6 * max(1.0, float(t64.abs().max()) / 0.02)
        # the rotation formed by the second launch itself == ortho9d2matrix of the o9 it wrote (the stand-alone kernel)
        o9b, t3b, R = dcl.ops.pose_heads(x, f["regressor_rot"], f["regressor_trans"], with_rotation=True)
        assert torch.equal(o9b, o9) and torch.equal(t3b, t3)
        assert torch.equal(R, dcl.ops.ortho9d_to_matrix(o9))


def test_linear_layer_writes_column_blocks_in_place(dcl):
    """ops.linear_lt (dcl_linear_fwd: vendor GEMM + bias / ReLU epilogue behind the C-ABI, what shapes the own core does not take
    fall back to) and ops.linear (the dispatcher: own core where it can) against torch on dense and on strided operands: x a
    column block of a wider buffer, out a column block of another -- the neighbouring columns must stay untouched"""
    g = torch.Generator().manual_seed(11)
    for fn in (dcl.ops.linear_lt, dcl.ops.linear):
        for M, K, n in ((1000, 256, 64), (4096, 480, 1024), (37, 128, 1), (2048, 512, 512), (5, 1024, 9), (300, 259, 512)):
            wide = torch.randn(M, K + 40, generator=g).cuda()
            x = wide[:, 8:8 + K]
            Wt = (torch.randn(K, n, generator=g) * 0.05).cuda()
            bias = torch.randn(n, generator=g).cuda()
            for relu, with_bias in ((True, True), (False, True), (False, False), (True, False)):
                want = x @ Wt + (bias if with_bias else 0.0)
                want = torch.relu(want) if relu else want
                got = fn(x, Wt, bias if with_bias else None, relu)
                tol = 2e-5 * max(1.0, float(want.abs().max()))
                assert float((got - want).abs().max()) <= tol, (M, K, n, relu, with_bias)
                buf = torch.full((M, n + 24), 7.0).cuda()
                fn(x, Wt, bias if with_bias else None, relu, out=buf[:, 16:16 + n])
                assert float((buf[:, 16:16 + n] - want).abs().max()) <= tol
                assert bool((buf[:, :16] == 7.0).all()) and bool((buf[:, 16 + n:] == 7.0).all())
        with pytest.raises(RuntimeError):
            fn(torch.zeros(4, 8), torch.zeros(8, 2))                       # host tensors are refused


def test_linear_group_equals_separate_layers(dcl):
    """dcl_linear_group_fwd (own fp32 MFMA kernel, several independent layers per launch) against float64 products: the
    model's small-batch shapes -- the four disengage second layers of a side (column blocks in, column blocks out), a
    confidence layer beside a fuser layer, a lone output column (zero-padded weight), ragged row counts"""
    g = torch.Generator().manual_seed(5)
    for M in (1024, 1000, 37, 2048):
        H = torch.randn(M, 1024 + 8, generator=g).cuda()[:, 8:]                      # a column block: pitch 1032, 32-byte offset
        fuse, conf_in = torch.full((M, 512), 7.0).cuda(), torch.full((M, 128), 7.0).cuda()
        jobs, want = [], []
        for j, n in enumerate((256, 64, 256, 64)):
            Wt = (torch.randn(256, n, generator=g) * 0.05).cuda()
            bias = torch.randn(n, generator=g).cuda()
            out = (fuse[:, :256], conf_in[:, :64], None, None)[j]
            jobs.append((H[:, 256 * j:256 * (j + 1)], Wt, bias, True, out))
            want.append(torch.relu(H[:, 256 * j:256 * (j + 1)].double() @ Wt.double() + bias.double()))
        got = dcl.ops.linear_group(jobs)
        for a, w in zip(got, want):
            assert float((a.double() - w).abs().max()) <= 2e-5 * max(1.0, float(w.abs().max())), M
        assert bool((fuse[:, 256:] == 7.0).all()) and bool((conf_in[:, 64:] == 7.0).all())   # neighbours untouched
        # mixed depths and a lone column; no bias / no ReLU
        x1, x2 = torch.randn(M, 128, generator=g).cuda(), torch.randn(M, 512, generator=g).cuda()
        W1 = dcl.ops.pad_linear_weight((torch.randn(128, 1, generator=g) * 0.1).cuda())
        W2 = (torch.randn(512, 1024, generator=g) * 0.03).cuda()
        W3 = (torch.randn(480, 96, generator=g) * 0.03).cuda()
        x3 = torch.randn(M, 480, generator=g).cuda()
        b2 = torch.randn(1024, generator=g).cuda()
        o1, o2, o3 = dcl.ops.linear_group([(x1, W1, None, False, None), (x2, W2, b2, True, None), (x3, W3, None, True, None)])
        for a, w in ((o1, x1.double() @ W1.double()), (o2, torch.relu(x2.double() @ W2.double() + b2.double())),
                     (o3, torch.relu(x3.double() @ W3.double()))):
            assert a.shape == w.shape
            assert float((a.double() - w).abs().max()) <= 2e-5 * max(1.0, float(w.abs().max())), M
    with pytest.raises(RuntimeError):
        dcl.ops.linear_group([(torch.zeros(64, 48).cuda(), torch.zeros(48, 64).cuda(), None, False, None)])   # K % 32 != 0


def test_own_gemm_core_matches_float64_on_every_tile_shape(request, dcl):
    """dcl_linear_dma_fwd (csrc/linear_dma.hip: the own fp32 MFMA GEMM core that replaces the vendor library on the forward)
    against float64 products: the model's layer shapes, ragged row counts (rows past M are fetched from the last valid row and
    never stored), column counts that are no multiple of a tile, a lone output column (zero-padded weight), operands and outputs
    that are column blocks of wider buffers (the neighbouring columns must stay untouched) -- for every tile shape"""
    lib = enter_diag(dcl, request)
    g = torch.Generator().manual_seed(21)
    try:
        for tile in (0, 1, 2, 3, 4, 5):
            lib.dcl_debug_linear_tile(tile)
            for M, K, n in ((1000, 256, 64), (4096, 480, 1024), (37, 128, 1), (2048, 512, 512), (333, 512, 96), (517, 256, 256),
                            (129, 32, 260)):
                wide = torch.randn(M, K + 40, generator=g).cuda()
                x = wide[:, 8:8 + K]                                               # pitch K + 40, 32-byte offset
                Wt = dcl.ops.pad_linear_weight((torch.randn(K, n, generator=g) * 0.05).cuda())
                bias = torch.randn(n, generator=g).cuda()
                for relu, with_bias in ((True, True), (False, False)):
                    want = x.double() @ Wt.double() + (bias.double() if with_bias else 0.0)
                    want = torch.relu(want) if relu else want
                    tol = 2e-5 * max(1.0, float(want.abs().max()))
                    got = dcl.ops.linear_dma(x, Wt, bias if with_bias else None, relu)
                    assert float((got.double() - want).abs().max()) <= tol, (tile, M, K, n, relu)
                    buf = torch.full((M, n + 24), 7.0).cuda()
                    dcl.ops.linear_dma(x, Wt, bias if with_bias else None, relu, out=buf[:, 16:16 + n])
                    assert float((buf[:, 16:16 + n].double() - want).abs().max()) <= tol, (tile, M, K, n, relu)
                    assert bool((buf[:, :16] == 7.0).all()) and bool((buf[:, 16 + n:] == 7.0).all())
    finally:
        lib.dcl_debug_linear_tile(0)
    with pytest.raises(RuntimeError):
        dcl.ops.linear_dma(torch.zeros(64, 48).cuda(), torch.zeros(48, 64).cuda())     # K % 32 != 0: stays on dcl_linear_fwd
    assert not dcl.ops.linear_dma_ok(torch.zeros(64, 48).cuda(), torch.zeros(48, 64).cuda())


@pytest.mark.timeout(300)
def test_own_gemm_core_at_the_benchmarked_row_counts(dcl):
    """the stress shape's rows (32 crops x 12288 points) through the three big layer shapes: sampled rows against float64, every
    row finite, and the same bits from two launches (one fmaf chain per element: no split-K, no atomics, no order that depends
    on scheduling)"""
    g = torch.Generator(device="cuda").manual_seed(3)
    M = 32 * 12288
    for K, n in ((480, 1024), (512, 512), (256, 64)):
        x = torch.randn(M, K, device="cuda", generator=g)
        Wt = torch.randn(K, n, device="cuda", generator=g) * 0.05
        bias = torch.randn(n, device="cuda", generator=g)
        y1 = dcl.ops.linear_dma(x, Wt, bias, True)
        y2 = dcl.ops.linear_dma(x, Wt, bias, True)
        assert torch.equal(y1, y2) and bool(torch.isfinite(y1).all())
        rows = torch.cat([torch.randint(0, M, (509,), device="cuda", generator=g), torch.tensor([0, 127, 128, M - 129, M - 1], device="cuda")])
        want = torch.relu(x[rows].double() @ Wt.double() + bias.double())
        assert float((y1[rows].double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
        del x, y1, y2


def test_last_fuser_layer_with_the_pooling_as_its_epilogue(dcl):
    """dcl_linear_pool_fwd: part[t] = sum over the rows j of row tile t of w[j] * relu(x[j] @ Wt + bias) -- against float64, for
    whole tiles, a ragged last tile (rows past M weigh nothing) and a column count that is no multiple of the tile; two
    launches give the same bits (fixed summation order)"""
    g = torch.Generator().manual_seed(8)
    for M, K, n in ((1024, 512, 1024), (4096, 512, 1024), (300, 128, 96), (2048 + 77, 256, 512)):
        x = torch.randn(M, K, generator=g).cuda()
        Wt = (torch.randn(K, n, generator=g) * 0.05).cuda()
        bias = torch.randn(n, generator=g).cuda()
        w = torch.rand(M, generator=g).cuda() / M
        part = dcl.ops.linear_pool(x, Wt, bias, w)
        assert torch.equal(part, dcl.ops.linear_pool(x, Wt, bias, w))
        F = torch.relu(x.double() @ Wt.double() + bias.double()) * w.double()[:, None]
        T = dcl.ops.LINEAR_POOL_TILE
        tiles = (M + T - 1) // T
        want = torch.stack([F[t * T:(t + 1) * T].sum(0) for t in range(tiles)])
        assert part.shape == (tiles, n)
        assert float((part.double() - want).abs().max()) <= 2e-5 * max(1e-3, float(want.abs().max())), (M, K, n)


def test_pad_copy_many_stages_and_hands_over_in_one_launch(dcl):
    """dcl_pad_copy_many (input staging / result hand-over of the whole-forward hipGraph): zero-padded 2-D copies, int64
    narrowing, column blocks of wider buffers, scalar fills -- all in one launch, bit for bit what the torch ops did"""
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(300, 7, generator=g).cuda()
    occ = torch.randint(0, 64, (123, 4), generator=g, dtype=torch.int64).cuda()
    v2p = torch.randint(0, 300, (123, 5), generator=g, dtype=torch.int32).cuda()
    wide = torch.randn(40, 512, generator=g).cuda()
    d_feats = torch.full((300, 7), 9.0).cuda()
    d_occ = torch.full((300, 4), 9, dtype=torch.int32).cuda()
    d_v2p = torch.full((300, 33), 9, dtype=torch.int32).cuda()
    d_v0 = torch.full((1,), 9, dtype=torch.int32).cuda()
    d_cols = torch.full((2, 20, 256), 9.0).cuda()
    d_r = torch.full((4, 3, 3), 9.0).cuda()
    src_r = torch.randn(4, 9, generator=g).cuda()
    dcl.ops.pad_copy_many([(d_feats, feats), (d_occ, occ), (d_v2p, v2p), (d_v0, 123), (d_cols, wide[:, 256:]), (d_r, src_r)])
    assert torch.equal(d_feats, feats)
    assert torch.equal(d_occ[:123], occ.int()) and int(d_occ[123:].abs().sum()) == 0
    assert torch.equal(d_v2p[:123, :5], v2p) and int(d_v2p[:123, 5:].abs().sum()) == 0 and int(d_v2p[123:].abs().sum()) == 0
    assert int(d_v0[0]) == 123
    assert torch.equal(d_cols.reshape(40, 256), wide[:, 256:]) and torch.equal(d_r.reshape(4, 9), src_r)
    with pytest.raises(RuntimeError):
        dcl.ops.pad_copy_many([(d_feats, feats)] * 13)                     # more jobs than one launch takes


# ------------------------------------------------------------------------------------------- native backbone runner
def _edge_voxels(rng, S=64):
    """crop 0: random blob touching the 0 and S-1 faces; crop 1: EMPTY; crop 2: one voxel in a corner; crop 3: dense 6^3 block
    plus scattered voxels; crop 4: a 1-voxel-thick plane"""
    rows = []
    blob = rng.integers(0, S, (400, 3))
    blob[:40, 0] = 0
    blob[40:80, 1] = S - 1
    blob[80:120, 2] = rng.choice([0, S - 1], 40)
    rows.append((0, np.unique(blob, axis=0)))
    rows.append((2, np.array([[S - 1, 0, S - 1]])))
    g = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), -1).reshape(-1, 3) + 20
    rows.append((3, np.unique(np.concatenate([g, rng.integers(0, S, (150, 3))]), axis=0)))
    pl = np.stack(np.meshgrid(np.arange(10, 40), np.arange(5, 45), indexing="ij"), -1).reshape(-1, 2)
    rows.append((4, np.concatenate([pl[:, :1], np.full((len(pl), 1), 31), pl[:, 1:]], 1)))
    occ = np.concatenate([np.concatenate([np.full((len(v), 1), b), v], 1) for b, v in rows]).astype(np.int32)
    out = []
    for b, v in rows:                                    # unsorted inside a crop, crops in order (like voxelize_idx output)
        sel = occ[occ[:, 0] == b]
        out.append(sel[rng.permutation(len(sel))])
    return np.concatenate(out), 5


def test_backbone_runner_edge_cases(dcl, oracle):
    """native runner (geometry + features + point read-out, implicit rulebooks, split-K) vs the oracle backbone on awkward
    active sets: an empty crop inside the batch, border voxels, a single-voxel crop, a dense block, a thin plane"""
    from oracle import graph as G
    rng = np.random.default_rng(17)
    occ, b = _edge_voxels(rng)
    S = 64
    cfg = dcl.synth.default_cfg(64, 64)
    net = dcl.DCL_Net.Network(cfg, mode="test")
    sd = dcl.synth.synth_state_dict(net, 5)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    f = net._fold()
    vox = rng.normal(size=(occ.shape[0], 7)).astype(np.float32)
    want = G.backbone(sd, "backbone_inp", vox, occ, [S] * 3, b)
    run = dcl.ops.BackboneRun(cuda(occ), b, S)
    run.set_counts(run.counts_dev.cpu().tolist())
    levels = run.features(cuda(vox), *f["backbone_inp_ptrs"])
    for m, (wf, wi) in enumerate(want):
        assert np.array_equal(run.level_indices(m).cpu().numpy(), wi), m
        got = levels[m].cpu().numpy()
        assert got.shape == wf.shape
        assert np.abs(got - wf).max() <= 5e-5 * max(1.0, np.abs(wf).max()), m
    # point read-out: points of every crop, including the crop that has no voxel at all (its rows stay unmatched: idx 0)
    n_per = 50
    pts = rng.uniform(-0.19, 0.19, (b * n_per, 3)).astype(np.float32)
    pb4 = np.concatenate([np.repeat(np.arange(b, dtype=np.float32), n_per)[:, None], pts], 1)
    unit = 0.006
    off = float(np.float32(-0.5 * unit * 64))
    extents = [float(np.float32(unit * sc)) for sc in (2, 4, 6, 8)]
    got_pf = run.point_features(cuda(pb4), extents, off).cpu().numpy()
    want_pf = G.point_feats(torch.from_numpy(pb4), want, [unit] * 3, [64] * 3).numpy()
    live = pb4[:, 0] != 1                                             # crop 1 is empty: the reference interpolates garbage there
    assert np.abs(got_pf[live] - want_pf[live]).max() <= 5e-5 * max(1.0, np.abs(want_pf[live]).max())


def _random_voxels(rng, S=64):
    """a batch of 1..6 crops of random shape families: gaussian blobs, shells, rods along an axis, sparse noise, near-full
    small cubes -- some crops hugging a face of the grid, row order shuffled inside each crop"""
    b = int(rng.integers(1, 7))
    out = []
    for c in range(b):
        kind = int(rng.integers(0, 5))
        if kind == 0:
            v = rng.normal(S / 2 + rng.integers(-20, 21, 3), rng.uniform(2, 9), (int(rng.integers(50, 900)), 3))
        elif kind == 1:
            d = rng.normal(size=(int(rng.integers(200, 1200)), 3))
            v = S / 2 + d / np.linalg.norm(d, axis=1, keepdims=True) * rng.uniform(5, 28)
        elif kind == 2:
            v = np.zeros((int(rng.integers(20, 200)), 3))
            ax = int(rng.integers(0, 3))
            v[:, ax] = rng.uniform(0, S, len(v))
            v[:, (ax + 1) % 3] = rng.integers(0, S) + rng.normal(0, 0.7, len(v))
            v[:, (ax + 2) % 3] = rng.integers(0, S) + rng.normal(0, 0.7, len(v))
        elif kind == 3:
            v = rng.uniform(0, S, (int(rng.integers(1, 300)), 3))
        else:
            e = int(rng.integers(3, 12))
            o = rng.integers(0, S - e, 3)
            v = np.stack(np.meshgrid(*[np.arange(e)] * 3, indexing="ij"), -1).reshape(-1, 3) + o
            v = v[rng.random(len(v)) < 0.9]
        v = np.unique(np.clip(np.floor(v), 0, S - 1).astype(np.int32), axis=0)
        v = v[rng.permutation(len(v))]
        out.append(np.concatenate([np.full((len(v), 1), c, np.int32), v], 1))
    return np.concatenate(out).astype(np.int32), b


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [101, 102, 103, 104, 105, 106])
def test_backbone_runner_on_random_active_sets(dcl, oracle, seed):
    """seeded fuzz of the whole sparse half: random crop counts and shape families through geometry + 8 convs + 4 pools of
    the native runner against the oracle backbone -- every level's voxel rows bit-exact, features within the GEMM tolerance"""
    from oracle import graph as G
    rng = np.random.default_rng(seed)
    occ, b = _random_voxels(rng)
    net = dcl.DCL_Net.Network(dcl.synth.default_cfg(64, 64), mode="test")
    sd = dcl.synth.synth_state_dict(net, seed)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    vox = rng.normal(size=(occ.shape[0], 7)).astype(np.float32)
    want = G.backbone(sd, "backbone_tmp", vox, occ, [64] * 3, b)
    run = dcl.ops.BackboneRun(cuda(occ), b, 64)
    run.set_counts(run.counts_dev.cpu().tolist())
    levels = run.features(cuda(vox), *net._fold()["backbone_tmp_ptrs"])
    for m, (wf, wi) in enumerate(want):
        assert np.array_equal(run.level_indices(m).cpu().numpy(), wi), (seed, m)
        got = levels[m].cpu().numpy()
        assert got.shape == wf.shape
        assert np.abs(got - wf).max() <= 5e-5 * max(1.0, np.abs(wf).max()), (seed, m)


@pytest.mark.gpu
@pytest.mark.parametrize("b,per", [(2, 300), (16, 900), (5, 40)])
def test_backbone_feature_stage_of_both_sides_in_one_launch_sequence(dcl, oracle, b, per):
    """dcl_backbone_features_pair: the observed and the template backbone (different active sets, different weights) with
    every layer as ONE launch over both sides' tiles -- each side's levels equal its own single-side pass within the GEMM
    tolerance and the oracle backbone; exact and capacity mode; a side that runs out of voxels at a deep level drops out"""
    from oracle import graph as G
    rng = np.random.default_rng(b * 1000 + per)
    net = dcl.DCL_Net.Network(dcl.synth.default_cfg(64, 64), mode="test")
    sd = dcl.synth.synth_state_dict(net, 7)
    net.load_state_dict(sd)
    net = net.cuda().eval()
    f = net._fold()
    occ = {"inp": rand_voxels(rng, b, 64, per), "tmp": rand_voxels(rng, b, 64, max(3, per // 3))}
    vox = {k: rng.normal(size=(v.shape[0], 7)).astype(np.float32) for k, v in occ.items()}
    runs, single = {}, {}
    for side in ("inp", "tmp"):
        runs[side] = dcl.ops.BackboneRun(cuda(occ[side]), b, 64)
        runs[side].set_counts(runs[side].counts_dev.cpu().tolist())
        solo = dcl.ops.BackboneRun(cuda(occ[side]), b, 64)
        solo.set_counts(solo.counts_dev.cpu().tolist())
        single[side] = [t.clone() for t in solo.features(cuda(vox[side]), *f["backbone_%s_ptrs" % side])]
    la, lb = dcl.ops.backbone_features_pair(runs["inp"], cuda(vox["inp"]), f["backbone_inp_ptrs"],
                                            runs["tmp"], cuda(vox["tmp"]), f["backbone_tmp_ptrs"])
    for side, levels in (("inp", la), ("tmp", lb)):
        want = G.backbone(sd, "backbone_" + side, vox[side], occ[side], [64] * 3, b)
        for m in range(4):
            tol = 5e-5 * max(1.0, float(single[side][m].abs().max())) if single[side][m].numel() else 0.0
            assert levels[m].shape == single[side][m].shape
            assert float((levels[m] - single[side][m]).abs().max()) <= tol if levels[m].numel() else True, (side, m)
            assert np.abs(levels[m].cpu().numpy() - want[m][0]).max() <= 5e-5 * max(1.0, np.abs(want[m][0]).max()), (side, m)
    # capacity mode (what a captured forward runs): same values in the live rows
    caps = {}
    for side in ("inp", "tmp"):
        V0 = occ[side].shape[0]
        pad = torch.zeros((V0 + 17, 4), dtype=torch.int32, device="cuda")
        pad[:V0] = cuda(occ[side])
        caps[side] = dcl.ops.BackboneRunCap(pad, torch.tensor([V0], dtype=torch.int32, device="cuda"), b, 64)
        caps[side].geometry()
        caps[side].vox = torch.zeros((V0 + 17, 7), device="cuda")
        caps[side].vox[:V0] = cuda(vox[side])
    ca, cb = dcl.ops.backbone_features_pair(caps["inp"], caps["inp"].vox, f["backbone_inp_ptrs"],
                                            caps["tmp"], caps["tmp"].vox, f["backbone_tmp_ptrs"])
    for side, levels in (("inp", ca), ("tmp", cb)):
        for m in range(4):
            n = single[side][m].shape[0]
            tol = 5e-5 * max(1.0, float(single[side][m].abs().max())) if n else 0.0
            if n:
                assert float((levels[m][:n] - single[side][m]).abs().max()) <= tol, (side, m)


@pytest.mark.gpu
def test_geometry_one_launch_mask_chain_equals_chained_launches(request, dcl):
    """the 8 active sets of a pass from the one-workgroup-per-crop LDS chain (default on 64^3 grids) and from the 8 chained
    mask launches: same counts, same voxel rows at every level -- on the awkward sets and on a 32-crop batch"""
    lib = enter_diag(dcl, request)      # kernel variants live in the diagnostic library only (tests/_diag)
    rng = np.random.default_rng(23)
    occ_e, b_e = _edge_voxels(rng)
    data = dcl.synth.make_batch(32, 1024, 64, first=3)
    cases = [(cuda(occ_e), b_e), (data["inp"]["occupied_voxels"].int().cuda().contiguous(), 32)]
    for occ, b in cases:
        got = {}
        try:
            for mode in (0, 1):
                lib.dcl_debug_geometry_chain(mode)
                run = dcl.ops.BackboneRun(occ, b, 64)
                counts = run.counts_dev.cpu().tolist()
                run.set_counts(counts)
                got[mode] = (counts, [run.level_indices(m).cpu().numpy() for m in range(4)])
        finally:
            lib.dcl_debug_geometry_chain(1)
        assert got[0][0] == got[1][0]
        assert all(c >= 0 for c in got[1][0]) and got[1][0][0] > 0
        for a, c in zip(got[0][1], got[1][1]):
            assert np.array_equal(a, c)


@pytest.mark.gpu
def test_geometry_stage_in_one_launch_equals_the_separate_launches(request, dcl):
    """a pass of at most 16 crops runs its whole geometry stage as ONE launch (k_geometry_small: mark, mask chain, counts
    exchanged between the crops' workgroups, prefixes, rows, permutation); against the separate launches: same counts, same
    voxel rows, and -- through masks, prefixes, conv-set rows and the level-0 permutation -- bit-identical features"""
    lib = enter_diag(dcl, request)
    rng = np.random.default_rng(5)
    occ_e, b_e = _edge_voxels(rng)
    cases = [(cuda(occ_e), b_e)]
    for b in (1, 2, 8, 16):
        d = dcl.synth.make_batch(b, 1024, 64, first=7 + b)
        occ = d["inp"]["occupied_voxels"].int()
        perm = torch.cat([torch.randperm(int((occ[:, 0] == i).sum())) + int((occ[:, 0] < i).sum()) for i in range(b)])
        cases.append((occ[perm].cuda().contiguous(), b))                     # unsorted inside a crop
    net = dcl.DCL_Net.Network(dcl.synth.default_cfg(64, 64), mode="test")
    net.load_state_dict(dcl.synth.synth_state_dict(net, 5))
    f = net.cuda().eval()._fold()
    for occ, b in cases:
        vox = torch.randn(occ.shape[0], 7, generator=torch.Generator().manual_seed(b)).cuda()
        got = {}
        try:
            for mode in (2, 1):                                              # 2 = separate launches, 1 = default
                lib.dcl_debug_geometry_chain(mode)
                run = dcl.ops.BackboneRun(occ, b, 64)
                counts = run.counts_dev.cpu().tolist()
                run.set_counts(counts)
                levels = [t.clone() for t in run.features(vox, *f["backbone_inp_ptrs"])]
                got[mode] = (counts, [run.level_indices(m).cpu() for m in range(4)], levels)
        finally:
            lib.dcl_debug_geometry_chain(1)
        assert got[1][0] == got[2][0] and got[1][0][0] > 0, b
        for a, c in zip(got[1][1], got[2][1]):
            assert torch.equal(a, c), b
        for a, c in zip(got[1][2], got[2][2]):
            assert torch.equal(a, c), b


@pytest.mark.gpu
@pytest.mark.parametrize("scales,unit", [((2, 4, 8, 16), 0.006), ((2, 4, 6, 8), 0.006), ((2, 4, 6, 8), 0.005)])
def test_point_neighbours_grid_search_is_exact(request, dcl, scales, unit):
    """the grid-pruned 3-NN of the point read-out returns bit-for-bit what the per-crop scan returns -- distances, rows and
    tie order -- for surface points, lattice points (8-way ties), isolated and out-of-grid queries (scan fallback), an
    empty crop and invalid crop ids; so does the split read-out API.  Extents: the true strides (2,4,8,16) and the
    PRODUCTION quirk (2,4,6,8) of models/DCL_Net.py:54, whose level-3/4 centre lattices (pitch 6u / 8u) do not coincide
    with the 8^3 / 4^3 occupancy cells (pitch 8u / 16u) -- the case a wrong bound / certification would hide in."""
    rng = np.random.default_rng(23)
    occ, b = _edge_voxels(rng)
    S = 64
    off = float(np.float32(-0.5 * unit * 64))
    extents = [float(np.float32(unit * sc)) for sc in scales]
    run = dcl.ops.BackboneRun(cuda(occ), b, S)
    run.set_counts(run.counts_dev.cpu().tolist())
    q = []
    for bi in range(b):
        own = occ[occ[:, 0] == bi][:, 1:4].astype(np.float32)
        if len(own):
            pick = own[rng.integers(0, len(own), 300)]
            q.append(np.c_[np.full(300, bi), (pick + rng.uniform(0, 1, pick.shape)) * unit + off])      # points inside occupied voxels
            q.append(np.c_[np.full(100, bi), (pick[:100] * unit + off).astype(np.float32)])             # voxel corners: lattice ties
            q.append(np.c_[np.full(100, bi), ((pick[:100] // 2 * 2 + 1) * unit + off)])                 # level-0 cell centres / corners
            for lvl, cell in ((2, 8), (3, 16)):                                                         # coarse levels: centres and
                ci = pick[:60] // cell                                                                  # mid-points of the centre lattice
                q.append(np.c_[np.full(60, bi), (ci * scales[lvl] + 0.5 * scales[lvl]) * unit + off])   # (pitch = scales[lvl] units)
                q.append(np.c_[np.full(60, bi), ((ci + 1) * scales[lvl]) * unit + off])
        q.append(np.c_[np.full(60, bi), rng.uniform(-0.192, 0.192, (60, 3))])                           # anywhere in the grid
        q.append(np.c_[np.full(20, bi), rng.uniform(-0.5, 0.5, (20, 3))])                               # partly outside the grid
    q.append(np.c_[np.array([-1.0, b, 0.5, np.nan]), np.zeros((4, 3))])                                 # crop ids that match nothing
    pb4 = cuda(np.concatenate(q).astype(np.float32))
    lib = enter_diag(dcl, request)      # kernel variants live in the diagnostic library only (tests/_diag)
    res = {}
    try:
        for mode in (0, 1, 2, 3, 4, 5):              # scan / automatic / forced fallback / 1, 4, 8 lanes per query
            lib.dcl_debug_three_nn_grid(mode)
            d, i = run.point_neighbours(pb4, extents, off)
            res[mode] = (d.cpu().numpy(), i.cpu().numpy())
    finally:
        lib.dcl_debug_three_nn_grid(1)
    for mode in (1, 2, 3, 4, 5):
        assert np.array_equal(res[mode][1], res[0][1]), mode
        assert np.array_equal(res[mode][0].view(np.uint32), res[0][0].view(np.uint32)), mode
    # the searched levels really had work to do, and the fallback was exercised (isolated queries exist)
    assert np.isfinite(res[0][0][0]).all(axis=1).sum() > 1000
    # the two halves compose to point_features
    vox = cuda(rng.normal(size=(occ.shape[0], 7)).astype(np.float32))
    cfg = dcl.synth.default_cfg(64, 64)
    net = dcl.DCL_Net.Network(cfg, mode="test")
    net.load_state_dict(dcl.synth.synth_state_dict(net, 5))
    f = net.cuda().eval()._fold()
    run.features(vox, *f["backbone_inp_ptrs"])
    whole = run.point_features(pb4, extents, off)
    halves = run.point_interpolate(*run.point_neighbours(pb4, extents, off))
    assert torch.equal(torch.nan_to_num(whole), torch.nan_to_num(halves))     # NaN rows: queries without any voxel


def test_split_bf16_gemm_core_matches_float64_like_the_fp32_core(dcl):
    """dcl_linear_split_fwd (csrc/linear_split.hip: three bf16 pieces per fp32 operand, six piece products, fp32 accumulators)
    against float64 products, beside the fp32-MFMA core on the same operands: errors of the same size (the bound asserted is the
    fp32 core's tolerance), ragged row counts, column counts that are no multiple of a tile, a lone output column, operands and
    outputs that are column blocks of wider buffers (neighbours untouched), values spread over many binades, exact zeros"""
    g = torch.Generator().manual_seed(33)
    worst = 0.0
    for M, K, n in ((1000, 256, 64), (4096, 480, 1024), (37, 128, 1), (2048, 512, 512), (333, 512, 96), (517, 256, 256), (129, 16, 260),
                    (70000, 48, 130)):
        wide = torch.randn(M, K + 40, generator=g) * torch.exp2(torch.randint(-6, 7, (M, 1), generator=g).float())
        wide[::7, ::5] = 0.0
        wide = wide.cuda()
        x = wide[:, 8:8 + K]
        Wt = (torch.randn(K, n, generator=g) * 0.05).cuda()
        bias = torch.randn(n, generator=g).cuda()
        sw = dcl.ops.SplitWeight(Wt)
        for relu, with_bias in ((True, True), (False, False)):
            want = x.double() @ Wt.double() + (bias.double() if with_bias else 0.0)
            want = torch.relu(want) if relu else want
            tol = 2e-5 * max(1.0, float(want.abs().max()))
            got = dcl.ops.linear_split(x, sw, bias if with_bias else None, relu)
            err = float((got.double() - want).abs().max())
            assert err <= tol, (M, K, n, relu, err, tol)
            if K % 32 == 0:
                ref = dcl.ops.linear_dma(x, dcl.ops.pad_linear_weight(Wt), bias if with_bias else None, relu)
                err32 = float((ref.double() - want).abs().max())
                worst = max(worst, err / max(err32, 1e-30))
            buf = torch.full((M, n + 24), 7.0).cuda()
            dcl.ops.linear_split(x, sw, bias if with_bias else None, relu, out=buf[:, 16:16 + n])
            assert torch.equal(buf[:, 16:16 + n], got)
            assert bool((buf[:, :16] == 7.0).all()) and bool((buf[:, 16 + n:] == 7.0).all())
    assert worst <= 4.0, "split-bf16 errors against float64 should be the size of the fp32 core's: worst ratio %.2f" % worst
    with pytest.raises(AssertionError):
        dcl.ops.SplitWeight(torch.zeros(40, 64).cuda())                                # K % 16 != 0


def test_split_bf16_pieces_sum_to_the_operand_exactly(dcl):
    """the three bf16 pieces dcl_linear_split_weight writes add up to the fp32 weight EXACTLY (h + m + l in float64 == w), for
    normal values over the whole exponent range the network sees and for zeros"""
    g = torch.Generator().manual_seed(5)
    K, n = 64, 200
    Wt = torch.randn(K, n, generator=g) * torch.exp2(torch.randint(-20, 21, (K, n), generator=g).float())
    Wt[::3, ::4] = 0.0
    sw = dcl.ops.SplitWeight(Wt.cuda())
    planes = sw.planes.cpu().view(torch.bfloat16).view(K // 16, 2, 3, 128, 2, 8)       # [chunk][column tile][piece][column][half slot][k]
    total = planes.double().sum(2)                                                       # h + m + l
    for kc in range(K // 16):
        for c in range(n):
            for hs in range(2):
                hh = hs ^ ((c % 128 >> 3) & 1)
                want = Wt[kc * 16 + 8 * hh: kc * 16 + 8 * hh + 8, c].double()
                assert torch.equal(total[kc, c // 128, c % 128, hs], want), (kc, c, hs)
    assert float(planes[:, 1, :, 200 - 128:].float().abs().max()) == 0.0                # columns past N are zero


def test_cross_attention_split_bf16_form_matches_float64_like_the_fp32_form(request, dcl):
    """k_cross_attn_split (csrc/dense.hip: the 8-wave attention with P.V as six bf16 piece products per fp32 product, V's pieces
    written by k_attn_split_v) beside k_cross_attn_dma<8> on the same operands, both against float64: ragged query / key counts
    (keys past nk are zero pieces and -inf scores), large logits, a score spike that forces the lazy rescale late in the key axis,
    a forced key split (partial records + combine), V2 aliasing K as in the forward; errors of the same size"""
    lib = enter_diag(dcl, request)
    lib.dcl_debug_attention_variant(3)                                   # the 8-wave form whatever the size
    worst = 0.0
    try:
        for b, nq, nk, scale, split in ((2, 256, 256, 1.0, 0), (1, 200, 500, 1.0, 0), (3, 64, 96, 6.0, 0), (1, 1000, 132, 0.3, 0),
                                        (2, 300, 1029, 1.0, 0), (2, 300, 1029, 1.0, 4), (1, 96, 320, 1.0, 0)):
            g = torch.Generator().manual_seed(nq + nk + split)
            Q = torch.randn(b, nq, 64, generator=g) * scale
            K = torch.randn(b, nk, 64, generator=g)
            if nk == 320:                                                # spikes in tiles 6 and 9
                K = K * 0.1
                K[0, 200] = Q[0, 5] * 3.0
                K[0, 300] = Q[0, 40] * 5.0
            V1 = torch.randn(b, nk, 256, generator=g) * torch.exp2(torch.randint(-4, 5, (b, nk, 1), generator=g).float())
            Q, K, V1 = Q.cuda(), K.cuda(), V1.cuda()
            want = _attn_ref(Q, K, torch.cat([V1, K], 2))
            tol = 2e-5 * max(1.0, float(want.abs().max()))
            errs = {}
            for bf16 in (1, 0):
                lib.dcl_debug_attention_bf16(bf16)
                lib.dcl_debug_attention_split(split)
                O1 = torch.full((b * nq, 256 + 8), 7.0, device="cuda")
                O2 = torch.empty(b * nq, 64, device="cuda")
                dcl.ops.cross_attention(b, Q.reshape(-1, 64), K.reshape(-1, 64), V1.reshape(-1, 256), O1[:, :256], K.reshape(-1, 64), O2)
                got = torch.cat([O1[:, :256].reshape(b, nq, 256), O2.view(b, nq, 64)], 2).double()
                errs[bf16] = float((got - want).abs().max())
                assert errs[bf16] <= tol, (b, nq, nk, split, bf16, errs[bf16], tol)
                assert bool((O1[:, 256:] == 7.0).all())
            worst = max(worst, errs[1] / max(errs[0], 1e-30))
    finally:
        lib.dcl_debug_attention_variant(0)
        lib.dcl_debug_attention_bf16(1)
        lib.dcl_debug_attention_split(0)
    assert worst <= 4.0, "split-bf16 attention errors against float64 should be the size of the fp32 form's: worst ratio %.2f" % worst


def test_v_stack_written_as_attention_pieces_equals_the_piece_pass(request, dcl):
    """dcl_linear_split_vpieces_fwd (csrc/linear_split.hip, EPI = 3): the GEMM that computes the attention's V1 writes it straight
    into the attention's scratch as bf16 pieces -- the same bits the piece pass (k_attn_split_v) makes from the fp32 output of the
    same GEMM, so the attention fed with V1 = None gives the same bits as the attention fed with that fp32 V1; and a call that does
    not take the split kernel refuses V1 = None"""
    lib = enter_diag(dcl, request)
    lib.dcl_debug_attention_variant(3)                                   # the 8-wave split form whatever the size
    try:
        g = torch.Generator().manual_seed(77)
        b, nq, nk, Kd = 3, 200, 512, 256
        H = torch.randn(b * nk, Kd, generator=g).cuda()
        Wt = (torch.randn(Kd, 256, generator=g) * 0.08).cuda()
        bias = torch.randn(256, generator=g).cuda()
        sw = dcl.ops.SplitWeight(Wt)
        Q = torch.randn(b * nq, 64, generator=g).cuda()
        Km = (torch.randn(b * nk, 64, generator=g) * 0.5).cuda()
        V1 = dcl.ops.linear_split(H, sw, bias, True)                     # the fp32 form of the same layer
        planes_a, whole = dcl.ops.attention_planes(b, nq, nk, 1)
        assert planes_a is not None and whole
        planes_b = torch.zeros_like(planes_a)
        O1a, O2a = torch.empty(b * nq, 256, device="cuda"), torch.empty(b * nq, 64, device="cuda")
        dcl.ops.cross_attention(b, Q, Km, V1, O1a, Km, O2a, planes=planes_a)        # piece pass over all 320 channels
        dcl.ops.linear_split_vpieces(H, sw, bias, planes_b, nk, relu=True)          # V1's pieces from the GEMM's epilogue
        O1b, O2b = torch.empty(b * nq, 256, device="cuda"), torch.empty(b * nq, 64, device="cuda")
        dcl.ops.cross_attention(b, Q, Km, None, O1b, Km, O2b, planes=planes_b)      # piece pass over V2 (and K) only
        nht = nk // 16
        va = planes_a[:b * nht * 30720].view(b * nht, 3, 320, 32)
        vb = planes_b[:b * nht * 30720].view(b * nht, 3, 320, 32)
        assert torch.equal(va, vb)                                                   # every piece of every channel, bit for bit
        assert torch.equal(O1a, O1b) and torch.equal(O2a, O2b)
        want = _attn_ref(Q.view(b, nq, 64), Km.view(b, nk, 64), torch.cat([V1.view(b, nk, 256), Km.view(b, nk, 64)], 2))
        got = torch.cat([O1b.view(b, nq, 256), O2b.view(b, nq, 64)], 2).double()
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    finally:
        lib.dcl_debug_attention_variant(0)
    with pytest.raises(RuntimeError):                                    # 3 crops of 200 queries: the fp32 kernel, which needs V1
        dcl.ops.cross_attention(b, Q, Km, None, O1b, Km, O2b, planes=planes_b)


def test_cross_attention_long_key_axis_below_a_full_round_takes_the_split_kernel(request, dcl):
    """fewer than 256 eight-wave workgroups but a long key axis (64 workgroups x 256 key tiles: what a stress-shape call of fewer
    than 32 crops is in its M -> N direction): the launcher takes the split-bf16 kernel with a key split (partial records +
    combine) instead of the fp32 4-wave kernel; both against float64, and the launch census says which kernel ran"""
    lib = enter_diag(dcl, request)
    g = torch.Generator().manual_seed(8)
    b, nq, nk = 2, 8192, 8192
    Q = (torch.randn(b * nq, 64, generator=g) * 0.5).cuda()
    K = (torch.randn(b * nk, 64, generator=g) * 0.5).cuda()
    V1 = torch.randn(b * nk, 256, generator=g).cuda()
    want = _attn_ref(Q.view(b, nq, 64), K.view(b, nk, 64), torch.cat([V1.view(b, nk, 256), K.view(b, nk, 64)], 2))
    tol = 2e-5 * max(1.0, float(want.abs().max()))
    try:
        for bf16 in (1, 0):
            lib.dcl_debug_attention_bf16(bf16)
            lib.dcl_debug_launch_census_reset()
            O1, O2 = torch.empty(b * nq, 256, device="cuda"), torch.empty(b * nq, 64, device="cuda")
            dcl.ops.cross_attention(b, Q, K, V1, O1, K, O2)
            got = torch.cat([O1.view(b, nq, 256), O2.view(b, nq, 64)], 2).double()
            assert float((got - want).abs().max()) <= tol, bf16
            import test_kernel_census as TC
            seen = TC.census(lib)
            assert (seen.get("k_cross_attn_split", 0) > 0) == (bf16 == 1), (bf16, {k: v for k, v in seen.items() if "attn" in k})
    finally:
        lib.dcl_debug_attention_bf16(1)

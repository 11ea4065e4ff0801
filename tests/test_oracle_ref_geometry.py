"""Pins the oracle's rulebooks (oracle/dclnet_oracle.c) against the REFERENCE's own spconv geometry code
(libs/spconv/include/spconv/geometry.h), compiled from /root/reference by oracle/Makefile into oracle/_ref/
(the prebuilt .so travels to the GPU box; if it is absent the committed fixture is used instead).

The reference CPU functions number output voxels in first-encounter order (geometry.h:181-187) while the GPU
path sorts them (spconv_ops.h:126): compare offset-wise (input row, output COORDINATE) pair sets and the output
coordinate set, then check the oracle's ordering rule (ascending linear index) on its own."""
import os

import numpy as np
import pytest

CASES = [(64, 3, 1, 1, False), (64, 3, 2, 1, False), (32, 3, 1, 1, False), (32, 3, 2, 1, False), (16, 3, 1, 1, True),
         (8, 3, 2, 1, False), (4, 3, 1, 1, True), (4, 3, 2, 1, False)]


def voxels(seed, b, S, n):
    rng = np.random.default_rng(seed)
    rows = []
    for bi in range(b):
        lin = rng.choice(S ** 3, size=min(n, S ** 3), replace=False)
        rows.append(np.stack([np.full(lin.shape, bi), lin // (S * S), (lin // S) % S, lin % S], 1))
    return np.concatenate(rows).astype(np.int32)


def coord_pairs(pairs, num, outids):
    return [set((int(i), tuple(outids[o])) for i, o in zip(pairs[k, 0, :num[k]], pairs[k, 1, :num[k]]))
            for k in range(pairs.shape[0])]


@pytest.mark.parametrize("S,ks,st,pad,subm", CASES)
def test_rulebook_sets_match_reference_geometry(oracle, S, ks, st, pad, subm):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref/libref_geometry.so absent (needs /root/reference at build time)")
    b = 2
    idx = voxels(S + st, b, S, 120)
    o_out, o_pairs, o_num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, ks, st, pad, 1, subm=subm)
    r_out, r_pairs, r_num, _ = oracle.get_indice_pairs(idx, b, [S] * 3, ks, st, pad, 1, subm=subm, use_ref=True)
    assert np.array_equal(o_num, r_num)
    assert set(map(tuple, o_out)) == set(map(tuple, r_out))
    assert coord_pairs(o_pairs, o_num, o_out) == coord_pairs(r_pairs, r_num, r_out)
    if not subm:                                   # the GPU ordering rule the oracle adds on top (spconv_ops.h:126)
        Sout = (S + 2 * pad - (ks - 1) - 1) // st + 1
        lin = ((o_out[:, 0].astype(np.int64) * Sout + o_out[:, 1]) * Sout + o_out[:, 2]) * Sout + o_out[:, 3]
        assert np.all(np.diff(lin) > 0)


def test_valid_out_pos_matches_reference(oracle):
    """getValidOutPos (geometry.h:23-85) for every position of a small grid, both strides"""
    import ctypes as C
    ref = oracle.ref_lib()
    if ref is None:
        pytest.skip("oracle/_ref absent")
    for S, st in ((8, 1), (8, 2)):
        Sout = (S + 2 - 2 - 1) // st + 1
        osh = (C.c_int * 3)(Sout, Sout, Sout)
        for x in range(S):
            pos = (C.c_int * 3)(x, (x * 3) % S, S - 1 - x)
            out = (C.c_int * (27 * 4))()
            n = ref.ref_valid_out_pos(pos, 3, st, 1, 1, osh, out)
            got = sorted(tuple(out[i * 4:i * 4 + 4]) for i in range(n))
            idx = np.array([[0, pos[0], pos[1], pos[2]]], np.int32)
            o_out, o_pairs, o_num, _ = oracle.get_indice_pairs(idx, 1, [S] * 3, 3, st, 1, 1)
            want = sorted(tuple(int(v) for v in o_out[o_pairs[k, 1, 0]][1:]) + (k,) for k in range(27) if o_num[k])
            assert got == want


def test_committed_rulebook_fixture(oracle, golden_dir):
    """reference-built rulebook statistics committed as a fixture (so this check also runs where _ref is absent)"""
    z = np.load(os.path.join(golden_dir, "rulebook_ref.npz"))
    for i, (S, ks, st, pad, subm) in enumerate(CASES):
        idx = voxels(S + st, 2, S, 120)
        o_out, o_pairs, o_num, _ = oracle.get_indice_pairs(idx, 2, [S] * 3, ks, st, pad, 1, subm=subm)
        assert np.array_equal(o_num, z["num_%d" % i])
        assert np.array_equal(np.array(sorted(map(tuple, o_out)), np.int32).reshape(-1, 4), z["outset_%d" % i])

"""Known-answer and edge-case checks of the C oracle itself (no GPU): semantics spelled out in SURVEY Appendix C."""
import numpy as np


def test_voxelize_idx_first_encounter_and_padding(oracle):
    coords = np.array([[0, 1, 1, 1], [0, 2, 2, 2], [0, 1, 1, 1], [1, 1, 1, 1], [0, 2, 2, 2], [0, 1, 1, 1]], np.int64)
    oc, im, om = oracle.voxelize_idx(coords, 2, 4)
    assert im.tolist() == [0, 1, 0, 2, 1, 0]
    assert oc.tolist() == [[0, 1, 1, 1], [0, 2, 2, 2], [1, 1, 1, 1]]
    assert om.tolist() == [[3, 0, 2, 5], [2, 1, 4, 0], [1, 3, 0, 0]]
    feats = np.arange(12, dtype=np.float32).reshape(6, 2)
    v = oracle.voxelize_fp(feats, om, 4)
    third = np.float32(1) / np.float32(3)
    want0 = (np.float32(0) + third * feats[0]) + third * feats[2]
    want0 = want0 + third * feats[5]
    assert np.array_equal(v[0], want0) and np.array_equal(v[2], feats[3])
    assert np.array_equal(oracle.voxelize_fp(feats, om, 3)[1], feats[1] + feats[4])


def test_ball_query_rules(oracle):
    xyz = np.array([[[0, 0, 0], [1, 0, 0], [0.1, 0, 0], [0.2, 0, 0], [5, 5, 5]]], np.float32)
    new = np.array([[[0, 0, 0], [9, 9, 9]]], np.float32)
    idx = oracle.ball_query(0.5, 4, xyz, new)
    assert idx[0, 0].tolist() == [0, 2, 3, 0]        # hits in ascending order, padded with the FIRST hit
    assert idx[0, 1].tolist() == [0, 0, 0, 0]        # no hit: the caller's zeros stay
    assert oracle.ball_query(0.1, 2, xyz, new)[0, 0].tolist() == [0, 0]   # strict d2 < r2 excludes the point at 0.1


def test_three_nn_sp_ties_and_empty_batches(oracle):
    known = np.array([[0, 1, 0, 0], [0, -1, 0, 0], [0, 0, 1, 0], [0, 0, -1, 0], [2, 0, 0, 0]], np.float32)
    unk = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [2, 0.5, 0, 0]], np.float32)
    d2, idx = oracle.three_nn_sp(unk, known)
    assert idx[0].tolist() == [0, 1, 2] and d2[0].tolist() == [1, 1, 1]      # equal distances keep the lowest index
    assert idx[1].tolist() == [0, 0, 0] and np.isinf(d2[1]).all()             # no voxel in batch 1
    assert idx[2].tolist() == [4, 0, 0] and d2[2, 0] == 0.25 and np.isinf(d2[2, 1:]).all()


def test_fps_reference_tie_rule(oracle):
    """points 1..4 are all at distance 1 from point 0: the winner is decided by the thread-strided scan + the
    left-biased tree of sampling_gpu.cu, not by 'smallest index'"""
    xyz = np.zeros((1, 8, 3), np.float32)
    xyz[0, 1:5] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0]]
    xyz[0, 5:] = [[0.1, 0, 0], [0, 0.1, 0], [0, 0, 0.1]]
    idx = oracle.furthest_point_sample(xyz, 3)
    assert idx[0, 0] == 0
    assert idx[0, 1] == 4        # T = 8 threads: tree keeps slot 0's side (t=0,4 -> 4 has d=1), then t=2/6, t=1/5/3/7
    assert oracle.lib().orc_fps_block_size(12288) == 1024 and oracle.lib().orc_fps_block_size(1000) == 512


def test_avgpool_divides_before_adding(oracle):
    idx = np.array([[0, 0, 0, 0], [0, 0, 0, 1], [0, 1, 1, 1]], np.int32)
    feat = np.array([[1.0], [2.0], [4.0]], np.float32)
    out_ids, pairs, num, _ = oracle.get_indice_pairs(idx, 1, [4, 4, 4], 3, 2, 1, 1)
    out, rf = oracle.indice_avgpool(feat, pairs, num, out_ids.shape[0])
    o000 = [i for i, r in enumerate(out_ids) if tuple(r) == (0, 0, 0, 0)][0]
    assert rf[o000] == 3
    third = np.float32(3)
    order = sorted((k for k in range(27) if (pairs[k, 1, :num[k]] == o000).any()))
    acc = np.float32(0)
    for k in order:
        j = int(np.where(pairs[k, 1, :num[k]] == o000)[0][0])
        acc = acc + feat[pairs[k, 0, j], 0] / third
    assert out[o000, 0] == acc


def test_subm_center_first_and_conv_order(oracle):
    rng = np.random.default_rng(0)
    idx = np.array([[0, 1, 1, 1], [0, 1, 1, 2], [0, 2, 1, 1]], np.int32)
    feat = rng.normal(size=(3, 4)).astype(np.float32)
    W = rng.normal(size=(3, 3, 3, 4, 5)).astype(np.float32)
    _, pairs, num, _ = oracle.get_indice_pairs(idx, 1, [4, 4, 4], 3, subm=True)
    assert int(np.argmax(num)) == 13 and num[13] == 3
    out = oracle.indice_conv(feat, W, pairs, num, 3, subm=True)
    dense = np.zeros((3, 5))
    for k in range(27):
        for c in range(num[k]):
            dense[pairs[k, 1, c]] += feat[pairs[k, 0, c]].astype(np.float64) @ W.reshape(27, 4, 5)[k].astype(np.float64)
    assert np.abs(out - dense).max() < 1e-5

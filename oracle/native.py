"""ctypes/numpy front end of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Each wrapper mirrors the reference pybind entry point it stands for (argument
meaning as in the reference; numpy arrays instead of torch tensors).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def _so_name(policy):
    return "liboracle.so" if policy == 0 else "liboracle_p%d.so" % policy


def build(force=False):
    """Compile liboracle.so, its three alternate-FMA-policy twins (and oracle/_ref when /root/reference is present)."""
    src = os.path.join(_HERE, "dclnet_oracle.c")
    for policy in (0, 1, 2, 3):
        so = os.path.join(_HERE, _so_name(policy))
        if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, _so_name(policy)], stdout=subprocess.DEVNULL)
    ref_so = os.path.join(_HERE, "_ref", "libref_geometry.so")
    if os.path.isdir("/root/reference/libs/spconv/include") and (force or not os.path.exists(ref_so)):
        subprocess.call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_POLICY = 0
_LIBS = {}


def lib():
    """liboracle.so under the FMA-association policy in effect (0 = the pinned one, see fma_policy)."""
    global _LIB
    if _POLICY not in _LIBS:
        build()
        L = C.CDLL(os.path.join(_HERE, _so_name(_POLICY)))
        assert L.orc_fma_policy() == _POLICY
        _LIBS[_POLICY] = L
    _LIB = _LIBS[_POLICY]
    return _LIB


class fma_policy:
    """`with fma_policy(p):` -- every oracle kernel called inside runs the build of dclnet_oracle.c made with
    -DORC_FMA_POLICY=p (0 pinned; 1 no contraction; 2 / 3 the other two FMA chains).  tests/test_fma_policy.py only."""

    def __init__(self, policy):
        assert policy in (0, 1, 2, 3)
        self.policy = policy

    def __enter__(self):
        global _POLICY
        self.prev, _POLICY = _POLICY, self.policy
        return self

    def __exit__(self, *exc):
        global _POLICY
        _POLICY = self.prev
        return False


def set_num_threads(n):
    """threads of the C kernels' OpenMP row loops; returns the count in effect"""
    return int(lib().orc_set_num_threads(int(n)))


def ref_lib():
    """The reference's own geometry.h build (oracle/_ref), or None if absent."""
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libref_geometry.so")
        if not os.path.exists(p):
            build()
        if os.path.exists(p):
            _REF = C.CDLL(p)
    return _REF


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# --------------------------------------------------------------------------- voxelization
def voxelize_idx(coords, batch_size, mode=4):
    """PG_OP.voxelize_idx (libs/pointgroup_ops/functions/pointgroup_ops.py:11-39).
    coords int64 (N,4) -> output_coords int64 (M,4), input_map i32 (N), output_map i32 (M,1+maxActive)."""
    assert mode in (0, 1, 2, 3, 4)
    coords = np.ascontiguousarray(coords, dtype=np.int64)
    n = coords.shape[0]
    input_map = np.zeros(n, np.int32)
    mx = C.c_int32(0)
    na = lib().orc_voxelize_idx_pass1(_p(coords, C.c_int64), n, _p(input_map, C.c_int32), C.byref(mx))
    if mode <= 2:                                          # voxelize.cpp:119-138: one point per voxel, maxActive = 1
        assert mode != 0 or mx.value == 1, "mode 0 promises unique coordinates (voxelize.cpp:120-124 asserts it)"
        out_coords = np.zeros((na, 4), np.int64)
        out_map = np.zeros((na, 2), np.int32)
        lib().orc_voxelize_idx_pass2_single(_p(coords, C.c_int64), n, _p(input_map, C.c_int32), int(mode),
                                            _p(out_coords, C.c_int64), _p(out_map, C.c_int32))
        return out_coords, input_map, out_map
    out_coords = np.zeros((na, 4), np.int64)
    out_map = np.zeros((na, mx.value + 1), np.int32)
    lib().orc_voxelize_idx_pass2(_p(coords, C.c_int64), n, _p(input_map, C.c_int32), na, mx.value,
                                 _p(out_coords, C.c_int64), _p(out_map, C.c_int32))
    return out_coords, input_map, out_map


def voxelize_fp(feats, rules, mode=4):
    """PG_OP.voxelize_fp (pointgroup_ops.py:42-62)."""
    feats = _f32(feats); rules = _i32(rules)
    V, ma = rules.shape[0], rules.shape[1] - 1
    Cn = feats.shape[1]
    out = np.zeros((V, Cn), np.float32)
    lib().orc_voxelize_fp(_p(feats, C.c_float), _p(rules, C.c_int32), _p(out, C.c_float), V, ma, Cn,
                          int(mode == 4))
    return out


# --------------------------------------------------------------------------- rulebooks
def conv_output_size(size, k, s, p, d=1):
    """spconv/ops.py:19-30"""
    return [(x + 2 * p - d * (k - 1) - 1) // s + 1 for x in size]


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1,
                     subm=False, use_ref=False):
    """torch.ops.spconv.get_indice_pairs_3d (spconv/ops.py:45-98) in the GPU order.
    Returns (outids (Vo,4) i32, indice_pairs (27,2,V) i32, indice_num (27) i32, out_shape).
    use_ref=True runs the reference's own geometry.h build instead (CPU order!)."""
    indices = _i32(indices)
    V = indices.shape[0]
    kv = ksize ** 3
    pairs = np.full((kv, 2, V), -1, np.int32)
    num = np.zeros(kv, np.int32)
    L = ref_lib() if use_ref else lib()
    if subm:
        shape = np.asarray(spatial_shape, np.int32)
        fn = L.ref_indice_pairs_subm if use_ref else L.orc_indice_pairs_subm
        if V:
            fn(_p(indices, C.c_int32), V, int(batch_size), _p(shape, C.c_int32), ksize, dilation,
               _p(pairs, C.c_int32), _p(num, C.c_int32))
        return indices, pairs, num, list(spatial_shape)
    oshape = conv_output_size(spatial_shape, ksize, stride, padding, dilation)
    osh = np.asarray(oshape, np.int32)
    outids = np.zeros((max(V * kv, 1), 4), np.int32)
    n_out = 0
    if V:
        fn = L.ref_indice_pairs_conv if use_ref else L.orc_indice_pairs_conv
        n_out = fn(_p(indices, C.c_int32), V, int(batch_size), _p(osh, C.c_int32), ksize, stride, padding,
                   dilation, _p(outids, C.c_int32), _p(pairs, C.c_int32), _p(num, C.c_int32))
    return outids[:n_out].copy(), pairs, num, oshape


def indice_conv(features, filters, indice_pairs, indice_num, num_act_out, subm=False):
    """torch.ops.spconv.indice_conv_fp32 (spconv/ops.py:101-118); filters (k,k,k,Cin,Cout)."""
    features = _f32(features); indice_pairs = _i32(indice_pairs); indice_num = _i32(indice_num)
    cin, cout = filters.shape[-2], filters.shape[-1]
    W = _f32(np.asarray(filters).reshape(-1, cin, cout))
    out = np.zeros((num_act_out, cout), np.float32)
    lib().orc_indice_conv(_p(features, C.c_float), _p(W, C.c_float), _p(indice_pairs, C.c_int32),
                          _p(indice_num, C.c_int32), indice_pairs.shape[2], num_act_out, cin, cout,
                          W.shape[0], int(subm), _p(out, C.c_float))
    return out


def indice_avgpool(features, indice_pairs, indice_num, num_act_out):
    """indiceSummaryRF + indice_avgpool_fp32 with use_gs=False (spconv/functional.py:137-162).
    Returns (out, summaryrf)."""
    features = _f32(features); indice_pairs = _i32(indice_pairs); indice_num = _i32(indice_num)
    c = features.shape[1]
    out = np.zeros((num_act_out, c), np.float32)
    rf = np.zeros(num_act_out, np.int32)
    lib().orc_indice_avgpool(_p(features, C.c_float), _p(indice_pairs, C.c_int32), _p(indice_num, C.c_int32),
                             indice_pairs.shape[2], num_act_out, c, indice_pairs.shape[0],
                             _p(rf, C.c_int32), _p(out, C.c_float))
    return out, rf


# --------------------------------------------------------------------------- pointnet_sp
def three_nn_sp(unknown, known):
    """pointnet2_cuda.three_nn_wrapper of libs/pointnet_sp (returns dist2, idx)."""
    unknown = _f32(unknown); known = _f32(known)
    n, m = unknown.shape[0], known.shape[0]
    d2 = np.empty((n, 3), np.float32); idx = np.empty((n, 3), np.int32)
    lib().orc_three_nn_sp(n, m, _p(unknown, C.c_float), _p(known, C.c_float), _p(d2, C.c_float),
                          _p(idx, C.c_int32))
    return d2, idx


def three_interpolate_sp(points, idx, weight):
    points = _f32(points); idx = _i32(idx); weight = _f32(weight)
    m, c = points.shape
    n = idx.shape[0]
    out = np.empty((n, c), np.float32)
    lib().orc_three_interpolate_sp(c, m, n, _p(points, C.c_float), _p(idx, C.c_int32), _p(weight, C.c_float),
                                   _p(out, C.c_float))
    return out


# --------------------------------------------------------------------------- pointnet_lib
def ball_query(radius, nsample, xyz, new_xyz):
    xyz = _f32(xyz); new_xyz = _f32(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().orc_ball_query(b, n, m, C.c_float(radius), nsample, _p(new_xyz, C.c_float), _p(xyz, C.c_float),
                         _p(idx, C.c_int32))
    return idx


def group_points(points, idx):
    points = _f32(points); idx = _i32(idx)
    b, c, n = points.shape
    _, npo, ns = idx.shape
    out = np.empty((b, c, npo, ns), np.float32)
    lib().orc_group_points(b, c, n, npo, ns, _p(points, C.c_float), _p(idx, C.c_int32), _p(out, C.c_float))
    return out


def gather_points(points, idx):
    points = _f32(points); idx = _i32(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.empty((b, c, m), np.float32)
    lib().orc_gather_points(b, c, n, m, _p(points, C.c_float), _p(idx, C.c_int32), _p(out, C.c_float))
    return out


def furthest_point_sample(xyz, npoint):
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, np.float32)
    idx = np.zeros((b, npoint), np.int32)
    lib().orc_furthest_point_sampling(b, n, npoint, _p(xyz, C.c_float), _p(temp, C.c_float), _p(idx, C.c_int32))
    return idx


def knn(k, unknown, known):
    unknown = _f32(unknown); known = _f32(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    assert k <= 200
    d2 = np.empty((b, n, k), np.float32); idx = np.empty((b, n, k), np.int32)
    lib().orc_knn(b, n, m, k, _p(unknown, C.c_float), _p(known, C.c_float), _p(d2, C.c_float), _p(idx, C.c_int32))
    return d2, idx


def three_nn(unknown, known):
    unknown = _f32(unknown); known = _f32(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.empty((b, n, 3), np.float32); idx = np.empty((b, n, 3), np.int32)
    lib().orc_three_nn(b, n, m, _p(unknown, C.c_float), _p(known, C.c_float), _p(d2, C.c_float), _p(idx, C.c_int32))
    return d2, idx


def three_interpolate(points, idx, weight):
    points = _f32(points); idx = _i32(idx); weight = _f32(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.empty((b, c, n), np.float32)
    lib().orc_three_interpolate(b, c, m, n, _p(points, C.c_float), _p(idx, C.c_int32), _p(weight, C.c_float),
                                _p(out, C.c_float))
    return out

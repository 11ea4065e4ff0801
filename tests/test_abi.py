"""The C-ABI library loads without a GPU and exports every symbol include/dclnet_hip.h declares (no compute calls
here); argument validation paths that never touch the device are exercised too."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(diag=False):
    """entry points include/dclnet_hip.h declares: for the product library (the `#ifdef DCL_DIAG` block of test / tuning
    hooks removed) or, diag=True, for the diagnostic library (everything)"""
    text = open(os.path.join(ROOT, "include", "dclnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    if not diag:
        text = re.sub(r"#ifdef DCL_DIAG.*?#endif", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dcl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("dcl_voxelize_idx_count", "dcl_voxelize_fp", "dcl_conv_out_grid", "dcl_rulebook_gather",
                 "dcl_sparse_conv_fwd", "dcl_sparse_avgpool_fwd", "dcl_three_nn_sp", "dcl_three_interpolate_sp",
                 "dcl_ball_query", "dcl_group_points", "dcl_gather_points", "dcl_furthest_point_sampling", "dcl_knn",
                 "dcl_three_nn", "dcl_three_interpolate", "dcl_cross_attention", "dcl_conf_pool",
                 "dcl_ortho9d_to_matrix", "dcl_backbone_geometry", "dcl_backbone_features", "dcl_point_features"):
        assert must in syms


def test_library_loads_and_exports_every_declared_symbol(dcl):
    lib = dcl._native.lib()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.dcl_abi_version() == 2 == dcl._native.ABI_VERSION        # include/dclnet_hip.h: DCL_ABI_VERSION


def test_product_library_has_no_hooks_and_the_diagnostic_library_has_them_all(dcl):
    """`nm -D libdclnet_hip.so | grep dcl_debug` is empty; variants are selected through the diagnostic twin"""
    import subprocess
    assert not [s for s in declared_symbols() if s.startswith("dcl_debug")]
    hooks = [s for s in declared_symbols(diag=True) if s.startswith("dcl_debug")]
    assert len(hooks) >= 10
    nm = subprocess.run(["nm", "-D", dcl._native.SO_PATH], capture_output=True, text=True).stdout
    assert "dcl_debug" not in nm
    assert os.path.dirname(dcl._native.DIAG_SO_PATH) != os.path.dirname(dcl._native.SO_PATH)     # not inside the package
    with dcl._native.diagnostic_library() as L:
        missing = [s for s in declared_symbols(diag=True) if not hasattr(L, s)]
        assert not missing, missing
    assert dcl._native.lib() is not L
    und = subprocess.run(["nm", "-D", "--undefined-only", dcl._native.SO_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in und, "the product library must not read the environment"


def test_no_one_launch_feature_stage_in_any_library(dcl):
    """the persistent one-launch feature stage (spin waits between the layers' work items, a pinned status
    word, a fall-back-for-good path) lost its A/B at every batch size (profiles/r4_stage_ab.txt) and is gone -- from the
    header, from the product and from the diagnostic library; Network takes no such switch"""
    import inspect
    import subprocess
    assert not [s for s in declared_symbols(diag=True) if "stage" in s and "backbone" in s]
    for path in (dcl._native.SO_PATH, dcl._native.DIAG_SO_PATH):
        if os.path.exists(path):
            nm = subprocess.run(["nm", path], capture_output=True, text=True).stdout
            assert "features_stage" not in nm and "k_feature_stage" not in nm and "k_stage_prepare" not in nm, path
    assert "feature_stage" not in inspect.signature(dcl.DCL_Net.Network.__init__).parameters
    assert not hasattr(dcl.ops, "backbone_features_stage")


def test_bad_arguments_return_einval_without_touching_the_gpu(dcl):
    lib = dcl._native.lib()
    assert lib.dcl_voxelize_fp(None, None, None, 4, 1, 0, 1, None) == -1          # n_planes must be > 0
    assert b"invalid argument" in lib.dcl_last_error()
    assert lib.dcl_knn(1, 4, 4, 201, None, None, None, None, None) == -1           # k <= 200
    assert lib.dcl_cross_attention(1, 8, 8, None, 64, None, 64, None, 48, 48, None, 48, None, 0, 0, None, 0, None) == -1
    n = C.c_int64(0)
    assert lib.dcl_backbone_ws_bytes(2, 48, 10, C.byref(n)) == -1                   # S must be a power of two
    assert lib.dcl_backbone_ws_bytes(2, 64, 1000, C.byref(n)) == 0 and n.value > 0


def test_host_voxelize_idx_runs_on_cpu(dcl, oracle):
    rng = np.random.default_rng(0)
    coords = np.concatenate([np.repeat(np.arange(4), 300)[:, None], rng.integers(0, 7, (1200, 3))], 1).astype(np.int64)
    oc, im, om = dcl.ops.voxelize_idx(torch.from_numpy(coords), 4, 4)
    rc, rm, rom = oracle.voxelize_idx(coords, 4, 4)
    assert np.array_equal(oc.numpy(), rc) and np.array_equal(im.numpy(), rm) and np.array_equal(om.numpy(), rom)
    c3 = torch.from_numpy(np.ascontiguousarray(coords[:, 1:]))
    oc3, im3, om3 = dcl.ops.voxelize_idx(c3, 1, 4)
    assert oc3.shape[1] == 3 and int(im3.max()) + 1 == oc3.shape[0]
    # modes 0 / 1 / 2 (voxelize.cpp:119-138): one point per voxel, rows [1, point], maxActive == 1
    for mode in (1, 2):
        oc1, im1, om1 = dcl.ops.voxelize_idx(torch.from_numpy(coords), 4, mode)
        r1 = oracle.voxelize_idx(coords, 4, mode)
        assert np.array_equal(oc1.numpy(), r1[0]) and np.array_equal(im1.numpy(), r1[1]) and np.array_equal(om1.numpy(), r1[2])
        assert om1.shape[1] == 2 and np.array_equal(oc1.numpy(), rc) and np.array_equal(im1.numpy(), rm)
        # known answer straight from the cited lines: .front() (mode 1) / .back() (mode 2) of the voxel's ascending point list
        pick = rom[:, 1] if mode == 1 else rom[np.arange(rom.shape[0]), rom[:, 0]]
        assert np.array_equal(om1.numpy()[:, 0], np.ones(rom.shape[0], np.int32)) and np.array_equal(om1.numpy()[:, 1], pick)
    with pytest.raises(RuntimeError):                                               # mode 0 promises unique coordinates
        dcl.ops.voxelize_idx(torch.from_numpy(coords), 4, 0)
    uniq = np.ascontiguousarray(coords[np.sort(np.unique(coords, axis=0, return_index=True)[1])])
    oc0, im0, om0 = dcl.ops.voxelize_idx(torch.from_numpy(uniq), 4, 0)
    r0 = oracle.voxelize_idx(uniq, 4, 0)
    assert np.array_equal(oc0.numpy(), r0[0]) and np.array_equal(im0.numpy(), r0[1]) and np.array_equal(om0.numpy(), r0[2])
    assert np.array_equal(im0.numpy(), np.arange(uniq.shape[0])) and np.array_equal(om0.numpy()[:, 1], np.arange(uniq.shape[0]))
    # ... and the voxel features they select through voxelize_fp's rule rows are single points (checked on the GPU side)


def test_product_never_imports_the_oracle():
    """the product path must not route through oracle/ (parity claims would be void)"""
    pkg = os.path.join(ROOT, "dcl-net_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(d, f)).read().lower()
                assert "oracle" not in text, (d, f)

import importlib
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")        # as dcl-net_amd/__init__.py does: before anything initialises the GPU

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """tests marked `gpu` are skipped (not failed) on a box without a GPU, so a plain `pytest` run is green there too"""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dcl():
    """the product package (its directory name has a hyphen)"""
    return importlib.import_module("dcl-net_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import native
    native.build()
    return native


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def load_golden_data(path):
    """tests/golden/dclnet_*.npz -> (loader-style data dict of torch CPU tensors, expected outputs, meta)"""
    import torch
    z = np.load(path)
    b, n_inp, n_tmp, wseed = [int(v) for v in z["meta"]]
    data = {"labels": {}, "batch_offsets": (torch.arange(b + 1) * n_inp).int(),
            "voxel_num_limit": torch.tensor([64, 64, 64]), "flags": torch.IntTensor([-1])}
    for side in ("inp", "tmp"):
        data[side] = {k: torch.from_numpy(z["%s_%s" % (side, k)]) for k in
                      ("feats", "occupied_voxels", "p2v_maps", "v2p_maps")}
    exp = {k: z[k] for k in z.files if not k.startswith(("inp_", "tmp_")) and k != "meta"}
    if "flags" in exp:                                    # train-mode fixtures carry the symmetry flags and the gt poses
        data["flags"] = torch.from_numpy(exp["flags"])
        data["labels"] = {"rot_gt": torch.from_numpy(exp["rot_gt"]), "trans_gt": torch.from_numpy(exp["trans_gt"])}
    return data, exp, (b, n_inp, n_tmp, wseed)


def load_stress_golden(dcl, path, voxelize_idx=None):
    """tests/golden/dclnet_stress_b1.npz: ONE crop of BASELINE configs[1]'s shape run through the reference's own Network
    (make_golden.py stress).  The inputs are procedural -- regenerated here and checked against the fixture's checksums."""
    import torch
    z = np.load(path)
    b, n_inp, n_tmp, wseed, first = [int(v) for v in z["meta"]]
    data = dcl.synth.make_batch(b, n_inp, n_tmp, first=first, voxelize_idx=voxelize_idx)
    for side in ("inp", "tmp"):
        chk = z["%s_check" % side]
        d = data[side]
        got = [float(d["feats"].double().sum()), float(d["occupied_voxels"].double().sum()), float(d["v2p_maps"].double().sum()),
               d["occupied_voxels"].shape[0], d["v2p_maps"].shape[1]]
        assert np.allclose(got, chk, rtol=0, atol=1e-6 * max(1.0, abs(chk[0]))), (side, got, chk)
    exp = {k: z[k] for k in ("trans_pred", "rot_pred", "conf", "F_Xo_p_sub", "F_Xo_p_sum")}
    return data, exp, (b, n_inp, n_tmp, wseed)

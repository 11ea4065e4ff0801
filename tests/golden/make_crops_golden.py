"""Generates tests/golden/crops_ref.npz by running the REFERENCE's own loaders' `__getitem__` (unmodified code of
YCBV/dataloader_test_YCBV.py and LM/dataloader_test_LM.py) on the synthetic frames of tests/crop_scene.py.

Runs only in the build container (needs /root/reference; nothing from it is copied).  The loader classes are imported
as they are; only their file I/O is replaced: `Image.open` / `scio.loadmat` return the arrays of the synthetic scene,
the dataset object is created without running `__init__` (which reads the real dataset) and given the attributes
`__init__` would have set, and `pointgroup_ops.voxelization_idx` is the oracle's restatement (pinned separately).
Absent third-party modules (open3d, transforms3d, cv2) are empty stubs -- no code path used here touches them.
The fixture stores only the loaders' OUTPUTS (the scenes regenerate from their seeds).

    python tests/golden/make_crops_golden.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from crop_scene import make_scene  # noqa: E402
from oracle import native as K  # noqa: E402

CFG = dict(input_size=96, tmp_size=64, unit_voxel_extent=[0.006] * 3, voxel_num_limit=[64] * 3, voxelization_mode=4)
YCBV_CASES = [(11, {}), (12, dict(tiny=0)), (13, dict(empty=1, undetected=2))]
LM_CASES = [(31, "test", False), (32, "eval", False), (33, "test", True), (34, "eval", True)]


def install_stubs():
    if not hasattr(np, "int"):
        np.int, np.float = int, float
    sys.dont_write_bytecode = True
    for name in ("open3d", "transforms3d", "transforms3d.euler", "cv2"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["transforms3d.euler"].euler2mat = None

    def voxelization_idx(coords, b, mode=4):
        return tuple(torch.from_numpy(a) for a in K.voxelize_idx(coords.numpy(), int(b), mode))
    pg = types.ModuleType("libs.pointgroup_ops.functions.pointgroup_ops")
    pg.voxelization_idx = voxelization_idx
    for pkg in ("libs", "libs.pointgroup_ops", "libs.pointgroup_ops.functions"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    sys.modules["libs.pointgroup_ops.functions"].pointgroup_ops = pg
    sys.modules["libs.pointgroup_ops.functions.pointgroup_ops"] = pg
    sys.path.insert(0, REF)


def ycbv_item(mod, sc, cfg, seed):
    ds = object.__new__(mod.YCBDataset)
    ds.npoint_inp, ds.npoint_tmp = cfg["input_size"], cfg["tmp_size"]
    ds.unit_voxel_extent = np.array(cfg["unit_voxel_extent"]).astype(float)
    ds.voxel_num_limit = np.array(cfg["voxel_num_limit"]).astype(float)
    ds.total_voxel_extent = ds.voxel_num_limit * ds.unit_voxel_extent
    ds.voxelization_mode = cfg["voxelization_mode"]
    ds.root, ds.path_mask, ds.list = "/scene", "/masks", ["frame"]
    ds.list_pc_CAD, ds.list_rgb_CAD = sc["cad_pts"], sc["cad_col"]
    H, W = sc["depth"].shape
    ds.xmap = np.array([[j for _ in range(W)] for j in range(H)])
    ds.ymap = np.array([[i for i in range(W)] for _ in range(H)])
    ds.cam_cx, ds.cam_cy, ds.cam_fx, ds.cam_fy, ds.cam_scale = 312.9869, 241.3109, 1066.778, 1067.487, 10000.0
    files = {"/scene/frame-color.png": sc["img"], "/scene/frame-depth.png": sc["depth"]}
    mats = {"/masks/000000.mat": {"labels": sc["label"], "rois": sc["rois"]},
            "/scene/frame-meta.mat": {"cls_indexes": sc["gt_obj"].reshape(-1, 1), "poses": sc["poses"]}}
    mod.Image.open = lambda path: files[path]
    mod.scio.loadmat = lambda path: mats[path]
    np.random.seed(seed)
    return ds[0]


def lm_item(mod, sc, cfg, seed, mode, depth, mask_label, bb, cls):
    ds = object.__new__(mod.Dataset)
    ds.npoint_inp, ds.npoint_tmp = cfg["input_size"], cfg["tmp_size"]
    ds.unit_voxel_extent = np.array(cfg["unit_voxel_extent"]).astype(float)
    ds.voxel_num_limit = np.array(cfg["voxel_num_limit"]).astype(float)
    ds.total_voxel_extent = ds.voxel_num_limit * ds.unit_voxel_extent
    ds.voxelization_mode, ds.mode = cfg["voxelization_mode"], mode
    ds.objlist = [cls]
    ds.symmetry_obj_idx = []
    ds.list_rgb, ds.list_depth, ds.list_label, ds.list_obj, ds.list_rank = ["rgb"], ["depth"], ["label"], [cls], [0]
    ds.meta = {cls: {0: [{"obj_bb": bb, "cam_R_m2c": list(np.eye(3).ravel()), "cam_t_m2c": [10.0, 20.0, 800.0],
                          "obj_id": cls}]}}
    ds.list_pc_CAD, ds.list_rgb_CAD = sc["cad_pts"], sc["cad_col"]
    H, W = depth.shape
    ds.xmap = np.array([[j for _ in range(W)] for j in range(H)])
    ds.ymap = np.array([[i for i in range(W)] for _ in range(H)])
    ds.cam_cx, ds.cam_cy, ds.cam_fx, ds.cam_fy = 325.26110, 242.04899, 572.41140, 573.57043
    white = (mask_label[:, :, None] * 255).astype(np.uint8).repeat(3, axis=2)          # the dataset's mask png
    label = (mask_label * 255).astype(np.uint8) if mode == "eval" else white
    files = {"rgb": sc["img"], "depth": depth, "label": label}
    mod.Image.open = lambda path: files[path]
    if mode == "eval":                                       # segnet label -> box via cv2 in the loader: feed the same box
        mod.mask_to_bbox = lambda mask: bb
    np.random.seed(seed)
    return ds[0]


def main():
    install_stubs()
    out = {}
    ymod = importlib.import_module("YCBV.dataloader_test_YCBV")
    for seed, kw in YCBV_CASES:
        sc = make_scene(seed, tmp_size=CFG["tmp_size"], **kw)
        d = ycbv_item(ymod, sc, CFG, 100 + seed)
        tag = "ycbv%d_" % seed
        for side in ("inp", "tmp"):
            for k in ("feats", "occupied_voxels", "p2v_maps", "v2p_maps"):
                out[tag + side + "_" + k] = d[side][k].numpy()
        out[tag + "centroids"] = d["all_centroids"].numpy()
        out[tag + "flags"] = d["all_flags"].numpy()
        out[tag + "rot_gt"], out[tag + "trans_gt"] = d["labels"]["rot_gt"].numpy(), d["labels"]["trans_gt"].numpy()
    lmod = importlib.import_module("LM.dataloader_test_LM")
    cfg_lm = dict(CFG, unit_voxel_extent=[0.005] * 3, input_size=128)
    for seed, mode, small in LM_CASES:
        sc = make_scene(seed, n_obj=1, tmp_size=cfg_lm["tmp_size"], tiny=0 if small else None)
        cls = int(sc["gt_obj"][0])
        mask_label = sc["label"] == cls
        depth = (sc["depth"].astype(np.float64) / 10).astype(np.uint16)
        ys, xs = np.nonzero(mask_label)
        bb = [int(xs.min()) - 3, int(ys.min()) - 2, int(xs.max() - xs.min()) + 7, int(ys.max() - ys.min()) + 5]
        item = lm_item(lmod, sc, cfg_lm, seed, mode, depth, mask_label, bb, cls)
        tag = "lm%d_" % seed
        out[tag + "flag"] = item[4].numpy()
        if float(item[4][0]) != -1:
            out[tag + "feat_inp"], out[tag + "vox_inp"] = item[0].numpy(), item[1].numpy()
            out[tag + "feat_tmp"], out[tag + "vox_tmp"] = item[2].numpy(), item[3].numpy()
            out[tag + "centroid"] = item[9].numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "crops_ref.npz"), **out)
    print("golden written: crops_ref.npz", len(out), "arrays")


if __name__ == "__main__":
    main()

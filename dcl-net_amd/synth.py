"""Procedural YCB-V-shaped inputs and seeded weights: the GPU box has neither datasets nor
checkpoints, so crops are synthesised the way YCBV/dataloader_test_YCBV.py:146-258 builds them from a
depth crop: camera-facing half of a posed object surface + 1 mm noise, centroid-centred, box-filtered to
the 64^3 x 6 mm voxel volume, resampled to N points, [1, rgb, xyz] features, voxel indices, voxelize_idx.

numpy only (deterministic across machines); the product's own host voxelize_idx (ops.voxelize_idx) is
the default hashing step, exactly where the reference's loader calls pointgroup_ops.voxelization_idx.
"""
import numpy as np
import torch

N_CLASSES = 21
RGB_MEAN = np.array([0.485, 0.456, 0.406], np.float32)     # YCBV/dataloader_test_YCBV.py:57-58


def _class_shape(cls):
    """(kind, half extents) spanning the measured YCB CAD range 0.039 .. 0.147 m."""
    rng = np.random.default_rng(7000 + cls)
    kind = ("box", "cylinder", "ellipsoid", "bar")[cls % 4]
    big = 0.039 + (0.147 - 0.039) * (cls / (N_CLASSES - 1))
    ext = np.array([big, big * rng.uniform(0.4, 0.9), big * rng.uniform(0.3, 0.8)], np.float64)
    if kind == "bar":
        ext[1:] = big * rng.uniform(0.08, 0.2, 2)
    return kind, ext


def _sample_surface(kind, ext, n, rng):
    """n points on the surface + outward unit normals."""
    a, b, c = ext
    if kind in ("box", "bar"):
        areas = np.array([b * c, b * c, a * c, a * c, a * b, a * b])
        face = rng.choice(6, size=n, p=areas / areas.sum())
        u = rng.uniform(-1, 1, (n, 3)) * ext
        nrm = np.zeros((n, 3))
        ax = face // 2
        sign = np.where(face % 2 == 0, 1.0, -1.0)
        u[np.arange(n), ax] = sign * ext[ax]
        nrm[np.arange(n), ax] = sign
        return u, nrm
    if kind == "cylinder":            # axis = x, radii (b, c)
        side_area = 2 * a * np.pi * (b + c)
        cap_area = np.pi * b * c
        on_side = rng.uniform(size=n) < side_area / (side_area + 2 * cap_area)
        th = rng.uniform(0, 2 * np.pi, n)
        r = np.sqrt(rng.uniform(size=n))
        x = rng.uniform(-a, a, n)
        sgn = np.where(rng.uniform(size=n) < 0.5, 1.0, -1.0)
        p = np.stack([np.where(on_side, x, sgn * a), np.where(on_side, 1.0, r) * b * np.cos(th),
                      np.where(on_side, 1.0, r) * c * np.sin(th)], 1)
        nrm = np.stack([np.where(on_side, 0.0, sgn), np.where(on_side, np.cos(th) / b, 0.0),
                        np.where(on_side, np.sin(th) / c, 0.0)], 1)
        return p, nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    nrm = d / ext
    return d * ext, nrm / np.linalg.norm(nrm, axis=1, keepdims=True)


def _rand_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def class_template(cls, M):
    """fixed template cloud of class `cls`: (M,3) xyz [m], (M,3) rgb (mean-subtracted)."""
    kind, ext = _class_shape(cls)
    rng = np.random.default_rng(cls)
    pts, _ = _sample_surface(kind, ext, M, rng)
    rgb = rng.uniform(0, 1, (M, 3)).astype(np.float32) - RGB_MEAN
    return pts.astype(np.float32), rgb


def make_crop(i, N, M, unit=0.006, S=64):
    """crop i: class i % 21, seed 1000+i."""
    cls = i % N_CLASSES
    kind, ext = _class_shape(cls)
    rng = np.random.default_rng(1000 + i)
    half = 0.5 * unit * S
    R = _rand_rotation(rng)
    t = rng.uniform(-0.03, 0.03, 3) + np.array([0.0, 0.0, 0.8])           # object ~0.8 m in front of the camera
    dense, nrm = _sample_surface(kind, ext, max(4 * N, 4096), rng)
    posed = dense @ R.T + t
    view = posed / np.linalg.norm(posed, axis=1, keepdims=True)
    vis = np.einsum("ij,ij->i", nrm @ R.T, view) < 0                      # camera-facing half
    cloud = posed[vis] + rng.normal(0, 0.001, (int(vis.sum()), 3))
    rgb = (rng.uniform(0, 1, (cloud.shape[0], 3)).astype(np.float32) - RGB_MEAN)
    centroid = cloud.mean(0)
    cloud = cloud - centroid
    t = t - centroid
    keep = (np.abs(cloud) < half).all(1)                                  # dataloader_test_YCBV.py:164-169
    if keep.sum() > 32:
        cloud, rgb = cloud[keep], rgb[keep]
    sel = rng.choice(cloud.shape[0], N, replace=cloud.shape[0] <= N)
    cloud, rgb = cloud[sel].astype(np.float32), rgb[sel]
    vox_inp = ((cloud + np.float32(half)) / np.float32(unit)).astype(np.int64)   # .long() truncation, :178
    if keep.sum() <= 32:
        vox_inp = np.clip(vox_inp, 0, S - 1)
    tmp, tmp_rgb = class_template(cls, M)
    vox_tmp = ((tmp + np.float32(half)) / np.float32(unit)).astype(np.int64)
    feat_inp = np.concatenate([np.ones((N, 1), np.float32), rgb, cloud], 1)
    feat_tmp = np.concatenate([np.ones((M, 1), np.float32), tmp_rgb, tmp], 1)
    return dict(feat_inp=feat_inp, vox_inp=vox_inp, feat_tmp=feat_tmp, vox_tmp=vox_tmp, R=R.astype(np.float32),
                t=t.astype(np.float32), cls=cls)


def make_batch(b, N=1024, M=1024, unit=0.006, S=64, first=0, voxelize_idx=None, mode=4):
    """The reference loader's `data` dict (YCBV/dataloader_test_YCBV.py:228-258) for crops first..first+b-1."""
    if voxelize_idx is None:
        from . import ops
        voxelize_idx = ops.voxelize_idx
    crops = [make_crop(first + i, N, M, unit, S) for i in range(b)]
    data = {"labels": {"rot_gt": torch.from_numpy(np.stack([c["R"] for c in crops])),
                       "trans_gt": torch.from_numpy(np.stack([c["t"] for c in crops]))},
            "batch_offsets": (torch.arange(b + 1) * N).int(),
            "voxel_num_limit": torch.tensor([S, S, S]),
            "obj_idx": torch.IntTensor([c["cls"] for c in crops]),
            "all_flags": torch.ones(b, dtype=torch.int32),
            "flags": torch.IntTensor([-1])}
    for side, n in (("inp", N), ("tmp", M)):
        feats = torch.from_numpy(np.concatenate([c["feat_" + side] for c in crops], 0))
        vox = np.concatenate([c["vox_" + side] for c in crops], 0)
        bid = np.repeat(np.arange(b, dtype=np.int64), n)[:, None]
        coords = torch.from_numpy(np.ascontiguousarray(np.concatenate([bid, vox], 1)))
        occ, p2v, v2p = voxelize_idx(coords, b, mode)
        data[side] = {"feats": feats, "occupied_voxels": occ, "p2v_maps": p2v, "v2p_maps": v2p}
    return data


def synth_state_dict(model, seed=1):
    """Seeded stand-in for the unavailable checkpoints: He-normal convs, non-trivial BatchNorm statistics.
    Deterministic across machines (numpy Generator), keyed by parameter name."""
    sd = {}
    rng = np.random.default_rng(seed)
    for name, ref in sorted(model.state_dict().items()):
        shape = tuple(ref.shape)
        if name.endswith("num_batches_tracked"):
            v = np.zeros(shape, np.int64)
        elif name.endswith("running_mean"):
            v = rng.normal(0, 0.1, shape)
        elif name.endswith("running_var"):
            v = rng.uniform(0.5, 1.5, shape)
        elif len(shape) == 1 and name.endswith("weight"):              # BatchNorm gamma
            v = rng.uniform(0.5, 1.5, shape)
        elif len(shape) == 1:                                          # biases / BatchNorm beta
            v = rng.normal(0, 0.1, shape)
        elif len(shape) == 5 and shape[0] == shape[1] == shape[2] == 3:   # sparse conv (3,3,3,Cin,Cout)
            v = rng.normal(0, np.sqrt(2.0 / (9 * shape[3])), shape)
        else:                                                          # (Cout, Cin, 1[,1,1])
            v = rng.normal(0, np.sqrt(2.0 / shape[1]), shape)
        if name.startswith(("regressor_trans.layers.4", "regressor_trans2.layers.4")):
            v = v * 0.02          # keep translations at the centimetre scale of real crops (tolerances are in metres)
        sd[name] = torch.from_numpy(np.asarray(v)).to(ref.dtype)
    return sd


def default_cfg(n_inp=1024, n_tmp=1024, unit=0.006):
    """cfg.model block of configs/config_YCBV_bs32.yaml:20-29 as an attribute dict."""
    class A(dict):
        __getattr__ = dict.__getitem__
    return A(voxelization_mode=4, unit_voxel_extent=[unit] * 3, voxel_num_limit=[64, 64, 64], n_inp=n_inp,
             n_tmp=n_tmp, backbone=A(downsample_by_pooling=True, kernel_size=3, bias=False))


def load_yaml_cfg(path):
    """Tiny stand-in for gorilla.Config.fromfile: yaml -> attribute dict (keys as in the reference configs)."""
    import yaml

    class A(dict):
        __getattr__ = dict.__getitem__

    def wrap(x):
        if isinstance(x, dict):
            return A({k: wrap(v) for k, v in x.items()})
        return x
    with open(path) as fh:
        return wrap(yaml.safe_load(fh))


def load_checkpoint(model, filename, map_location="cpu"):
    """Loads a reference checkpoint (gorilla.solver.save_checkpoint layout {"model": state_dict, "optimizer": ...,
    "meta": ...}, tools/train_YCBV_stage1.py:102-104) or a bare state_dict into `model`."""
    ckpt = torch.load(filename, map_location=map_location)
    sd = ckpt.get("model", ckpt.get("state_dict", ckpt)) if isinstance(ckpt, dict) else ckpt
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}      # nn.DataParallel prefix
    model.load_state_dict(sd)
    return ckpt.get("meta", {}) if isinstance(ckpt, dict) else {}


# ---------------------------------------------------------------------------------------------- synthetic eval frames
# Stand-in for one YCB-V test frame (colour, 16-bit depth, label image, PoseCNN rois, CAD clouds): only the array shapes /
# dtypes of what YCBV/dataloader_test_YCBV.py:99-106 reads from disk matter to the crop builder (dcl-net_amd/crops.py).
# Used by the crop-builder tests (tests/crop_scene.py re-exports it) and by bench.py's eval-stream legs.
FRAME_H, FRAME_W = 480, 640


def make_frame(seed, n_obj=4, tmp_size=64, tiny=None, empty=None, undetected=None, rgba=False):
    """tiny / empty / undetected: instance numbers that get a <=32-pixel mask / a mask fully at depth 0 / no roi."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (FRAME_H, FRAME_W, 4 if rgba else 3), dtype=np.uint8)
    depth = rng.integers(6000, 14000, (FRAME_H, FRAME_W)).astype(np.uint16)
    depth[rng.random((FRAME_H, FRAME_W)) < 0.1] = 0                                   # sensor holes
    label = np.zeros((FRAME_H, FRAME_W), np.int32)
    classes = rng.permutation(np.arange(1, 22))[:n_obj]
    rois = []
    for k, cls in enumerate(classes):
        if k == tiny:
            h, w = 4, 6
        else:
            h, w = int(rng.integers(50, 170)), int(rng.integers(50, 200))
        r0, c0 = int(rng.integers(0, FRAME_H - h)), int(rng.integers(0, FRAME_W - w))
        yy, xx = np.mgrid[0:h, 0:w]
        blob = ((yy - h / 2) / (h / 2)) ** 2 + ((xx - w / 2) / (w / 2)) ** 2 <= 1.0 if k != tiny else np.ones((h, w), bool)
        sub = label[r0:r0 + h, c0:c0 + w]
        sub[blob] = cls
        # object surface: smooth depth + noise, so that most points fall inside the 0.384 m voxel grid
        z0 = int(rng.integers(7000, 12000))
        surf = (z0 + 300 * np.sin(yy / 17.0) + 200 * np.cos(xx / 23.0) + rng.normal(0, 15, (h, w))).astype(np.uint16)
        dsub = depth[r0:r0 + h, c0:c0 + w]
        holes = dsub == 0
        dsub[blob] = surf[blob]
        dsub[holes] = 0
        if k == tiny:                                                     # far outliers keep valid_num <= 32 interesting
            dsub[0, 0] = 30000
        if k == empty:
            dsub[blob] = 0
        if k != undetected:
            rois.append([0, cls, c0 - 3, r0 - 2, c0 + w + 2, r0 + h + 3, 0.9])
    rois.append([0, 99, 10, 10, 60, 60, 0.5])                             # a detection of a class that is not in the frame
    poses = rng.normal(size=(3, 4, n_obj))
    cad_pts = {c: rng.uniform(-90, 90, (tmp_size, 3)) for c in range(1, 23)}
    cad_col = {c: rng.uniform(0, 1, (tmp_size, 3)) - np.array([0.485, 0.456, 0.406]) for c in range(1, 23)}
    return dict(img=img, depth=depth, label=label, rois=np.asarray(rois, np.float64), gt_obj=classes.astype(np.int32),
                poses=poses, cad_pts=cad_pts, cad_col=cad_col)

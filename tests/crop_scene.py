"""Synthetic stand-in for one YCB-V test frame (colour, 16-bit depth, label image, PoseCNN rois, CAD clouds): only the
array shapes/dtypes of what YCBV/dataloader_test_YCBV.py:99-106 reads from disk matter to the crop builder."""
import numpy as np

H, W = 480, 640


def make_scene(seed, n_obj=4, tmp_size=64, tiny=None, empty=None, undetected=None, rgba=False):
    """tiny / empty / undetected: instance numbers that get a <=32-pixel mask / a mask fully at depth 0 / no roi."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (H, W, 4 if rgba else 3), dtype=np.uint8)
    depth = rng.integers(6000, 14000, (H, W)).astype(np.uint16)
    depth[rng.random((H, W)) < 0.1] = 0                                   # sensor holes
    label = np.zeros((H, W), np.int32)
    classes = rng.permutation(np.arange(1, 22))[:n_obj]
    rois = []
    for k, cls in enumerate(classes):
        if k == tiny:
            h, w = 4, 6
        else:
            h, w = int(rng.integers(50, 170)), int(rng.integers(50, 200))
        r0, c0 = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
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

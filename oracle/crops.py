"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's per-image crop construction
(YCBV/dataloader_test_YCBV.py:99-258 `YCBDataset.__getitem__`, `get_bbox` :263-303) in numpy / torch-CPU, with the
file I/O replaced by arguments.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

Every arithmetic statement keeps the reference's operand types (float32 arrays, Python-float camera constants, the
float64 colour mean, torch float32 voxel arithmetic) because the device builder is required to be bit-identical.
Parity pin: tests/golden/crops_ref.npz holds the outputs of the reference's OWN loader code -- `YCBDataset.__getitem__`
and the LineMOD `Dataset.__getitem__`, imported unmodified in the build container with only their file I/O replaced
(tests/golden/make_crops_golden.py) -- on the seeded synthetic frames of tests/crop_scene.py; this restatement reproduces
them bit for bit (tests/test_crops.py).
"""
import numpy as np
import numpy.ma as ma
import torch

from . import native

IMG_H, IMG_W = 480, 640
BORDERS = [-1, 40, 80, 120, 160, 200, 240, 280, 320, 360, 400, 440, 480, 520, 560, 600, 640, 680]
RGB_MEAN = np.array([0.485, 0.456, 0.406])
CAM = dict(cx=312.9869, cy=241.3109, fx=1066.778, fy=1067.487, scale=10000.0)      # dataloader_test_YCBV.py:77-81


def get_bbox(rois, idx, img_h=IMG_H, img_w=IMG_W):
    """dataloader_test_YCBV.py:266-303: detection box -> window whose sides are snapped up to a multiple of 40,
    centred on the detection and shifted back inside the image."""
    rmin = max(int(rois[idx][3]) + 1, 0)
    rmax = min(int(rois[idx][5]) - 1, img_h)
    cmin = max(int(rois[idx][2]) + 1, 0)
    cmax = min(int(rois[idx][4]) - 1, img_w)

    def snap(v):
        for lo, hi in zip(BORDERS[:-1], BORDERS[1:]):
            if lo < v < hi:
                return hi
        return v
    r_b, c_b = snap(rmax - rmin), snap(cmax - cmin)
    cr, cc = int((rmin + rmax) / 2), int((cmin + cmax) / 2)
    rmin, rmax = cr - int(r_b / 2), cr + int(r_b / 2)
    cmin, cmax = cc - int(c_b / 2), cc + int(c_b / 2)
    if rmin < 0:
        rmin, rmax = 0, rmax - rmin
    if cmin < 0:
        cmin, cmax = 0, cmax - cmin
    if rmax > img_h:
        rmin, rmax = rmin - (rmax - img_h), img_h
    if cmax > img_w:
        cmin, cmax = cmin - (cmax - img_w), img_w
    return rmin, rmax, cmin, cmax


def build_image(img, depth, label, rois, gt_obj, cad_points_mm, cad_colors, cfg, poses=None, cam=CAM):
    """One image -> the loader's `data` dict (:99-258).  img (H,W,>=3) u8, depth (H,W) u16, label (H,W) int,
    rois (k,>=6) [_, cls, x1, y1, x2, y2], gt_obj (n) class ids, cad_points_mm / cad_colors: {cls: (M,3) float64} as
    held in list_pc_CAD / list_rgb_CAD (:57-58), poses (3,4,n) or None.  Draws from np.random like the reference."""
    npoint_inp, npoint_tmp = cfg["input_size"], cfg["tmp_size"]
    unit = np.array(cfg["unit_voxel_extent"]).astype(float)                        # :17-19
    limit = np.array(cfg["voxel_num_limit"]).astype(float)
    extent = limit * unit
    H, W = depth.shape
    xmap = np.array([[j for _ in range(W)] for j in range(H)])                     # :75-76
    ymap = np.array([[i for i in range(W)] for _ in range(H)])
    mask_depth = ma.getmaskarray(ma.masked_not_equal(depth, 0))                    # :102
    n = len(gt_obj)
    flags = np.zeros(n, np.int8)
    feats_inp, vox_inp, feats_tmp, vox_tmp, centroids, rot, trans, counts = [], [], [], [], [], [], [], []
    for idx in range(n):
        if np.sum(rois[:, 1] == gt_obj[idx]) == 0:                                 # :116
            continue
        rmin, rmax, cmin, cmax = get_bbox(rois, np.where(rois[:, 1] == gt_obj[idx])[0][0], H, W)
        mask = ma.getmaskarray(ma.masked_equal(label, gt_obj[idx])) * mask_depth   # :126-127
        choose = mask[rmin:rmax, cmin:cmax].flatten().nonzero()[0]                 # :132
        if choose.shape[0] == 0:                                                   # :134
            continue
        flags[idx] = 1
        img_masked = np.array(img)[:, :, :3][rmin:rmax, cmin:cmax, :].astype(np.float32).reshape((-1, 3))[choose, :]
        img_masked = img_masked / 255.0 - RGB_MEAN[np.newaxis, :]                  # :145 (float64 from here)
        depth_masked = depth[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
        xmap_masked = xmap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
        ymap_masked = ymap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
        pt2 = depth_masked / cam["scale"]                                          # :151-154
        pt0 = (ymap_masked - cam["cx"]) * pt2 / cam["fx"]
        pt1 = (xmap_masked - cam["cy"]) * pt2 / cam["fy"]
        cloud = np.concatenate((pt0, pt1, pt2), axis=1)
        assert cloud.dtype == np.float32
        centroid = np.mean(cloud, axis=0)                                          # :156
        cloud = cloud - centroid[np.newaxis, :]
        inside = (np.abs(cloud[:, 0]) < extent[0] * 0.5) & (np.abs(cloud[:, 1]) < extent[1] * 0.5) & \
                 (np.abs(cloud[:, 2]) < extent[2] * 0.5)                           # :160
        valid_num = np.sum(inside)
        if valid_num > 32:                                                         # :163-165
            cloud, img_masked = cloud[inside, :], img_masked[inside, :]
        if cloud.shape[0] > npoint_inp:                                            # :166-169
            pick = np.random.choice(cloud.shape[0], npoint_inp, replace=False)
        else:
            pick = np.random.choice(cloud.shape[0], npoint_inp)
        counts.append((choose.shape[0], int(valid_num), cloud.shape[0]))
        cloud_t = torch.FloatTensor(cloud[pick, :])
        rgb_t = torch.FloatTensor(img_masked[pick, :])
        feats_inp.append(torch.cat([torch.ones(npoint_inp, 1), rgb_t, cloud_t], 1))          # :173
        v = (cloud_t + extent[0] * 0.5) / torch.FloatTensor(unit)                            # :174
        if valid_num <= 32:
            v = torch.clamp(v, min=0, max=limit[0] - 1)                                      # :176-177
        vox_inp.append(v.long())
        model_points = torch.FloatTensor(cad_points_mm[int(gt_obj[idx])] / 1000.0)           # :180-183
        model_colors = torch.FloatTensor(cad_colors[int(gt_obj[idx])])
        feats_tmp.append(torch.cat([torch.ones(npoint_tmp, 1), model_colors, model_points], 1))
        vox_tmp.append(((model_points + extent[0] * 0.5) / torch.FloatTensor(unit)).long())
        centroids.append(torch.tensor(centroid))
        if poses is not None:
            rot.append(torch.FloatTensor(np.array(poses[:, :, idx][:, 0:3])))
            trans.append(torch.FloatTensor(np.array([poses[:, :, idx][:, 3:4].flatten()]).reshape(3) - centroid))
    b = len(feats_inp)
    data = {"batch_offsets": (torch.arange(b + 1) * 1024).int(), "voxel_num_limit": torch.tensor(limit),
            "obj_idx": torch.IntTensor(np.asarray(gt_obj) - 1), "all_flags": torch.IntTensor(flags),
            "flags": torch.IntTensor([-1]), "all_centroids": torch.stack(centroids), "labels": {},
            "counts": np.asarray(counts, np.int32)}
    if poses is not None:
        data["labels"] = {"rot_gt": torch.stack(rot), "trans_gt": torch.stack(trans)}
    for side, fl, vl, npnt in (("inp", feats_inp, vox_inp, npoint_inp), ("tmp", feats_tmp, vox_tmp, npoint_tmp)):
        feats = torch.stack(fl).reshape(b * npnt, 7)                                         # :201-215
        vox = torch.stack(vl).reshape(b * npnt, 3)
        ids = torch.arange(b).unsqueeze(1).repeat(1, npnt).view(b * npnt, 1).long()
        coords = torch.cat([ids, vox], 1)
        occ, p2v, v2p = native.voxelize_idx(coords.numpy(), b, cfg["voxelization_mode"])
        data[side] = {"feats": feats, "coords": coords, "occupied_voxels": torch.from_numpy(occ),
                      "p2v_maps": torch.from_numpy(p2v), "v2p_maps": torch.from_numpy(v2p)}
    return data


LM_CAM = dict(cx=325.26110, cy=242.04899, fx=572.41140, fy=573.57043)                # LM/dataloader_test_LM.py:104-107


def lm_get_bbox(bbox, img_h=480, img_w=640):
    """LM/dataloader_test_LM.py:287-333."""
    bbx = [bbox[1], bbox[1] + bbox[3], bbox[0], bbox[0] + bbox[2]]
    if bbx[0] < 0:
        bbx[0] = 0
    if bbx[1] >= img_h:
        bbx[1] = img_h - 1
    if bbx[2] < 0:
        bbx[2] = 0
    if bbx[3] >= img_w:
        bbx[3] = img_w - 1
    rmin, rmax, cmin, cmax = bbx

    def snap(v):
        for lo, hi in zip(BORDERS[:-1], BORDERS[1:]):
            if lo < v < hi:
                return hi
        return v
    r_b, c_b = snap(rmax - rmin), snap(cmax - cmin)
    cr, cc = int((rmin + rmax) / 2), int((cmin + cmax) / 2)
    rmin, rmax = cr - int(r_b / 2), cr + int(r_b / 2)
    cmin, cmax = cc - int(c_b / 2), cc + int(c_b / 2)
    if rmin < 0:
        rmin, rmax = 0, rmax - rmin
    if cmin < 0:
        cmin, cmax = 0, cmax - cmin
    if rmax > img_h:
        rmin, rmax = rmin - (rmax - img_h), img_h
    if cmax > img_w:
        cmin, cmax = cmin - (cmax - img_w), img_w
    return rmin, rmax, cmin, cmax


def build_lm_sample(img, depth, mask_label, obj_bb, obj, cad_points_mm, cad_colors, cfg, eval_mode=False):
    """`__getitem__` of the LineMOD loader (LM/dataloader_test_LM.py:116-214; test/eval modes, i.e. without the training
    augmentation): returns (feat_inp, voxel_inp, feat_tmp, voxel_tmp, centroid) or None for its all-zero dummy sample."""
    npoint_inp = cfg["input_size"]
    unit = np.array(cfg["unit_voxel_extent"]).astype(float)
    extent = np.array(cfg["voxel_num_limit"]).astype(float) * unit
    H, W = depth.shape
    xmap = np.array([[j for _ in range(W)] for j in range(H)])
    ymap = np.array([[i for i in range(W)] for _ in range(H)])
    mask = mask_label * ma.getmaskarray(ma.masked_not_equal(depth, 0))                     # :130-135
    rmin, rmax, cmin, cmax = lm_get_bbox(obj_bb, H, W)
    choose = mask[rmin:rmax, cmin:cmax].flatten().nonzero()[0]                             # :147
    if len(choose) == 0:
        return None
    img_masked = np.array(img)[:, :, :3][rmin:rmax, cmin:cmax, :].astype(np.float32).reshape((-1, 3))[choose, :]
    img_masked = img_masked / 255.0 - RGB_MEAN[np.newaxis, :]
    depth_masked = depth[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    xmap_masked = xmap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    ymap_masked = ymap[rmin:rmax, cmin:cmax].flatten()[choose][:, np.newaxis].astype(np.float32)
    cam_scale = 1.0                                                                        # :155-160
    pt2 = depth_masked / cam_scale
    pt0 = (ymap_masked - LM_CAM["cx"]) * pt2 / LM_CAM["fx"]
    pt1 = (xmap_masked - LM_CAM["cy"]) * pt2 / LM_CAM["fy"]
    cloud = np.concatenate((pt0, pt1, pt2), axis=1)
    cloud = cloud / 1000.0
    assert cloud.dtype == np.float32
    centroid = np.mean(cloud, axis=0)
    cloud = cloud - centroid[np.newaxis, :]
    inside = (np.abs(cloud[:, 0]) < extent[0] * 0.5) & (np.abs(cloud[:, 1]) < extent[1] * 0.5) & \
             (np.abs(cloud[:, 2]) < extent[2] * 0.5)                                       # :196
    if not (np.sum(inside) > 128 or eval_mode):                                            # :197
        return None
    cloud, img_masked = cloud[inside, :], img_masked[inside, :]
    if cloud.shape[0] > npoint_inp:
        pick = np.random.choice(cloud.shape[0], npoint_inp, replace=False)
    else:
        pick = np.random.choice(cloud.shape[0], npoint_inp)
    cloud_t, rgb_t = torch.FloatTensor(cloud[pick, :]), torch.FloatTensor(img_masked[pick, :])
    feat_inp = torch.cat([torch.ones(npoint_inp, 1), rgb_t, cloud_t], 1)
    vox_inp = ((cloud_t + extent[0] * 0.5) / torch.FloatTensor(unit)).long()
    model_points = torch.FloatTensor(cad_points_mm[obj] / 1000.0)
    model_colors = torch.FloatTensor(cad_colors[obj])
    feat_tmp = torch.cat([torch.ones(cfg["tmp_size"], 1), model_colors, model_points], 1)
    vox_tmp = ((model_points + extent[0] * 0.5) / torch.FloatTensor(unit)).long()
    return feat_inp, vox_inp, feat_tmp, vox_tmp, torch.FloatTensor(centroid)

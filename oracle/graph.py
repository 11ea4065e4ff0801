"""CPU restatement of DCL-Net's forward graph -- TEST INFRASTRUCTURE ONLY.

Functional (state_dict-driven) restatement of the reference's Python graph:
models/DCL_Net.py:155-259 (Network.forward), models/Modules.py (backbone,
point-feature interpolation, Aligner, heads) and models/refiner.py:78-95, on top
of oracle/native.py (the C restatement of the reference's CUDA kernels) and
torch-CPU fp32 for the dense algebra (conv1x1, BN, bmm, softmax, svd) -- the
stand-in for the reference's cuBLAS/cuDNN/MAGMA calls.

Pinned against the reference itself by tests/golden/make_golden.py, which runs
the reference's own models/*.py (imported in the build container) on the same
inputs and weights; see tests/test_oracle_golden.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import native as K

SCALE_LISTS = (2, 4, 6, 8)            # models/DCL_Net.py:54 (sic: true strides are 2,4,8,16)
BN_EPS = 1e-5


def _bn(x, sd, prefix):
    """eval-mode BatchNorm over dim 1 (nn.BatchNorm1d/3d)."""
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, BN_EPS)


def backbone(sd, prefix, feats, indices, spatial_shape, batch_size, trace=None):
    """Backbone_SPCONV.forward (models/Modules.py:153-159): 4 x {conv, subm conv, avg-pool}.
    feats np (V0,7), indices np i32 (V0,4).  Returns [(features np (V_l,C_l), indices np (V_l,4))]*4."""
    x = np.asarray(feats, np.float32)
    idx = np.asarray(indices, np.int32)
    shape = list(spatial_shape)
    levels = []
    for m in range(1, 5):
        for j, subm in ((0, False), (1, True)):
            key = "%s.module%d.%d.layers" % (prefix, m, j)
            W = sd[key + ".0.weight"].numpy()
            if subm:
                outids, pairs, num, oshape = K.get_indice_pairs(idx, batch_size, shape, 3, subm=True)
            else:                                                  # stride is always 1 (Modules.py:138)
                outids, pairs, num, oshape = K.get_indice_pairs(idx, batch_size, shape, 3, 1, 1, 1)
            x = K.indice_conv(x, W, pairs, num, outids.shape[0], subm=subm)
            if outids.shape[0] != 0:                               # spconv/modules.py:125-127
                x = torch.relu(_bn(torch.from_numpy(x), sd, key + ".1")).numpy()
            idx, shape = outids, oshape
            if trace is not None:
                trace["%s.m%d.%s" % (prefix, m, "subm" if subm else "conv")] = (x, idx, pairs, num)
        outids, pairs, num, oshape = K.get_indice_pairs(idx, batch_size, shape, 3, 2, 1, 1)   # Modules.py:151
        x, rf = K.indice_avgpool(x, pairs, num, outids.shape[0])
        idx, shape = outids, oshape
        if trace is not None:
            trace["%s.m%d.pool" % (prefix, m)] = (x, idx, pairs, num, rf)
        levels.append((x, idx))
    return levels


def voxel_centres(indices, unit_voxel_extent, voxel_num_limit, scale):
    """Ops_tensor2points (models/Modules.py:204-211): fp32, left to right."""
    ind = torch.from_numpy(np.asarray(indices, np.int32)).float()
    unit = np.asarray(unit_voxel_extent, np.float64)
    offset = torch.Tensor(-0.5 * unit * np.asarray(voxel_num_limit))      # Modules.py:234
    ve = torch.Tensor(unit * scale)
    ind[:, 1:] = ind[:, 1:] * ve + offset + .5 * ve
    return ind


def point_feats(points_b4, levels, unit_voxel_extent, voxel_num_limit, trace=None, tag=""):
    """Ops_GetPointFeat_spconv.forward (models/Modules.py:236-251) -> (n, 480)."""
    outs = []
    for l, (vf, vi) in enumerate(levels):
        centres = voxel_centres(vi, unit_voxel_extent, voxel_num_limit, SCALE_LISTS[l])
        d2, idx = K.three_nn_sp(points_b4.numpy(), centres.numpy())
        dist = torch.sqrt(torch.from_numpy(d2))                             # pointnet_sp/pointnet2_utils.py:31
        recip = 1.0 / (dist + 1e-8)                                         # Modules.py:222-224
        w = recip / torch.sum(recip, dim=1, keepdim=True)
        outs.append(torch.from_numpy(K.three_interpolate_sp(vf, idx, w.numpy())))
        if trace is not None:
            trace["%snn%d" % (tag, l)] = (d2, idx, w.numpy())
    return torch.cat(outs, dim=1)


def _disengage(sd, name, x):
    """two BasicBlock_3DCONV: 1x1x1 conv (no bias) -> BN3d -> ReLU (Modules.py:58-97)."""
    for j in (0, 1):
        W = sd["%s.%d.layers.0.weight" % (name, j)]
        x = F.conv1d(x, W.reshape(W.shape[0], W.shape[1], 1))
        x = torch.relu(_bn(x, sd, "%s.%d.layers.1" % (name, j)))
    return x


def _head(sd, name, x, acts, bns):
    """Head_MultiLayerPerceptron (Modules.py:173-201): Conv1d -> act -> [BN]."""
    li = 0
    for act, bn in zip(acts, bns):
        x = F.conv1d(x, sd["%s.layers.%d.weight" % (name, li)], sd["%s.layers.%d.bias" % (name, li)])
        li += 1
        if act == "relu":
            x = torch.relu(x); li += 1
        if bn:
            x = _bn(x, sd, "%s.layers.%d" % (name, li)); li += 1
    return x


def ortho9d2matrix(x_raw, y_raw, z_raw):
    """models/DCL_Net.py:15-36 with utils/transform3D.py:6-30."""
    def nrm(v):
        return v / (torch.sqrt(v.pow(2).sum(1, keepdim=True)) + 1e-8)
    M = torch.stack((nrm(x_raw), nrm(y_raw), nrm(z_raw)), dim=2)
    U, S, V = torch.svd(M)
    sigma = torch.ones(M.shape[0], 3)
    sigma[:, -1] = torch.bmm(U, V.transpose(1, 2)).det()
    return U @ torch.diag_embed(sigma) @ V.transpose(1, 2)


RELU3 = ("relu", "relu", "none")


def forward(sd, cfg, data, mode="test", trace=None):
    """Network.forward (models/DCL_Net.py:155-259).  sd: dict name->torch CPU tensor;
    cfg: dict with voxelization_mode, unit_voxel_extent, n_inp, n_tmp; data: the loader dict."""
    n_inp, n_tmp = cfg["n_inp"], cfg["n_tmp"]
    unit = cfg["unit_voxel_extent"]
    vlim = [int(v) for v in np.asarray(data["voxel_num_limit"]).astype(np.int64)]
    b = int(data["batch_offsets"].shape[0]) - 1
    pf = {}
    for side, bb, n in (("inp", "backbone_inp", n_inp), ("tmp", "backbone_tmp", n_tmp)):
        feats = data[side]["feats"].float()
        vox = K.voxelize_fp(feats.numpy(), data[side]["v2p_maps"].numpy(), cfg["voxelization_mode"])
        occ = data[side]["occupied_voxels"].int().numpy()
        levels = backbone(sd, bb, vox, occ, vlim, b, trace)
        pts = feats[:, 4:].reshape(b, n, 3).reshape(-1, 3)
        bid = torch.arange(b).unsqueeze(1).repeat(1, n).view(-1, 1).float()
        pf[side] = point_feats(torch.cat([bid, pts], 1), levels, unit, vlim, trace, side + ".")
        pf[side + "_pts"] = pts.view(b, n, 3)
        if trace is not None:
            trace[side + ".vox"] = vox
            trace[side + ".point_feats"] = pf[side].numpy()
    F_Xc = pf["inp"].view(b, n_inp, -1).transpose(1, 2)
    F_Yo = pf["tmp"].view(b, n_tmp, -1).transpose(1, 2)
    Xc = {k: _disengage(sd, "disengage_Xc_" + k, F_Xc) for k in ("p1", "m1", "p2", "m2")}
    Yo = {k: _disengage(sd, "disengage_Yo_" + k, F_Yo) for k in ("p1", "m1", "p2", "m2")}

    def aligner(RI_1, RI_2, RE_2):                                          # Modules.py:162-169
        A = torch.softmax(torch.bmm(RI_2.transpose(1, 2), RI_1), dim=1)
        return torch.bmm(RE_2, A), A

    F_Xo_p, A1 = aligner(Xc["m1"], Yo["m1"], Yo["p1"])
    F_Yc_p, A2 = aligner(Yo["m2"], Xc["m2"], Xc["p2"])
    F_Xo_m = torch.bmm(Yo["m1"], A1)
    F_Yc_m = torch.bmm(Xc["m2"], A2)
    conf_1 = _head(sd, "regressor_conf", torch.cat([Xc["m1"], F_Xo_m], 1), RELU3, (0, 0, 0))
    conf_2 = _head(sd, "regressor_conf_bi", torch.cat([F_Yc_m, Yo["m2"]], 1), RELU3, (0, 0, 0))
    conf = torch.sigmoid(torch.cat([conf_1, conf_2], dim=2))
    conf_softmax = torch.softmax(conf, dim=2)
    F_p1 = _head(sd, "neck_fuser", torch.cat([Xc["p1"], F_Xo_p], 1), ("relu",) * 3, (1, 1, 1))
    F_p2 = _head(sd, "neck_fuser_bi", torch.cat([F_Yc_p, Yo["p2"]], 1), ("relu",) * 3, (1, 1, 1))
    F_p_wei = torch.sum(torch.cat([F_p1, F_p2], dim=2) * conf_softmax, dim=2, keepdim=True)
    o9 = _head(sd, "regressor_rot", F_p_wei, RELU3, (0, 0, 0)).squeeze(-1)
    rot = ortho9d2matrix(o9[:, :3], o9[:, 3:6], o9[:, 6:])
    trans = _head(sd, "regressor_trans", F_p_wei, RELU3, (0, 0, 0)).squeeze(-1)
    pred = {"trans_pred": trans, "rot_pred": rot, "conf": conf.squeeze(1), "F_Xo_p": F_Xo_p}
    if mode != "test":
        pred["Xo_pred"] = _head(sd, "regressor_Xo", F_Xo_p, RELU3, (0, 0, 0)).transpose(1, 2)
        pred["Yc_pred"] = _head(sd, "regressor_Yc", F_Yc_p, RELU3, (0, 0, 0)).transpose(1, 2)
    if trace is not None:
        trace.update({"ortho9d": o9, "F_p_wei": F_p_wei, "F_Yc_p": F_Yc_p})
    data.setdefault("labels", {})
    data["labels"]["points_tmp"] = pf["tmp_pts"]
    data["labels"]["points_inp"] = pf["inp_pts"]
    return pred


def refiner_forward(sd, input_features, conf):
    """Refiner.forward (models/refiner.py:78-95)."""
    conf_softmax = torch.softmax(conf.unsqueeze(1), dim=2)[:, :, :1024]       # refiner.py:81 (no renorm)
    shared = _head(sd, "MLP_share", input_features, ("relu",) * 3, (0, 0, 0))
    shared = (shared * conf_softmax).sum(dim=2, keepdim=True)
    o9 = _head(sd, "regressor_rot2", shared, RELU3, (0, 0, 0)).squeeze(-1)
    dt = _head(sd, "regressor_trans2", shared, RELU3, (0, 0, 0)).squeeze(-1)
    return {"trans_pred": dt, "rot_pred": ortho9d2matrix(o9[:, :3], o9[:, 3:6], o9[:, 6:])}


def refine_loop(sd_ref, pred, points_inp, iteration=2):
    """tools/test_YCBV_stage2.py:204-225: iterative pose refinement."""
    rot, trans = pred["rot_pred"], pred["trans_pred"]
    F_Xo_p, conf = pred["F_Xo_p"], pred["conf"]
    cur = torch.bmm(points_inp - trans.unsqueeze(1), rot)
    inp = torch.cat([cur.transpose(1, 2), F_Xo_p], dim=1)
    for _ in range(iteration):
        out = refiner_forward(sd_ref, inp, conf)
        trans = (rot @ out["trans_pred"].unsqueeze(2)).squeeze(2) + trans
        rot = rot @ out["rot_pred"]
        cur = torch.bmm(points_inp - trans.unsqueeze(1), rot)
        inp = torch.cat([cur.transpose(1, 2), F_Xo_p], dim=1)
    return rot, trans


# --------------------------------------------------------------------------- metric (a19)
def vocap_auc(dis_list, max_dis=0.1):
    """Literal restatement of cal_auc_acc + VOCap (tools/test_YCBV_stage1.py:83-110)."""
    D = np.array(dis_list, dtype=np.float64)
    n = len(dis_list)
    if n == 0:
        return 0.0, 0.0
    D[np.where(D > max_dis)] = np.inf
    D = np.sort(D)
    acc = np.cumsum(np.ones((1, n)), dtype=np.float32) / n
    ok = np.where(D != np.inf)
    if len(ok[0]) == 0:
        aps = 0.0
    else:
        rec, prec = D[ok], acc[ok]
        mrec = np.array([0.0] + list(rec) + [0.1])
        mpre = np.array([0.0] + list(prec) + [prec[-1]])
        for i in range(1, prec.shape[0]):
            mpre[i] = max(mpre[i], mpre[i - 1])
        i = np.where(mrec[1:] != mrec[0:-1])[0] + 1
        aps = np.sum((mrec[i] - mrec[i - 1]) * mpre[i]) * 10
    acc2 = (D < 0.02).sum() / n
    return aps * 100, acc2 * 100

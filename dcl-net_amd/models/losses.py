"""Training objectives of DCL-Net (`losses`, models/DCL_Net.py:261-318) and of the refiner (`losses_refiner`,
models/refiner.py:98-139) as plain torch modules over the GPU outputs of `Network` / `Refiner`.

Point-set terms: `L2_Dis` = per-point Euclidean distance between corresponding points, `CD_Dis` = symmetric
nearest-neighbour (Chamfer) distance for symmetric objects, from coordinate differences in row chunks."""
import torch
import torch.nn as nn

from .. import ops


def l2_dis(pred, target):
    """(b,n,3) x (b,n,3) -> (b,n) point-to-corresponding-point distance."""
    return torch.norm(pred - target, dim=2)


def cd_dis(pred, target, chunk=256):
    """(b,n,3) x (b,n,3) -> (b,n): 0.5 * (nearest target of every pred point + nearest pred point of every target point).
    The pairwise Euclidean matrix is formed from coordinate differences (as the reference does), `chunk` pred rows at a
    time, so the (b,N,M,3) difference tensor never exists in full."""
    near_t, near_p = [], None
    for s0 in range(0, pred.shape[1], chunk):
        d = torch.norm(pred[:, s0:s0 + chunk].unsqueeze(2) - target.unsqueeze(1), dim=3)      # (b, chunk, M)
        near_t.append(d.min(dim=2)[0])
        m = d.min(dim=1)[0]
        near_p = m if near_p is None else torch.minimum(near_p, m)
    return 0.5 * (torch.cat(near_t, dim=1) + near_p)


def get_cano_label(points_tmp, points_inp, rot_pred, trans_gt):
    """Observed points mapped to the canonical frame and snapped to their nearest template point (kNN, k = 1)."""
    cano = torch.bmm(points_inp - trans_gt, rot_pred)
    _, idx = ops.knn(1, cano.contiguous(), points_tmp.contiguous())
    return torch.gather(points_tmp, 1, idx.long().repeat(1, 1, 3))


class _PoseLoss(nn.Module):
    L2_Dis = staticmethod(l2_dis)
    CD_Dis = staticmethod(cd_dis)
    get_cano_label = staticmethod(get_cano_label)

    @staticmethod
    def _pose_term(posed, posed_gt, sym):
        sym = sym.unsqueeze(1)
        return ((1 - sym) * l2_dis(posed, posed_gt) + sym * cd_dis(posed, posed_gt)).mean(dim=1).mean()


class losses(_PoseLoss):
    """loss_all = loss_pose + 5 loss_Xo + loss_Yc + loss_conf (models/DCL_Net.py:264-304)."""

    def __init__(self, cfg=None):
        super().__init__()

    def forward(self, pred, gt):
        R, t, sym, conf = pred["rot_pred"], pred["trans_pred"], pred["sym_flag"], pred["conf"]
        dev = R.device
        R_gt, t_gt = gt["rot_gt"].to(dev), gt["trans_gt"].to(dev)
        tmp, inp = gt["points_tmp"], gt["points_inp"]
        s1 = sym.unsqueeze(1)
        posed = torch.bmm(tmp, R.transpose(1, 2)) + t.unsqueeze(1)
        posed_gt = torch.bmm(tmp, R_gt.transpose(1, 2)) + t_gt.unsqueeze(1)
        loss_pose = self._pose_term(posed, posed_gt, sym)
        Xo, Yc = pred["Xo_pred"], pred["Yc_pred"]
        inp_cano_pred = torch.bmm(inp - t.unsqueeze(1), R).detach()
        inp_cano_gt = torch.bmm(inp - t_gt.unsqueeze(1), R_gt).detach()
        loss_Xo = (1 - s1) * l2_dis(Xo, inp_cano_gt) + 0.5 * s1 * (cd_dis(Xo, tmp) + l2_dis(Xo, inp_cano_pred))
        loss_Yc = (1 - s1) * l2_dis(Yc, posed_gt) + 0.5 * s1 * (cd_dis(Yc, posed_gt) + l2_dis(Yc, posed.detach()))
        loss_conf = torch.mean(torch.cat([loss_Xo, loss_Yc], dim=1).detach() * conf - 0.01 * torch.log(conf))
        out = {"loss_pose": loss_pose, "loss_Xo": loss_Xo.mean(), "loss_Yc": loss_Yc.mean(), "loss_conf": loss_conf}
        out["loss_all"] = out["loss_pose"] + 5 * out["loss_Xo"] + 1 * out["loss_Yc"] + 1 * out["loss_conf"]
        return out


class losses_refiner(_PoseLoss):
    """Pose loss of the template posed by the refiner's delta and then by the current pose (models/refiner.py:101-125)."""

    def __init__(self, cfg=None):
        super().__init__()

    def forward(self, pred_refiner, trans_cur, rot_cur, points_tmp, sym_flag, gt):
        dR, dt = pred_refiner["rot_pred"], pred_refiner["trans_pred"]
        dev = dR.device
        R_gt, t_gt = gt["rot_gt"].to(dev), gt["trans_gt"].to(dev)
        posed_delta = torch.bmm(points_tmp, dR.transpose(1, 2)) + dt.unsqueeze(1)
        posed_gt = torch.bmm(points_tmp, R_gt.transpose(1, 2)) + t_gt.unsqueeze(1)
        refined = torch.bmm(posed_delta, rot_cur.transpose(1, 2)) + trans_cur.unsqueeze(1)
        loss_pose = self._pose_term(refined, posed_gt, sym_flag)
        return {"loss_pose": loss_pose, "loss_all": loss_pose}

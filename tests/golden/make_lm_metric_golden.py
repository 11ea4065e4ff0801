"""Generates tests/golden/lm_metric_ref.npz with the REFERENCE's own LineMOD metric code: the per-frame body of the eval
loop in tools/test_LM.py (:112-141: pose the template cloud by the predicted and the ground-truth pose, ADD for
non-symmetric / ADD-S for symmetric objects, `dis < diameter[idx]`, success_count / num_count), executed from the
reference source in the build container (nothing is copied): the statements are compiled out of the script's AST (the
script itself needs gorilla / open3d / a dataset to run) and fed seeded poses; `.cuda()` is a no-op.

    python tests/golden/make_lm_metric_golden.py
"""
import ast
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = "/root/reference/tools/test_LM.py"


def loop_body():
    """the statements of `for i, data in enumerate(dataloder)` between `pred = model(data)` and the progress-bar update"""
    tree = ast.parse(open(SRC).read())
    test = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "test"][0]
    loop = [n for w in ast.walk(test) if isinstance(w, ast.With) for n in w.body if isinstance(n, ast.For)][0]
    keep, on = [], False
    for st in loop.body:
        src = ast.unparse(st)
        if src.startswith("points_tmp ="):
            on = True
        if src.startswith("t.set_description"):
            break
        if on:
            keep.append(st)
    return compile(ast.Module(body=keep, type_ignores=[]), SRC, "exec")


def rand_rot(rng, angle_scale=None):
    if angle_scale is None:
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        return (q * np.sign(np.linalg.det(q))).astype(np.float32)
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    a = rng.normal() * angle_scale
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return (np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K).astype(np.float32)


class _Sink(object):
    def write(self, *_):
        pass


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self
    code = loop_body()
    rng = np.random.default_rng(11)
    n_obj, P, frames = 13, 160, 60
    diameter = (rng.uniform(0.1, 0.3, n_obj) * 0.1).tolist()              # metres * 0.1, as tools/test_LM.py:68-75 forms them

    class Cfg(object):
        pass
    cfg = Cfg()
    cfg.diameter, cfg.num_objects = diameter, n_obj
    ns = {"torch": torch, "np": np, "cfg": cfg, "fw": _Sink(), "count": 0,
          "success_count": [0] * n_obj, "num_count": [0] * n_obj}
    out = {"diameter": np.array(diameter, np.float64), "n_frames": np.array([frames])}
    clouds = (rng.normal(size=(n_obj, P, 3)) * rng.uniform(0.02, 0.06, (n_obj, 1, 1))).astype(np.float32)
    out["clouds"] = clouds
    for f in range(frames):
        nb = int(rng.integers(1, 4))
        flags = rng.choice([0, 1, -1], size=nb, p=[0.55, 0.35, 0.10])
        valid = flags != -1
        nv = int(valid.sum())
        if nv == 0:                                                   # the reference skips such a frame before model(data)
            flags[0], valid[0], nv = 0, True, 1
        idx = rng.integers(0, n_obj, nv)
        Rg = np.stack([rand_rot(rng) for _ in range(nv)])
        tg = rng.normal(0, 0.3, (nv, 3)).astype(np.float32)
        scale = rng.choice([0.01, 0.05, 0.2, 1.0], size=nv)
        Rp = np.stack([rand_rot(rng, s) @ R for s, R in zip(scale, Rg)]).astype(np.float32)
        tp = (tg + rng.normal(0, 0.004, (nv, 3)) * scale[:, None] * 5).astype(np.float32)
        ns["data"] = {"labels": {"points_tmp": torch.from_numpy(clouds[idx]), "rot_gt": torch.from_numpy(Rg),
                                 "trans_gt": torch.from_numpy(tg)},
                      "flags": torch.from_numpy(flags.astype(np.int64)), "obj_idx": torch.from_numpy(idx.astype(np.int64))}
        ns["pred"] = {"rot_pred": torch.from_numpy(Rp), "trans_pred": torch.from_numpy(tp)}
        exec(code, ns)
        out["f%d_flags" % f], out["f%d_idx" % f] = flags.astype(np.int32), idx.astype(np.int32)
        out["f%d_Rp" % f], out["f%d_tp" % f], out["f%d_Rg" % f], out["f%d_tg" % f] = Rp, tp, Rg, tg
        out["f%d_l2" % f], out["f%d_cd" % f] = ns["l2_dis"].numpy(), ns["cd_dis"].numpy()
    out["success_count"] = np.array(ns["success_count"], np.int64)
    out["num_count"] = np.array(ns["num_count"], np.int64)
    out["count"] = np.array([ns["count"]], np.int64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lm_metric_ref.npz"), **out)
    print("golden written: lm_metric_ref.npz  success", out["success_count"].tolist(), "of", out["num_count"].tolist(),
          "frames counted", int(out["count"][0]))


if __name__ == "__main__":
    main()

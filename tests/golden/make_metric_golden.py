"""Generates tests/golden/metric_ref.npz with the REFERENCE's own metric functions (tools/test_YCBV_stage1.py:83-125:
VOCap, cal_dis_acc, cal_auc_acc, cal_metric_auc_acc), executed from the reference source in the build container (nothing is
copied): the functions are compiled out of the script's AST (the script itself needs open3d / gorilla to import).

    python tests/golden/make_metric_golden.py
"""
import ast
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = "/root/reference/tools/test_YCBV_stage1.py"
WANT = ("VOCap", "cal_dis_acc", "cal_auc_acc", "cal_metric_auc_acc")


class _Log(object):
    def warning(self, *a, **k):
        pass


def main():
    tree = ast.parse(open(SRC).read())
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT], type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, SRC, "exec"), ns)
    out = {}
    for case in range(6):
        rng = np.random.default_rng(case)
        n = int(rng.integers(60, 400))
        idx = rng.integers(0, 21, n)
        idx[:21] = np.arange(21)                                   # every class present
        d = np.abs(rng.normal(0.02, 0.03, n))
        d[rng.random(n) < 0.08] = np.inf                           # missed detections (:192-194)
        if case == 5:
            d[idx == 3] = np.inf                                   # a class with no valid distance at all
        per_auc, per_acc = [], []
        for c in range(21):
            a, acc = ns["cal_auc_acc"](list(d[idx == c]))
            per_auc.append(a)
            per_acc.append(acc)
        mean_auc = ns["cal_metric_auc_acc"](list(d), list(idx), _Log())
        out["d%d" % case], out["idx%d" % case] = d, idx
        out["auc%d" % case], out["acc%d" % case] = np.array(per_auc, np.float64), np.array(per_acc, np.float64)
        out["mean_auc%d" % case] = np.array([mean_auc], np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metric_ref.npz"), **out)
    print("golden written: metric_ref.npz", [float(out["mean_auc%d" % c][0]) for c in range(6)])


if __name__ == "__main__":
    main()

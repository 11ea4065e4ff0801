"""A/B of the feature stage as ONE launch (Network(feature_stage=True)) against the per-layer launches, whole forward, same
process: ms per call at several batch sizes (N = M = 1024), launch by launch and as graph replay.
usage: python tools/stage_ab.py [b ...]"""
import copy
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dcl = importlib.import_module("dcl-net_amd")


def bench(net, data, reps):
    with torch.no_grad():
        for _ in range(3):
            net(data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            net(data)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    bs = [int(a) for a in sys.argv[1:]] or [1, 2, 6, 16, 32, 40]
    n = int(os.environ.get("N", "1024"))
    cfg = dcl.synth.default_cfg(n, n)
    variants = [("per-layer (default)", {}), ("per-layer pair", dict(pair_features=True)),
                ("stage s512", dict(feature_stage=True, stage_slots=512)), ("stage s384", dict(feature_stage=True, stage_slots=384))]
    for graph in (1, 0):
        for b in bs:
            data = dcl.synth.make_batch(b, n, n)
            dev = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) and k in ("inp", "tmp") else v) for k, v in data.items()}
            line = []
            for name, kw in variants:
                net = dcl.DCL_Net.Network(cfg, mode="test", graph_max_batch=(64 if graph else 0), **kw)
                net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
                net = net.cuda().eval()
                ms = bench(net, dev, 30 if b <= 8 else 15)
                ok = net.check_feature_stage()
                line.append("%s %.3f%s" % (name, ms, "" if ok else " TIMEOUT"))
                del net
            print("b=%d %s: %s" % (b, "graph" if graph else "eager", " | ".join(line)), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Every batch size b0..b1 of N = M = 1024 crops once through the default Network (whole-forward graph, two branches) and once
launch by launch, each under a watchdog: a batch size whose GEMM / attention / conv launch plans do not tile the chip evenly is
where a scheduling hazard would show (33 crops did, before csrc/linear.cpp stopped taking workspace-exchanging GEMM algorithms).
usage: tools/batch_sweep.py [b0 b1 [n_inp n_tmp]]"""
import faulthandler, importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
b0, b1 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 48)
n_inp, n_tmp = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1024, 1024)
dev = torch.device("cuda:0")
cfg = dcl.synth.default_cfg(n_inp, n_tmp)
nets = {}
for name, kw in (("default (graph up to 98304 points)" if n_inp != 1024 else "graph", dict() if n_inp != 1024 else dict(graph_max_batch=1 << 20)),
                 ("launch by launch", dict(graph_max_batch=0, graph_max_points=0))):
    net = dcl.DCL_Net.Network(cfg, mode="test", **kw)
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
    nets[name] = net.to(dev).eval()
for b in range(b0, b1 + 1):
    data = bench.to_device(dcl.synth.make_batch(b, n_inp, n_tmp), dev)
    line = []
    for name, net in nets.items():
        faulthandler.dump_traceback_later(60, exit=True)
        with torch.no_grad():
            for _ in range(2):
                out = net(data)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                out = net(data)
            torch.cuda.synchronize()
        faulthandler.cancel_dump_traceback_later()
        ok = bool(torch.isfinite(out["trans_pred"]).all()) and bool(torch.isfinite(out["rot_pred"]).all())
        line.append("%s %.3f ms%s" % (name, (time.perf_counter() - t0) / 5 * 1e3, "" if ok else " NON-FINITE"))
        if hasattr(net, "_invalidate"):
            net._invalidate()
    print("b=%2d: %s" % (b, "; ".join(line)), flush=True)
print("sweep done")

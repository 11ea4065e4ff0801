#!/usr/bin/env python3
"""Do the two per-side branches of the captured forward really overlap?  Replays the whole-forward hipGraph of b crops as
captured by default (two branches) and with Network(single_stream=True) (one chain).  usage: graph_branches.py [b ...]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
dev = torch.device("cuda:0")
if os.environ.get("DCL_DIAG"):                     # A/B of kernel switches (DCL_CONV_* etc., tools/_diag.py): diagnostic library
    from _diag import use_diag
    use_diag(dcl)
CONFIGS = ({},) if os.environ.get("DCL_DIAG") else ({}, {"single_stream": True}, {"single_stream": True, "pair_features": False})
for b in [int(x) for x in sys.argv[1:]] or [1, 6]:
    data = bench.to_device(dcl.synth.make_batch(b, 1024, 1024, unit=0.005), dev)
    for kw in CONFIGS:
        net = dcl.DCL_Net.Network(dcl.synth.default_cfg(1024, 1024, unit=0.005), mode="test", **kw)
        net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
        net = net.to(dev).eval()
        with torch.no_grad():
            for _ in range(3):
                net.forward_graphed(data)
            ent = next(iter(net._graphs.values()))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                ent["graph"].replay()
            torch.cuda.synchronize()
            bare = (time.perf_counter() - t0) / 200 * 1e3
            t0 = time.perf_counter()
            for _ in range(200):
                net.forward_graphed(data)
            torch.cuda.synchronize()
            full = (time.perf_counter() - t0) / 200 * 1e3
        print("b=%d %-50s bare replay %.3f ms, call %.3f ms, nodes %s" % (b, kw or "default (two branches)", bare, full, ent.get("nodes")), flush=True)

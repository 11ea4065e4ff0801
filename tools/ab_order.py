#!/usr/bin/env python3
"""A/B of the row ordering on the whole forward (diagnostic library): ms per bs-32 step at N = M = 1024 with the ordering on
(default) and off, alternating, launch by launch and as graph replays."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
L = use_diag(dcl)
dev = torch.device("cuda:0")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = dcl.synth.default_cfg(1024, 1024)
data = bench.to_device(dcl.synth.make_batch(b, 1024, 1024), dev)
def make(graph):
    net = dcl.DCL_Net.Network(cfg, mode="test", **({} if graph else {"graph_max_batch": 0}))
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
    return net.to(dev).eval()
def run(net, steps=40):
    with torch.no_grad():
        for _ in range(5): net(data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): net(data)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for graph in (False, True):
    res = {"on": [], "off": []}
    for rep in range(3):
        for name, mb in (("on", 12), ("off", 1 << 30)):
            L.dcl_debug_order_min_batch(mb)
            net = make(graph)                       # (a captured graph has the choice baked in: new instance per setting)
            res[name].append(run(net))
            del net
    L.dcl_debug_order_min_batch(12)
    print("%s: ordering on %s ms, off %s ms per step" % ("graph replay" if graph else "launch by launch",
          ["%.3f" % v for v in res["on"]], ["%.3f" % v for v in res["off"]]), flush=True)

#!/usr/bin/env python3
"""What really overlaps in an UNPROFILED replay of the whole-forward hipGraph: one-thread stamp launches (100 MHz wall clock,
diagnostic library) behind every stage of the two branches.  usage: graph_timeline.py [b ...]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
use_diag(dcl)
dev = torch.device("cuda:0")
for b in [int(x) for x in sys.argv[1:]] or [32]:
    data = bench.to_device(dcl.synth.make_batch(b, 1024, 1024), dev)
    net = dcl.DCL_Net.Network(dcl.synth.default_cfg(1024, 1024), mode="test", graph_max_batch=64)
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
    net = net.to(dev).eval()
    buf = torch.zeros(64, dtype=torch.int64, device=dev)
    names = []

    def stage_done(name, stream, buf=buf, names=names):          # replaces Network._stage_done on this instance
        if name not in names:
            names.append(name)
        with torch.cuda.stream(stream):
            dcl.ops.N.check(dcl.ops.N.lib().dcl_debug_stamp(dcl.ops.C.c_void_p(buf[names.index(name):].data_ptr()), dcl.ops.N.stream()), "stamp")
    net._stage_done = stage_done
    with torch.no_grad():
        for _ in range(4):
            net.forward_graphed(data)
        ent = next(iter(net._graphs.values()))
        torch.cuda.synchronize()
        for _ in range(5):
            ent["graph"].replay()
        torch.cuda.synchronize()
    t = buf.cpu().numpy()[:len(names)].astype(np.int64)
    t0 = t[0]
    print("b=%d%s (stamps of the last replay, us after the first):" % (b, ""))
    for nm, x in sorted(zip(names, t), key=lambda p: p[1]):
        print("  %9.1f  %s" % ((x - t0) * 0.01, nm))

#!/usr/bin/env python3
"""The whole-forward graph of 32 reference-shape crops, captured several times in one process: replay time of every capture and the
stage stamps inside it (one-thread stamp launches behind every stage of the two branches, diagnostic library) -- does a capture
replay in a different mode than another one, and if so which stage differs?  usage: tools/recapture_probe.py [captures] [crops] [GRAPH_TRIES]"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
use_diag(dcl)
dev = torch.device("cuda:0")
if os.environ.get("DCL_PROBE_PRIO") is not None:               # side-stream priority (0 = normal) for this probe
    dcl.DCL_Net.Network.SIDE_STREAM_PRIORITY = int(os.environ["DCL_PROBE_PRIO"])
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
tries = int(sys.argv[3]) if len(sys.argv) > 3 else None
data = bench.to_device(dcl.synth.make_batch(b, 1024, 1024), dev)
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(1024, 1024), mode="test", graph_max_batch=64)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.to(dev).eval()
if tries is not None:
    net.GRAPH_TRIES = tries                                   # 0 = take every capture as it comes
buf = torch.zeros(64, dtype=torch.int64, device=dev)
names = []


def stage_done(name, stream):
    if name not in names:
        names.append(name)
    with torch.cuda.stream(stream):
        dcl.ops.N.check(dcl.ops.N.lib().dcl_debug_stamp(dcl.ops.C.c_void_p(buf[names.index(name):].data_ptr()), dcl.ops.N.stream()), "stamp")


net._stage_done = stage_done
for cap in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    with torch.no_grad():
        for _ in range(4):
            net.forward_graphed(data)
        ent = next(iter(net._graphs.values()))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            ent["graph"].replay()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 40 * 1e3
    t = buf.cpu().numpy()[:len(names)].astype(np.int64)
    order = sorted(zip(names, t), key=lambda p: p[1])
    print("capture %d (tries %s): replay %.3f ms;  " % (cap, ent.get("capture_ms"), ms) + "  ".join("%s %.0f" % (nm.replace(" done", "").replace("stage ", "s"), (x - order[0][1]) * 0.01) for nm, x in order), flush=True)
    net._invalidate()

#!/usr/bin/env python3
"""Small-batch latency of Network.forward: eager vs whole-forward hipGraph (the reference's eval loop feeds one image =
a few crops per call)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
n = 1024
cfg = dcl.synth.default_cfg(n, n)
net = dcl.DCL_Net.Network(cfg, mode="test", graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
for b in (1, 4, 8, 32):
    data = dcl.synth.make_batch(b, n, n)
    res = {}
    for name, fn in (("eager", lambda: net(data)), ("hipgraph", lambda: net.forward_graphed(data))):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 20 * 1e3
    print("b=%d N=M=%d: eager %.3f ms, hipgraph %.3f ms per forward (host data dict -> pose)" % (b, n, res["eager"], res["hipgraph"]), flush=True)

# breakdown at b=1: bare graph replay (GPU time of the captured forward) vs staging of the host dict
b = 1
data = dcl.synth.make_batch(b, n, n)
net.forward_graphed(data)
ent = net._graphs[(b, n, n, 64)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    ent["graph"].replay()
torch.cuda.synchronize()
print("b=1 bare replay: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
gdata = {k: ({kk: vv.cuda() if torch.is_tensor(vv) else vv for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in data.items()}
for name, d in (("host dict", data), ("device dict", gdata)):
    for fn_name, fn in (("eager", lambda: net(d)), ("hipgraph", lambda: net.forward_graphed(d))):
        with torch.no_grad():
            for _ in range(3): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): fn()
            torch.cuda.synchronize()
        print("b=1 %s %s: %.3f ms" % (name, fn_name, (time.perf_counter() - t0) / 30 * 1e3), flush=True)

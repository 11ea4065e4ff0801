#!/usr/bin/env python3
"""Does a live whole-forward hipGraph (bs 32) slow down the launch-by-launch paths of other Network instances of the
process?  (No: 4.09 / 4.11 / 4.10 ms for pipelined eager calls before / beside / after it.)  usage: pipe_probe.py [1]"""
import importlib, sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dcl = importlib.import_module("dcl-net_amd")
dev = torch.device("cuda")
cfg = dcl.synth.default_cfg(1024, 1024)
data = bench.to_device(dcl.synth.make_batch(32, 1024, 1024), dev)
def mk(**kw):
    net = dcl.DCL_Net.Network(cfg, mode="test", **kw)
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1)); return net.cuda().eval()
def t(net, steps=30):
    dt, _ = bench.run_forward_bench(dcl, net, data, steps, 5, False)
    return dt / steps * 1e3
a = mk(async_inputs=True)
print("async eager, fresh process: %.3f ms" % t(a))
if len(sys.argv) > 1:
    g = mk()
    print("default (graph): %.3f ms" % t(g))
    print("async eager after a b=32 graph exists: %.3f ms" % t(a))
    del g
    torch.cuda.empty_cache()
    print("async eager after the graph is deleted: %.3f ms" % t(a))
    l = mk(graph_max_batch=0)
    print("serial eager: %.3f ms" % t(l))

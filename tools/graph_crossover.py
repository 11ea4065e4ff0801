#!/usr/bin/env python3
"""Launch-by-launch forward vs whole-forward hipGraph replay per batch size: where the default routing of
Network.forward (graph_max_batch / graph_max_points) comes from.  usage: graph_crossover.py [N M [b1,b2,...]]"""
import importlib, sys, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dcl = importlib.import_module("dcl-net_amd")
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (12288, 2048)
bs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [32]
cfg = dcl.synth.default_cfg(n, m)
nets = {}
for name, g in (("eager", 0), ("graph", 1024)):
    net = dcl.DCL_Net.Network(cfg, mode="test", graph_max_batch=g)
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1)); nets[name] = net.cuda().eval()
for b in bs:
    data = bench.to_device(dcl.synth.make_batch(b, n, m), torch.device("cuda"))
    res = {}
    for name, net in nets.items():
        t0 = time.perf_counter()
        for _ in range(3): net(data)
        torch.cuda.synchronize(); first = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(20): out = net(data)
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / 20 * 1e3
        print("  %s: first 3 calls %.2f s, reserved %d MiB" % (name, first, torch.cuda.memory_reserved() >> 20), flush=True)
    print("N=%d M=%d b=%2d: eager %.3f ms, graph %.3f ms per call" % (n, m, b, res["eager"], res["graph"]), flush=True)

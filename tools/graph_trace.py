"""One whole-forward graph replay under `rocprofv3 --kernel-trace`: the kernels of the LAST replay as a timeline (start / end
relative to the replay's first kernel, queue, name) -- where the branches of the captured forward really overlap.
usage: rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python3 tools/graph_trace.py [b]
       python3 tools/graph_trace.py --read DIR/.../t_kernel_trace.csv"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read(path):
    import csv
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last replay: everything after the last long pause (> 2 ms) between kernels
    cut = 0
    for i in range(1, len(rows)):
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 2000000:
            cut = i
    rows = rows[cut:]
    t0 = int(rows[0]["Start_Timestamp"])
    queues = {}
    busy_until, idle = t0, 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = queues.setdefault(r["Queue_Id"], len(queues))
        if s > busy_until:
            idle += s - busy_until
        busy_until = max(busy_until, e)
        print("%8.1f %8.1f %7.1f q%d %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, r["Kernel_Name"][:90]))
    print("replay: %d kernels, %.1f us first start to last end, %.1f us with no kernel running" % (
        len(rows), (busy_until - t0) / 1e3, idle / 1e3))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        return read(sys.argv[2])
    import time
    import torch
    dcl = importlib.import_module("dcl-net_amd")
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    n = 1024
    cfg = dcl.synth.default_cfg(n, n)
    net = dcl.DCL_Net.Network(cfg, mode="test", graph_max_batch=64)
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
    net = net.cuda().eval()
    data = dcl.synth.make_batch(b, n, n)
    dev = {k: ({kk: vv.cuda() for kk, vv in v.items()} if isinstance(v, dict) and k in ("inp", "tmp") else v) for k, v in data.items()}
    with torch.no_grad():
        for _ in range(4):
            net(dev)
            torch.cuda.synchronize()
            time.sleep(0.01)


if __name__ == "__main__":
    main()

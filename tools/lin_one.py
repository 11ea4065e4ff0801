#!/usr/bin/env python3
"""One linear-layer shape, one implementation, a few launches -- the subject of a rocprofv3 pass.
usage: tools/lin_one.py M K N own|lib|pool [reps] [tile]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
dcl = importlib.import_module("dcl-net_amd")
M, K, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
which = sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
if len(sys.argv) > 6:
    from _diag import use_diag
    L = use_diag(dcl)
    L.dcl_debug_linear_tile(int(sys.argv[6]))

g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(M, K, device="cuda", generator=g)
Wt = torch.randn(K, n, device="cuda", generator=g) * 0.05
bias = torch.randn(n, device="cuda", generator=g)
y = torch.empty(M, n, device="cuda")
w = torch.rand(M, device="cuda", generator=g)
fn = {"own": lambda: dcl.ops.linear_dma(x, Wt, bias, True, out=y), "lib": lambda: dcl.ops.linear_lt(x, Wt, bias, True, out=y),
      "pool": lambda: dcl.ops.linear_pool(x, Wt, bias, w)}[which]
for _ in range(reps):
    fn()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    fn()
e.record(); torch.cuda.synchronize()
t = a.elapsed_time(e) / reps * 1e3
print("%s M=%d K=%d N=%d: %.1f us  %.1f TF/s" % (which, M, K, n, t, 2.0 * M * K * n / t / 1e6))

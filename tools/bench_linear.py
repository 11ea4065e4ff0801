#!/usr/bin/env python3
"""The per-point linear layers at the bench shapes: torch._addmm_activation (torch's hipBLASLt call) vs dcl_linear_fwd (the
same library through the C-ABI, free leading dimensions) -- us per call and TFLOP/s, dense and column-block outputs."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
for M, K, n, ldx, ldy in ((393216, 480, 1024, 480, 1024), (393216, 256, 256, 1024, 256), (393216, 256, 256, 1024, 512),
                          (393216, 256, 64, 1024, 128), (458752, 512, 512, 512, 512), (458752, 512, 1024, 512, 1024),
                          (32768, 480, 1024, 480, 1024), (32768, 256, 256, 1024, 512), (65536, 512, 512, 512, 512),
                          (1024, 480, 1024, 480, 1024), (1024, 256, 256, 1024, 512), (2048, 512, 512, 512, 512), (2048, 128, 128, 128, 128)):
    xw = torch.randn(M, ldx, device="cuda")
    x = xw[:, :K]
    Wt = torch.randn(K, n, device="cuda") * 0.05
    bias = torch.randn(n, device="cuda")
    yw = torch.empty(M, ldy, device="cuda")
    y = yw[:, ldy - n:]
    yc = torch.empty(M, n, device="cuda")
    t_torch = timeit(lambda: torch._addmm_activation(bias, x, Wt, out=yc))
    t_lin = timeit(lambda: dcl.ops.linear(x, Wt, bias, True, out=yc))
    t_blk = timeit(lambda: dcl.ops.linear(x, Wt, bias, True, out=y))
    fl = 2.0 * M * K * n
    print("M=%6d K=%4d N=%4d ldx=%4d ldy=%4d: torch %8.1f us (%5.1f TF)  dcl dense out %8.1f us (%5.1f TF)  dcl column-block out %8.1f us (%5.1f TF)" %
          (M, K, n, ldx, ldy, t_torch, fl / t_torch / 1e6, t_lin, fl / t_lin / 1e6, t_blk, fl / t_blk / 1e6), flush=True)

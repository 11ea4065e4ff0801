#!/usr/bin/env python3
"""The own GEMM cores (fp32 MFMA: dcl_linear_dma_fwd, csrc/linear_dma.hip; split-bf16: dcl_linear_split_fwd, csrc/linear_split.hip)
against the vendor library (dcl_linear_fwd ->
hipBLASLt) on the linear layers of a forward: us per call and TFLOP/s per layer shape, at the stress shape (32 crops of 12288 /
2048 points), the reference shape (32 x 1024) and a handful of crops.  usage: tools/bench_linear_dma.py [--tiles] [--check]
  --tiles  also every tile shape of the own core (diagnostic library)   --check  compare both with a float64 product"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
dcl = importlib.import_module("dcl-net_amd")
TILES = "--tiles" in sys.argv or "--whatif" in sys.argv
CHECK = "--check" in sys.argv
L = None
if TILES:
    from _diag import use_diag
    L = use_diag(dcl)


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3


SHAPES = [  # M, K, N, ldx, ldy
    (393216, 480, 1024, 480, 1024), (393216, 256, 256, 1024, 512), (393216, 256, 64, 1024, 128), (393216, 512, 512, 512, 512),
    (393216, 512, 1024, 512, 1024), (65536, 480, 1024, 480, 1024), (65536, 256, 256, 1024, 512), (65536, 256, 64, 1024, 128),
    (65536, 512, 512, 512, 512), (65536, 512, 1024, 512, 1024), (32768, 480, 1024, 480, 1024), (32768, 256, 256, 1024, 512),
    (32768, 256, 64, 1024, 128), (32768, 512, 512, 512, 512), (32768, 512, 1024, 512, 1024), (32768, 256, 512, 256, 512),
    (6144, 480, 1024, 480, 1024), (6144, 512, 512, 512, 512), (6144, 512, 1024, 512, 1024), (6144, 256, 256, 1024, 512),
    (1024, 480, 1024, 480, 1024), (1024, 512, 512, 512, 512), (1024, 256, 64, 1024, 128)]
if "--quick" in sys.argv:
    SHAPES = SHAPES[:5] + SHAPES[10:15]
for M, K, n, ldx, ldy in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(M + K + n)
    xw = torch.randn(M, ldx, device="cuda", generator=g)
    x = xw[:, :K]
    Wt = torch.randn(K, n, device="cuda", generator=g) * 0.05
    bias = torch.randn(n, device="cuda", generator=g)
    yw = torch.empty(M, ldy, device="cuda")
    y = yw[:, ldy - n:]
    fl = 2.0 * M * K * n
    t_lib = timeit(lambda: dcl.ops.linear_lt(x, Wt, bias, True, out=y))
    t_own = timeit(lambda: dcl.ops.linear_dma(x, Wt, bias, True, out=y))
    sw = dcl.ops.SplitWeight(Wt)
    t_sp = timeit(lambda: dcl.ops.linear_split(x, sw, bias, True, out=y))
    line = "M=%6d K=%4d N=%4d: hipBLASLt %8.1f us (%5.1f TF)  own fp32-MFMA %8.1f us (%5.1f TF)  own split-bf16 %8.1f us (%5.1f TF)  own/lib %.3f" % (
        M, K, n, t_lib, fl / t_lib / 1e6, t_own, fl / t_own / 1e6, t_sp, fl / t_sp / 1e6, min(t_own, t_sp) / t_lib)
    if n == 1024 and K == 512:                            # the last fuser layer: the pooling epilogue instead of the store
        w = torch.rand(M, device="cuda", generator=g)
        t_pool = timeit(lambda: dcl.ops.linear_pool(x, Wt, bias, w))
        t_pool_sp = timeit(lambda: dcl.ops.linear_split_pool(x, sw, bias, w))
        line += "  pool-epilogue %8.1f us (%5.1f TF), split-bf16 %8.1f us (%5.1f TF)" % (t_pool, fl / t_pool / 1e6, t_pool_sp, fl / t_pool_sp / 1e6)
    if "--tiles" in sys.argv:
        for t, name in ((1, "128x128"), (2, "128x64"), (3, "64x64"), (4, "64x64 K/2")):
            L.dcl_debug_linear_tile(t)
            tt = timeit(lambda: dcl.ops.linear_dma(x, Wt, bias, True, out=y))
            line += "  %s %.1f" % (name, fl / tt / 1e6)
        L.dcl_debug_linear_tile(0)
    if "--whatif" in sys.argv and L is not None and M >= 32768 and n >= 256:
        for bits, what in ((1, "no DMA"), (2, "no split"), (4, "no stores"), (7, "none of the three")):
            L.dcl_debug_linear_split_whatif(bits)
            tt = timeit(lambda: dcl.ops.linear_split(x, sw, bias, True, out=y))
            line += "  [%s %.1f]" % (what, fl / tt / 1e6)
        L.dcl_debug_linear_split_whatif(0)
    if CHECK:
        rows = torch.randint(0, M, (256,), device="cuda")
        want = torch.relu(x[rows].double() @ Wt.double() + bias.double())
        dcl.ops.linear_dma(x, Wt, bias, True, out=y)
        e_own = float((y[rows].double() - want).abs().max())
        dcl.ops.linear_lt(x, Wt, bias, True, out=y)
        e_lib = float((y[rows].double() - want).abs().max())
        dcl.ops.linear_split(x, sw, bias, True, out=y)
        e_sp = float((y[rows].double() - want).abs().max())
        line += "  |err| fp32-MFMA %.2e split-bf16 %.2e lib %.2e" % (e_own, e_sp, e_lib)
    print(line, flush=True)

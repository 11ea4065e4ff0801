#!/usr/bin/env python3
"""Both directions of the correspondence attention alone, at a given shape (default: stress, b=32, N=12288, M=2048):
time per launch and TFLOP/s of 2*(64+320)*nq*nk*b.  usage: bench_attention.py [b] [N] [M] [reps]
(DCL_ATTN_XCD=0 switches the XCD-aware workgroup numbering off; run under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE`
for the traffic of the two numberings.)"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
DIAG = use_diag(dcl)          # kernel-variant hooks exist in the diagnostic library only
ops = dcl.ops
args = [x for x in sys.argv[1:] if not x.startswith('--')]
b = int(args[0]) if len(args) > 0 else 32
n = int(args[1]) if len(args) > 1 else 12288
m = int(args[2]) if len(args) > 2 else 2048
reps = int(args[3]) if len(args) > 3 else 10
g = torch.Generator(device="cuda").manual_seed(1)
def rnd(rows, c, s=1.0):
    return torch.randn(rows, c, device="cuda", generator=g) * s
Xm, Ym = rnd(b * n, 64, 0.3), rnd(b * m, 64, 0.3)
Xp, Yp = rnd(b * n, 256), rnd(b * m, 256)
for name, (Q, K, V1, V2) in (("N->M", (Xm, Ym, Yp, Ym)), ("M->N", (Ym, Xm, Xp, Xm))):
    nq, nk = Q.shape[0] // b, K.shape[0] // b
    for bf16, form in ((1, "split-bf16 P.V (incl. the V piece pass)"), (0, "fp32 MFMA")):
        DIAG.dcl_debug_attention_bf16(bf16)
        O1 = torch.empty(b * nq, 256, device="cuda"); O2 = torch.empty(b * nq, 64, device="cuda")
        ops.cross_attention(b, Q, K, V1, O1, V2, O2); torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            ops.cross_attention(b, Q, K, V1, O1, V2, O2)
        e.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(e) / reps
        print("%s b=%d nq=%d nk=%d  %-42s %.3f ms  %.1f TFLOP/s" % (name, b, nq, nk, form + ":", ms, 2.0 * 384 * nq * nk * b / ms / 1e9), flush=True)
    DIAG.dcl_debug_attention_bf16(1)
    if "--whatif" in sys.argv:
        for bits, what in ((1, "no P.V phase"), (2, "no S phase"), (4, "no DMA after the first tile"), (5, "no P.V, no DMA"), (6, "no S, no DMA"), (3, "neither phase")):
            DIAG.dcl_debug_attention_whatif(bits)
            ops.cross_attention(b, Q, K, V1, O1, V2, O2); torch.cuda.synchronize()
            a.record()
            for _ in range(reps):
                ops.cross_attention(b, Q, K, V1, O1, V2, O2)
            e.record(); torch.cuda.synchronize()
            print("    what-if %-28s %.3f ms" % (what + ":", a.elapsed_time(e) / reps), flush=True)
        DIAG.dcl_debug_attention_whatif(0)

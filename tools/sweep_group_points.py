#!/usr/bin/env python3
"""Sweep the group_points launch configuration at the north-star primitive shape (B=32, C=64, N=12288, np=2048, ns=64)."""
import importlib, os, sys, itertools
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
DIAG = use_diag(dcl)          # kernel-variant hooks exist in the diagnostic library only
lib = dcl.ops.N.lib()
B, N, NP, NS, C = 32, 12288, 2048, 64, 64
feats = torch.randn(B, C, N, device="cuda")
idx = torch.randint(0, N, (B, NP, NS), device="cuda", dtype=torch.int32)
nbytes = 4 * B * C * N + 4 * B * NP * NS + 4 * B * C * NP * NS
ref = None
for cc, xb, thr, nt in itertools.product((0, 1, 2, 3), (0, 1, 2), (0, 512), (0, 2)):
    lib.dcl_debug_group_points_cfg(cc, xb, thr, nt)
    out = dcl.ops.group_points(feats, idx)
    if ref is None: ref = out.clone()
    assert torch.equal(out, ref)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): dcl.ops.group_points(feats, idx)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("cc=%d xb=%d thr=%4d nt=%d: %.4f ms %.0f GB/s" % (cc, xb, thr, nt, ms, nbytes / ms / 1e6), flush=True)

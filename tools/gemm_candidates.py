#!/usr/bin/env python3
"""How much do the GEMM library's other heuristic candidates differ from the one dcl_linear_fwd takes, for the linear layers of a
forward at b crops of 1024 points?  (diagnostic library)  usage: tools/gemm_candidates.py [b ...]"""
import ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
L = use_diag(dcl)
torch.zeros(1, device="cuda")
shapes = [("disengage 480->1024", 480, 1024), ("disengage 256->256", 256, 256), ("disengage 256->64", 256, 64),
          ("fuser 512->512", 512, 512), ("fuser 512->1024", 512, 1024), ("conf 128->128", 128, 128)]
for b in [int(a) for a in sys.argv[1:]] or [1, 6, 32]:
    M = b * 1024
    for name, K, N in shapes:
        ms = np.zeros(32, np.float32)
        wsz = np.zeros(32, np.int64)
        found = ctypes.c_int(0)
        rc = L.dcl_debug_linear_candidates(M, N, K, 32, ms.ctypes.data_as(ctypes.c_void_p), wsz.ctypes.data_as(ctypes.c_void_p),
                                           ctypes.byref(found))
        if rc:
            print("b=%d %s: error" % (b, name)); continue
        t = ms[:found.value]
        ok = t[t > 0]
        fl = 2.0 * M * N * K
        free = [i for i in range(found.value) if wsz[i] == 0 and t[i] > 0]      # what dcl_linear_fwd may take: no workspace
        taken = free[0] if free else -1
        print("b=%-2d %-20s M=%6d: first zero-workspace candidate (#%d) %7.1f us (%5.1f TF/s); best of %2d: %7.1f us (%5.1f TF/s, candidate %d, "
              "workspace %d B); within 5%% of best: %d" % (
                  b, name, M, taken, t[taken] * 1e3 if taken >= 0 else float("nan"), fl / t[taken] / 1e9 if taken >= 0 else float("nan"),
                  found.value, ok.min() * 1e3, fl / ok.min() / 1e9, int(np.argmin(np.where(t > 0, t, 1e9))),
                  int(wsz[int(np.argmin(np.where(t > 0, t, 1e9)))]), int((ok <= ok.min() * 1.05).sum())))

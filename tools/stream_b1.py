#!/usr/bin/env python3
"""One-crop stream (LineMOD-style b=1 calls): ms per call and kernel nodes of the captured forward -- the lm_stream leg of
bench.py on its own.  usage: stream_b1.py [reps] [crops per call]"""
import importlib, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
if os.environ.get("DCL_USE_DIAG"):                 # A/B through the diagnostic library's switches (tools/_diag.py reads DCL_CONV_* here)
    from _diag import use_diag
    use_diag(dcl)
print(json.dumps(bench.lm_stream_bench(dcl, torch.device("cuda:0"), reps=int(sys.argv[1]) if len(sys.argv) > 1 else 100,
                                       b=int(sys.argv[2]) if len(sys.argv) > 2 else 1)))

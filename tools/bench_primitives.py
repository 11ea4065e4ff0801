#!/usr/bin/env python3
"""north-star primitives standalone (ball_query + group_points + FPS), same routine bench.py reports as `primitives`."""
import importlib, importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
dcl = importlib.import_module("dcl-net_amd")
if os.environ.get("DCL_USE_DIAG"):                 # A/B through the diagnostic library's switches (tools/_diag.py reads DCL_* here)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _diag import use_diag
    use_diag(dcl)
print(json.dumps(bench.primitives_roofline(dcl, reps=10)))

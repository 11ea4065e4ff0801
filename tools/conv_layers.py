#!/usr/bin/env python3
"""Per-layer time of the sparse-conv stack inside ordinary forwards (the backbone runner's implicit rulebooks, product library):
bench.py's roofline_sparse_conv for one shape, printed layer by layer.  usage: conv_layers.py [ref|stress] [b]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dcl = importlib.import_module("dcl-net_amd")
if os.environ.get("DCL_USE_DIAG"):                 # A/B through the diagnostic library's switches (tools/_diag.py reads DCL_CONV_* here)
    from _diag import use_diag
    use_diag(dcl)
shape = sys.argv[1] if len(sys.argv) > 1 else "ref"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
n_inp, n_tmp = (1024, 1024) if shape == "ref" else (12288, 2048)
dev = torch.device("cuda:0")
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n_inp, n_tmp), mode="test", graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.to(dev).eval()
data = bench.to_device(dcl.synth.make_batch(b, n_inp, n_tmp), dev)
r = bench.sparse_conv_roofline(dcl, net, data, dev, steps=5)
g = r["grouped_single_stream"]
print("%s b=%d: conv %.4f ms per forward, frac %.4f (each backbone's own launches: the default schedule); grouped one-stream "
      "schedule %.4f ms, frac %.4f (the layer table below); density %.3f; feature stage %s" % (
          shape, b, r["conv_ms_per_forward"], r["frac"], g["conv_ms_per_forward"], g["frac"], r["rulebook_density"],
          {k: v for k, v in r["feature_stage"].items() if k != "what"}))
for L in r["layers"]:
    print("  %-18s rows %7d dens %.2f  %7.1f us  %5.1f TF  mfma %6.1f us  gather %6.1f us  %-9s frac %.3f" % (
        L["layer"], L["rows"], L["density"], L["ms"] * 1e3, L["TFLOPs"], L["mfma_bound_ms"] * 1e3, L["l2_gather_bound_ms"] * 1e3,
        L["bound"], L["frac_of_bound"]))

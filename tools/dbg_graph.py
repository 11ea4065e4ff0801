import importlib, os, sys, time
import torch
sys.path.insert(0, '.')
dcl = importlib.import_module("dcl-net_amd")
n = 1024
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = dcl.synth.default_cfg(n, n)
net = dcl.DCL_Net.Network(cfg, mode="test")
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
data = dcl.synth.make_batch(b, n, n)
print("eager", flush=True)
with torch.no_grad():
    w = net(data)
torch.cuda.synchronize()
print("eager ok; capture", flush=True)
g = net.forward_graphed(data)
torch.cuda.synchronize()
print("graph ok", float((g["rot_pred"] - w["rot_pred"]).abs().max()), flush=True)
for i in range(5):
    g = net.forward_graphed(dcl.synth.make_batch(b, n, n, first=i))
    torch.cuda.synchronize()
    print("replay", i, "ok", flush=True)

#!/usr/bin/env python3
"""Training-step timing (forward in train mode + loss + backward + Adam) on synthetic crops: tools/train_step.py [b] [N]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n, n), mode="train")
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().train()
crit = dcl.DCL_Net.losses(None)
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
data = dcl.synth.make_batch(b, n, n)
data["flags"] = torch.zeros(b)
def step():
    opt.zero_grad()
    pred = net(data)
    loss = crit(pred, data["labels"])["loss_all"]
    loss.backward()
    opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): l = step()
torch.cuda.synchronize()
print("train step b=%d N=M=%d: %.1f ms (loss %.4f), peak mem %.2f GB" % (b, n, (time.perf_counter() - t0) / 10 * 1e3, float(l.detach()), torch.cuda.max_memory_allocated() / 1e9))

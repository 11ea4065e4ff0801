import importlib, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
dcl = importlib.import_module("dcl-net_amd")
b = int(sys.argv[1]); graph = int(sys.argv[2])
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(1024, 1024, unit=0.005 if b == 1 else 0.006), mode="test", graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
data = bench.to_device(dcl.synth.make_batch(b, 1024, 1024, unit=0.005 if b == 1 else 0.006), torch.device("cuda"))
with torch.no_grad():
    for _ in range(30):
        (net.forward_graphed(data) if graph else net(data))
torch.cuda.synchronize()

"""furthest_point_sampling alone at the north-star shape (B = 32, N = 12288, npoint = 2048) and two smaller ones: ms per call.
Reference: libs/pointnet_lib/src/sampling_gpu.cu:93-209."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dcl = importlib.import_module("dcl-net_amd")
for (B, N, M) in ((32, 12288, 2048), (32, 2048, 512), (8, 12288, 2048), (32, 1024, 256)):
    g = torch.Generator(device="cpu").manual_seed(1)
    xyz = (torch.rand(B, N, 3, generator=g) * 0.3).cuda()
    for _ in range(2):
        idx = dcl.ops.furthest_point_sampling(xyz, M)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        idx = dcl.ops.furthest_point_sampling(xyz, M)
    e.record(); torch.cuda.synchronize()
    print("FPS B=%d N=%d npoint=%d: %.3f ms per call, checksum %d" % (B, N, M, a.elapsed_time(e) / 5, int(idx.long().sum())))

"""The stage-2 refine loop alone (2 iterations, 32 crops of 1024 points), launch by launch: ms per loop; under
`rocprofv3 --kernel-trace --stats` the per-kernel breakdown.  usage: python tools/refiner_loop.py [reps]"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dcl = importlib.import_module("dcl-net_amd")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
b, n = 32, 1024
dev = torch.device("cuda:0")
ref = dcl.refiner.Refiner().to(dev).eval()
g = torch.Generator(device="cpu").manual_seed(1)
pred = {"rot_pred": torch.linalg.qr(torch.randn(b, 3, 3, generator=g))[0].to(dev), "trans_pred": (torch.randn(b, 3, generator=g) * 0.01).to(dev),
        "F_Xo_p": torch.randn(b, 256, n, generator=g).to(dev), "conf": torch.randn(b, 2 * n, generator=g).to(dev)}
pts = (torch.randn(b, n, 3, generator=g) * 0.05).to(dev)
for graph in (False, True):
    for _ in range(3):
        dcl.refiner.refine_loop(ref, pred, pts, 2, graph=graph)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = dcl.refiner.refine_loop(ref, pred, pts, 2, graph=graph)
    torch.cuda.synchronize()
    print("refine loop (2 iterations, b=%d): %.3f ms %s" % (b, (time.perf_counter() - t) / reps * 1e3, "hipGraph" if graph else "eager"))

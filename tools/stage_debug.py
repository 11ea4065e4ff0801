#!/usr/bin/env python3
"""The feature stage of BOTH backbones stand-alone: per-layer launches (each side alone, and the grouped pair form) against the
ONE-launch stage (ops.backbone_features_stage) -- the four pooled levels of both sides compared word by word, and the time of
each form (us per feature stage, exact mode and capacity mode).
usage: python tools/stage_debug.py [N] [batch] [slots]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
if os.environ.get("STAGE_DIAG"):
    from _diag import use_diag
    use_diag(dcl)          # timing-experiment flags (bits 1.. of `flags`) exist in the diagnostic library only
ops = dcl.ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = int(sys.argv[2]) if len(sys.argv) > 2 else 32
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 512
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cfg = dcl.synth.default_cfg(n, n)
net = dcl.DCL_Net.Network(cfg, mode="test")
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.cuda().eval()
f = net._fold()
data = dcl.synth.make_batch(b, n, n)
sides = ("inp", "tmp")
occ = {s: data[s]["occupied_voxels"].int().cuda().contiguous() for s in sides}
x = {s: ops.voxelize_fp(data[s]["feats"].cuda(), data[s]["v2p_maps"].cuda(), 4) for s in sides}
ptrs = {s: f["backbone_%s_ptrs" % s] for s in sides}
CAP = bool(os.environ.get("STAGE_CAP"))      # capacity mode (what a captured whole-forward graph runs): device-side counts only
def geo(s):
    if CAP:
        v0 = occ[s].shape[0]
        pad = torch.zeros((b * n, 4), dtype=torch.int32, device="cuda"); pad[:v0] = occ[s]
        xs = torch.zeros((b * n, x[s].shape[1]), dtype=torch.float32, device="cuda"); xs[:v0] = x[s]; x[s] = xs
        run = ops.BackboneRunCap(pad, torch.tensor([v0], dtype=torch.int32, device="cuda"), b, 64)
        run.geometry()
        run.counts = run.counts_dev.cpu().tolist()
        return run
    run = ops.BackboneRun(occ[s], b, 64)
    run.set_counts(run.counts_dev.cpu().tolist())
    return run
runs = {s: geo(s) for s in sides}
print("flags=%d " % flags, end="")
print("b=%d N=%d slots=%d counts inp %s tmp %s" % (b, n, slots, runs["inp"].counts, runs["tmp"].counts))
status = ops.stage_status_buffer()
def per_layer():
    for s in sides:
        runs[s].features(x[s], *ptrs[s])
def pair():
    if CAP:
        return per_layer()
    ops.backbone_features_pair(runs["inp"], x["inp"], ptrs["inp"], runs["tmp"], x["tmp"], ptrs["tmp"])
def stage():
    assert ops.backbone_features_stage([runs[s] for s in sides], [x[s] for s in sides], [ptrs[s] for s in sides], status, slots=slots, flags=flags)
def stage1():
    for s in sides:
        assert ops.backbone_features_stage([runs[s]], [x[s]], [ptrs[s]], status, slots=slots, flags=flags)
def levels():
    torch.cuda.synchronize()
    return {s: [t[:runs[s].counts[2 * m + 1]].clone() for m, t in enumerate(runs[s].levels)] for s in sides}
per_layer(); want = levels()
for name, fn in (("pair", pair), ("stage (both sides, one launch)", stage), ("stage (a launch per side)", stage1)):
    for rep in range(3):
        fn(); got = levels()
        line = []
        for s in sides:
            for m in range(4):
                d = (got[s][m] - want[s][m]).abs()
                bad = int((got[s][m] != want[s][m]).sum())
                line.append("%s%d %d/%d %.2e" % (s[0], m, bad, d.numel(), float(d.max()) if d.numel() else 0.0))
        print("%s rep %d status %d: words that differ / max |diff| per level: %s" % (name, rep, int(status[0]), " | ".join(line)))
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps * 1e3
for name, fn in (("per-layer, side after side", per_layer), ("per-layer pair", pair), ("stage", stage), ("stage per side", stage1)):
    print("%-28s %.1f us" % (name, timeit(fn)))
print("status", int(status[0]))
if os.environ.get("STAGE_DIAG"):
    import ctypes as C
    import numpy as np
    stage(); torch.cuda.synchronize()
    nmax = 1 << 16
    buf = (C.c_ulonglong * (4 * nmax))()
    dcl._native.lib().dcl_debug_stage_stamps(buf, nmax)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(nmax, 4).astype(np.int64)
    live = a[:, 1] > 0
    a = a[live]
    t0 = a[:, 0].min()
    names = {0: "stem", 1: "wlds16", 2: "dma", 3: "reduce", 4: "pool"}
    print("phase kind shape items | first start, last start, first end, last end (us) | mean / median / max item us | sum item us / 512 slots")
    for p in sorted(set(a[:, 2].tolist())):
        r = a[a[:, 2] == p]
        kind, shape = int(r[0, 3] >> 32), int(r[0, 3] & 0xffffffff)
        d = (r[:, 1] - r[:, 0]) / 100.0
        print("%2d %-6s %6d %5d | %7.1f %7.1f %7.1f %7.1f | %6.1f %6.1f %6.1f | %6.1f" % (
            p, names[kind], shape if kind in (2, 3) else 0, len(r), (r[:, 0].min() - t0) / 100.0, (r[:, 0].max() - t0) / 100.0,
            (r[:, 1].min() - t0) / 100.0, (r[:, 1].max() - t0) / 100.0, d.mean(), np.median(d), d.max(), d.sum() / 512.0))
        if os.environ.get("STAGE_DIAG") == "2" and kind == 2:
            st = np.sort((r[:, 0] - t0) / 100.0)
            print("     starts (deciles):", " ".join("%.0f" % st[int(q * (len(st) - 1) / 10)] for q in range(11)),
                  "| items (deciles):", " ".join("%.0f" % v for v in np.percentile(d, range(0, 101, 10))))

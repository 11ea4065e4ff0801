#!/usr/bin/env python3
"""Where the workgroups of ONE LDS-DMA sparse-conv launch of the backbone runner spend their time (the runner's own launches:
implicit rulebooks, both backbones grouped or one side).  Needs the stamps library (make -C dcl-net_amd/csrc stamps:
in-kernel s_memrealtime stamps, 100 MHz ticks, 8 per segment of a workgroup).
usage: DCL_HIP_LIB=tests/_diag/libdclnet_hip_stamps.so tools/conv_stamps.py [launch 0-5] [pair|inp|tmp] [ref|stress] [b]
launch = index among the LDS-DMA launches of a feature stage: 0 L1 conv 32->32 (pair only), 1 L1 subm 32->64, 2 L2 conv,
3 L2 subm, 4 L3 conv, 5 L3 subm (one side: L1 conv runs the filter-resident kernel, so 0 = L1 subm, ...)."""
import ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
os.environ.setdefault("DCL_HIP_LIB", os.path.join(ROOT, "tests", "_diag", "libdclnet_hip_stamps.so"))
DIAG = use_diag(dcl)
ops = dcl.ops
sel = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = sys.argv[2] if len(sys.argv) > 2 else "pair"
shape = sys.argv[3] if len(sys.argv) > 3 else "ref"
b = int(sys.argv[4]) if len(sys.argv) > 4 else 32
n_inp, n_tmp = (1024, 1024) if shape == "ref" else (12288, 2048)
dev = torch.device("cuda:0")
net = dcl.DCL_Net.Network(dcl.synth.default_cfg(n_inp, n_tmp), mode="test", graph_max_batch=0)
net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
net = net.to(dev).eval()
f = net._fold()
data = dcl.synth.make_batch(b, n_inp, n_tmp)
runs, xs, ptrs = {}, {}, {}
for s in ("inp", "tmp"):
    occ = data[s]["occupied_voxels"].to(dev).int().contiguous()
    xs[s] = ops.voxelize_fp(data[s]["feats"].to(dev).float().contiguous(), data[s]["v2p_maps"].to(dev).int().contiguous(), 4)
    run = ops.BackboneRun(occ, b, 64)
    run.set_counts(run.counts_dev.cpu().tolist())
    runs[s], ptrs[s] = run, f["backbone_%s_ptrs" % s]


def stage():
    if mode == "pair":
        ops.backbone_features_pair(runs["inp"], xs["inp"], ptrs["inp"], runs["tmp"], xs["tmp"], ptrs["tmp"])
    else:
        runs[mode].features(xs[mode], *ptrs[mode])


lib = ops.N.lib()
lib.dcl_debug_conv_stamps_select(1 << 20)           # nothing stamped while warming up
for _ in range(3):
    stage()
torch.cuda.synchronize()
lib.dcl_debug_conv_stamps(None, 0, 1)
lib.dcl_debug_conv_stamps_select(sel)
stage()
torch.cuda.synchronize()
WG, SEG = 1024, 4
st = np.zeros((WG, SEG, 16), np.uint64)
ph = np.zeros((WG, SEG, 16), np.uint64)
lib.dcl_debug_conv_stamps(st.ctypes.data_as(ctypes.c_void_p), 0, 0)
lib.dcl_debug_conv_stamps(ph.ctypes.data_as(ctypes.c_void_p), 1, 0)
st, ph = st.astype(np.int64), ph.astype(np.float64)
live = st[:, :, 0] > 0
if not live.any():
    sys.exit("no stamps: launch %d of this stage is not an LDS-DMA launch" % sel)
tick = 0.01                                          # us per tick
t0 = st[:, :, 0][live].min()
end_all = st[:, :, 1:7].max()
print("launch %d (%s, %s, b=%d): %d workgroups, %d segments; first start -> last end %.1f us" % (
    sel, mode, shape, b, int(live.any(1).sum()), int(live.sum()), (end_all - t0) * tick))
names = ["head: rows + nbr table + masks", "first operand fetch", "chunk loop", "publish partial", "ticket", "combine + epilogue"]
for sg in range(SEG):
    L = live[:, sg]
    if not L.any():
        continue
    S = st[L, sg]
    print(" segment %d: %d workgroups, starts %.1f..%.1f us after the launch's first" % (
        sg, int(L.sum()), (S[:, 0].min() - t0) * tick, (S[:, 0].max() - t0) * tick))
    for i, nm in enumerate(names):
        ok = (S[:, i + 1] > 0) & (S[:, i] > 0)
        if ok.any():
            d = (S[ok, i + 1] - S[ok, i]) * tick
            print("   %-32s avg %7.2f  p50 %7.2f  max %7.2f us  (n=%d)" % (nm, d.mean(), np.median(d), d.max(), int(ok.sum())))
    for a, z, nm in ((0, 8, "  rows + first barrier"), (8, 9, "  table loop (look-ups -> LDS)"), (9, 10, "  mask reduce + barrier"),
                     (10, 11, "  step masks (27 ballots)"), (11, 1, "  accumulators, first chunk pick")):
        ok = (S[:, z] > 0) & (S[:, a] > 0)
        if ok.any():
            d = (S[ok, z] - S[ok, a]) * tick
            print("   %-32s avg %7.2f  p50 %7.2f  max %7.2f us" % (nm, d.mean(), np.median(d), d.max()))
    e = S[:, 1:7].max(1)
    # a whole-tile segment's epilogue (stores) lies after stamp 3 and carries no stamp of its own
    print("   segment life (to its last stamp)  avg %7.2f us" % ((e - S[:, 0]).mean() * tick))
    P = ph[L, sg]
    ok = P[:, 4] > 0
    if ok.any():
        per = P[ok, :4] / P[ok, 4:5]
        loop_us = ((S[:, 3] - S[:, 2]) * tick)[ok]
        cyc = P[ok, :4].sum(1)
        good = loop_us > 0
        print("   per chunk (wave 0, shader cycles): DMA wait %.0f  barrier %.0f  issue %.0f  MFMA block %.0f | used chunks %.1f | clock %.2f GHz" % (
            per[:, 0].mean(), per[:, 1].mean(), per[:, 2].mean(), per[:, 3].mean(), P[ok, 4].mean(),
            (cyc[good] / (loop_us[good] * 1e3)).mean() if good.any() else float("nan")))
# per workgroup: time inside segments vs life of the workgroup
first = np.where(live, st[:, :, 0], np.iinfo(np.int64).max).min(1)
last = np.where(live, st[:, :, 1:7].max(2), 0).max(1)
w = live.any(1)
life = (last[w] - first[w]) * tick
print(" workgroup life avg %.1f us (min %.1f max %.1f); ends spread: p50 %.1f  p90 %.1f  max %.1f us after the launch's first start" % (
    life.mean(), life.min(), life.max(), *[np.percentile((last[w] - t0) * tick, q) for q in (50, 90, 100)]))
starts = (first[w] - t0) * tick
print(" workgroup starts: p50 %.1f  p90 %.1f  max %.1f us" % (np.percentile(starts, 50), np.percentile(starts, 90), starts.max()))
# ---- who is slow?  per-chunk loop time of every segment against where the workgroup ran (XCC, CU) and what shared its CU
hw = st[:, :, 7]
xcc = (hw >> 32) & 0xF
hwid = hw & 0xFFFFFFFF
cu = xcc * 1000 + ((hwid >> 13) & 0x7) * 100 + ((hwid >> 12) & 1) * 50 + ((hwid >> 8) & 0xF)       # xcc, se, sh, cu
rows = []
for w in range(WG):
    for sg in range(SEG):
        if live[w, sg] and ph[w, sg, 4] > 0 and st[w, sg, 3] > st[w, sg, 2] > 0:
            rows.append((int(xcc[w, sg]), int(cu[w, sg]), (st[w, sg, 3] - st[w, sg, 2]) * tick / ph[w, sg, 4], ph[w, sg, 4], w, sg))
if rows:
    R = np.array([(r[0], r[1], r[2], r[3]) for r in rows], dtype=np.float64)
    big = R[:, 3] >= 8                                            # segments of at least 8 chunks
    print(" us per used chunk (segments of >= 8 chunks): avg %.3f  p10 %.3f  p50 %.3f  p90 %.3f  max %.3f" % (
        R[big, 2].mean(), *[np.percentile(R[big, 2], q) for q in (10, 50, 90, 100)]))
    print(" by XCC: " + "  ".join("%d: %.3f (n=%d)" % (x, R[big & (R[:, 0] == x), 2].mean(), int((big & (R[:, 0] == x)).sum())) for x in range(8) if (big & (R[:, 0] == x)).any()))
    per_cu = {}
    for x, c, t, n_ in R[big]:
        per_cu.setdefault(int(c), []).append(t)
    occ = np.array([len(v) for v in per_cu.values()])
    mean_by_occ = {k: np.mean([np.mean(v) for v in per_cu.values() if len(v) == k]) for k in sorted(set(occ.tolist()))}
    print(" distinct CUs %d; segments per CU -> mean us per chunk: %s" % (len(per_cu), {k: round(v, 3) for k, v in mean_by_occ.items()}))
    cu_means = np.array([np.mean(v) for v in per_cu.values()])
    print(" CU-to-CU spread of the mean: p10 %.3f p50 %.3f p90 %.3f" % tuple(np.percentile(cu_means, q) for q in (10, 50, 90)))
# ---- per CU: when do its workgroups end?  (a CU is done when its last workgroup is; a workgroup that outlives its mate runs alone)
wg_cu = {}
for w in range(WG):
    if live[w].any():
        sg0 = int(np.argmax(live[w]))
        wg_cu.setdefault(int(cu[w, sg0]), []).append(((first[w] - t0) * tick, (last[w] - t0) * tick, w))
ends = np.array([max(e for _, e, _ in v) for v in wg_cu.values()])
pairs = [v for v in wg_cu.values() if len(v) == 2]
if pairs:
    gaps = np.array([abs(v[0][1] - v[1][1]) for v in pairs])
    print(" CU end times: p10 %.1f p50 %.1f p90 %.1f max %.1f us; CUs with two workgroups %d: gap between their two ends p50 %.1f p90 %.1f max %.1f us" % (
        *[np.percentile(ends, q) for q in (10, 50, 90, 100)], len(pairs), *[np.percentile(gaps, q) for q in (50, 90, 100)]))
    ids = np.array([[v[0][2], v[1][2]] for v in pairs])
    print(" workgroup ids sharing a CU (first five pairs): %s" % ids[:5].tolist())
half = WG // 2 if int(live.any(1).sum()) > 300 else 0
if half and rows:
    Rw = np.array([(r[4], r[2], r[3]) for r in rows], dtype=np.float64)
    bigw = Rw[:, 2] >= 8
    n_live = int(live.any(1).sum())
    lo_ids, hi_ids = Rw[:, 0] < 256, Rw[:, 0] >= 256
    print(" us per chunk, workgroups 0..255 (first on their CU): %.3f   workgroups 256..: %.3f;  life: %.1f vs %.1f us" % (
        Rw[bigw & lo_ids, 1].mean(), Rw[bigw & hi_ids, 1].mean(), life[:256].mean() if len(life) > 256 else float("nan"),
        life[256:].mean() if len(life) > 256 else float("nan")))

#!/usr/bin/env python3
"""Where a sparse-conv workgroup spends its time: runs one backbone layer on the real active sets (bs 32) with the
diagnostic library (make -C dcl-net_amd/csrc stamps; in-kernel s_memrealtime stamps, 100 MHz ticks) and prints the average
phase durations over the workgroups.  usage: DCL_HIP_LIB=tests/_diag/libdclnet_hip_stamps.so tools/conv_stamps.py [level 0-3] [conv|subm] [split]"""
import ctypes, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
from _diag import use_diag
DIAG = use_diag(dcl)          # kernel-variant hooks exist in the diagnostic library only
ops, sp = dcl.ops, dcl.spconv.ops
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
which = sys.argv[2] if len(sys.argv) > 2 else "subm"
if len(sys.argv) > 3:
    ops.N.lib().dcl_debug_conv_split(int(sys.argv[3]))
b, S = int(os.environ.get("DCL_BENCH_B", "32")), 64
data = dcl.synth.make_batch(b, 1024, 64)
aset = ops.grid_from_indices(data["inp"]["occupied_voxels"].int().cuda().contiguous(), b, S)
chans = [7, 16, 32, 32, 64, 64, 128, 128, 256]
for lvl in range(level + 1):
    out, nbr1 = sp.build_rulebook(aset, 3, 1, 1, False)
    _, nbr2 = sp.build_rulebook(out, 3, 1, 1, True)
    pool, _ = sp.build_rulebook(out, 3, 2, 1, False)
    aset = pool
c0, c1, c2 = chans[2 * level], chans[2 * level + 1], chans[2 * level + 2]
if which == "conv":
    nbr, cin, cout, subm, rows_in = nbr1, c0, c1, False, nbr1.max().item() + 1
else:
    nbr, cin, cout, subm, rows_in = nbr2, c1, c2, True, out.n
feat = torch.randn(rows_in, cin, device="cuda")
W = torch.randn(27, cin, cout, device="cuda") * 0.05
lib = ops.N.lib()
for _ in range(3):
    ops.sparse_conv(feat, nbr, out.n, W, subm)
torch.cuda.synchronize()
lib.dcl_debug_conv_stamps(None, 0, 1)
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
ops.sparse_conv(feat, nbr, out.n, W, subm)
e.record()
torch.cuda.synchronize()
n_wg = 16384
buf = np.zeros((n_wg, 8), np.uint64)
lib.dcl_debug_conv_stamps(buf.ctypes.data_as(ctypes.c_void_p), n_wg, 0)
live = buf[buf[:, 0] > 0].astype(np.int64)
if len(live) == 0:
    sys.exit("layer L%d %s %d->%d: no stamps -- this launch ran a kernel family without them (filter-resident / stem)" % (level, which, cin, cout))
t0 = live[:, 0].min()
names = ["nbr table", "first fetch", "chunk loop", "publish", "ticket", "combine/exit"]
print("layer L%d %s %d->%d rows %d: %.1f us, %d workgroups stamped" % (level, which, cin, cout, out.n, a.elapsed_time(e) * 1e3, len(live)))
tick = 0.01                                         # s_memrealtime runs at 100 MHz: us per tick
for i, nm in enumerate(names):
    d = (live[:, i + 1] - live[:, i])
    ok = (live[:, i + 1] > 0) & (live[:, i] > 0)
    if ok.any():
        print("  %-14s avg %7.2f us  max %7.2f us  (n=%d)" % (nm, d[ok].mean() * tick, d[ok].max() * tick, ok.sum()))
end = live[:, 1:].max(axis=1)
print("  (stamps are overwritten per segment: all figures are those of each workgroup's LAST segment)")
print("  last-segment life avg %.2f us; first start -> last end %.2f us; starts spread %.2f us" % (
    ((end - live[:, 0]).mean()) * tick, (end.max() - t0) * tick, (live[:, 0].max() - t0) * tick))
st = np.sort(live[:, 0] - t0) * tick
print("  last-segment starts by 20-us bucket:", np.histogram(st, bins=np.arange(0, st.max() + 20, 20))[0].tolist())
ev = sorted([(x, 1) for x in (live[:, 0] - t0)] + [(x, -1) for x in (end - t0)])
cur = peak = 0
for _, d in ev:
    cur += d
    peak = max(peak, cur)
print("  peak concurrent last segments: %d" % peak)
hw = buf[buf[:, 0] > 0][:, 7]
xcc, hwid = (hw >> np.uint64(32)).astype(np.int64) & 0xF, hw.astype(np.int64) & 0xFFFFFFFF
cu_key = xcc * 100000 + ((hwid >> 8) & 0xF) * 1000 + ((hwid >> 13) & 0x7) * 100 + ((hwid >> 12) & 1) * 50   # cu_id, se_id, sh_id
print("  distinct (xcc, se, sh, cu): %d; workgroups per XCC: %s" % (len(set(cu_key.tolist())), np.bincount(xcc, minlength=8).tolist()))
ph = np.zeros((n_wg, 8), np.uint64)
lib.dcl_debug_conv_stamps(ph.ctypes.data_as(ctypes.c_void_p), -n_wg, 0)
ph = ph[buf[:, 0] > 0].astype(np.float64)
ok = ph[:, 4] > 0
if ok.any():
    per = ph[ok, :4] / ph[ok, 4:5]
    loop_us = ((live[:, 3] - live[:, 2]) * tick)[ok]
    cyc = ph[ok, :4].sum(1)
    print("  per chunk (wave 0, shader cycles): DMA wait %.0f  barrier %.0f  issue %.0f  MFMA block %.0f  | chunks/wg %.1f | clock %.2f GHz" % (
        per[:, 0].mean(), per[:, 1].mean(), per[:, 2].mean(), per[:, 3].mean(), ph[ok, 4].mean(), (cyc / (loop_us * 1e3)).mean()))
last = live[:, 6] > 0
print("  last arrivers: %d" % int(last.sum()))

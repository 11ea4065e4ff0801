#!/usr/bin/env python3
"""bench.py -- frames/s of DCL_Net.forward on synthetic YCB-V-shaped crops (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W [--shape stress|ref]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one Network.forward over one batch of b=32 crops whose `data` dict is already resident in HBM.
One process per GPU; ranks own disjoint crops (weak scaling, no data-path collective); the only RCCL
traffic is the ADD-S metric all-reduce after the timed region.  Rank 0 prints ONE JSON line.

Workloads: "stress" = BASELINE.json configs[1] (bs 32, N=12288 observed, M=2048 model points) -- the headline;
"ref" = what configs/config_YCBV_bs32.yaml really defines (N=M=1024), reported alongside as `ref_shape`.
"""
import argparse
import importlib
import json
import os
import sys
import time

# two hardware queues per device (the package's own default, set here too because torch is imported first: dcl-net_amd/__init__.py,
# DESIGN.md section 6 -- a hipGraph branch on a third or fourth queue makes the graph 1.3-4x slower); a value in the environment wins
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32 = 157.3          # TFLOP/s, MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_MFMA_BF16 = 2500.0        # TFLOP/s dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
# the split-bf16 kernels (csrc/linear_split.hip): an fp32 product = six bf16 piece products, so their fp32-equivalent bound is
PEAK_SPLIT = PEAK_MFMA_BF16 / 6.0
OWN_GEMM = "own_gemm (k_linear_split*, k_linear_dma*, k_linear_group, k_mlp128_to1)"
PEAK_HBM = 8000.0              # GB/s spec
SHAPES = {"stress": (12288, 2048), "ref": (1024, 1024)}


def measure_traffic_bytes(kernel_substr, shape, batch, timeout_s=300):
    """HBM bytes per launch of a kernel, MEASURED IN THIS RUN: two child processes of this same script (one forward pass of
    the same workload each, `--pmc-child`) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes with
    --kernel-trace only, as MI355X_MICROARCH.md's rocprofv3 section prescribes (the two counters do not fit one pass) --
    while this process idles.  Corrections from the guide's HBM section: both counters are KiB; on gfx950 FETCH_SIZE
    reports half of the bytes of a wide coalesced read, so it is doubled.  -> (bytes or None, source string)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "unmeasured: rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "unmeasured: this run is itself under a rocprof tool (no nested counter passes)"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="dcl_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", tmp, "-o", "pmc", "--output-format", "csv", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--shape", shape, "--batch", str(batch)]
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("DCL_FORCE_DIST", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=timeout_s, check=True)
            per_launch = []
            for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        per_launch.append(float(r["Counter_Value"]))
            if per_launch:
                vals[counter] = sum(per_launch) / len(per_launch)
        except (subprocess.SubprocessError, OSError) as e:
            shutil.rmtree(tmp, ignore_errors=True)
            return None, "unmeasured: %s pass failed (%s)" % (counter, type(e).__name__)
        shutil.rmtree(tmp, ignore_errors=True)
    if len(vals) != 2:
        return None, "unmeasured: kernel not found in the counter files"
    return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), (
        "measured in this run: child rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over one forward of the same "
        "workload, average per launch of the kernel; FETCH_SIZE x2 (gfx950), KiB -> B")


def measure_kernel_shares(shape, batch, timeout_s=300):
    """Whose kernels the step's GPU time goes to, MEASURED IN THIS RUN: one more child process of this script (`--pmc-child`:
    two launch-by-launch forwards of the same workload, a pause between them) under `rocprofv3 --kernel-trace` alone (no
    counters: kernels keep their concurrency and their durations), this process idle meanwhile.  Only the kernels of the LAST
    forward count -- everything before the trace's last pause of more than 20 ms (set-up kernels, uploads, the cold first forward) is cut off.
    -> (dict or None, source string): per kernel family the summed kernel time and its share of that forward's summed kernel time."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "unmeasured: rocprofv3 not on PATH"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "unmeasured: this run is itself under a rocprof tool"
    tmp = tempfile.mkdtemp(prefix="dcl_kt_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "-d", tmp, "-o", "kt", "--output-format", "csv", "--",
           sys.executable, os.path.abspath(__file__), "--pmc-child", "--shape", shape, "--batch", str(batch)]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("DCL_FORCE_DIST", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    SPARSE = "sparse feature stage (k_sparse_conv*, k_conv_frag*, k_sparse_avgpool*)"
    fam = {"attention (k_cross_attn*)": 0.0, OWN_GEMM: 0.0,
           "vendor_gemm (hipBLASLt Cijk_*)": 0.0, SPARSE: 0.0, "other": 0.0}
    launches = {k: 0 for k in fam}
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s,
                       check=True)
        trace = []
        for f in glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                trace.append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]), r["Kernel_Name"]))
        trace.sort()
        cut, busy_until = 0, None                            # the last forward starts behind the LAST long pause of the trace (the child
        for i, (t0, t1, _) in enumerate(trace):              # sleeps 50 ms in front of it; plan creation inside the cold first forward pauses too)
            if busy_until is not None and t0 - busy_until > 2e7:
                cut = i
            busy_until = t1 if busy_until is None else max(busy_until, t1)
        rows = 0
        for t0, t1, name in trace[cut:]:
            key = ("attention (k_cross_attn*)" if "k_cross_attn" in name else
                   OWN_GEMM if ("k_linear_split" in name or "k_linear_dma" in name or "k_linear_group" in name
                                                                                 or "k_mlp128_to1" in name) else
                   "vendor_gemm (hipBLASLt Cijk_*)" if name.startswith("Cijk_") else
                   SPARSE if ("k_sparse_conv" in name or "k_conv_frag" in name or "k_sparse_avgpool" in name) else "other")
            fam[key] += t1 - t0
            launches[key] += 1
            rows += 1
    except (subprocess.SubprocessError, OSError, KeyError, ValueError) as e:
        shutil.rmtree(tmp, ignore_errors=True)
        return None, "unmeasured: kernel-trace pass failed (%s)" % type(e).__name__
    shutil.rmtree(tmp, ignore_errors=True)
    total = sum(fam.values())
    if rows == 0 or total <= 0:
        return None, "unmeasured: empty kernel trace"
    out = {k: {"ns": v, "share": v / total, "launches": launches[k]} for k, v in fam.items()}
    out["_forwards"] = 1
    return out, ("measured in this run: child rocprofv3 --kernel-trace pass, the kernels of ONE launch-by-launch forward of the same "
                 "workload (the trace behind its longest pause); share = the family's summed kernel time / the summed kernel "
                 "time of that forward (kernels of the two side streams overlap: a share of summed kernel time, not of wall time)")


def _flush_c_stdio(unbuffer=False):
    """C-level stdout (libraries that printf, e.g. RCCL's banner): flush it, optionally switch it to unbuffered."""
    import ctypes
    try:
        libc = ctypes.CDLL(None)
        if unbuffer:
            libc.setvbuf(ctypes.c_void_p.in_dll(libc, "stdout"), None, 2, 0)          # _IONBF
        libc.fflush(None)
    except (OSError, ValueError):
        pass


def to_device(data, dev):
    out = {}
    for k, v in data.items():
        if isinstance(v, dict):
            out[k] = to_device(v, dev)
        elif torch.is_tensor(v) and k != "voxel_num_limit":
            out[k] = v.to(dev)
        else:
            out[k] = v
    return out


def run_forward_bench(dcl, net, data, steps, warmup, distributed):
    """returns (seconds for `steps` forwards, max over ranks; per-launch attention kernel times in ms)"""
    dev = torch.device("cuda", torch.cuda.current_device())
    with torch.no_grad():
        for _ in range(warmup):
            net(data)
        dcl.ops.PROFILE_EVENTS = []
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net(data)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ev, dcl.ops.PROFILE_EVENTS = dcl.ops.PROFILE_EVENTS, None
    att_ms = [a.elapsed_time(b) for (name, a, b) in ev if name == "cross_attention"]
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, att_ms


def primitives_roofline(dcl, reps=5):
    """north-star primitives standalone (SURVEY 8d 'P'): ball_query + group_points at B=32, N=12288, npoint=2048,
    r=0.03, nsample=64, C=64; algorithmic bytes per SURVEY 8d."""
    B, N, NP, NS, C, r = 32, 12288, 2048, 64, 64, 0.03
    data = dcl.synth.make_batch(B, N, 64)
    xyz = data["inp"]["feats"][:, 4:7].reshape(B, N, 3).contiguous().cuda()
    fps = dcl.ops.furthest_point_sampling(xyz, NP)
    new_xyz = torch.gather(xyz, 1, fps.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    feats = torch.randn(B, C, N, device="cuda")
    out = {}
    for name, fn, nbytes in (
            ("ball_query", lambda: dcl.ops.ball_query(r, NS, xyz, new_xyz), 12 * B * (N + NP) + 4 * B * NP * NS),
            ("group_points", None, 4 * B * C * N + 4 * B * NP * NS + 4 * B * C * NP * NS)):
        if fn is None:
            idx = dcl.ops.ball_query(r, NS, xyz, new_xyz)
            fn = lambda: dcl.ops.group_points(feats, idx)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        out[name] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1), "bytes": nbytes}
    # ball_query is an exact search: its bound is the vector-instruction issue rate, not HBM.  Per candidate
    # test the kernel issues 4.5 vector instructions (two tests per 3 packed subtractions + packed mul + 2 packed fma + packed
    # compare-subtract + 2 v_alignbit); the chip issues 256 CUs x 64 lanes x 2.4 GHz lane-instructions per second.  `frac` is
    # priced on the tests the kernel EXECUTES: a workgroup of 16 centres scans the cloud in super-tiles of 1024
    # candidates and stops before the first one that starts with all its centres full -- which the kernel's own output tells:
    # a centre is full iff its last slot differs from its first (slots past the hit count repeat the first hit, ascending
    # indices), and then the index in its last slot is where its scan could end.
    tests = float(B) * NP * N
    idx = dcl.ops.ball_query(r, NS, xyz, new_xyz).long()
    full = idx[:, :, -1] != idx[:, :, 0]
    end_pt = torch.where(full, idx[:, :, -1] + 1, torch.full_like(idx[:, :, -1], N))           # candidates a centre needs scanned
    wg_end = end_pt.view(B, NP // 16, 16).max(dim=2).values                                     # ... its workgroup (16 centres)
    tiles = torch.clamp((wg_end + 1023) // 1024, max=(N + 1023) // 1024)
    executed = float((torch.clamp(tiles * 1024, max=N) * 16).sum().item())
    bq_s = out["ball_query"]["ms"] * 1e-3
    valu_bound = 256 * 64 * 2.4e9 / 4.5
    out["ball_query"].update({"bound": "valu (packed f32 issue)", "candidate_tests_brute_force": tests,
                              "candidate_tests_executed": executed,
                              "tests_per_s": round(executed / bq_s, 1), "valu_bound_tests_per_s": round(valu_bound, 1),
                              "frac_of_valu_bound": round(executed / bq_s / valu_bound, 3),
                              "speedup_over_scan": round(tests / executed, 3),
                              "note": "frac = executed (centre, candidate) tests per second / the vector-issue bound; "
                                      "speedup_over_scan = brute-force pairs / executed pairs (a workgroup stops scanning once "
                                      "its 16 centres hold nsample hits)"})
    tot_b = out["ball_query"]["bytes"] + out["group_points"]["bytes"]
    tot_ms = out["ball_query"]["ms"] + out["group_points"]["ms"]
    out["ball_query+group_points"] = {"bound": "hbm", "achieved": round(tot_b / tot_ms / 1e6, 1), "peak": PEAK_HBM,
                                      "unit": "GB/s", "frac": round(tot_b / tot_ms / 1e6 / PEAK_HBM, 4)}
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        dcl.ops.furthest_point_sampling(xyz, NP)
    b.record()
    torch.cuda.synchronize()
    out["fps_ms"] = round(a.elapsed_time(b) / 3, 3)
    # the remaining pointnet_lib primitives of SURVEY 8a row a17 (feature propagation direction: N unknown <- np known)
    d2, nn_idx = dcl.ops.three_nn(xyz, new_xyz)
    w = 1.0 / (d2.sqrt() + 1e-8)
    w = (w / w.sum(2, keepdim=True)).contiguous()
    kfeats = torch.randn(B, C, NP, device="cuda")
    for name, fn, nbytes in (
            ("gather_points", lambda: dcl.ops.gather_points(feats, fps), 4 * B * C * N + 4 * B * NP + 4 * B * C * NP),
            ("three_nn", lambda: dcl.ops.three_nn(xyz, new_xyz), 12 * B * (N + NP) + 24 * B * N),
            ("three_interpolate", lambda: dcl.ops.three_interpolate(kfeats, nn_idx, w), 4 * B * C * NP + 24 * B * N + 4 * B * C * N),
            ("knn1", lambda: dcl.ops.knn(1, xyz, new_xyz), 12 * B * (N + NP) + 8 * B * N)):     # get_cano_label's call
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        out[name] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1), "bytes": nbytes}
    # three_nn / knn (k = 1) are exact searches over the same B x N x npoint pairs as ball_query; the bucketed walk executes a
    # fraction of them.  `frac` is priced on the EXECUTED (query, point) tests -- counted by the diagnostic build of the same
    # kernels on the same inputs (tests/_diag: an atomic per scanned range; deterministic for given inputs) -- at 4 vector
    # instructions per test (3 subtractions, mul, 2 fma: dcl_dist2; the key compare-exchanges are the kept minority);
    # speedup_over_scan = brute-force pairs / executed pairs.
    executed_nn = {}
    try:
        import ctypes
        with dcl._native.diagnostic_library() as L:
            L.dcl_debug_nn_tests_executed.restype = ctypes.c_ulonglong
            for name, fn in (("three_nn", lambda: dcl.ops.three_nn(xyz, new_xyz)), ("knn1", lambda: dcl.ops.knn(1, xyz, new_xyz))):
                L.dcl_debug_nn_tests_executed(1)
                fn()
                executed_nn[name] = float(L.dcl_debug_nn_tests_executed(1))
    except (RuntimeError, AttributeError, OSError) as e:                  # no diagnostic library on this box: say so
        executed_nn = {"error": str(e)[:120]}
    valu_bound_nn = 256 * 64 * 2.4e9 / 6.0
    for name in ("three_nn", "knn1"):
        s_ = out[name]["ms"] * 1e-3
        ex = executed_nn.get(name)
        out[name].update({"bound": "valu (f32 issue), 6 vector instructions per executed test", "candidate_tests_brute_force": tests,
                          "candidate_tests_executed": ex, "valu_bound_tests_per_s": round(valu_bound_nn, 1),
                          "tests_per_s": round(ex / s_, 1) if ex else None,
                          "frac_of_valu_bound": round(ex / s_ / valu_bound_nn, 3) if ex else None,
                          "speedup_over_scan": round(tests / ex, 3) if ex else None})
        if not ex:
            out[name]["note"] = "diagnostic library unavailable: executed tests not counted (%s)" % executed_nn.get("error")
    return out


def pipelined_bench(dcl, dev, sd, cfg, data, b, steps, warmup, ref):
    """same workloads with Network(async_inputs=True): resident, never rewritten inputs let back-to-back calls overlap (the
    sparse half of call k+1 runs on side streams under the dense half of call k); all work of the K calls is inside the
    timed region, results are bit-identical to serial calls (tests/test_gpu_network.py)"""
    out = {"what": "K back-to-back forward calls, Network(async_inputs=True)"}
    net = dcl.DCL_Net.Network(cfg, mode="test", async_inputs=True)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    dt, _ = run_forward_bench(dcl, net, data, steps, warmup, False)
    out["headline_workload"] = {"value": round(b * steps / dt, 2), "unit": "frames/s",
                                                     "ms_per_step": round(dt / steps * 1e3, 3)}
    del net
    if ref is not None:
        rcfg, rdata = ref
        rnet = dcl.DCL_Net.Network(rcfg, mode="test", async_inputs=True)
        rnet.load_state_dict(dcl.synth.synth_state_dict(rnet, 1))
        rnet = rnet.to(dev).eval()
        rsteps = max(steps, 20)
        rdt, _ = run_forward_bench(dcl, rnet, rdata, rsteps, max(warmup, 3), False)
        out["ref_shape"] = {"value": round(b * rsteps / rdt, 2), "unit": "frames/s", "ms_per_step": round(rdt / rsteps * 1e3, 3)}
    return out


def whole_forward_rate(n_inp, n_tmp, b, step_s):
    """All MFMA-shaped work of one forward call against the step time (SURVEY.md 8d figures, test mode: the regressor_Xo/Yc
    heads -- 197 376 FLOP per point -- are not executed and not counted): per-point MLPs 3 473 664 FLOP x (N + M), the two
    attention directions 1536 N M; the sparse convolutions (~1 % of this) are left out.  One GPU's share."""
    per_frame = 3473664.0 * (n_inp + n_tmp) + 1536.0 * n_inp * n_tmp
    ach = per_frame * b / step_s / 1e12
    return {"dense_flop_per_frame": per_frame, "achieved": round(ach, 2), "unit": "TFLOP/s per GPU", "peak": round(PEAK_SPLIT, 1),
            "frac": round(ach / PEAK_SPLIT, 4), "x_fp32_mfma_peak": round(ach / PEAK_MFMA_F32, 4),
            "note": "fp32-product work of the whole step / step time.  peak = dense bf16 MFMA peak / 6 (the bound of the split-bf16 "
                    "kernels, which carry the per-point layers); x_fp32_mfma_peak = against what the fp32 matrix pipe could "
                    "do (157.3): beyond 1 since the split kernels"}


def lm_stream_bench(dcl, dev, reps=30, b=1):
    """BASELINE config 4 (S3): LineMOD eval stream -- one object crop per call (tools/test_LM.py:104-112), N=M=1024,
    5 mm voxels (configs/config_LM.yaml:17-20); forward() vs the whole-forward hipGraph replay, inputs resident in HBM."""
    cfg = dcl.synth.default_cfg(1024, 1024, unit=0.005)
    net = dcl.DCL_Net.Network(cfg, mode="test", graph_max_batch=0)      # "eager" below is the plain launch-by-launch call
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
    net = net.to(dev).eval()
    out = {"workload": "LineMOD stream: %d crop%s per call, N=M=1024, 64^3 x 5 mm voxels" % (b, "" if b == 1 else "s")}
    data = to_device(dcl.synth.make_batch(b, 1024, 1024, unit=0.005), dev)
    # (the graph path: a STREAM of calls -- 10 calls to settle behind the capture, then 8 x reps timed; 30 calls right behind
    #  the capture measured 0.47 ms where 300 measure 0.43)
    for name, fn, warm, n in (("eager", lambda: net(data), 3, reps), ("hipgraph", lambda: net.forward_graphed(data), 10, 8 * reps)):
        with torch.no_grad():
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        out[name] = {"ms_per_frame": round(ms, 3), "frames_per_s": round(1e3 / ms, 1), "calls_timed": n}
    ent = next(iter(getattr(net, "_graphs", {}).values()), None)
    out["hipgraph"]["kernel_nodes"] = None if ent is None else ent.get("nodes")
    return out


def eval_stream_bench(dcl, dev, images=40, n_obj=6):
    """The reference's eval loop as ONE pipeline (tools/test_YCBV_stage1.py:173-199; SURVEY 8f-1 + path + 8f-2): per 480x640
    frame with `n_obj` objects  CropBuilder.build -> Network.forward -> ADD-S -> per-class table,  frames resident in HBM
    (decoded ahead, like the forward's own inputs), N = M = 1024, 6 mm voxels.  Three schedules: `serial` = the reference's
    order, one frame after the other on one stream; `pipelined` = frame k+1 is built on a second stream (its two host
    synchronisations and the loader's np.random.choice draws then run underneath frame k's forward; the network waits for the
    builder's ready_event); `prefetch_thread` = crops.CropPrefetcher, the role of the reference's DataLoader workers (:133-137):
    a builder thread keeps two frames ahead of the network's thread.  Same crops, same results.  The ADD-S values stay on the device until the stream ends (one
    read-back for the table).  Also reported: each stage alone (device-synchronised), and the LineMOD-regime variant
    (tools/test_LM.py:104-141: one object per frame, 5 mm voxels, ADD / ADD-S by symmetry flag, success counts)."""
    out = {}
    bstream = torch.cuda.Stream(dev)
    for tag, unit, n_o in (("ycbv", 0.006, n_obj), ("linemod", 0.005, 1)):
        cfg_b = dict(input_size=1024, tmp_size=1024, unit_voxel_extent=[unit] * 3, voxel_num_limit=[64] * 3, voxelization_mode=4)
        frames = [dcl.synth.make_frame(500 + i, n_obj=n_o, tmp_size=1024) for i in range(4)]
        # capacity form: the observed side's voxel rows keep their capacity shape and their count stays on the device -- ONE
        # host read-back per frame (the point counts the loader's random draws need) instead of two; the network's graph
        # path takes that form as it is.  The overflow / range flags of every frame are checked after the stream.
        builder = dcl.crops.CropBuilder(cfg_b, frames[0]["cad_pts"], frames[0]["cad_col"], device=dev, capacity=True)
        res = [dcl.crops.CropBuilder.resident(f["img"], f["depth"], f["label"], dev) for f in frames]
        net = dcl.DCL_Net.Network(dcl.synth.default_cfg(1024, 1024, unit=unit), mode="test")     # default routing: graph replay
        net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
        net = net.to(dev).eval()
        sym = torch.tensor([0, 1] * 16, dtype=torch.int32, device=dev)
        main = torch.cuda.current_stream(dev)

        def build(i):
            f, (im, de, la) = frames[i % 4], res[i % 4]
            return builder.build(im, de, la, f["rois"], f["gt_obj"], poses=f["poses"])

        def build_ahead(i):                                      # on the builder stream; what main will read is marked
            with torch.cuda.stream(bstream):
                d = build(i)
            for side in ("inp", "tmp"):
                for k in ("feats", "occupied_voxels", "v2p_maps", "v0_dev"):
                    if k in d[side]:
                        d[side][k].record_stream(main)
            for k in ("rot_gt", "trans_gt"):
                d["labels"][k].record_stream(main)
            return d

        def metric(data, pred):
            b = pred["rot_pred"].shape[0]
            cld = data["tmp"]["feats"][:, 4:7].reshape(b, 1024, 3)
            if tag == "linemod":
                return dcl.sharding.add_lm(cld, pred["rot_pred"], pred["trans_pred"], data["labels"]["rot_gt"],
                                           data["labels"]["trans_gt"], sym[:b])
            return dcl.sharding.add_s(cld, pred["rot_pred"], pred["trans_pred"], data["labels"]["rot_gt"],
                                      data["labels"]["trans_gt"])

        def tabulate(dist_dev):                                  # the per-class lists of :190-199, from one read-back each
            table = dcl.sharding.AddsTable() if tag == "ycbv" else dcl.sharding.LmTable([0.01] * 22)
            for obj_idx, flags, dd in dist_dev:
                cls = [int(c) for c, f in zip(obj_idx.tolist(), flags.tolist()) if f]
                for c, x in zip(cls, dd.cpu().tolist()):
                    table.add(c, float(x))
            return table
        res_tag = {"workload": "%d object(s) per 480x640 frame, N=M=1024, %g mm voxels" % (n_o, unit * 1e3)}
        with torch.no_grad():
            np.random.seed(1)
            for i in range(6):                                   # warm-up: graph capture, builder caches
                d = build(i)
                metric(d, net(d))
            torch.cuda.synchronize()
            ref_d = None
            for sched in ("serial", "pipelined", "prefetch_thread"):
                np.random.seed(2)
                dist_dev, crops, vi_flags = [], 0, []
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if sched == "serial":
                    for i in range(images):
                        d = build(i)
                        p = net(d)
                        dist_dev.append((d["obj_idx"], d["all_flags"], metric(d, p)))
                        vi_flags.append(d["inp"]["vi_info"])
                        crops += int(p["rot_pred"].shape[0])
                elif sched == "pipelined":
                    d = build_ahead(0)
                    for i in range(images):
                        main.wait_event(d["ready_event"])
                        p = net(d)                                # queued; the host goes on to build the next frame
                        dist_dev.append((d["obj_idx"], d["all_flags"], metric(d, p)))
                        vi_flags.append(d["inp"]["vi_info"])
                        crops += int(p["rot_pred"].shape[0])
                        if i + 1 < images:
                            d = build_ahead(i + 1)
                else:                                             # the loader-worker role: a builder thread, two frames ahead
                    args = ((res[i % 4][0], res[i % 4][1], res[i % 4][2], frames[i % 4]["rois"], frames[i % 4]["gt_obj"],
                             {"poses": frames[i % 4]["poses"]}) for i in range(images))
                    with dcl.crops.CropPrefetcher(builder, args, depth=2, stream=bstream) as feed:
                        for d in feed:
                            p = net(d)
                            dist_dev.append((d["obj_idx"], d["all_flags"], metric(d, p)))
                            vi_flags.append(d["inp"]["vi_info"])
                            crops += int(p["rot_pred"].shape[0])
                torch.cuda.synchronize()
                table = tabulate(dist_dev)
                dt = time.perf_counter() - t0
                assert int(torch.stack(vi_flags)[:, 2].sum()) == 0, "a crop's voxel rows exceeded the builder's v2p_pitch"
                dd = torch.cat([x[2] for x in dist_dev]).cpu()
                if ref_d is None:
                    ref_d = dd
                res_tag[sched] = {"images_per_s": round(images / dt, 1), "crops_per_s": round(crops / dt, 1),
                                  "ms_per_image": round(dt / images * 1e3, 3)}
                if sched != "serial":
                    res_tag[sched]["same_distances_as_serial"] = bool(torch.equal(dd, ref_d))
            # stages alone, device-synchronised
            stage = {}
            builder.draw_seconds = 0.0
            t = time.perf_counter()
            for i in range(20):
                d = build(i)
            torch.cuda.synchronize()
            stage["builder_ms"] = (time.perf_counter() - t) / 20 * 1e3
            stage["builder_ms_of_which_loader_rng_draws_on_the_host"] = builder.draw_seconds / 20 * 1e3
            t = time.perf_counter()
            for _ in range(20):
                p = net(d)
            torch.cuda.synchronize()
            stage["forward_ms"] = (time.perf_counter() - t) / 20 * 1e3
            t = time.perf_counter()
            for _ in range(20):
                metric(d, p)
            torch.cuda.synchronize()
            stage["metric_ms"] = (time.perf_counter() - t) / 20 * 1e3
        res_tag["stages_alone_ms"] = {k: round(v, 3) for k, v in stage.items()}
        res_tag["builder_host_syncs_per_frame"] = 1
        res_tag["builder_note"] = ("capacity-form crops: the one read-back left is the per-instance point counts that the loader's "
                                   "np.random.choice draws need (a seeded run consumes the global generator like the original loader)")
        out[tag] = res_tag
    return out


def refiner_bench(dcl, dev, b, iters=2, reps=20):
    """BASELINE config 5 (S4): the stage-2 refine loop (2 iterations, tools/test_YCBV_stage2.py:214-225) on b crops of
    1024 points, eager vs hipGraph-captured; ms per loop and crops/s."""
    ref = dcl.refiner.Refiner()
    ref.load_state_dict(dcl.synth.synth_state_dict(ref, 2))
    ref = ref.to(dev).eval()
    g = torch.Generator().manual_seed(0)
    n = 1024
    pred = {"F_Xo_p": torch.randn(b, 256, n, generator=g).to(dev), "conf": torch.rand(b, 2 * n, generator=g).to(dev),
            "rot_pred": dcl.ops.ortho9d_to_matrix(torch.randn(b, 9, generator=g).to(dev)),
            "trans_pred": (torch.randn(b, 3, generator=g) * 0.02).to(dev)}
    pts = (torch.randn(b, n, 3, generator=g) * 0.05).to(dev)
    # the features as the fused Network hands them over: point-major storage behind the (b, 256, n) view (no transposing copy in
    # front of the loop's feature GEMM)
    pred_pm = dict(pred, F_Xo_p=pred["F_Xo_p"].transpose(1, 2).contiguous().transpose(1, 2))

    def timed(p, graph):
        best = None
        for _ in range(2):                               # the lower of two runs (a stray first-use cost is not the loop's)
            for _ in range(3):
                dcl.refiner.refine_loop(ref, p, pts, iters, graph=graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                dcl.refiner.refine_loop(ref, p, pts, iters, graph=graph)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            best = ms if best is None else min(best, ms)
        return {"ms_per_loop": round(best, 4), "crops_per_s": round(b / best * 1e3, 1)}
    out = {"features": "channel-first storage (the reference's own layout: one transposing copy per loop)"}
    for name, graph in (("eager", False), ("hipgraph", True)):
        out[name] = timed(pred, graph)
    out["point_major_features"] = {"what": "the (b, 256, n) view over point-major storage that Network.forward hands over (stage2_chain)",
                                   "eager": timed(pred_pm, False), "hipgraph": timed(pred_pm, True)}
    return out


def usable_cores():
    """host cores this process may really use: the affinity mask, capped by the cgroup CPU quota (os.cpu_count() reports
    the machine's 256 even inside a container that owns a handful)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def stage2_chain_bench(dcl, dev, net, data, b, iters=2, reps=20):
    """BASELINE configs[4] as ONE workload: stage-1 forward chained into the 2-iteration refiner loop with pose composition
    (tools/test_YCBV_stage2.py:204-225; refiner.stage2_chain), frames/s including both refine iterations; the refine loop
    eager and as its hipGraph."""
    ref = dcl.refiner.Refiner()
    ref.load_state_dict(dcl.synth.synth_state_dict(ref, 2))
    ref = ref.to(dev).eval()
    out = {"workload": "DCL_Net.forward (bs=%d, N=M=1024) -> %d x Refiner with pose composition" % (b, iters)}
    for name, graph in (("eager_loop", False), ("hipgraph_loop", True)):
        for _ in range(3):
            dcl.refiner.stage2_chain(net, ref, data, iters, graph=graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dcl.refiner.stage2_chain(net, ref, data, iters, graph=graph)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        out[name] = {"ms_per_step": round(ms, 3), "frames_per_s": round(b / ms * 1e3, 1)}
    return out


def cpu_baseline(dcl, sd, cfg, n_inp, n_tmp, budget_s=12.0, chunk=4, max_crops=32):
    """the CPU oracle (kind 'port': the reference has no runnable CPU path) on a bounded sample of
    the same workload, on the host cores this process may use: the C kernels' row loops run under OpenMP, the dense
    algebra is torch-CPU fp32, both on every usable core.  Crops of the workload are processed `chunk` at a time until
    ~budget_s seconds of CPU work are done (at most the whole batch)."""
    from oracle import graph as G
    from oracle import native as oracle_native
    cores = usable_cores()
    torch.set_num_threads(cores)
    oracle_native.set_num_threads(cores)
    done, dt = 0, 0.0
    while done < max_crops and dt < budget_s:
        data = dcl.synth.make_batch(chunk, n_inp, n_tmp, first=done)
        t0 = time.perf_counter()
        G.forward(sd, dict(cfg), data, mode="test")
        dt += time.perf_counter() - t0
        done += chunk
    return {"value": round(done / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d crops of the same workload (N=%d, M=%d), %d per call, through oracle/graph.py: C kernels with "
                      "OpenMP row loops, dense ops torch-CPU fp32, %d threads; %.1f s of CPU work" % (
                          done, n_inp, n_tmp, chunk, cores, dt)}


def conv_pairs_flop(dcl, net, data, dev):
    """algorithmic sparse-conv work of one forward over `data` = 2 * sum_layers pairs * Cin * Cout with the MEASURED pair
    counts of this batch: the module mirrors (spconv shim) build every layer's gather table once, its
    non-negative entries are the rulebook pairs.  Both backbones.  -> (flop, pairs per layer list)"""
    import numpy as np_
    vlim = np_.asarray(data["voxel_num_limit"]).astype(np_.int64)
    b = int(data["batch_offsets"].size(0)) - 1
    flop, per_layer = 0.0, []
    with torch.no_grad():
        for side, bb in (("inp", net.backbone_inp), ("tmp", net.backbone_tmp)):
            feats = data[side]["feats"].to(dev).float().contiguous()
            v2p = data[side]["v2p_maps"].to(dev).int().contiguous()
            occ = data[side]["occupied_voxels"].to(dev).int().contiguous()
            x = dcl.spconv.SparseConvTensor(dcl.ops.voxelize_fp(feats, v2p, 4), occ, vlim, b)
            for m in range(1, 5):
                for blk in getattr(bb, "module%d" % m):
                    conv = blk.layers[0]
                    aset = x.active_set()
                    out_set, nbr = dcl.spconv.ops.build_rulebook(aset, 3, 1, 1, conv.subm)
                    n_out = x.indices.shape[0] if conv.subm else out_set.n
                    pairs = int((nbr[:, :n_out] >= 0).sum().item())
                    per_layer.append((side, conv.in_channels, conv.out_channels, bool(conv.subm), n_out, pairs))
                    flop += 2.0 * pairs * conv.in_channels * conv.out_channels
                    x = blk(x)
                x = bb.pool(x)
    return flop, per_layer


PEAK_L2_GATHER = 17800.0       # GB/s chip-wide, rows gathered from the XCDs' L2 into LDS (MI355X_MICROARCH.md "Indexed rows: gather into LDS": 16.8-18.8 TB/s)


def sparse_conv_roofline(dcl, net, data, dev, steps=3):
    """roofline entry of the sparse-conv kernel family for one workload: algorithmic flop (measured pairs, above) divided by
    the summed device time of all conv calls of a forward, measured with HIP events on the launch streams inside the
    library (dcl_profile_conv_begin / _end_calls) during `steps` ordinary forwards run on ONE stream (net.single_stream, so that
    the two backbones' kernels do not overlap each other inside the bracketed intervals).  layers[]: the same run call by
    call -- per layer (both backbones in one grouped launch) its time, useful flop, gathered bytes (pairs x Cin x 4: every
    pair's input row is fetched once per column tile at least) and the bound that binds it: the fp32 MFMA peak or the rate at
    which rows can be gathered from L2 into LDS."""
    import ctypes
    flop, per_layer = conv_pairs_flop(dcl, net, data, dev)
    lib = dcl._native.lib()
    old, old_pair = net.single_stream, net._pair_features
    net.single_stream = True
    timed, calls_of = {}, {}
    CAP = 16 * steps + 16
    try:
        # one stream; first as the library schedules one stream (both backbones' layers grouped: 8 launches per forward),
        # then with each backbone's own launches (16): the second figure is the one comparable with earlier rounds
        for name, pair in (("grouped", None), ("separate", False)):
            net._pair_features = pair
            with torch.no_grad():
                net(data)
                torch.cuda.synchronize()
                lib.dcl_profile_conv_begin()
                for _ in range(steps):
                    net(data)
                torch.cuda.synchronize()
                ms, calls = ctypes.c_double(0), ctypes.c_int32(0)
                per_ms, what = (ctypes.c_float * CAP)(), (ctypes.c_int32 * (4 * CAP))()
                lib.dcl_profile_conv_end_calls(ctypes.byref(ms), ctypes.byref(calls), per_ms, what, CAP)
            timed[name] = (ms.value / steps, int(calls.value))
            calls_of[name] = [(float(per_ms[i]), tuple(int(what[4 * i + q]) for q in range(4))) for i in range(min(int(calls.value), CAP))]
    finally:
        net.single_stream, net._pair_features = old, old_pair
    ms_fwd = timed["grouped"][0]
    ach = flop / (ms_fwd * 1e-3) / 1e12 if ms_fwd > 0 else float("nan")
    issued = sum(2.0 * 27 * n_out * ci * co for (_, ci, co, _, n_out, _) in per_layer)
    # per layer: the calls of the grouped run in call order are layer 0..7 of both sides, `steps` times over
    layers = []
    nl = len(per_layer) // 2
    per_fwd = timed["grouped"][1] // steps if steps else 0
    if per_fwd == nl:
        for i in range(nl):
            (_, ci, co, subm, n_a, p_a), (_, _, _, _, n_b, p_b) = per_layer[i], per_layer[nl + i]
            t = [calls_of["grouped"][k * nl + i] for k in range(steps)]
            assert all(w[0] == ci and w[1] == co and w[2] == int(subm) for _, w in t), (t, ci, co, subm)
            ms_l = sum(x for x, _ in t) / steps
            fl = 2.0 * (p_a + p_b) * ci * co
            gath = 4.0 * (p_a + p_b) * ci
            t_mfma, t_gather = fl / (PEAK_MFMA_F32 * 1e12) * 1e3, gath / (PEAK_L2_GATHER * 1e9) * 1e3
            layers.append({"layer": "L%d %s %d->%d" % (i // 2, "subm" if subm else "conv", ci, co), "rows": n_a + n_b,
                           "pairs": p_a + p_b, "density": round((p_a + p_b) / (27.0 * (n_a + n_b)), 3) if n_a + n_b else None,
                           "ms": round(ms_l, 4), "useful_flop": fl, "TFLOPs": round(fl / (ms_l * 1e-3) / 1e12, 2) if ms_l > 0 else None,
                           "gathered_bytes": gath, "flop_per_gathered_byte": round(fl / gath, 1) if gath else None,
                           "mfma_bound_ms": round(t_mfma, 4), "l2_gather_bound_ms": round(t_gather, 4),
                           "bound": "mfma" if t_mfma >= t_gather else "l2-gather",
                           "frac_of_bound": round(max(t_mfma, t_gather) / ms_l, 4) if ms_l > 0 else None})
    ms_sep = timed["separate"][0]
    ach_sep = flop / (ms_sep * 1e-3) / 1e12 if ms_sep > 0 else float("nan")
    # `achieved` / `frac` = the schedule the DEFAULT forward runs: each backbone's own launches; the grouped
    # one-stream schedule (Network(single_stream=True)) under its own key, with the per-layer table taken on it
    return {"kernel": "k_sparse_conv_* (8 conv layers x 2 backbones)", "bound": "mfma", "achieved": round(ach_sep, 2),
            "peak": PEAK_MFMA_F32, "unit": "TFLOP/s", "frac": round(ach_sep / PEAK_MFMA_F32, 4),
            "flop_per_forward": flop, "conv_ms_per_forward": round(ms_sep, 4), "conv_calls_timed": timed["separate"][1],
            "rulebook_density": round(flop / issued, 4) if issued else None,
            "pairs_per_forward": int(sum(p[5] for p in per_layer)),
            "schedule": "each backbone's own launches (16 conv launches per forward: what the default two-stream forward "
                        "issues), timed on one stream by HIP events inside the library",
            "separate_launches": {"conv_ms_per_forward": round(ms_sep, 4), "conv_calls_timed": timed["separate"][1],
                                  "frac": round(ach_sep / PEAK_MFMA_F32, 4) if ms_sep > 0 else None,
                                  "what": "= the headline figures above (key kept for comparison with earlier rounds)"},
            "grouped_single_stream": {
                "conv_ms_per_forward": round(ms_fwd, 4), "conv_calls_timed": timed["grouped"][1], "achieved": round(ach, 2),
                "frac": round(ach / PEAK_MFMA_F32, 4),
                "schedule": "one stream; every layer of the two backbones as ONE grouped launch (what Network(single_stream=True) "
                            "runs); rows of the two deep levels ordered on the device (the ordering launches run in the geometry "
                            "stage, outside the timed conv calls: 2 launches per backbone, see profiles/)"},
            "layers": layers,
            "layers_note": "per layer of the GROUPED schedule (both backbones per launch).  bound per layer: max(useful flop / fp32 "
                           "MFMA peak, pairs x Cin x 4 B / %.1f TB/s L2->LDS gather rate); frac_of_bound = that time / measured time"
                           % (PEAK_L2_GATHER / 1e3),
            "feature_stage": dict(feature_stage_times(dcl, net, data, dev),
                                  what="convs + pools of both backbones stand-alone: a launch per layer and side / one grouped "
                                       "launch per layer")}


def feature_stage_times(dcl, net, data, dev, reps=10):
    """The sparse feature stage of BOTH backbones (8 convs + 4 pools each) on this batch, stand-alone, two ways: one launch
    per layer and side, and one grouped launch per layer (dcl_backbone_features_pair).  HIP events around the calls."""
    ops = dcl.ops
    f = net._fold()
    b = int(data["batch_offsets"].size(0)) - 1
    runs, xs, ptrs = [], [], []
    with torch.no_grad():
        for s in ("inp", "tmp"):
            occ = data[s]["occupied_voxels"].to(dev).int().contiguous()
            xs.append(ops.voxelize_fp(data[s]["feats"].to(dev).float().contiguous(), data[s]["v2p_maps"].to(dev).int().contiguous(), 4))
            run = ops.BackboneRun(occ, b, 64)
            run.set_counts(run.counts_dev.cpu().tolist())
            runs.append(run)
            ptrs.append(f["backbone_%s_ptrs" % s])

        def per_layer():
            for r, x, p in zip(runs, xs, ptrs):
                r.features(x, *p)

        def pair():
            ops.backbone_features_pair(runs[0], xs[0], ptrs[0], runs[1], xs[1], ptrs[1])

        out = {}
        for name, fn in (("per_layer_side_after_side", per_layer), ("per_layer_grouped", pair)):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out[name + "_ms"] = round(e0.elapsed_time(e1) / reps, 4)
    return out


def collective_report(distributed, rank, world, mine):
    """what the collective library saw, gathered from every rank (outside the timed region): world size and backend as
    torch.distributed reports them (not WORLD_SIZE echoed), and per rank its device and the host cores it is pinned to"""
    if not distributed:
        return {"world": 1, "backend": None, "initialized": False, "device_per_rank": [mine["device"]],
                "device_name": mine.get("name"), "cpus_per_rank": [_cpu_ranges(mine["cpus"])]}
    per = [None] * dist.get_world_size()
    dist.all_gather_object(per, dict(mine, rank=rank))
    per.sort(key=lambda d: d["rank"])
    return {"world": dist.get_world_size(), "backend": dist.get_backend(), "initialized": True,
            "device_per_rank": [d["device"] for d in per], "device_name": per[0].get("name"),
            "cpus_per_rank": [_cpu_ranges(d["cpus"]) for d in per]}


def _cpu_ranges(cpus):
    out, cpus = [], sorted(cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def dry_run(args, rank, local_rank, world, cpus):
    """`--dry-run`: the N-rank plumbing of this script WITHOUT a GPU (tests/test_host.py runs it under torch.distributed.run
    with 2 processes): rank / world from the launcher, host-core placement, a gloo group, the barrier-bracketed timing loop
    with the max over ranks, disjoint crops per rank, the exact metric reduction -- around an EMPTY step (no forward runs:
    the product path needs the GPU and there is no CPU fallback), so `value` is null and the line says "dry_run": true."""
    dcl = importlib.import_module("dcl-net_amd")
    distributed = world > 1
    if distributed:
        dist.init_process_group("gloo")
    b = args.batch
    n_inp, n_tmp = SHAPES[args.shape] if args.shape in SHAPES else SHAPES["ref"]
    n_inp, n_tmp = min(n_inp, 64), min(n_tmp, 64)              # tiny crops: only their class ids / poses are used
    host_data = dcl.synth.make_batch(b, n_inp, n_tmp, first=rank * b, voxelize_idx=lambda c, bs, mode: (
        torch.zeros((1, 4), dtype=torch.int64), torch.zeros(c.shape[0], dtype=torch.int32), torch.zeros((1, 2), dtype=torch.int32)))
    if distributed:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass                                                   # the step: nothing (see above)
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    table = dcl.sharding.AddsTable()
    for i, c in enumerate(host_data["obj_idx"].tolist()):      # a deterministic stand-in distance per crop
        table.add(int(c), 0.001 * ((rank * b + i) % 97))
    table.reduce()
    auc, _, _, _ = table.finalize()
    assert int(table.sums[:, 0].sum()) == world * b, "metric reduction lost frames"
    rccl = collective_report(distributed, rank, world, {"device": "cpu (dry run; cuda:%d on a GPU box)" % local_rank,
                                                        "name": None, "cpus": cpus})
    line = {"metric": "frames/sec DCL_Net.forward @ YCB-V bs32", "value": None, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
            "config": {"workload": "dry run of the rank plumbing (no forward)", "global_batch": world * b,
                       "frames_per_step_per_gpu": b, "parallelism": "frames sharded x%d" % world},
            "loop_seconds_max_over_ranks": dt, "adds_auc_stand_in": auc,
            "metric_frames_reduced": int(table.sums[:, 0].sum()), "rccl": rccl}
    if distributed:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def main():
    _flush_c_stdio(unbuffer=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape", choices=list(SHAPES), default="stress")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-extras", action="store_true", help="skip ref-shape / primitives / cpu baseline legs")
    ap.add_argument("--pipelined-calls", action="store_true",
                    help="let back-to-back forward calls overlap on the GPU (Network(async_inputs=True))")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run PMC passes behind roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true",
                    help="(internal) one warm-up + one forward of the workload and exit: the body of the PMC passes")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the launcher / rank / affinity / sharding / metric-reduction plumbing over gloo with an "
                         "empty step; prints the same JSON line with \"dry_run\": true and no rate (CPU test of the N > 1 path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.pmc_child:
        # called without a launcher: start one rank per GPU as a CHILD (nothing has touched the GPU yet; never exec) and
        # hand its output and exit code on
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29517"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or os.environ.get("DCL_FORCE_DIST") == "1"     # the latter: exercise the RCCL path on 1 GPU
    # one process per GPU: before anything touches the GPU, every rank moves onto the host cores of its GPU's NUMA node
    # (KFD topology in sysfs; an even split of the allowed cores if that cannot be read) -- sharding.rank_cpu_set
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cpus = importlib.import_module("dcl-net_amd").sharding.pin_rank_to_gpu_numa(local_rank, local_world)
    if args.dry_run:
        return dry_run(args, rank, local_rank, world, cpus)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        if world == 1:                                         # DCL_FORCE_DIST=1 without a launcher: a one-rank RCCL group
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)         # RCCL over xGMI
    dcl = importlib.import_module("dcl-net_amd")
    b = args.batch
    n_inp, n_tmp = SHAPES[args.shape]
    cfg = dcl.synth.default_cfg(n_inp, n_tmp)
    # headline: strictly serial forward calls (the attention roofline is then measured on an otherwise idle GPU);
    # --pipelined-calls lets back-to-back calls overlap (async_inputs: sparse half of call k+1 under the dense half of
    # call k) -- reported as an extra at N=1
    net = dcl.DCL_Net.Network(cfg, mode="test", async_inputs=args.pipelined_calls,
                              **({"graph_max_batch": 0} if args.pmc_child else {}))   # counter passes: separate launches
    sd = dcl.synth.synth_state_dict(net, 1)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    host_data = dcl.synth.make_batch(b, n_inp, n_tmp, first=rank * b)          # disjoint crops per rank
    data = to_device(host_data, dev)
    if args.pmc_child:
        with torch.no_grad():
            net(data)
            torch.cuda.synchronize()
            time.sleep(0.05)                              # the pause measure_kernel_shares cuts the trace at
            net(data)
        torch.cuda.synchronize()
        return

    dt, att_ms = run_forward_bench(dcl, net, data, args.steps, args.warmup, distributed)
    graphed = bool(net.__dict__.get("_graphs"))          # calls this small in points replay a whole-forward hipGraph (default)
    net_l = net
    if graphed:                                          # per-kernel timings (attention events, conv hooks) need separate launches
        net_l = dcl.DCL_Net.Network(cfg, mode="test", async_inputs=args.pipelined_calls, graph_max_batch=0)
        net_l.load_state_dict(sd)
        net_l = net_l.to(dev).eval()
        _, att_ms = run_forward_bench(dcl, net_l, data, max(5, args.steps // 2), 2, False)
    frames = world * b * args.steps
    value = frames / dt

    # roofline of the dominant hand-written kernel: the correspondence attention (2 launches per forward)
    flop_dir = [2.0 * (64 + 320) * n_inp * n_tmp * b] * 2     # dir 1: nq=N,nk=M ; dir 2: nq=M,nk=N -- same product
    att_avg_ms = float(np.mean(att_ms)) if att_ms else float("nan")
    achieved = flop_dir[0] / (att_avg_ms * 1e-3) / 1e12 if att_ms else float("nan")
    traffic, traffic_source = None, "not collected (N > 1, --no-extras or --no-traffic)"
    if rank == 0 and world == 1 and not args.no_extras and not args.no_traffic:
        traffic, traffic_source = measure_traffic_bytes("k_cross_attn", args.shape, b)
    # since round 6 the large calls run k_cross_attn_split: both products (S = K Q^T and P.V) as six bf16 piece products per fp32
    # product, so the kernel's fp32-equivalent bound is the dense bf16 MFMA peak / 6
    roofline = {"kernel": "k_cross_attn_split (+ k_attn_split_v / _k: the V and K piece passes, inside the timed span)", "bound": "mfma",
                "achieved": round(achieved, 2), "peak": round(PEAK_SPLIT, 1),
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_SPLIT, 4), "x_fp32_mfma_peak": round(achieved / PEAK_MFMA_F32, 4),
                "peak_note": "every fp32 product of the attention = six bf16 piece products (three exact bf16 pieces per operand, fp32 "
                             "accumulators): peak = 2500 / 6; x_fp32_mfma_peak = achieved / 157.3, the bound of the all-fp32 kernel "
                             "this replaces (which ran at 0.866 of it)",
                "traffic": traffic, "traffic_source": traffic_source,
                "traffic_algorithmic": int(b * (n_inp + n_tmp) * (4 * 64 + (6 * 64 + 6 * 320 + 4 * 320) // 2)),
                "traffic_note": "main kernel only, average of the two directions: Q fp32 (re-read from L2 per key tile, once from HBM), "
                                "K and V as three bf16 pieces (6 B per value), O fp32; the piece passes read K, V (4 B) and write the "
                                "pieces (6 B) once per launch on top",
                "avg_launch_ms": round(att_avg_ms, 4), "launches_timed": len(att_ms),
                "flop_per_launch": flop_dir[0]}
    # whose kernels the step's GPU time goes to: since round 6 the per-point linear layers run on the
    # library's OWN fp32 MFMA GEMM core (csrc/linear_dma.hip; the last fuser layer with the pooling as its epilogue) -- no
    # vendor GEMM is left on the forward, `vendor_gemm` reports what the trace still finds of it (expected: nothing)
    roofline["share_of_gpu_time"], roofline["own_gemm"], roofline["vendor_gemm"] = None, None, None
    roofline["share_source"] = "not collected (N > 1, --no-extras or --no-traffic)"
    if rank == 0 and world == 1 and not args.no_extras and not args.no_traffic:
        shares, roofline["share_source"] = measure_kernel_shares(args.shape, b)
        if shares is not None:
            att, gem = shares["attention (k_cross_attn*)"], shares["vendor_gemm (hipBLASLt Cijk_*)"]
            own = shares[OWN_GEMM]
            gemm_flop = 3473664.0 * b * (n_inp + n_tmp) * shares["_forwards"]       # test-mode dense flop per point
            roofline["share_of_gpu_time"] = round(att["share"], 4)
            own_tf = gemm_flop / (own["ns"] * 1e-9) / 1e12 if own["ns"] > 0 else None
            roofline["own_gemm"] = {"kernels": "k_linear_split<store | pooling | row-dot epilogue> (split-bf16: launches of >= 192 tiles), "
                                               "k_linear_dma<128,128 | 128,64 | 64,64> (fp32 MFMA: the rest), k_linear_group, k_mlp128_to1",
                                    "share_of_gpu_time": round(own["share"], 4), "bound": "mfma",
                                    "achieved": round(own_tf, 1) if own_tf else None, "peak": round(PEAK_SPLIT, 1), "unit": "TFLOP/s",
                                    "frac": round(own_tf / PEAK_SPLIT, 4) if own_tf else None,
                                    "x_fp32_mfma_peak": round(own_tf / PEAK_MFMA_F32, 3) if own_tf else None,
                                    "launches_per_forward": own["launches"] // shares["_forwards"],
                                    "note": "every 1x1x1-conv / head layer of the dense half: SURVEY 8d's dense flop per point (fp32 "
                                            "products, counted once) over the family's summed kernel time in the traced forward.  The "
                                            "big launches compute an fp32 product as six bf16 piece products (three exact bf16 pieces "
                                            "per operand, fp32 accumulators: errors against float64 the size of the fp32 core's), so "
                                            "peak = the dense bf16 MFMA peak / 6; x_fp32_mfma_peak = achieved / 157.3 (what the "
                                            "fp32 matrix pipe could do at most)"}
            roofline["vendor_gemm"] = {"share_of_gpu_time": round(gem["share"], 4),
                                       "launches_per_forward": gem["launches"] // shares["_forwards"],
                                       "note": "hipBLASLt Tensile kernels (Cijk_*) found in the traced forward: none expected since round 6"}
            sp_key = [k for k in shares if k.startswith("sparse feature stage")][0]
            roofline["sparse_stage_share_of_gpu_time"] = round(shares[sp_key]["share"], 4)     # convs + combines + pools
            roofline["note"] = ("k_cross_attn is the dominant single kernel; the own GEMM family follows -- both hand-written MFMA "
                                "kernels of this library")

    # metric reduction over RCCL (outside the timed region): ADD-S table of this rank's crops
    with torch.no_grad():
        pred = net(data)
        cld = host_data["tmp"]["feats"][:, 4:7].reshape(b, n_tmp, 3).to(dev)
        d = dcl.sharding.add_s(cld, pred["rot_pred"], pred["trans_pred"], host_data["labels"]["rot_gt"].to(dev),
                               host_data["labels"]["trans_gt"].to(dev)).cpu().numpy()
    table = dcl.sharding.AddsTable()
    for c, x in zip(host_data["obj_idx"].tolist(), d.tolist()):
        table.add(int(c), float(x))
    table.reduce(device=dev)
    auc, acc2, _, _ = table.finalize()
    assert int(table.sums[:, 0].sum()) == world * b, "metric reduction lost frames"
    rccl = collective_report(distributed, rank, world, {"device": "cuda:%d" % local_rank,
                                                        "name": torch.cuda.get_device_name(dev), "cpus": cpus})

    line = {"metric": "frames/sec DCL_Net.forward @ YCB-V bs32", "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (exact 3xbf16-split products on the bf16 MFMA, f32 accumulate)",
            "data": "synthetic",
            "dtype_note": "fp32 tensors, fp32 accumulators everywhere.  Sparse convs: fp32 MFMA.  The big per-point linear layers (and, "
                          "where noted under `roofline`, the attention) form each fp32 product from the EXACT three-way bf16 split of "
                          "both operands -- six bf16 piece products of weight >= 2^-16 on the bf16 MFMA, dropped terms <= 2^-25 of the "
                          "product -- with errors against float64 equal to an fp32 FMA chain's (tests/test_gpu_ops.py: split core "
                          "beside the fp32 core; every golden at the reference tolerances 1e-4 / 1e-5)",
            "config": {"workload": "YCB-V bs=32 (config_YCBV_bs32.yaml), N=%d observed / M=%d model points per crop, "
                                   "64^3 x 6 mm voxels; shape=%s" % (n_inp, n_tmp, args.shape),
                       "global_batch": world * b, "frames_per_step_per_gpu": b, "parallelism": "frames sharded x%d" % world,
                       "weights": "seeded random (no checkpoints offline)",
                       "calls": "pipelined (async_inputs)" if args.pipelined_calls else "serial",
                       "path": "whole-forward hipGraph replay" if graphed else "launch by launch"},
            "roofline": roofline,
            "whole_forward": whole_forward_rate(n_inp, n_tmp, b, dt / args.steps),
            "adds_auc_synthetic_weights": auc, "metric_frames_reduced": int(table.sums[:, 0].sum()), "rccl": rccl}
    rdata_for_pipe = None
    if rank == 0 and world == 1 and not args.no_extras:
        line["roofline_sparse_conv"] = {args.shape: sparse_conv_roofline(dcl, net_l, data, dev)}
        # the conv family's share of the STEP: event-bracketed conv time of a forward (each backbone's own launches, what the
        # timed forward issues) over the measured step time -- not a share of summed kernel time under a profiler
        rsc = line["roofline_sparse_conv"][args.shape]
        rsc["share_of_step_time"] = round(rsc["separate_launches"]["conv_ms_per_forward"] / (dt / args.steps * 1e3), 4)
        if args.shape != "ref":
            rn, rm = SHAPES["ref"]
            rcfg = dcl.synth.default_cfg(rn, rm)
            # two instances: `rnet` as a caller constructs it (calls this small in points replay the whole-forward hipGraph,
            # Network.__init__), `rnet_l` launch by launch -- what the per-kernel timings (attention events, conv hooks)
            # need, and the number the graph replay is compared with
            rnet = dcl.DCL_Net.Network(rcfg, mode="test", async_inputs=args.pipelined_calls)
            rnet_l = dcl.DCL_Net.Network(rcfg, mode="test", async_inputs=args.pipelined_calls, graph_max_batch=0)
            for m_ in (rnet, rnet_l):
                m_.load_state_dict(dcl.synth.synth_state_dict(m_, 1))
            rnet, rnet_l = rnet.to(dev).eval(), rnet_l.to(dev).eval()
            rdata = to_device(dcl.synth.make_batch(b, rn, rm), dev)
            rsteps = max(args.steps, 20)
            rdt, _ = run_forward_bench(dcl, rnet, rdata, rsteps, max(args.warmup, 3), False)
            ldt, _ = run_forward_bench(dcl, rnet_l, rdata, rsteps, max(args.warmup, 3), False)
            # (the attention launches are timed by events around each of them: with the tail's two directions side by side --
            # what a launch-by-launch call of this shape does too -- two launches would share one interval; timed serially)
            rnet_l.PAR_TAIL = False
            _, ratt = run_forward_bench(dcl, rnet_l, rdata, rsteps, max(args.warmup, 3), False)
            rnet_l.PAR_TAIL = None
            rflop = 2.0 * (64 + 320) * rn * rm * b
            graphed = bool(rnet.__dict__.get("_graphs"))
            line["ref_shape"] = {"workload": "N=M=1024 (what config_YCBV_bs32.yaml defines), bs=32",
                                 "value": round(b * rsteps / rdt, 2), "unit": "frames/s",
                                 "ms_per_step": round(rdt / rsteps * 1e3, 3),
                                 "path": "whole-forward hipGraph replay (default for calls of <= 98304 points)" if graphed
                                         else "launch by launch",
                                 "launch_by_launch": {"value": round(b * rsteps / ldt, 2), "ms_per_step": round(ldt / rsteps * 1e3, 3)},
                                 "attention_TFLOPs": round(rflop / (np.mean(ratt) * 1e-3) / 1e12, 2) if ratt else None}
            line["roofline_sparse_conv"]["ref"] = sparse_conv_roofline(dcl, rnet_l, rdata, dev)
            line["roofline_sparse_conv"]["ref"]["share_of_step_time"] = round(
                line["roofline_sparse_conv"]["ref"]["separate_launches"]["conv_ms_per_forward"] / (ldt / rsteps * 1e3), 4)
            # BASELINE configs[2]'s per-GPU shape: 40 crops per call (config_YCBV_bs40.yaml)
            d40 = to_device(dcl.synth.make_batch(40, rn, rm), dev)
            dt40, _ = run_forward_bench(dcl, rnet, d40, 20, 3, False)
            lt40, _ = run_forward_bench(dcl, rnet_l, d40, 20, 3, False)
            line["bs40"] = {"workload": "N=M=1024, bs=40 (config_YCBV_bs40.yaml batch)", "unit": "frames/s",
                            "value": round(40 * 20 / dt40, 2), "ms_per_step": round(dt40 / 20 * 1e3, 3),
                            "launch_by_launch": {"value": round(40 * 20 / lt40, 2), "ms_per_step": round(lt40 / 20 * 1e3, 3)}}
            line["stage2_chain"] = stage2_chain_bench(dcl, dev, rnet, rdata, b)
            rdata_for_pipe = (rcfg, rdata)
            del rnet, rnet_l, d40
        # the same forward fed from the loader's HOST tensors (pageable memory, H2D inside forward) -- never `value`
        hdt, _ = run_forward_bench(dcl, net, host_data, max(3, args.steps // 2), 1, False)
        hsteps = max(3, args.steps // 2)
        # ... and from PINNED host tensors (what DataLoader(pin_memory=True) hands over; the copies inside forward() are then
        # real asynchronous DMAs instead of staged pageable copies)
        def pin(v):
            return v.pin_memory() if torch.is_tensor(v) and not v.is_cuda else v
        pinned = {k: ({kk: pin(vv) for kk, vv in v.items()} if isinstance(v, dict) else pin(v)) for k, v in host_data.items()}
        pinned["labels"] = {}
        pdt, _ = run_forward_bench(dcl, net, pinned, max(3, args.steps // 2), 1, False)
        line["h2d_inclusive"] = {"value": round(b * hsteps / pdt, 2), "unit": "frames/s",
                                 "ms_per_step": round(pdt / hsteps * 1e3, 3),
                                 "what": "data dict in PINNED host memory (a pinning loader), uploaded inside forward()",
                                 "pageable": {"value": round(b * hsteps / hdt, 2), "ms_per_step": round(hdt / hsteps * 1e3, 3),
                                              "what": "the same from pageable host tensors (pin_memory: False in config_YCBV_bs32.yaml:47,62 -- what the shipped configs hand over)"}}
        if not args.pipelined_calls:
            line["pipelined_calls"] = pipelined_bench(dcl, dev, sd, cfg, data, b, args.steps, args.warmup, rdata_for_pipe)
        line["lm_stream"] = lm_stream_bench(dcl, dev)
        line["eval_stream"] = eval_stream_bench(dcl, dev)
        line["primitives"] = primitives_roofline(dcl)
        line["refiner"] = refiner_bench(dcl, dev, b)
        line["cpu_baseline"] = cpu_baseline(dcl, sd, cfg, n_inp, n_tmp)
    if world > 1 and not args.no_extras:
        # BASELINE configs[2] ("YCB-V bs=40, 8 x MI355X frame-sharded"): every rank runs 40 crops of the shipped shape per call
        # (N = M = 1024), same barrier-bracketed timing, MAX over ranks -- so that the first multi-GPU run measures the config
        # BASELINE names and not only the stress one.  Every rank takes part (collectives inside).
        rn, rm = SHAPES["ref"]
        net40 = dcl.DCL_Net.Network(dcl.synth.default_cfg(rn, rm), mode="test")
        net40.load_state_dict(dcl.synth.synth_state_dict(net40, 1))
        net40 = net40.to(dev).eval()
        d40 = to_device(dcl.synth.make_batch(40, rn, rm, first=40 * rank), dev)
        dt40, _ = run_forward_bench(dcl, net40, d40, 20, 3, distributed)
        line["bs40_per_rank"] = {"workload": "N=M=1024, bs=40 per GPU x %d GPUs (config_YCBV_bs40.yaml batch, frames sharded)" % world,
                                 "unit": "frames/s", "value": round(world * 40 * 20 / dt40, 2), "n_gpus": world,
                                 "ms_per_step": round(dt40 / 20 * 1e3, 3), "scaling": "weak"}
        del net40, d40
    # the JSON line is the LAST thing on stdout: RCCL writes its banner ("RCCL version ...", "Librccl path ...") through C
    # stdio, which on a pipe is only flushed at exit -- i.e. behind a line printed here.  C stdout was made unbuffered in
    # main(); every rank flushes once more, and rank 0 prints after a barrier behind those flushes.
    _flush_c_stdio()
    if distributed:
        dist.barrier()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

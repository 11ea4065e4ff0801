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

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32 = 157.3          # TFLOP/s, MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM = 8000.0              # GB/s spec
SHAPES = {"stress": (12288, 2048), "ref": (1024, 1024)}


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per launch of a kernel from the committed PMC passes (profiles/r*_pmc_summary.csv, separate
    rocprofv3 --pmc runs of this same bench at the stress shape): FETCH_SIZE is doubled (gfx950 reports half of a wide
    coalesced read, MI355X_MICROARCH.md HBM section), both counters are KiB.  None if no summary is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.csv")))
    if not files:
        return None
    fetch = write = None
    for r in csv.DictReader(open(files[-1])):
        if kernel_substr in r["kernel"]:
            if r["counter"] == "FETCH_SIZE":
                fetch = float(r["avg_value"])
            elif r["counter"] == "WRITE_SIZE":
                write = float(r["avg_value"])
    if fetch is None or write is None:
        return None
    return int((2.0 * fetch + write) * 1024)


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


def lm_stream_bench(dcl, dev, reps=30):
    """BASELINE config 4 (S3): LineMOD eval stream -- one object crop per call (tools/test_LM.py:104-112), N=M=1024,
    5 mm voxels (configs/config_LM.yaml:17-20); forward() vs the whole-forward hipGraph replay, inputs resident in HBM."""
    cfg = dcl.synth.default_cfg(1024, 1024, unit=0.005)
    net = dcl.DCL_Net.Network(cfg, mode="test")
    net.load_state_dict(dcl.synth.synth_state_dict(net, 1))
    net = net.to(dev).eval()
    out = {"workload": "LineMOD stream: 1 crop per call, N=M=1024, 64^3 x 5 mm voxels"}
    data = to_device(dcl.synth.make_batch(1, 1024, 1024, unit=0.005), dev)
    for name, fn in (("eager", lambda: net(data)), ("hipgraph", lambda: net.forward_graphed(data))):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        out[name] = {"ms_per_frame": round(ms, 3), "frames_per_s": round(1e3 / ms, 1)}
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
    out = {}
    for name, graph in (("eager", False), ("hipgraph", True)):
        for _ in range(3):
            dcl.refiner.refine_loop(ref, pred, pts, iters, graph=graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            dcl.refiner.refine_loop(ref, pred, pts, iters, graph=graph)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        out[name] = {"ms_per_loop": round(ms, 4), "crops_per_s": round(b / ms * 1e3, 1)}
    return out


def cpu_baseline(dcl, sd, cfg, n_inp, n_tmp, crops=1):
    """the CPU oracle (kind 'port': the reference has no runnable CPU path) on a bounded sample of
    the same workload, host cores of this box."""
    from oracle import graph as G
    torch.set_num_threads(os.cpu_count() or 1)
    data = dcl.synth.make_batch(crops, n_inp, n_tmp)
    t0 = time.perf_counter()
    G.forward(sd, dict(cfg), data, mode="test")
    dt = time.perf_counter() - t0
    return {"value": round(crops / dt, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d crops of the same workload (N=%d, M=%d), oracle/graph.py: C kernels single-threaded, dense "
                      "ops torch-CPU fp32 on %d threads; %.1f s" % (crops, n_inp, n_tmp, torch.get_num_threads(), dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape", choices=list(SHAPES), default="stress")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-extras", action="store_true", help="skip ref-shape / primitives / cpu baseline legs")
    ap.add_argument("--pipelined-calls", action="store_true",
                    help="let back-to-back forward calls overlap on the GPU (Network(async_inputs=True))")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or os.environ.get("DCL_FORCE_DIST") == "1"     # the latter: exercise the RCCL path on 1 GPU
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
    net = dcl.DCL_Net.Network(cfg, mode="test", async_inputs=args.pipelined_calls)
    sd = dcl.synth.synth_state_dict(net, 1)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    host_data = dcl.synth.make_batch(b, n_inp, n_tmp, first=rank * b)          # disjoint crops per rank
    data = to_device(host_data, dev)

    dt, att_ms = run_forward_bench(dcl, net, data, args.steps, args.warmup, distributed)
    frames = world * b * args.steps
    value = frames / dt

    # roofline of the dominant hand-written kernel: the correspondence attention (2 launches per forward)
    flop_dir = [2.0 * (64 + 320) * n_inp * n_tmp * b] * 2     # dir 1: nq=N,nk=M ; dir 2: nq=M,nk=N -- same product
    att_avg_ms = float(np.mean(att_ms)) if att_ms else float("nan")
    achieved = flop_dir[0] / (att_avg_ms * 1e-3) / 1e12 if att_ms else float("nan")
    roofline = {"kernel": "k_cross_attn", "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_MFMA_F32,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_MFMA_F32, 4),
                "traffic": pmc_traffic_bytes("k_cross_attn") if args.shape == "stress" else None,
                "avg_launch_ms": round(att_avg_ms, 4), "launches_timed": len(att_ms),
                "flop_per_launch": flop_dir[0]}

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

    line = {"metric": "frames/sec DCL_Net.forward @ YCB-V bs32", "value": round(value, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "YCB-V bs=32 (config_YCBV_bs32.yaml), N=%d observed / M=%d model points per crop, "
                                   "64^3 x 6 mm voxels; shape=%s" % (n_inp, n_tmp, args.shape),
                       "global_batch": world * b, "frames_per_step_per_gpu": b, "parallelism": "frames sharded x%d" % world,
                       "weights": "seeded random (no checkpoints offline)",
                       "calls": "pipelined (async_inputs)" if args.pipelined_calls else "serial"},
            "roofline": roofline,
            "adds_auc_synthetic_weights": auc, "metric_frames_reduced": int(table.sums[:, 0].sum())}
    rdata_for_pipe = None
    if rank == 0 and world == 1 and not args.no_extras:
        if args.shape != "ref":
            rn, rm = SHAPES["ref"]
            rcfg = dcl.synth.default_cfg(rn, rm)
            rnet = dcl.DCL_Net.Network(rcfg, mode="test", async_inputs=args.pipelined_calls)
            rnet.load_state_dict(dcl.synth.synth_state_dict(rnet, 1))
            rnet = rnet.to(dev).eval()
            rdata = to_device(dcl.synth.make_batch(b, rn, rm), dev)
            rdt, ratt = run_forward_bench(dcl, rnet, rdata, max(args.steps, 20), max(args.warmup, 3), False)
            rsteps = max(args.steps, 20)
            rflop = 2.0 * (64 + 320) * rn * rm * b
            line["ref_shape"] = {"workload": "N=M=1024 (what config_YCBV_bs32.yaml defines), bs=32",
                                 "value": round(b * rsteps / rdt, 2), "unit": "frames/s",
                                 "ms_per_step": round(rdt / rsteps * 1e3, 3),
                                 "attention_TFLOPs": round(rflop / (np.mean(ratt) * 1e-3) / 1e12, 2) if ratt else None}
            rdata_for_pipe = (rcfg, rdata)
            del rnet
        # the same forward fed from the loader's HOST tensors (pageable memory, H2D inside forward) -- never `value`
        hdt, _ = run_forward_bench(dcl, net, host_data, max(3, args.steps // 2), 1, False)
        hsteps = max(3, args.steps // 2)
        line["h2d_inclusive"] = {"value": round(b * hsteps / hdt, 2), "unit": "frames/s",
                                 "ms_per_step": round(hdt / hsteps * 1e3, 3),
                                 "what": "data dict on the host (pageable), uploaded inside forward()"}
        if not args.pipelined_calls:
            line["pipelined_calls"] = pipelined_bench(dcl, dev, sd, cfg, data, b, args.steps, args.warmup, rdata_for_pipe)
        line["lm_stream"] = lm_stream_bench(dcl, dev)
        line["primitives"] = primitives_roofline(dcl)
        line["refiner"] = refiner_bench(dcl, dev, b)
        line["cpu_baseline"] = cpu_baseline(dcl, sd, cfg, n_inp, n_tmp)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Host logic on CPU: frame sharding + exact metric reduction (2-process gloo), the AUC closed form against the literal
VOCap restatement, synthetic-data determinism, config loading, state_dict layout."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_closed_form_auc_equals_vocap(dcl):
    from oracle import graph as G
    rng = np.random.default_rng(0)
    for trial in range(300):
        n = int(rng.integers(1, 40))
        d = rng.uniform(0, 0.16, n)
        d[rng.uniform(size=n) < 0.15] = np.inf                          # missed detections
        if trial % 7 == 0:
            d[:] = np.inf
        if trial % 5 == 0 and n > 3:
            d[1] = d[2]                                                  # ties
        t = dcl.sharding.AddsTable()
        for x in d:
            t.add(3, float(x))
        _, _, auc, acc = t.finalize()
        want_auc, want_acc = G.vocap_auc(list(d))
        assert abs(auc[3] - want_auc) <= 1e-4 and abs(acc[3] - want_acc) <= 1e-9, (trial, auc[3], want_auc)


def test_shard_indices_partition():
    dcl_sh = __import__("importlib").import_module("dcl-net_amd").sharding
    for n, w in ((10, 3), (8, 8), (5, 8), (0, 2)):
        parts = [dcl_sh.shard_indices(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))


WORKER = textwrap.dedent('''
    import importlib, os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    dcl = importlib.import_module("dcl-net_amd")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    cls = rng.integers(0, 21, 200); d = rng.uniform(0, 0.14, 200); d[rng.uniform(size=200) < 0.1] = np.inf
    full = dcl.sharding.AddsTable()
    for c, x in zip(cls, d): full.add(int(c), float(x))
    mine = dcl.sharding.AddsTable()
    for i in dcl.sharding.shard_indices(200, rank, world): mine.add(int(cls[i]), float(d[i]))
    mine.reduce()
    a, b = mine.finalize(), full.finalize()
    assert a[0] == b[0] and a[1] == b[1], (a[:2], b[:2])
    assert np.allclose(a[2], b[2], atol=1e-9) and np.allclose(a[3], b[3])
    dist.destroy_process_group()
    print("rank", rank, "ok", a[0], a[1])
''')


def test_metric_allreduce_two_processes_gloo(tmp_path):
    """N>1 path: world_size 2 on CPU (gloo stands in for RCCL); sharded tables reduce to the single-process result"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_state_dict_layout_matches_survey_appendix_a(dcl):
    net = dcl.DCL_Net.Network(dcl.synth.default_cfg(), mode="test")
    sd = net.state_dict()
    assert len(sd) == 270
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k) == 8393972
    assert tuple(sd["backbone_inp.module1.0.layers.0.weight"].shape) == (3, 3, 3, 7, 16)
    assert tuple(sd["backbone_tmp.module4.1.layers.0.weight"].shape) == (3, 3, 3, 128, 256)
    assert tuple(sd["disengage_Xc_m1.1.layers.0.weight"].shape) == (64, 256, 1, 1, 1)
    assert tuple(sd["neck_fuser.layers.6.weight"].shape) == (1024, 512, 1)
    assert tuple(sd["regressor_rot.layers.4.weight"].shape) == (9, 128, 1)
    ref = dcl.refiner.Refiner()
    assert len(ref.state_dict()) == 18
    assert sum(v.numel() for v in ref.state_dict().values()) == 2103564
    assert tuple(ref.state_dict()["MLP_share.layers.0.weight"].shape) == (512, 259, 1)


def test_yaml_config_loader(dcl, tmp_path):
    p = tmp_path / "cfg.yaml"
    p.write_text("model:\n  voxelization_mode: 4\n  unit_voxel_extent: [0.006, 0.006, 0.006]\n  n_inp: 1024\n"
                 "  n_tmp: 1024\n  backbone:\n    downsample_by_pooling: True\n    kernel_size: 3\n    bias: False\n")
    cfg = dcl.synth.load_yaml_cfg(str(p))
    net = dcl.DCL_Net.Network(cfg.model, mode="test")
    assert net.n_inp == 1024 and cfg.model.backbone.kernel_size == 3


def test_yaml_loader_on_the_reference_configs(dcl):
    """the reference's own yaml files (read here in the build container only; skipped where /root/reference is absent):
    every model block builds a Network / Refiner with the shapes the configs promise"""
    import glob
    import pytest
    files = sorted(glob.glob("/root/reference/configs/*.yaml"))
    if not files:
        pytest.skip("reference configs not present on this box")
    for f in files:
        cfg = dcl.synth.load_yaml_cfg(f)
        m = cfg.model
        assert m.voxelization_mode == 4 and len(m.unit_voxel_extent) == 3 and m.backbone.kernel_size == 3
        net = dcl.DCL_Net.Network(m, mode="test")
        assert (net.n_inp, net.n_tmp) == (m.n_inp, m.n_tmp) == (1024, 1024)
        assert len(net.state_dict()) == 270


def test_load_checkpoint_layouts(dcl, tmp_path):
    """synth.load_checkpoint against the layouts the reference's tools produce and read: gorilla.solver.save_checkpoint's
    dict {"model": state_dict, "optimizer": ..., "scheduler": ..., "meta": ...} (tools/train_YCBV_stage1.py:102-104, read
    back by tools/test_YCBV_stage1.py:233-235), the same saved from an nn.DataParallel wrapper ("module." prefixes,
    tools/test_YCBV_stage1.py:230-231), a {"state_dict": ...} dict, and a bare state_dict; wrong shapes must raise"""
    import pytest
    cfg = dcl.synth.default_cfg(64, 64)
    src = dcl.DCL_Net.Network(cfg, mode="test")
    sd = dcl.synth.synth_state_dict(src, 11)
    layouts = {
        "gorilla": {"model": sd, "optimizer": {"state": {}, "param_groups": []}, "scheduler": {"last_epoch": 3},
                    "meta": {"epoch": 30, "iter": 1234}},
        "dataparallel": {"model": {"module." + k: v for k, v in sd.items()}, "meta": {"epoch": 2}},
        "state_dict": {"state_dict": sd},
        "bare": sd,
    }
    for name, obj in layouts.items():
        path = str(tmp_path / (name + ".pth"))
        torch.save(obj, path)
        net = dcl.DCL_Net.Network(cfg, mode="test")
        meta = dcl.synth.load_checkpoint(net, path)
        got = net.state_dict()
        assert sorted(got.keys()) == sorted(sd.keys())
        for k, v in sd.items():
            assert torch.equal(got[k], v), (name, k)
        assert meta == (obj.get("meta", {}) if name in ("gorilla", "dataparallel") else {}), name
    ref = dcl.refiner.Refiner()
    sdr = dcl.synth.synth_state_dict(ref, 12)
    path = str(tmp_path / "refiner.pth")
    torch.save({"model": sdr, "meta": {"epoch": 9}}, path)
    ref2 = dcl.refiner.Refiner()
    assert dcl.synth.load_checkpoint(ref2, path) == {"epoch": 9}
    assert all(torch.equal(ref2.state_dict()[k], v) for k, v in sdr.items())
    bad = dict(sd)
    bad["neck_fuser.layers.6.weight"] = torch.zeros(1024, 256, 1)
    torch.save({"model": bad}, str(tmp_path / "bad.pth"))
    with pytest.raises(RuntimeError):
        dcl.synth.load_checkpoint(dcl.DCL_Net.Network(cfg, mode="test"), str(tmp_path / "bad.pth"))
    missing = {k: v for k, v in sd.items() if not k.startswith("regressor_rot")}
    torch.save({"model": missing}, str(tmp_path / "missing.pth"))
    with pytest.raises(RuntimeError):
        dcl.synth.load_checkpoint(dcl.DCL_Net.Network(cfg, mode="test"), str(tmp_path / "missing.pth"))


def test_synth_batch_contract(dcl):
    d = dcl.synth.make_batch(3, 128, 96)
    assert d["inp"]["feats"].shape == (3 * 128, 7) and d["tmp"]["feats"].shape == (3 * 96, 7)
    for side in ("inp", "tmp"):
        occ, v2p, p2v = d[side]["occupied_voxels"], d[side]["v2p_maps"], d[side]["p2v_maps"]
        assert occ.dtype == torch.int64 and v2p.dtype == torch.int32 and p2v.dtype == torch.int32
        assert int(occ.min()) >= 0 and int(occ[:, 1:].max()) < 64
        assert int(v2p[:, 0].sum()) == d[side]["feats"].shape[0]          # every point in exactly one voxel
        assert bool((occ[:-1, 0] <= occ[1:, 0]).all())                    # batch-sorted (loaders guarantee it)
    assert torch.equal(d["inp"]["feats"][:, 0], torch.ones(3 * 128))
    again = dcl.synth.make_batch(3, 128, 96)
    assert torch.equal(again["inp"]["feats"], d["inp"]["feats"])


def test_metric_matches_reference_functions_golden(dcl, golden_dir):
    """AddsTable's closed form vs the outputs of the reference's own cal_auc_acc / cal_metric_auc_acc
    (tests/golden/make_metric_golden.py executed them from tools/test_YCBV_stage1.py:83-125 in the build container)"""
    import os
    z = np.load(os.path.join(golden_dir, "metric_ref.npz"))
    for case in range(6):
        d, idx = z["d%d" % case], z["idx%d" % case]
        table = dcl.sharding.AddsTable()
        for c, x in zip(idx, d):
            table.add(int(c), float(x))
        mean_auc, mean_acc, auc, acc = table.finalize()
        assert np.abs(auc - z["auc%d" % case]).max() <= 1e-4          # the reference accumulates its ramp in float32
        assert np.abs(acc - z["acc%d" % case]).max() <= 1e-9
        assert abs(mean_auc - float(z["mean_auc%d" % case][0])) <= 0.0100001     # both rounded to 2 decimals


def _lm_frames(golden_dir):
    z = np.load(os.path.join(golden_dir, "lm_metric_ref.npz"))
    frames = []
    for f in range(int(z["n_frames"][0])):
        frames.append({k: z["f%d_%s" % (f, k)] for k in ("flags", "idx", "Rp", "tp", "Rg", "tg", "l2", "cd")})
    return z, frames


def test_linemod_metric_matches_reference_loop_golden(dcl, golden_dir):
    """sharding.add_lm + LmTable vs the per-frame body of the reference's LineMOD eval loop (tools/test_LM.py:112-141, run
    from the reference source by tests/golden/make_lm_metric_golden.py): ADD for non-symmetric, ADD-S for symmetric
    objects, `dis < diameter[idx]`, lost detections skipped -> the same success_count / num_count"""
    z, frames = _lm_frames(golden_dir)
    clouds = torch.from_numpy(z["clouds"])
    table = dcl.sharding.LmTable(z["diameter"])
    for fr in frames:
        flags = fr["flags"]
        sym = torch.from_numpy(flags[flags != -1].astype(np.int32))
        cld = clouds[torch.from_numpy(fr["idx"]).long()]
        args = [torch.from_numpy(fr[k]) for k in ("Rp", "tp", "Rg", "tg")]
        d = dcl.sharding.add_lm(cld, *args, sym)
        want = np.where(sym.numpy() != 0, fr["cd"], fr["l2"])
        assert np.abs(d.numpy() - want).max() <= 1e-6
        table.add_batch(fr["idx"], d.tolist(), flags)
    assert np.array_equal(table.counts[:, 0], z["num_count"]) and np.array_equal(table.counts[:, 1], z["success_count"])
    all_rate, per = table.finalize()
    assert abs(all_rate - z["success_count"].sum() / z["num_count"].sum()) <= 1e-12
    assert np.allclose(per, z["success_count"] / z["num_count"])
    assert len(dcl.sharding.LmTable.OBJLIST) == 13


LM_WORKER = textwrap.dedent('''
    import importlib, os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    dcl = importlib.import_module("dcl-net_amd")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(%r)
    n = int(z["n_frames"][0])
    mine = dcl.sharding.LmTable(z["diameter"])
    for f in dcl.sharding.shard_indices(n, rank, world):               # frames r, r+W, ... on this rank
        flags = z["f%%d_flags" %% f]
        d = np.where(flags[flags != -1] != 0, z["f%%d_cd" %% f], z["f%%d_l2" %% f])
        mine.add_batch(z["f%%d_idx" %% f], d.tolist(), flags)
    assert mine.counts[:, 0].sum() < z["num_count"].sum()                # really a shard
    mine.reduce()
    assert np.array_equal(mine.counts[:, 0], z["num_count"]) and np.array_equal(mine.counts[:, 1], z["success_count"])
    dist.destroy_process_group()
    print("rank", rank, "ok", mine.finalize()[0])
''')


def test_linemod_table_allreduce_two_processes_gloo(tmp_path, golden_dir):
    """LineMOD variant of the metric reduction: SUM of success_count[13] / num_count[13] over 2 gloo ranks
    reproduces the reference's single-process counts exactly"""
    script = tmp_path / "lm_worker.py"
    script.write_text(LM_WORKER % (ROOT, os.path.join(golden_dir, "lm_metric_ref.npz")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_graph_routing_rule(dcl):
    """which eval-mode calls replay a whole-forward hipGraph (host logic only): up to graph_max_batch crops always, larger
    batches while the call is small in points (pipelining instances too: a replayed small call beats the pipelined
    launch-by-launch one since the split-bf16 kernels); graph_max_batch = 0 = never"""
    mk = lambda n, m, **kw: dcl.DCL_Net.Network(dcl.synth.default_cfg(n, m), mode="test", **kw)    # noqa: E731
    ref = mk(1024, 1024)
    assert [ref.replays_graph(b) for b in (0, 1, 8, 32, 40, 48, 49)] == [False, True, True, True, True, True, False]
    stress = mk(12288, 2048)
    assert [stress.replays_graph(b) for b in (1, 6, 8, 9, 32)] == [True, True, True, False, False]
    assert [mk(1024, 1024, async_inputs=True).replays_graph(b) for b in (1, 8, 9, 32, 49)] == [True, True, True, True, False]
    assert [mk(12288, 2048, async_inputs=True).replays_graph(b) for b in (6, 9, 32)] == [True, False, False]
    assert not any(mk(1024, 1024, graph_max_batch=0).replays_graph(b) for b in (1, 8, 32))
    assert [mk(1024, 1024, graph_max_batch=4, graph_max_points=0).replays_graph(b) for b in (4, 5)] == [True, False]


def test_own_gemm_core_eligibility_rule(dcl):
    """which layers ops.linear sends to the own GEMM core (host logic only: shapes, pitches, alignment -- csrc/linear_dma.hip wants
    K in whole 32-chunks, 16-byte aligned operands, row pitches of whole float4s, a Wt row holding N rounded up to 4 floats);
    every layer shape of the forward qualifies, the refiner's reference-layout entry (K = 259) and unpadded thin weights do not"""
    import torch
    ok = dcl.ops.linear_dma_ok
    z = lambda r, c: torch.zeros(r, c)                                     # noqa: E731
    for K, n in ((480, 1024), (256, 256), (256, 64), (512, 512), (512, 1024), (128, 128), (1024, 512), (512, 128)):
        assert ok(z(64, K), z(K, n)), (K, n)
    assert ok(z(64, 1024)[:, 256:512], z(256, 64))                          # a column block of a wider buffer: pitch 1024
    assert ok(z(64, 128), dcl.ops.pad_linear_weight(z(128, 9))) and ok(z(64, 128), dcl.ops.pad_linear_weight(z(128, 1)))
    assert not ok(z(64, 128), z(128, 9))                                    # rows of 9 floats: not whole float4s
    assert not ok(z(64, 259), z(259, 512)) and not ok(z(64, 48), z(48, 64)) and not ok(z(64, 16), z(16, 64))
    assert not ok(z(64, 260)[:, 2:258], z(256, 64))                         # 8-byte offset: x not 16-byte aligned
    # the folded head weights of a Network: every last layer that goes through _mlp sits in padded rows
    net = dcl.DCL_Net.Network(dcl.synth.default_cfg(256, 256), mode="test")
    f = net._fold()
    for name in ("regressor_conf", "regressor_conf_bi", "regressor_Xo", "regressor_Yc", "regressor_rot_padded", "regressor_trans_padded"):
        Wt = f[name][2][0]
        assert Wt.stride(0) % 4 == 0 and ok(z(8, Wt.shape[0]), Wt), name
    assert all(t.is_contiguous() for pair in f["regressor_rot"] + f["regressor_trans"] for t in pair)   # what dcl_pose_heads takes


def test_split_bf16_core_dispatch_rule(dcl):
    """which launches ops.linear sends to the split-bf16 core (host logic only): a PREPARED weight (ops.prepare_linear: CUDA, more
    than 64 output columns, K in whole 16-chunks) and at least ops.SPLIT_MIN_TILES tiles of 256 x 128; nothing is prepared on the
    CPU, and an unprepared weight never takes the split core"""
    import torch
    ops = dcl.ops
    assert ops.prepare_linear(torch.zeros(512, 512)) is None                 # CPU tensor: nothing to prepare
    assert ops.prepared_linear(torch.zeros(512, 512), torch.zeros(65536, 512)) is None
    assert ops.linear_split_ok(torch.zeros(64, 512), 512) and ops.linear_split_ok(torch.zeros(64, 1024)[:, 256:512], 256)
    assert not ops.linear_split_ok(torch.zeros(64, 40), 40) and not ops.linear_split_ok(torch.zeros(64, 260)[:, 2:258], 256)
    tiles = lambda M, n: ((M + 255) // 256) * ((n + 127) // 128)            # noqa: E731
    assert tiles(6144, 1024) >= ops.SPLIT_MIN_TILES > tiles(6144, 512)       # six crops: the 1024-column layers only
    assert tiles(32768, 256) >= ops.SPLIT_MIN_TILES > tiles(1024, 1024)      # 32 crops: every layer of > 64 columns; one crop: none


def test_tail_parallel_rule(dcl):
    """do the dense tail's two directions run side by side?  (host logic only) -- always while both attention launches are
    small (every N = M = 1024 call), at N = 12288 only for the batch sizes whose grid of 256-query workgroups ends in a
    mostly empty round (8 crops = 1.5 rounds, 24 = 4.5), not for whole rounds (16, 32); PAR_TAIL forces either way"""
    mk = lambda n, m: dcl.DCL_Net.Network(dcl.synth.default_cfg(n, m), mode="test")    # noqa: E731
    ref, stress = mk(1024, 1024), mk(12288, 2048)
    assert all(ref._tail_parallel(b) for b in (1, 6, 32, 40, 63))
    assert [ref._tail_parallel(b, launch_by_launch=True) for b in (1, 2, 3, 32)] == [False, False, True, True]
    assert [stress._tail_parallel(b) for b in (1, 4, 8, 16, 24, 32)] == [True, True, True, False, True, False]
    stress.PAR_TAIL = True
    assert stress._tail_parallel(32)
    ref.PAR_TAIL = False
    assert not ref._tail_parallel(1)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_two_rank_dry_run_line(tmp_path, world):
    """bench.py's N > 1 plumbing under the driver's own launcher, on the CPU (gloo, --dry-run: empty
    step), for 2 ranks and for the 8 ranks of a full node: one JSON line from rank 0, last on stdout, with what
    torch.distributed saw -- world, backend, one device and one DISJOINT host-core set per rank -- and a metric reduction
    that lost no frames (metric_frames_reduced == world * b)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--dry-run", "--no-extras", "--shape", "ref", "--batch", "4"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    if len(os.sched_getaffinity(0)) < world:
        pytest.skip("fewer host cores than ranks: the sets cannot be disjoint")
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [l for l in r.stdout.strip().splitlines() if l.strip()][-1]
    line = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "rccl", "metric_frames_reduced"):
        assert k in line, k
    assert line["dry_run"] is True and line["value"] is None          # no rate is claimed without a GPU
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["metric_frames_reduced"] == world * 4
    rc = line["rccl"]
    assert rc["world"] == world and rc["backend"] == "gloo" and rc["initialized"] is True
    assert len(rc["device_per_rank"]) == world
    assert all("cuda:%d" % i in rc["device_per_rank"][i] for i in range(world))
    sets = [set(dcl_cpus(c)) for c in rc["cpus_per_rank"]]
    assert all(sets), "every rank reports its cores"
    for i in range(world):
        for j in range(i + 1, world):
            assert not (sets[i] & sets[j]), "ranks must be pinned to disjoint host cores"


def dcl_cpus(text):
    out = []
    for part in text.split(","):
        lo, _, hi = part.partition("-")
        out += list(range(int(lo), int(hi or lo) + 1))
    return out


def test_rank_cpu_sets_follow_the_gpu_numa_topology(dcl, tmp_path):
    """sharding.rank_cpu_set on a fake sysfs: 4 GPUs on 2 NUMA nodes -> every rank inside its GPU's node, peers on a node
    split it evenly, nothing shared"""
    base = tmp_path / "class" / "kfd" / "kfd" / "topology" / "nodes"
    def node(i, cpu_cores, simd, links=()):
        d = base / str(i)
        (d / "io_links").mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\n" % (cpu_cores, simd))
        for j, (to, w) in enumerate(links):
            (d / "io_links" / str(j)).mkdir()
            (d / "io_links" / str(j) / "properties").write_text("node_from %d\nnode_to %d\nweight %d\n" % (i, to, w))
    node(0, 8, 0); node(1, 8, 0)
    node(2, 0, 1024, [(0, 20), (1, 40)]); node(3, 0, 1024, [(0, 20)]); node(4, 0, 1024, [(1, 20), (0, 40)]); node(5, 0, 1024, [(1, 20)])
    for n, cl in ((0, "0-7"), (1, "8-15")):
        d = tmp_path / "devices" / "system" / "node" / ("node%d" % n)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cl + "\n")
    S = dcl.sharding
    assert S.gpu_numa_nodes(str(tmp_path)) == [0, 0, 1, 1]
    sets = [S.rank_cpu_set(r, 4, allowed=set(range(16)), sysfs_root=str(tmp_path)) for r in range(4)]
    assert sets == [{0, 1, 2, 3}, {4, 5, 6, 7}, {8, 9, 10, 11}, {12, 13, 14, 15}]
    # a cgroup that grants only part of a node, and a host without KFD: even split of what is allowed
    assert S.rank_cpu_set(1, 4, allowed={0, 1, 8, 9, 10, 11}, sysfs_root=str(tmp_path)) == {1}
    assert S.rank_cpu_set(1, 2, allowed=set(range(8)), sysfs_root=str(tmp_path / "nowhere")) == {4, 5, 6, 7}
    # a leased subset -- HIP ordinal i is not KFD GPU i.  A plain ordinal list is followed (HIP 0 = GPU 2 on node 1,
    # HIP 1 = GPU 0 on node 0); a list that cannot be resolved (uuids) falls back to the even split of the allowed cores
    env = {"HIP_VISIBLE_DEVICES": "2,0"}
    assert S.visible_device_map(env, 4) == [2, 0]
    assert S.rank_cpu_set(0, 2, allowed=set(range(16)), sysfs_root=str(tmp_path), env=env) == set(range(8, 16))
    assert S.rank_cpu_set(1, 2, allowed=set(range(16)), sysfs_root=str(tmp_path), env=env) == set(range(0, 8))
    assert S.visible_device_map({"ROCR_VISIBLE_DEVICES": "1,2,3", "HIP_VISIBLE_DEVICES": "2,0"}, 4) == [3, 1]
    env = {"ROCR_VISIBLE_DEVICES": "GPU-deadbeef,GPU-cafe"}
    assert S.visible_device_map(env, 4) is None
    assert S.rank_cpu_set(1, 2, allowed=set(range(16)), sysfs_root=str(tmp_path), env=env) == set(range(8, 16))
    assert S.visible_device_map({}, 4) == [0, 1, 2, 3]


def test_legacy_choice_heads_equal_numpy_and_advance_the_global_stream_alike(dcl):
    """ops.legacy_choice_heads (csrc/legacy_rng.cpp, host code) == [np.random.choice(m, n, replace=False) ...] on the global
    legacy generator, for sizes around the powers of two (where the mask of random_interval changes), after an arbitrary
    number of earlier draws (generator position mid-block), and the draws that FOLLOW are the same too"""
    rng = np.random.RandomState(5)
    for trial in range(40):
        seed = int(rng.randint(0, 2 ** 31))
        ms = [int(rng.randint(1025, 40000)) for _ in range(int(rng.randint(1, 7)))]
        if trial % 5 == 0:
            ms[0] = 1024
        if trial % 3 == 0:
            ms[-1] = (1 << int(rng.randint(11, 16))) + int(rng.randint(-1, 2))
        pre = int(rng.randint(0, 1500))
        np.random.seed(seed)
        np.random.rand(pre)
        want = [np.random.choice(m, 1024, replace=False) for m in ms]
        tail_w = np.random.randint(0, 1 << 30, 5)
        np.random.seed(seed)
        np.random.rand(pre)
        got = dcl.ops.legacy_choice_heads(ms, 1024)
        tail_g = np.random.randint(0, 1 << 30, 5)
        assert all((got[i] == want[i]).all() for i in range(len(ms))), (trial, ms)
        assert (tail_w == tail_g).all(), trial

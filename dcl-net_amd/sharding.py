"""Frame sharding across the GPUs of a node + exact reduction of the ADD-S metric.

Crops/images are independent: rank r owns items r, r+W, r+2W, ... and runs the whole forward on its own
GPU with no traffic.  The only collective is the metric reduction at the end, made exact by the closed form
of the reference's VOCap AUC (tools/test_YCBV_stage1.py:83-125):
    AUC_c = 100 * (10/n_c) * (0.1*m_c - (sum_valid D - max_valid D)),   valid: D <= 0.1
so per class only [n, m, sum D, #(D<0.02)] (SUM) and max D (MAX) cross the wire: 21x4 + 21 fp64 values
over RCCL (backend "nccl" on ROCm) or gloo in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist

N_CLASSES = 21
MAX_DIS = 0.1


def shard_indices(n_items, rank, world):
    return list(range(rank, n_items, world))


def add_s(cld, R_pred, t_pred, R_gt, t_gt, chunk=512):
    """ADD-S per object (tools/test_YCBV_stage1.py:186-189): cld (b,P,3); mean_i min_j |pred_i - gt_j|.
    CUDA tensors go to the fused HIP kernel (ops.add_s); the torch form below is the CPU-side helper of the gloo tests."""
    if cld.is_cuda:
        from . import ops
        return ops.add_s(cld, R_pred, t_pred, R_gt, t_gt)
    pred = torch.bmm(cld, R_pred.transpose(1, 2)) + t_pred.unsqueeze(1)
    gt = torch.bmm(cld, R_gt.transpose(1, 2)) + t_gt.unsqueeze(1)
    mins = []
    for s in range(0, pred.shape[1], chunk):
        mins.append(torch.cdist(pred[:, s:s + chunk], gt, compute_mode="donot_use_mm_for_euclid_dist").min(dim=2)[0])
    return torch.cat(mins, dim=1).mean(dim=1)


def add_lm(cld, R_pred, t_pred, R_gt, t_gt, sym_flag):
    """LineMOD distance per object (tools/test_LM.py:123-135): ADD (`l2_dis`) where sym_flag == 0, ADD-S (`cd_dis`) where it
    is 1.  CUDA tensors go to the fused HIP kernel; the torch form is the CPU-side helper of the gloo tests."""
    if cld.is_cuda:
        from . import ops
        return ops.add_s(cld, R_pred, t_pred, R_gt, t_gt, sym_flag=sym_flag)
    pred = torch.bmm(cld, R_pred.transpose(1, 2)) + t_pred.unsqueeze(1)
    gt = torch.bmm(cld, R_gt.transpose(1, 2)) + t_gt.unsqueeze(1)
    l2 = torch.norm(pred - gt, dim=2).mean(dim=1)
    cd = torch.cdist(pred, gt, compute_mode="donot_use_mm_for_euclid_dist").min(dim=2)[0].mean(dim=1)
    return torch.where(sym_flag.to(l2.device) != 0, cd, l2)


class LmTable(object):
    """LineMOD ADD(S) < 0.1 d success table (tools/test_LM.py:99-100,126-141,148-153): per object `num_count` and
    `success_count` with the threshold `dis < diameter[idx]` (diameter[] already holds 0.1 * object diameter in metres,
    tools/test_LM.py:68-75); lost detections (flag -1) are not counted at all.  Integer counts: the SUM all-reduce over
    RCCL / gloo is exact."""
    OBJLIST = (1, 2, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 15)          # tools/test_LM.py:71

    def __init__(self, diameter):
        self.diameter = np.asarray(diameter, np.float64)
        self.counts = np.zeros((len(self.diameter), 2), np.int64)        # num_count, success_count

    def add(self, idx, dis, flag=0):
        if flag == -1:
            return
        self.counts[idx, 0] += 1
        if dis < self.diameter[idx]:
            self.counts[idx, 1] += 1

    def add_batch(self, obj_idx, dis, flags=None):
        """the per-frame loop of tools/test_LM.py:126-141: `flags` covers every object of the frame (incl. -1 = lost),
        obj_idx / dis only the detected ones, in order"""
        flags = [0] * len(dis) if flags is None else [int(f) for f in flags]
        k = 0
        for f in flags:
            if f == -1:
                continue
            self.add(int(obj_idx[k]), float(dis[k]))
            k += 1
        assert k == len(dis)

    def reduce(self, device=None, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            return self
        c = torch.from_numpy(self.counts).to(device or "cpu")
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        self.counts = c.cpu().numpy()
        return self

    def finalize(self):
        """-> (ALL success rate, per-object success rates) as tools/test_LM.py:148-153 logs them"""
        num, suc = self.counts[:, 0], self.counts[:, 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            per = np.where(num > 0, suc / np.maximum(num, 1), np.nan)
        return (float(suc.sum()) / float(num.sum()) if num.sum() else float("nan")), per


class AddsTable(object):
    """per-class sufficient statistics of the YCB-V ADD-S AUC / <2cm metric."""

    def __init__(self, n_classes=N_CLASSES):
        self.sums = np.zeros((n_classes, 4), np.float64)      # n, m, sum D valid, count D < 0.02
        self.maxd = np.zeros(n_classes, np.float64)

    def add(self, cls, d):
        """d = ADD-S distance, np.inf for a missed detection (tools/test_YCBV_stage1.py:192-194)."""
        self.sums[cls, 0] += 1
        if d <= MAX_DIS:
            self.sums[cls, 1] += 1
            self.sums[cls, 2] += d
            self.maxd[cls] = max(self.maxd[cls], d)
        if d < 0.02:
            self.sums[cls, 3] += 1

    def reduce(self, device=None, group=None):
        """all_reduce over the process group (no-op when torch.distributed is not initialised; a one-rank group still
        goes through the collective, so a single-GPU smoke run exercises the RCCL path)."""
        if not (dist.is_available() and dist.is_initialized()):
            return self
        s = torch.from_numpy(self.sums).to(device or "cpu")
        m = torch.from_numpy(self.maxd).to(device or "cpu")
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=group)
        self.sums, self.maxd = s.cpu().numpy(), m.cpu().numpy()
        return self

    def finalize(self):
        """-> (mean AUC, mean <2cm, per-class AUC, per-class <2cm) as cal_metric_auc_acc reports them."""
        n, m, sd, c2 = self.sums.T
        with np.errstate(divide="ignore", invalid="ignore"):
            auc = np.where(n > 0, 100.0 * (10.0 / np.maximum(n, 1)) * (MAX_DIS * m - (sd - self.maxd)), 0.0)
            acc = np.where(n > 0, 100.0 * c2 / np.maximum(n, 1), 0.0)
        return round(float(np.mean(auc)), 2), round(float(np.mean(acc)), 2), auc, acc


# ------------------------------------------------------------------------------------------ rank placement on the host
def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_nodes(sysfs_root="/sys"):
    """NUMA node of every GPU in KFD enumeration order (= HIP device order unless *_VISIBLE_DEVICES reorders), read from the
    KFD topology in sysfs -- no HIP call, so it can run before the process touches a GPU.  A GPU node is a topology node
    with simd_count > 0; its NUMA node is the CPU node at the other end of its cheapest io_link.  [] if there is no KFD."""
    import os
    base = os.path.join(sysfs_root, "class", "kfd", "kfd", "topology", "nodes")
    if not os.path.isdir(base):
        return []

    def props(path):
        out = {}
        try:
            for line in open(path):
                k, _, v = line.strip().partition(" ")
                out[k] = v
        except OSError:
            pass
        return out
    nodes = sorted((int(d) for d in os.listdir(base) if d.isdigit()))
    info = {n: props(os.path.join(base, str(n), "properties")) for n in nodes}
    cpu_nodes = [n for n in nodes if int(info[n].get("cpu_cores_count", "0") or 0) > 0]
    numa = []
    for n in nodes:
        if int(info[n].get("simd_count", "0") or 0) <= 0:
            continue
        best, best_w = (cpu_nodes[0] if cpu_nodes else 0), None
        links = os.path.join(base, str(n), "io_links")
        if os.path.isdir(links):
            for l in sorted(os.listdir(links)):
                p = props(os.path.join(links, l, "properties"))
                to, w = int(p.get("node_to", "-1") or -1), int(p.get("weight", "0") or 0)
                if to in cpu_nodes and (best_w is None or w < best_w):
                    best, best_w = to, w
        numa.append(cpu_nodes.index(best) if best in cpu_nodes else 0)
    return numa


def visible_device_map(env, n_gpus):
    """KFD GPU index of every HIP ordinal under ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (ROCR first:
    it filters what HIP then numbers): list of ints, all of 0..n_gpus-1 when nothing is set; None when a variable cannot be
    resolved to plain ordinals (then the topology is not used)."""
    ids = list(range(n_gpus))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = env.get(var)
        if val is None or val.strip() == "":
            continue
        picked = []
        for tok in val.split(","):
            tok = tok.strip()
            if not tok.isdigit() or int(tok) >= len(ids):
                return None
            picked.append(ids[int(tok)])
        ids = picked
        if var == "HIP_VISIBLE_DEVICES":
            break                                              # (CUDA_VISIBLE_DEVICES is an alias of it: not applied twice)
    return ids


def rank_cpu_set(local_rank, local_world, allowed=None, sysfs_root="/sys", env=None):
    """the host cores rank `local_rank` of `local_world` ranks on this node should run on: the cores of its GPU's NUMA node
    (inside the process's allowed set), divided evenly among the ranks whose GPUs share that node -- with one process per GPU
    and a per-call host read-back on the launch-by-launch path, eight ranks hopping over 2 sockets become launch-bound
    otherwise.  Falls back to an even split of the allowed cores when the topology cannot be read.  Pure function."""
    import os
    if allowed is None:
        allowed = os.sched_getaffinity(0)
    allowed = sorted(allowed)
    numa = gpu_numa_nodes(sysfs_root)
    peers, mine = list(range(local_world)), allowed
    # HIP ordinal i is KFD GPU i only while no *_VISIBLE_DEVICES selects or reorders a subset (a leased N-of-8 box): an
    # explicit plain list "a,b,c" of ordinals is followed; anything else (uuids, ...) takes the even split below
    vis = visible_device_map(env if env is not None else os.environ, len(numa))
    if vis is None:
        numa = []
    else:
        numa = [numa[g] for g in vis]
    if len(numa) >= local_world and local_world > 0:
        node = numa[local_rank]
        try:
            cpus = _parse_cpulist(open(os.path.join(sysfs_root, "devices", "system", "node", "node%d" % node, "cpulist")).read())
        except OSError:
            cpus = set()
        on_node = [c for c in allowed if c in cpus]
        if on_node:
            mine = on_node
            peers = [r for r in range(local_world) if numa[r] == node]
    k, i = len(peers), peers.index(local_rank) if local_rank in peers else 0
    per = max(1, len(mine) // max(k, 1))
    chunk = mine[i * per:(i + 1) * per] if i * per < len(mine) else mine
    return set(chunk or mine)


def pin_rank_to_gpu_numa(local_rank, local_world, sysfs_root="/sys"):
    """call BEFORE the first GPU call of a rank (bench.py does): sets the process's CPU affinity to rank_cpu_set(...).
    Returns the sorted core list in effect (what the bench line reports per rank)."""
    import os
    try:
        cpus = rank_cpu_set(local_rank, local_world, sysfs_root=sysfs_root)
        if local_world > 1:
            os.sched_setaffinity(0, cpus)
    except OSError:
        pass
    return sorted(os.sched_getaffinity(0))

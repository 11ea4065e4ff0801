"""Thin torch-tensor front end of the C ABI (include/dclnet_hip.h).

Every function takes/returns CUDA tensors (PyTorch is the allocator and stream provider), checks
arguments the way the reference's Python wrappers do (contiguity, dtypes) and calls the HIP
library on torch's current stream.  No function here has a CPU fallback.
"""
import ctypes as C
import threading

import weakref

import torch

from . import _native as N

_c_int = C.c_int
_c_float = C.c_float

# bench.py sets this to a list to collect (name, start_event, end_event) around selected launches on torch's
# current stream (the stream the kernels are launched on); None = off.
PROFILE_EVENTS = None


def _i32_ptr_or_null(t):
    return N.ptr(t) if t is not None else N.vp(0)


# ------------------------------------------------------------------------------------ PG_OP
def voxelize_idx(coords, batch_size, mode=4):
    """PG_OP.voxelize_idx (libs/pointgroup_ops/functions/pointgroup_ops.py:11-39): HOST op.
    coords long (N, 3|4) CPU -> (output_coords long (M,ncol), input_map int (N), output_map int (M,1+maxActive))."""
    if coords.is_cuda:
        raise RuntimeError("voxelize_idx is a host (DataLoader-side) op: pass a CPU tensor")
    assert coords.is_contiguous() and coords.dtype == torch.int64 and coords.dim() == 2
    n, ncol = coords.shape
    input_map = torch.zeros(n, dtype=torch.int32)
    na, ma = C.c_int32(0), C.c_int32(0)
    N.check(N.lib().dcl_voxelize_idx_count(N.ptr(coords), n, ncol, int(batch_size), int(mode), N.ptr(input_map),
                                           C.byref(na), C.byref(ma)), "voxelize_idx_count")
    out_coords = torch.zeros((na.value, ncol), dtype=torch.int64)
    out_map = torch.zeros((na.value, ma.value + 1), dtype=torch.int32)
    N.check(N.lib().dcl_voxelize_idx_fill_mode(N.ptr(coords), n, ncol, N.ptr(input_map), na.value, ma.value, int(mode),
                                               N.ptr(out_coords), N.ptr(out_map)), "voxelize_idx_fill")
    return out_coords, input_map, out_map


def voxelize_idx_gpu(coords, batch_size, S=64, mode=4):
    """Device version of voxelize_idx for coords (N,4) int64 CUDA inside a batch x S^3 grid: identical outputs
    (first-encounter voxel ids, ascending point lists), one host read-back of {V, maxActive}."""
    N.need_cuda(coords)
    assert coords.is_contiguous() and coords.dtype == torch.int64 and coords.shape[1] == 4
    n, dev = coords.shape[0], coords.device
    nbytes = C.c_int64(0)
    N.check(N.lib().dcl_voxelize_idx_gpu_ws_bytes(n, int(batch_size), int(S), C.byref(nbytes)), "voxelize_idx_gpu_ws_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    input_map = torch.empty(max(n, 1), dtype=torch.int32, device=dev)[:n]
    info = torch.empty(3, dtype=torch.int32, device=dev)
    N.check(N.lib().dcl_voxelize_idx_gpu_count(N.ptr(coords), n, int(batch_size), int(S), int(mode), N.ptr(ws), nbytes.value,
                                               N.ptr(input_map), N.ptr(info), N.stream()), "voxelize_idx_gpu_count")
    V, ma, err = info.cpu().tolist()
    if err:
        raise RuntimeError("voxelize_idx_gpu: a coordinate lies outside the batch x S^3 grid")
    ma = max(ma, 1)
    out_coords = torch.empty((V, 4), dtype=torch.int64, device=dev)
    out_map = torch.empty((V, ma + 1), dtype=torch.int32, device=dev)
    N.check(N.lib().dcl_voxelize_idx_gpu_fill(N.ptr(coords), n, int(batch_size), int(S), N.ptr(ws), N.ptr(input_map), V, ma,
                                              N.ptr(out_coords), N.ptr(out_map), N.stream()), "voxelize_idx_gpu_fill")
    return out_coords, input_map, out_map


VI_CROPS_MAX_BATCH, VI_CROPS_MAX_POINTS, VI_CROPS_S = 64, 1024, 64
_VI_COMM = {}
_VI_LOCK = threading.Lock()


def voxelize_idx_crops(coords, batch_size, n_per, S=64, mode=4, pitch=33, occ_dtype=torch.int64):
    """voxelize_idx for the crop builder's layout -- batch_size crops of exactly n_per <= 1024 points each (rows c*n_per .. of
    `coords` (b*n_per, 4) int64 belong to crop c), 64^3 grids, at most 64 crops -- in ONE launch and WITHOUT a host read-back
    (csrc/voxelize_idx.hip: k_vi_crops).  Returns CAPACITY-shaped tensors: occ (b*n_per, 4) of occ_dtype, input_map (b*n_per),
    v2p (b*n_per, pitch) and info = device int32 {V, maxActive, error}: the first V rows of occ / v2p are live and equal
    voxelize_idx_gpu's rows bit for bit (v2p[:V, :maxActive+1]); error = a point outside its grid or a voxel with more than
    pitch-1 points."""
    N.need_cuda(coords)
    b, n_per = int(batch_size), int(n_per)
    assert coords.is_contiguous() and coords.dtype == torch.int64 and tuple(coords.shape) == (b * n_per, 4)
    assert 1 <= b <= VI_CROPS_MAX_BATCH and 1 <= n_per <= VI_CROPS_MAX_POINTS and int(S) == VI_CROPS_S
    dev = coords.device
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    with _VI_LOCK:         # (the builder thread of a CropPrefetcher and the caller's thread may both arrive here first)
        ent = _VI_COMM.get(key)
        if ent is None:    # a ring of 64 regions of 2 ints per crop that persist between calls + the call counter: consecutive calls
            ent = _VI_COMM[key] = [torch.zeros(64 * 2 * VI_CROPS_MAX_BATCH, dtype=torch.int32, device=dev), 0]   # (also on other streams / threads) use different words
        ent[1] = ent[1] % 0x3fffffff + 1
        gen = ent[1]
    comm = ent[0][(gen % 64) * 2 * VI_CROPS_MAX_BATCH:]
    occ = torch.empty((b * n_per, 4), dtype=occ_dtype, device=dev)
    input_map = torch.empty(b * n_per, dtype=torch.int32, device=dev)
    v2p = torch.empty((b * n_per, int(pitch)), dtype=torch.int32, device=dev)
    info = torch.empty(3, dtype=torch.int32, device=dev)
    assert occ_dtype in (torch.int64, torch.int32)
    N.check(N.lib().dcl_voxelize_idx_crops(N.ptr(coords), b, n_per, int(S), int(mode), int(pitch), N.ptr(comm), gen,
                                           N.ptr(input_map), N.ptr(occ), int(occ_dtype == torch.int32), N.ptr(v2p), N.ptr(info),
                                           N.stream()), "voxelize_idx_crops")
    return occ, input_map, v2p, info


def voxelize_fp(feats, map_rule, mode=4):
    """PG_OP.voxelize_fp (pointgroup_ops.py:42-62): feats (N,C) f32, map_rule (M,1+maxActive) i32 -> (M,C)."""
    N.need_cuda(feats, map_rule)
    assert feats.is_contiguous() and map_rule.is_contiguous()
    assert feats.dtype == torch.float32 and map_rule.dtype == torch.int32
    M, ma = map_rule.shape[0], map_rule.shape[1] - 1
    out = torch.empty((M, feats.shape[1]), dtype=torch.float32, device=feats.device)
    N.check(N.lib().dcl_voxelize_fp(N.ptr(feats), N.ptr(map_rule), N.ptr(out), M, ma, feats.shape[1],
                                    int(mode == 4), N.stream()), "voxelize_fp")
    return out


class _PadCopyJob(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("rows_dst", C.c_int32), ("cols_dst", C.c_int32),
                ("rows_src", C.c_int32), ("cols_src", C.c_int32), ("src_pitch", C.c_int32), ("src_is_i64", C.c_int32),
                ("fill_value", C.c_int32), ("dst_pitch", C.c_int32)]


def pad_copy_many(jobs):
    """jobs: list of (dst, src) 2-D CUDA tensor pairs of 4-byte elements (src may be int64 -> narrowed; may be smaller than
    dst -> zero padded; may be a column block of a wider row-major tensor) or (dst, int) fills -- ONE launch for all."""
    arr = (_PadCopyJob * len(jobs))()
    for j, (dst, src) in enumerate(jobs):
        assert dst.is_cuda and dst.element_size() == 4
        pitch = 0
        if dst.dim() == 2 and not dst.is_contiguous():                 # a column block of a wider row-major buffer
            assert dst.stride(1) == 1 and dst.stride(0) >= dst.shape[1]
            d2, pitch = dst, dst.stride(0)
        elif dst.dim() == 2:
            d2 = dst
        elif torch.is_tensor(src):
            d2 = dst.reshape(-1, src.shape[1])         # same logical rows as the source
        else:
            d2 = dst.reshape(1, -1)
        q = arr[j]
        assert pitch or dst.is_contiguous()
        q.dst, q.rows_dst, q.cols_dst, q.dst_pitch = dst.data_ptr(), d2.shape[0], d2.shape[1], pitch
        if torch.is_tensor(src):
            assert src.is_cuda and src.dim() == 2 and src.stride(1) == 1 and src.element_size() in (4, 8)
            assert src.element_size() == 4 or src.dtype == torch.int64
            q.src, q.rows_src, q.cols_src = src.data_ptr(), src.shape[0], src.shape[1]
            q.src_pitch = src.stride(0) if src.shape[0] > 1 else src.shape[1]
            q.src_is_i64 = int(src.dtype == torch.int64)
        else:
            q.src, q.fill_value = None, int(src)
    N.check(N.lib().dcl_pad_copy_many(arr, len(jobs), N.stream()), "pad_copy_many")


# ------------------------------------------------------------------------------------ rulebooks
def grid_words(batch, S):
    return (batch * S * S * S + 31) // 32


def scan_scratch(nwords, device):
    return torch.empty(((nwords + 1023) // 1024) + 1, dtype=torch.int32, device=device)


class ActiveSet(object):
    """Device-side description of an active voxel set (see include/dclnet_hip.h, spconv section)."""
    __slots__ = ("indices", "n", "n_dev", "cap", "mask", "wprefix", "perm", "S", "batch")

    def __init__(self, indices, n, n_dev, cap, mask, wprefix, perm, S, batch):
        self.indices, self.n, self.n_dev, self.cap = indices, n, n_dev, cap
        self.mask, self.wprefix, self.perm, self.S, self.batch = mask, wprefix, perm, S, batch

    def segments(self):
        """i32[batch+1]: rows of crop b are [seg[b], seg[b+1]) (rows are in ascending linear index
        unless perm is set)."""
        wpc = (self.S ** 3) // 32
        assert wpc * 32 == self.S ** 3 and self.perm is None
        return self.wprefix[::wpc].contiguous()


def grid_from_indices(indices, batch, S):
    """ActiveSet of an explicit voxel list (rows in any order)."""
    N.need_cuda(indices)
    assert indices.dtype == torch.int32 and indices.is_contiguous() and indices.shape[1] == 4
    V = indices.shape[0]
    dev = indices.device
    nw = grid_words(batch, S)
    mask = torch.empty(nw, dtype=torch.int32, device=dev)
    wprefix = torch.empty(nw + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(max(V, 1), dtype=torch.int32, device=dev)
    N.check(N.lib().dcl_grid_from_indices(N.ptr(indices), V, batch, S, N.ptr(mask), N.ptr(wprefix), N.ptr(perm),
                                          N.ptr(scan_scratch(nw, dev)), N.stream()), "grid_from_indices")
    return ActiveSet(indices, V, None, V, mask, wprefix, perm, S, batch)


def conv_out_size(S, k, s, p):
    return (S + 2 * p - (k - 1) - 1) // s + 1


def conv_out_grid(inp, ksize, stride, padding, cap=None):
    """Output active set of a non-submanifold conv/pool over `inp` (count stays on the device)."""
    S_out = conv_out_size(inp.S, ksize, stride, padding)
    dev = inp.indices.device
    bound = inp.batch * S_out ** 3
    if cap is None:
        cap = min(inp.cap * ksize ** 3, bound)
    cap = max(int(cap), 1)
    nw = grid_words(inp.batch, S_out)
    mask = torch.empty(nw, dtype=torch.int32, device=dev)
    wprefix = torch.empty(nw + 1, dtype=torch.int32, device=dev)
    out_idx = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    n_in_dev = inp.n_dev if inp.n is None else None
    n_in_host = inp.cap if inp.n is None else inp.n
    N.check(N.lib().dcl_conv_out_grid(N.ptr(inp.indices), _i32_ptr_or_null(n_in_dev), int(n_in_host),
                                      N.ptr(inp.mask), inp.batch, inp.S, ksize, stride, padding, N.ptr(mask), N.ptr(wprefix), N.ptr(out_idx),
                                      N.ptr(wprefix[nw:]), cap, N.ptr(scan_scratch(nw, dev)), N.stream()),
            "conv_out_grid")
    return ActiveSet(out_idx, None, wprefix[nw:], cap, mask, wprefix, None, S_out, inp.batch)


def rulebook_gather(out, inp, ksize, stride, padding):
    """Gather-form rulebook nbr i32 (kvol, rows(out)) of `out` rows against the `inp` set."""
    dev = out.indices.device
    rows = out.cap if out.n is None else out.n
    rows_alloc = max(rows, 1)
    nbr = torch.empty((ksize ** 3, rows_alloc), dtype=torch.int32, device=dev)
    n_dev = out.n_dev if out.n is None else None
    N.check(N.lib().dcl_rulebook_gather(N.ptr(out.indices), _i32_ptr_or_null(n_dev), int(rows if out.n is not None else 0),
                                        N.ptr(inp.mask), N.ptr(inp.wprefix), _i32_ptr_or_null(inp.perm), inp.batch,
                                        inp.S, ksize, stride, padding, N.ptr(nbr), rows_alloc, N.stream()),
            "rulebook_gather")
    return nbr


def rulebook_to_pairs(nbr, n_out, n_in):
    """Reference-format rulebook: indice_pairs i32 (kvol,2,n_in) (-1 padded), indice_num i32 (kvol)."""
    kvol, cap = nbr.shape
    pairs = torch.empty((kvol, 2, max(n_in, 1)), dtype=torch.int32, device=nbr.device)
    num = torch.empty(kvol, dtype=torch.int32, device=nbr.device)
    N.check(N.lib().dcl_rulebook_to_pairs(N.ptr(nbr), cap, N.vp(0), int(n_out), kvol, N.ptr(pairs), int(n_in),
                                          N.ptr(num), N.stream()), "rulebook_to_pairs")
    return pairs[:, :, :n_in], num


def rulebook_from_pairs(indice_pairs, indice_num, n_in, n_out, check=False):
    """Reference-format rulebook (indice_pairs i32 (kvol,2,V) -1 padded, indice_num i32 (kvol)) -> gather table
    nbr i32 (kvol, max(n_out,1)).  indice_num may live on the host (it does in the reference after spconv_ops.h:264) or
    on the device; it is used from the device, without a read-back.  check=True reads back the count of pairs that
    pointed outside the row ranges and raises if there were any."""
    N.need_cuda(indice_pairs)
    assert indice_pairs.dtype == torch.int32 and indice_pairs.dim() == 3 and indice_pairs.shape[1] == 2
    pairs = indice_pairs.contiguous()
    kvol, _, stride = pairs.shape
    num = indice_num.to(device=pairs.device, dtype=torch.int32, non_blocking=True).contiguous()
    assert num.numel() == kvol
    cap = max(int(n_out), 1)
    nbr = torch.empty((kvol, cap), dtype=torch.int32, device=pairs.device)
    bad = torch.empty(1, dtype=torch.int32, device=pairs.device) if check else None
    N.check(N.lib().dcl_rulebook_from_pairs(N.ptr(pairs), int(stride), N.ptr(num), kvol, int(n_in), int(n_out), N.ptr(nbr),
                                            cap, N.ptr(bad), N.stream()), "rulebook_from_pairs")
    if check and int(bad.item()):
        raise ValueError("rulebook_from_pairs: %d pairs point outside the feature / output rows" % int(bad.item()))
    return nbr


def indice_summary_rf(indice_pairs, indice_num, n_out):
    """torch.ops.spconv.indiceSummaryRF: receptive-field count i32 (n_out) from the pair format."""
    N.need_cuda(indice_pairs)
    pairs = indice_pairs.contiguous()
    kvol, _, stride = pairs.shape
    num = indice_num.to(device=pairs.device, dtype=torch.int32, non_blocking=True).contiguous()
    rf = torch.empty(int(n_out), dtype=torch.int32, device=pairs.device)
    N.check(N.lib().dcl_indice_summary_rf(N.ptr(pairs), int(stride), N.ptr(num), kvol, int(n_out), N.ptr(rf), N.stream()),
            "indice_summary_rf")
    return rf


def sparse_avgpool_rf(feat, nbr, n_out, summaryrf):
    """indice_avgpool_fp32 with the caller's divisor tensor (use_gs=True passes the kernel volume)."""
    N.need_cuda(feat, nbr, summaryrf)
    kvol, cap = nbr.shape
    c = feat.shape[1]
    out = torch.empty((n_out, c), dtype=torch.float32, device=feat.device)
    if n_out:
        rf = N.i32c(summaryrf)
        assert rf.numel() >= n_out
        N.check(N.lib().dcl_sparse_avgpool_fwd_rf(N.ptr(feat.contiguous()), N.ptr(nbr), cap, N.vp(0), int(n_out), c, kvol,
                                                  N.ptr(rf), N.ptr(out), N.stream()), "sparse_avgpool_fwd_rf")
    return out


def order_rows(out_set, in_mask, subm):
    """Row order of a k3 / s1 / p1 conv layer with output set `out_set` (ActiveSet) and the INPUT set's occupancy bits
    `in_mask` on the same batch x S^3 grid (csrc/row_order.hip) -> (order (n,) i32, bal (ntiles+1,) i32, smask (ntiles,) i32)
    for sparse_conv(..., order=...): tile slot i of the launch computes output row order[i]."""
    N.need_cuda(out_set.indices, in_mask)
    n, dev = int(out_set.n), out_set.indices.device
    cap = max(n, 1)
    tiles = (cap + 127) // 128
    nbytes = C.c_int64(0)
    N.check(N.lib().dcl_order_rows_ws_bytes(cap, C.byref(nbytes)), "order_rows_ws_bytes")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
    order = torch.empty(cap, dtype=torch.int32, device=dev)
    bal = torch.zeros(tiles + 2, dtype=torch.int32, device=dev)
    smask = torch.zeros(tiles + 1, dtype=torch.int32, device=dev)
    N.check(N.lib().dcl_order_rows(N.ptr(out_set.indices), N.vp(0), n, cap, N.ptr(in_mask), int(out_set.S), int(bool(subm)),
                                   N.ptr(ws), nbytes.value, N.ptr(order), N.ptr(bal), N.ptr(smask), N.stream()), "order_rows")
    nt = (n + 127) // 128
    return order[:n], bal[:nt + 1], smask[:nt]


def sparse_conv(feat, nbr, n_out, W, subm, scale=None, shift=None, relu=False, order=None):
    """indice_conv_fp32 (+ folded BatchNorm1d(eval) + ReLU).  feat (V_in,Cin); W (kvol,Cin,Cout).  order: the triple of
    order_rows -- the launch then computes its rows in that order and deals its work in used chunks (same output)."""
    N.need_cuda(feat, nbr, W)
    kvol, cap = nbr.shape
    cin, cout = W.shape[-2], W.shape[-1]
    assert feat.is_contiguous() and W.is_contiguous() and feat.shape[1] == cin
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feat.device)
    if n_out == 0:
        return out
    # stream-K scratch (partial-tile slots + tile tickets, csrc/sparse_conv.hip::launch_conv_dma)
    scratch = None
    if cout % 32 == 0:
        need = C.c_int64(0)
        N.check(N.lib().dcl_sparse_conv_scratch_floats(int(cap), int(cout), C.byref(need)), "sparse_conv_scratch_floats")
        scratch = torch.empty(need.value, dtype=torch.float32, device=feat.device)
    if order is not None:
        o, bal, smask = order
        assert o.numel() >= n_out and o.dtype == torch.int32 and o.is_cuda
        N.check(N.lib().dcl_sparse_conv_fwd_ordered(N.ptr(feat), N.ptr(nbr), cap, N.vp(0), int(n_out), N.ptr(W), cin, cout, kvol,
                                                    int(bool(subm)), N.ptr(scale), N.ptr(shift), int(bool(relu)), N.ptr(out),
                                                    N.ptr(scratch), C.c_int64(0 if scratch is None else scratch.numel()),
                                                    N.ptr(o), N.ptr(bal), N.ptr(smask), N.stream()), "sparse_conv_fwd_ordered")
        return out
    N.check(N.lib().dcl_sparse_conv_fwd_ws(N.ptr(feat), N.ptr(nbr), cap, N.vp(0), int(n_out), N.ptr(W), cin, cout, kvol,
                                           int(bool(subm)), N.ptr(scale), N.ptr(shift), int(bool(relu)), N.ptr(out),
                                           N.ptr(scratch), C.c_int64(0 if scratch is None else scratch.numel()),
                                           N.stream()), "sparse_conv_fwd")
    return out


def sparse_avgpool(feat, nbr, n_out, want_rf=False):
    """indiceSummaryRF + indice_avgpool_fp32 (use_gs=False)."""
    N.need_cuda(feat, nbr)
    kvol, cap = nbr.shape
    c = feat.shape[1]
    out = torch.empty((n_out, c), dtype=torch.float32, device=feat.device)
    rf = torch.empty(max(n_out, 1), dtype=torch.int32, device=feat.device) if want_rf else None
    if n_out:
        N.check(N.lib().dcl_sparse_avgpool_fwd(N.ptr(feat), N.ptr(nbr), cap, N.vp(0), int(n_out), c, kvol, N.ptr(out),
                                               N.ptr(rf), N.stream()), "sparse_avgpool_fwd")
    return (out, rf[:n_out]) if want_rf else out


# ------------------------------------------------------------------------------------ native backbone runner
BACKBONE_CHANNELS = (7, 16, 32, 32, 64, 64, 128, 128, 256)


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class BackboneRun(object):
    """State of one backbone pass driven by the native runner (csrc/backbone.hip)."""

    def __init__(self, occ, batch, S, batch_lo=0, counts_dev=None):
        """occ (V0,4) i32 [b,x,y,z]; batch_lo > 0 (or batch < number of crops in occ): pass over the crop window
        batch_lo .. batch_lo+batch-1 only (its crops are re-based to 0).  counts_dev: optional i32[8] slice of a caller's
        tensor that receives the level sizes -- a device tensor (several passes can then be read back with one copy) or
        PINNED host memory, which the geometry kernels write directly (the caller waits for the stream and reads it)."""
        N.need_cuda(occ)
        assert occ.dtype == torch.int32 and occ.is_contiguous()
        self.occ, self.batch, self.S, self.V0 = occ, int(batch), int(S), occ.shape[0]
        nbytes = C.c_int64(0)
        N.check(N.lib().dcl_backbone_ws_bytes(self.batch, self.S, self.V0, C.byref(nbytes)), "backbone_ws_bytes")
        self.ws = torch.empty(nbytes.value, dtype=torch.uint8, device=occ.device)
        if counts_dev is not None and not counts_dev.is_cuda and not counts_dev.is_pinned():
            raise RuntimeError("BackboneRun: a host counts buffer must be pinned (the geometry kernels write into it)")
        self.counts_dev = torch.empty(8, dtype=torch.int32, device=occ.device) if counts_dev is None else counts_dev
        self.chan = (C.c_int32 * 9)(*BACKBONE_CHANNELS)
        N.check(N.lib().dcl_backbone_geometry_window(N.ptr(occ), self.V0, int(batch_lo), self.batch, self.S, N.ptr(self.ws),
                                                     nbytes.value, N.ptr(self.counts_dev), N.stream()),
                "backbone_geometry")
        self.counts = None
        self.levels = None

    def set_counts(self, counts):
        self.counts = [int(c) for c in counts]
        self.ccounts = (C.c_int32 * 8)(*self.counts)

    def features(self, vox_feats, weights_arr, scales_arr, shifts_arr):
        """-> list of 4 level feature tensors (n_pool_m, C_m)."""
        dev = vox_feats.device
        nbytes = C.c_int64(0)
        N.check(N.lib().dcl_backbone_ws2_bytes(self.ccounts, self.chan, C.byref(nbytes)), "backbone_ws2_bytes")
        ws2 = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        self.levels = [torch.empty((self.counts[2 * m + 1], BACKBONE_CHANNELS[2 * m + 2]), dtype=torch.float32,
                                   device=dev) for m in range(4)]
        N.check(N.lib().dcl_backbone_features(N.ptr(self.occ), self.V0, self.batch, self.S, N.ptr(self.ws), self.ccounts,
                                              self.chan, N.ptr(vox_feats), weights_arr, scales_arr, shifts_arr,
                                              N.ptr(ws2), nbytes.value, _ptr_array(self.levels), N.stream()),
                "backbone_features")
        return self.levels

    def level_indices(self, m):
        """(n_pool_m, 4) int32 view of pooled level m's voxel rows inside the workspace."""
        off, wp, Sl = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        N.check(N.lib().dcl_backbone_level_info(self.batch, self.S, self.V0, m, C.byref(off), C.byref(wp), C.byref(Sl)),
                "backbone_level_info")
        n = self.counts[2 * m + 1]
        return self.ws[off.value:off.value + 16 * n].view(torch.int32).view(n, 4)

    def point_features(self, points_b4, voxel_extents, offset, out=None):
        """points_b4 (n,4) -> (n,480) interpolated multi-scale features."""
        n = points_b4.shape[0]
        dev = points_b4.device
        if out is None:
            out = torch.empty((n, 480), dtype=torch.float32, device=dev)
        tmp_bytes = 2 * (((n * 48) + 255) // 256 * 256)                       # dist2 + idx of the 4 levels
        tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=dev)
        ve = (C.c_float * 4)(*[float(v) for v in voxel_extents])
        N.check(N.lib().dcl_point_features(n, N.ptr(points_b4), self.batch, self.S, self.V0, N.ptr(self.ws), self.ccounts,
                                           self.chan, _ptr_array(self.levels), ve, _c_float(offset), N.ptr(out),
                                           out.stride(0), N.ptr(tmp), tmp_bytes, N.stream()), "point_features")
        return out

    def point_neighbours(self, points_b4, voxel_extents, offset):
        """first half of point_features: the four 3-NN searches (need geometry + counts only, not the level features).
        Returns (dist2, idx), each (4, n, 3)."""
        n = points_b4.shape[0]
        dev = points_b4.device
        dist2 = torch.empty((4, n, 3), dtype=torch.float32, device=dev)
        idx = torch.empty((4, n, 3), dtype=torch.int32, device=dev)
        ve = (C.c_float * 4)(*[float(v) for v in voxel_extents])
        N.check(N.lib().dcl_point_neighbours(n, N.ptr(points_b4), self.batch, self.S, self.V0, N.ptr(self.ws), self.ccounts,
                                             ve, _c_float(offset), N.ptr(dist2), N.ptr(idx), N.stream()),
                "point_neighbours")
        return dist2, idx

    def point_interpolate(self, dist2, idx, out=None):
        """second half of point_features: inverse-distance interpolation of the four levels' features -> (n, 480)."""
        n = idx.shape[1]
        if out is None:
            out = torch.empty((n, 480), dtype=torch.float32, device=idx.device)
        N.check(N.lib().dcl_point_interpolate(n, self.ccounts, self.chan, _ptr_array(self.levels), N.ptr(dist2),
                                              N.ptr(idx), N.ptr(out), out.stride(0), N.stream()), "point_interpolate")
        return out


def backbone_features_pair(run_a, vox_a, ptrs_a, run_b, vox_b, ptrs_b):
    """The feature stage of BOTH backbones of a forward (observed / template side; two BackboneRun or two BackboneRunCap of
    the same batch and grid): every layer is one launch over both sides' tiles (dcl_backbone_features_pair), so the fixed
    part of a launch is paid once per layer, not once per layer and side.  ptrs_*: (weights, scales, shifts) pointer
    arrays of a side.  Fills run_*.levels like run.features()."""
    cap_mode = isinstance(run_a, BackboneRunCap)
    assert isinstance(run_b, type(run_a)) and run_a.batch == run_b.batch and run_a.S == run_b.S
    dev = vox_a.device
    runs, voxs, ptrs = (run_a, run_b), (vox_a, vox_b), (ptrs_a, ptrs_b)
    if cap_mode:
        ws2 = [r.ws2 for r in runs]
        ws2_bytes = [r.ws2_bytes for r in runs]
        counts_host, counts_dev = None, (C.c_void_p * 2)(*[r.counts_dev.data_ptr() for r in runs])
    else:
        ws2, ws2_bytes = [], []
        for r in runs:
            nbytes = C.c_int64(0)
            N.check(N.lib().dcl_backbone_ws2_bytes(r.ccounts, r.chan, C.byref(nbytes)), "backbone_ws2_bytes")
            ws2.append(torch.empty(nbytes.value, dtype=torch.uint8, device=dev))
            ws2_bytes.append(nbytes.value)
            r.levels = [torch.empty((r.counts[2 * m + 1], BACKBONE_CHANNELS[2 * m + 2]), dtype=torch.float32, device=dev)
                        for m in range(4)]
        counts_host = (C.POINTER(C.c_int32) * 2)(*[C.cast(r.ccounts, C.POINTER(C.c_int32)) for r in runs])
        counts_dev = None
    level_ptrs = [r.level_ptrs if cap_mode else _ptr_array(r.levels) for r in runs]
    pp = lambda arrs: (C.c_void_p * 2)(*[C.cast(a, C.c_void_p) for a in arrs])                # noqa: E731
    N.check(N.lib().dcl_backbone_features_pair(
        run_a.batch, run_a.S, run_a.chan, (C.c_int32 * 2)(run_a.V0, run_b.V0),
        (C.c_void_p * 2)(*[r.ws.data_ptr() for r in runs]), counts_host, counts_dev,
        (C.c_void_p * 2)(*[v.data_ptr() for v in voxs]), pp([p[0] for p in ptrs]), pp([p[1] for p in ptrs]),
        pp([p[2] for p in ptrs]), (C.c_void_p * 2)(*[w.data_ptr() for w in ws2]), (C.c_int64 * 2)(*ws2_bytes),
        pp(level_ptrs), N.stream()), "backbone_features_pair")
    if not cap_mode:
        for r, w in zip(runs, ws2):
            r._ws2_keep = w                                # the levels are separate tensors; the scratch may go when the run goes
    return run_a.levels, run_b.levels


class BackboneRunCap(object):
    """Capacity-mode backbone pass (whole-forward hipGraph capture): every buffer is sized from (batch, S, V0_cap) alone,
    live row counts stay on the device, no host read-back.  `occ` is a STATIC (V0_cap,4) buffer whose first *v0_dev rows
    are live.  All methods only enqueue work, so they can run under torch.cuda.graph()."""

    def __init__(self, occ, v0_dev, batch, S):
        N.need_cuda(occ, v0_dev)
        assert occ.dtype == torch.int32 and occ.is_contiguous() and v0_dev.dtype == torch.int32
        self.occ, self.v0_dev, self.batch, self.S, self.V0 = occ, v0_dev, int(batch), int(S), occ.shape[0]
        dev = occ.device
        nbytes = C.c_int64(0)
        N.check(N.lib().dcl_backbone_ws_bytes(self.batch, self.S, self.V0, C.byref(nbytes)), "backbone_ws_bytes")
        self.ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        self.ws_bytes = nbytes.value
        self.counts_dev = torch.zeros(8, dtype=torch.int32, device=dev)
        self.chan = (C.c_int32 * 9)(*BACKBONE_CHANNELS)
        self.caps = (C.c_int32 * 8)()
        N.check(N.lib().dcl_backbone_caps(self.batch, self.S, self.V0, self.caps), "backbone_caps")
        n2 = C.c_int64(0)
        N.check(N.lib().dcl_backbone_ws2_bytes(self.caps, self.chan, C.byref(n2)), "backbone_ws2_bytes")
        self.ws2 = torch.empty(n2.value, dtype=torch.uint8, device=dev)
        self.ws2_bytes = n2.value
        self.levels = [torch.empty((self.caps[2 * m + 1], BACKBONE_CHANNELS[2 * m + 2]), dtype=torch.float32, device=dev)
                       for m in range(4)]
        self.level_ptrs = _ptr_array(self.levels)

    def geometry(self, voxelize=None):
        """voxelize = (feats (N,C) f32, map_rule (M, 1 + maxActive) i32, mode): also returns voxelize_fp(feats, map_rule, mode),
        carried by the geometry stage's own launch where that is one launch (up to 16 crops)"""
        if voxelize is None:
            N.check(N.lib().dcl_backbone_geometry_cap(N.ptr(self.occ), N.ptr(self.v0_dev), self.V0, self.batch, self.S,
                                                      N.ptr(self.ws), self.ws_bytes, N.ptr(self.counts_dev), N.stream()),
                    "backbone_geometry_cap")
            return None
        feats, rule, mode = voxelize
        N.need_cuda(feats, rule)
        assert feats.is_contiguous() and rule.is_contiguous() and feats.dtype == torch.float32 and rule.dtype == torch.int32
        out = torch.empty((rule.shape[0], feats.shape[1]), dtype=torch.float32, device=feats.device)
        N.check(N.lib().dcl_backbone_geometry_cap_vox(N.ptr(self.occ), N.ptr(self.v0_dev), self.V0, self.batch, self.S,
                                                      N.ptr(self.ws), self.ws_bytes, N.ptr(self.counts_dev), N.ptr(feats),
                                                      N.ptr(rule), N.ptr(out), rule.shape[0], rule.shape[1] - 1, feats.shape[1],
                                                      int(mode == 4), N.stream()), "backbone_geometry_cap_vox")
        return out

    def features(self, vox_feats, weights_arr, scales_arr, shifts_arr):
        N.check(N.lib().dcl_backbone_features_cap(N.ptr(self.occ), self.V0, self.batch, self.S, N.ptr(self.ws),
                                                  N.ptr(self.counts_dev), self.chan, N.ptr(vox_feats), weights_arr,
                                                  scales_arr, shifts_arr, N.ptr(self.ws2), self.ws2_bytes,
                                                  self.level_ptrs, N.stream()), "backbone_features_cap")
        return self.levels

    def point_features(self, points_b4, voxel_extents, offset, out, tmp):
        n = points_b4.shape[0]
        ve = (C.c_float * 4)(*[float(v) for v in voxel_extents])
        N.check(N.lib().dcl_point_features_cap(n, N.ptr(points_b4), self.batch, self.S, self.V0, N.ptr(self.ws),
                                               N.ptr(self.counts_dev), self.chan, self.level_ptrs, ve, _c_float(offset),
                                               N.ptr(out), out.stride(0), N.ptr(tmp), tmp.numel(), N.stream()),
                "point_features_cap")
        return out

    def tmp_bytes(self, n):
        return 2 * (((n * 48) + 255) // 256 * 256)

    def point_neighbours(self, points_b4, voxel_extents, offset, dist2, idx):
        """dist2 / idx: static (4, n, 3) buffers (see BackboneRun.point_neighbours)"""
        n = points_b4.shape[0]
        ve = (C.c_float * 4)(*[float(v) for v in voxel_extents])
        N.check(N.lib().dcl_point_neighbours_cap(n, N.ptr(points_b4), self.batch, self.S, self.V0, N.ptr(self.ws), ve,
                                                 _c_float(offset), N.ptr(dist2), N.ptr(idx), N.stream()),
                "point_neighbours_cap")

    def point_interpolate(self, dist2, idx, out):
        n = idx.shape[1]
        N.check(N.lib().dcl_point_interpolate(n, self.caps, self.chan, self.level_ptrs, N.ptr(dist2), N.ptr(idx),
                                              N.ptr(out), out.stride(0), N.stream()), "point_interpolate")
        return out


# ------------------------------------------------------------------------------------ pointnet_sp
def three_nn_sp(unknown, known, known_seg=None):
    """pointnet2_cuda.three_nn_wrapper of libs/pointnet_sp: returns (dist2 (N,3), idx (N,3) i32)."""
    N.need_cuda(unknown, known)
    assert unknown.is_contiguous() and known.is_contiguous()
    assert unknown.shape[1] == 4 and known.shape[1] == 4
    n, m = unknown.shape[0], known.shape[0]
    dist2 = torch.empty((n, 3), dtype=torch.float32, device=unknown.device)
    idx = torch.empty((n, 3), dtype=torch.int32, device=unknown.device)
    nb = 0 if known_seg is None else known_seg.numel() - 1
    N.check(N.lib().dcl_three_nn_sp(n, m, N.ptr(unknown), N.ptr(known), N.ptr(dist2), N.ptr(idx),
                                    _i32_ptr_or_null(known_seg), nb, N.stream()), "three_nn_sp")
    return dist2, idx


def three_interpolate_sp(features, idx, weight, out=None, from_dist2=False):
    """three_interpolate_wrapper of libs/pointnet_sp: features (M,C), idx (n,3), weight (n,3) -> (n,C).
    `out` may be a column block of a wider row-major buffer (its row stride is honoured);
    from_dist2=True treats `weight` as three_nn's dist2 and forms the weights in-kernel."""
    N.need_cuda(features, idx, weight)
    assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
    m, c = features.shape
    n = idx.shape[0]
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=features.device)
    assert out.shape == (n, c) and out.stride(1) == 1
    fn = N.lib().dcl_three_interpolate_dist2_sp if from_dist2 else N.lib().dcl_three_interpolate_sp
    N.check(fn(c, m, n, N.ptr(features), N.ptr(idx), N.ptr(weight), N.ptr(out), out.stride(0) if n > 1 else c,
               N.stream()), "three_interpolate_sp")
    return out


def voxel_centres(aset_or_indices, ve, off, n=None):
    """Ops_tensor2points (models/Modules.py:204-211) on device."""
    ind = aset_or_indices
    n = ind.shape[0] if n is None else n
    out = torch.empty((max(n, 1), 4), dtype=torch.float32, device=ind.device)
    N.check(N.lib().dcl_voxel_centres(N.ptr(ind), N.vp(0), int(n), _c_float(ve), _c_float(off), N.ptr(out), N.stream()),
            "voxel_centres")
    return out[:n]


# ------------------------------------------------------------------------------------ pointnet_lib
def ball_query(radius, nsample, xyz, new_xyz):
    N.need_cuda(xyz, new_xyz)
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    B, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.empty((B, m, nsample), dtype=torch.int32, device=xyz.device)
    N.check(N.lib().dcl_ball_query(B, n, m, _c_float(radius), int(nsample), N.ptr(new_xyz), N.ptr(xyz), N.ptr(idx),
                                   N.stream()), "ball_query")
    return idx


def group_points(features, idx, out=None):
    """out (optional): a channel block out_full[:, c0:c0+c] of a contiguous (B, C_total, npoint, nsample) tensor, filled
    in place (no concatenation copy afterwards)."""
    N.need_cuda(features, idx, out)
    assert features.is_contiguous() and idx.is_contiguous() and idx.dtype == torch.int32
    B, c, n = features.shape
    _, npoint, ns = idx.shape
    if out is None:
        out = torch.empty((B, c, npoint, ns), dtype=torch.float32, device=features.device)
    assert out.shape == (B, c, npoint, ns) and out.stride(3) == 1 and out.stride(2) == ns and out.stride(1) == npoint * ns
    oc = out.stride(0) // (npoint * ns) if B > 1 else c
    assert B == 1 or out.stride(0) == oc * npoint * ns
    N.check(N.lib().dcl_group_points_into(B, c, n, npoint, ns, N.ptr(features), N.ptr(idx), N.ptr(out), int(oc),
                                          N.stream()), "group_points")
    return out


def gather_points(features, idx):
    N.need_cuda(features, idx)
    assert features.is_contiguous() and idx.is_contiguous() and idx.dtype == torch.int32
    B, c, n = features.shape
    m = idx.shape[1]
    out = torch.empty((B, c, m), dtype=torch.float32, device=features.device)
    N.check(N.lib().dcl_gather_points(B, c, n, m, N.ptr(features), N.ptr(idx), N.ptr(out), N.stream()), "gather_points")
    return out


def furthest_point_sampling(xyz, npoint):
    N.need_cuda(xyz)
    assert xyz.is_contiguous()
    B, n, _ = xyz.shape
    temp = torch.full((B, n), 1e10, dtype=torch.float32, device=xyz.device)
    idx = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    N.check(N.lib().dcl_furthest_point_sampling(B, n, int(npoint), N.ptr(xyz), N.ptr(temp), N.ptr(idx), N.stream()),
            "furthest_point_sampling")
    return idx


def knn(k, unknown, known):
    N.need_cuda(unknown, known)
    assert unknown.is_contiguous() and known.is_contiguous()
    B, n, _ = unknown.shape
    m = known.shape[1]
    if k <= 3:
        # the sorted strict-'<' list of length k is the first k entries of the 3-NN cascade (same tie rule, same
        # defaults for missing neighbours): use the tiled three_nn kernel (DCL-Net only calls knn with k = 1)
        d2, idx = three_nn(unknown, known)
        return d2[:, :, :k].contiguous(), idx[:, :, :k].contiguous()
    dist2 = torch.empty((B, n, k), dtype=torch.float32, device=unknown.device)
    idx = torch.empty((B, n, k), dtype=torch.int32, device=unknown.device)
    N.check(N.lib().dcl_knn(B, n, m, int(k), N.ptr(unknown), N.ptr(known), N.ptr(dist2), N.ptr(idx), N.stream()), "knn")
    return dist2, idx


def three_nn(unknown, known):
    N.need_cuda(unknown, known)
    assert unknown.is_contiguous() and known.is_contiguous()
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknown.device)
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknown.device)
    N.check(N.lib().dcl_three_nn(B, n, m, N.ptr(unknown), N.ptr(known), N.ptr(dist2), N.ptr(idx), N.stream()),
            "three_nn")
    return dist2, idx


def three_interpolate(features, idx, weight):
    N.need_cuda(features, idx, weight)
    assert features.is_contiguous() and idx.is_contiguous() and weight.is_contiguous()
    B, c, m = features.shape
    n = idx.shape[1]
    out = torch.empty((B, c, n), dtype=torch.float32, device=features.device)
    N.check(N.lib().dcl_three_interpolate(B, c, m, n, N.ptr(features), N.ptr(idx), N.ptr(weight), N.ptr(out),
                                          N.stream()), "three_interpolate")
    return out


# ------------------------------------------------------------------------------------ dense path
def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1
    return t.stride(0) if t.shape[0] > 1 else t.shape[1]


ATTENTION_SPLIT = True        # large attention calls: P.V on the bf16 matrix pipe at fp32-sized errors (False: fp32 MFMA everywhere)


def attention_planes(b, nq, nk, concurrent=1, device=None):
    """scratch for the K / V pieces of a cross_attention call of this size that takes the split-bf16 kernel: (planes, whole) --
    planes None = the call keeps the fp32 kernel; whole = ALL b crops take the split kernel, i.e. the caller may have V1 written
    as pieces straight into `planes` by the GEMM that makes it (linear_split_vpieces) and pass V1 = None."""
    if not ATTENTION_SPLIT:
        return None, False
    lib = N.lib()
    lib.dcl_cross_attention_planes_bytes.restype = C.c_int64
    pb = int(lib.dcl_cross_attention_planes_bytes(int(b), int(nq), int(nk), int(concurrent)))
    if not pb:
        return None, False
    whole = int(lib.dcl_cross_attention_split_crops(int(b), int(nq), int(nk), int(concurrent))) == int(b)
    return torch.empty(pb, dtype=torch.uint8, device=device if device is not None else torch.device("cuda")), whole


def cross_attention(b, Q, K, V1, O1, V2=None, O2=None, concurrent=1, planes=None):
    """One direction of the correspondence attention on POINT-major 2-D operands (row = point):
    Q (b*nq, 64), K (b*nk, 64), V1 (b*nk, dv1) -> O1 (b*nq, dv1) [, V2 -> O2].  Operands may be
    column blocks of wider buffers (row stride honoured).  The PRODUCT library carries DCL-Net's own channel split only,
    dv1 = 256 with dv2 = 64 (anything else: DCL_EINVAL at run time -- include/dclnet_hip.h says so at dcl_cross_attention);
    the general-shape kernels (dv1, dv2 multiples of 32) live in the diagnostic library.
    concurrent = 2: another launch of the same size runs side by side on a second stream (the other direction of a forward) --
    a hint for the launcher's choice of workgroup shape (dcl_cross_attention_ws2)."""
    N.need_cuda(Q, K, V1, O1, V2, O2)
    nq, nk = Q.shape[0] // b, K.shape[0] // b
    assert Q.shape[1] == 64 and K.shape[1] == 64 and O1.shape[0] == Q.shape[0]
    assert V1 is not None or planes is not None, "V1 = None: its pieces are expected in `planes` (attention_planes, linear_split_vpieces)"
    assert V1 is None or V1.shape[0] == K.shape[0]
    dv1 = 256 if V1 is None else V1.shape[1]
    dv2 = 0 if V2 is None else V2.shape[1]
    ev = None
    if PROFILE_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    scratch = None
    if dv1 == 256 and dv2 == 64:          # key-split partial records (small launches, badly quantised large ones)
        need = C.c_int64(0)
        N.check(N.lib().dcl_cross_attention_scratch_floats(b, nq, C.byref(need)), "cross_attention_scratch_floats")
        if need.value:
            scratch = torch.empty(need.value, dtype=torch.float32, device=Q.device)
    if planes is None and dv1 == 256 and dv2 == 64:    # large calls: scratch for the K / V pieces of the split-bf16 kernel (csrc/dense.hip)
        planes, _ = attention_planes(b, nq, nk, concurrent, Q.device)
    N.check(N.lib().dcl_cross_attention_ws3(b, nq, nk, N.ptr(Q), _ld(Q), N.ptr(K), _ld(K), N.ptr(V1), dv1, 256 if V1 is None else _ld(V1),
                                            N.ptr(O1), _ld(O1), N.ptr(V2), dv2, 0 if V2 is None else _ld(V2), N.ptr(O2),
                                            0 if O2 is None else _ld(O2), N.ptr(scratch),
                                            C.c_int64(0 if scratch is None else scratch.numel()), int(concurrent), N.ptr(planes),
                                            C.c_int64(0 if planes is None else planes.numel()), N.stream()),
            "cross_attention")
    if ev is not None:
        ev[1].record()
        PROFILE_EVENTS.append(("cross_attention", ev[0], ev[1]))
    return O1, O2


def conf_pool(b, logit1, logit2, F1, F2, affine=None, finish=True):
    """Confidence pooling: logits (b*n1,)/(b*n2,), F1 (b*n1,C), F2 (b*n2,C) point-major ->
    conf (b,n1+n2), pooled1 (b,C), pooled2 (b,C), wsum (b,2).
    affine = (s1, t1, s2, t2), each (C,): returns (conf, ((s1*P1 + t1*wsum1) + s2*P2) + t2*wsum2) instead -- the pooled
    feature behind trailing BatchNorms that were not applied to F1 / F2 -- finished in one launch; finish=False: (conf, parts)
    with the slice partials for pose_heads_parts, which finishes them inside its first launch."""
    N.need_cuda(logit1, logit2, F1, F2)
    n1, n2 = F1.shape[0] // b, F2.shape[0] // b
    c = F1.shape[1]
    assert logit1.is_contiguous() and logit2.is_contiguous() and logit1.numel() == b * n1 and logit2.numel() == b * n2
    dev = F1.device
    # enough (crop, channel-chunk, slice) blocks to fill 256 CUs several times over
    nslices = max(1, min(64, 2048 // max(1, b * ((c + 255) // 256))))
    conf = torch.empty((b, n1 + n2), dtype=torch.float32, device=dev)
    w = torch.empty((b, n1 + n2), dtype=torch.float32, device=dev)
    part1 = torch.empty((b, nslices, c), dtype=torch.float32, device=dev)
    part2 = torch.empty((b, nslices, c), dtype=torch.float32, device=dev)
    ws = torch.empty((b, 2), dtype=torch.float32, device=dev)
    N.check(N.lib().dcl_conf_pool(b, c, n1, n2, N.ptr(logit1), N.ptr(logit2), N.ptr(F1), _ld(F1), N.ptr(F2), _ld(F2),
                                  N.ptr(conf), N.ptr(w), nslices, N.ptr(part1), N.ptr(part2), N.ptr(ws), N.stream()),
            "conf_pool")
    if affine is not None:
        s1, t1, s2, t2 = affine
        N.need_cuda(s1, t1, s2, t2)
        assert all(t.is_contiguous() and t.numel() == c and t.dtype == torch.float32 for t in affine)
        if not finish:                                 # the caller folds the finish into its next launch (pose_heads_parts)
            return conf, (nslices, part1, part2, ws)
        out = torch.empty((b, c), dtype=torch.float32, device=dev)
        N.check(N.lib().dcl_pool_finish(b, c, nslices, N.ptr(part1), N.ptr(part2), N.ptr(ws), N.ptr(s1), N.ptr(t1),
                                        N.ptr(s2), N.ptr(t2), N.ptr(out), N.stream()), "pool_finish")
        return conf, out
    return conf, part1.sum(dim=1), part2.sum(dim=1), ws


def affine3_relu(xyz, W3, term):
    """relu(term + xyz @ W3): xyz (rows,3), W3 (3,c), term (rows,c) -> (rows,c), one pass (csrc/dense.hip: k_affine3_relu) -- the
    xyz part of the refiner's first shared layer (models/refiner.py:78-80)."""
    N.need_cuda(xyz, W3, term)
    assert xyz.is_contiguous() and W3.is_contiguous() and term.is_contiguous() and xyz.shape[1] == 3
    rows, c = term.shape
    assert tuple(W3.shape) == (3, c) and xyz.shape[0] == rows
    out = torch.empty_like(term)
    N.check(N.lib().dcl_affine3_relu(C.c_int64(rows), int(c), N.ptr(xyz), N.ptr(W3), N.ptr(term), N.ptr(out), N.stream()),
            "affine3_relu")
    return out


def pose_heads(pooled, rot_layers, trans_layers, with_rotation=False):
    """regressor_rot / regressor_trans on the pooled (b,1024) feature for a handful of crops: both 3-layer heads in two
    launches.  *_layers: [(W1t, b1), (W2t, b2), (W3t, b3)] with (in, out) matrices -> (o9 (b,9), trans (b,3)); with_rotation:
    also R (b,3,3) = ortho9d_to_matrix(o9), formed by the second launch."""
    N.need_cuda(pooled)
    pooled = pooled.contiguous()
    b, dev = pooled.shape[0], pooled.device
    assert pooled.shape[1] == 1024 and tuple(rot_layers[0][0].shape) == (1024, 512) and tuple(rot_layers[2][0].shape) == (128, 9)
    assert tuple(trans_layers[1][0].shape) == (512, 128) and tuple(trans_layers[2][0].shape) == (128, 3)
    flat = lambda layers: _ptr_array([t for pair in layers for t in pair])                    # noqa: E731
    assert all(t.is_contiguous() and t.dtype == torch.float32 for layers in (rot_layers, trans_layers) for p in layers for t in p)
    h1 = torch.empty((2, b, 512), dtype=torch.float32, device=dev)
    o9 = torch.empty((b, 9), dtype=torch.float32, device=dev)
    trans = torch.empty((b, 3), dtype=torch.float32, device=dev)
    R = torch.empty((b, 3, 3), dtype=torch.float32, device=dev) if with_rotation else None
    N.check(N.lib().dcl_pose_heads(b, N.ptr(pooled), flat(rot_layers), flat(trans_layers), N.ptr(h1), N.ptr(o9), N.ptr(trans),
                                   N.ptr(R), N.stream()), "pose_heads")
    return (o9, trans, R) if with_rotation else (o9, trans)


def pose_heads_parts(parts, affine, rot_layers, trans_layers, with_rotation=False):
    """pose_heads on the pooled feature still in parts (conf_pool(..., affine, finish=False)): the pooling's finish runs inside
    the heads' first launch -- a launch less for a handful of crops, bit-identical to conf_pool(finish=True) + pose_heads."""
    nslices, part1, part2, ws = parts
    s1, t1, s2, t2 = affine
    b, dev = part1.shape[0], part1.device
    assert part1.shape[2] == 1024 and tuple(rot_layers[0][0].shape) == (1024, 512) and tuple(rot_layers[2][0].shape) == (128, 9)
    assert tuple(trans_layers[1][0].shape) == (512, 128) and tuple(trans_layers[2][0].shape) == (128, 3)
    flat = lambda layers: _ptr_array([t for pair in layers for t in pair])                    # noqa: E731
    assert all(t.is_contiguous() and t.dtype == torch.float32 for layers in (rot_layers, trans_layers) for p in layers for t in p)
    h1 = torch.empty((2, b, 512), dtype=torch.float32, device=dev)
    o9 = torch.empty((b, 9), dtype=torch.float32, device=dev)
    trans = torch.empty((b, 3), dtype=torch.float32, device=dev)
    R = torch.empty((b, 3, 3), dtype=torch.float32, device=dev) if with_rotation else None
    N.check(N.lib().dcl_pose_heads_parts(b, int(nslices), N.ptr(part1), N.ptr(part2), N.ptr(ws), N.ptr(s1), N.ptr(t1), N.ptr(s2),
                                         N.ptr(t2), flat(rot_layers), flat(trans_layers), N.ptr(h1), N.ptr(o9), N.ptr(trans),
                                         N.ptr(R), N.stream()), "pose_heads_parts")
    return (o9, trans, R) if with_rotation else (o9, trans)


VENDOR_GEMM = False           # tools / tests only: True sends linear() to the vendor library whatever the shape (A/B runs)


def linear(x, Wt, bias=None, relu=False, out=None):
    """One per-point linear layer (Conv1d k=1 / 1x1x1 Conv3d + folded BN of the reference's MLP stacks): act(x @ Wt + bias).
    x (M,K), Wt (K,N), out (M,N) are row-major 2-D tensors whose rows may be strided (column blocks of wider buffers):
    `out=buf[:, 256:512]` is written in place, no copy.  Runs on the library's OWN fp32 MFMA GEMM core (linear_dma,
    csrc/linear_dma.hip) -- every layer shape of the forward does; a shape that core does not take (K no multiple of 32,
    operands not 16-byte aligned) goes to the vendor library (linear_lt).  A layer whose weight has been PREPARED
    (prepare_linear: the Network's and the Refiner's folded weights are) runs its big launches on the split-bf16 core
    (linear_split, csrc/linear_split.hip: fp32-sized errors at 1.5x the fp32 MFMA's rate)."""
    if not VENDOR_GEMM and x.is_cuda and Wt.is_cuda and x.dim() == 2 and Wt.dim() == 2:
        sw = prepared_linear(Wt, x)
        if sw is not None:
            return linear_split(x, sw, bias, relu, out)
        if linear_dma_ok(x, Wt):
            return linear_dma(x, Wt, bias, relu, out)
    return linear_lt(x, Wt, bias, relu, out)


def linear_lt(x, Wt, bias=None, relu=False, out=None):
    """linear() as a vendor-library GEMM (hipBLASLt, bias / ReLU epilogue) called through the C-ABI (dcl_linear_fwd): only
    algorithms that ask for NO workspace are ever taken (csrc/linear.cpp: two workspace-exchanging stream-K kernels side by
    side hang the GPU); the call fails when the library has none for the shape."""
    N.need_cuda(x, Wt)
    assert x.dim() == 2 and Wt.dim() == 2 and x.shape[1] == Wt.shape[0] and x.dtype == Wt.dtype == torch.float32
    M, K = x.shape
    n = Wt.shape[1]
    if out is None:
        out = torch.empty((M, n), dtype=torch.float32, device=x.device)
    assert out.shape == (M, n) and out.dtype == torch.float32 and out.is_cuda
    for t in (x, Wt, out):
        assert t.stride(1) == 1 or t.shape[1] == 1, "rows must be dense"
    if bias is not None:
        assert bias.is_cuda and bias.dtype == torch.float32 and bias.numel() == n and bias.is_contiguous()
    pitch = lambda t: int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))   # noqa: E731
    N.check(N.lib().dcl_linear_fwd(N.ptr(x), C.c_int64(pitch(x)), N.ptr(Wt), C.c_int64(pitch(Wt)), N.ptr(bias), N.ptr(out),
                                   C.c_int64(pitch(out)), int(M), int(n), int(K), int(bool(relu)), None,
                                   C.c_int64(0), N.stream()), "linear_fwd")
    return out


def linear_dma_ok(x, Wt):
    """can linear_dma take this layer?  (K in whole 32-chunks, 16-byte aligned operands and pitches: csrc/linear_dma.hip)"""
    K, n = Wt.shape
    pw = int(Wt.stride(0)) if K > 1 else n
    px = int(x.stride(0)) if x.shape[0] > 1 else K
    return (K >= 32 and K % 32 == 0 and px % 4 == 0 and pw % 4 == 0 and x.data_ptr() % 16 == 0 and Wt.data_ptr() % 16 == 0 and
            pw >= (n + 3) // 4 * 4)


def linear_dma(x, Wt, bias=None, relu=False, out=None):
    """linear() on the library's OWN fp32 MFMA GEMM core (csrc/linear_dma.hip; no vendor library): act(x @ Wt + bias), same
    conventions -- x (M,K), Wt (K,N), out (M,N) row-major with free row pitches."""
    N.need_cuda(x, Wt)
    assert x.dim() == 2 and Wt.dim() == 2 and x.shape[1] == Wt.shape[0] and x.dtype == Wt.dtype == torch.float32
    M, K = x.shape
    n = Wt.shape[1]
    if out is None:
        out = torch.empty((M, n), dtype=torch.float32, device=x.device)
    assert out.shape == (M, n) and out.dtype == torch.float32 and out.is_cuda
    for t in (x, Wt, out):
        assert t.stride(1) == 1 or t.shape[1] == 1, "rows must be dense"
    if bias is not None:
        assert bias.is_cuda and bias.dtype == torch.float32 and bias.numel() == n and bias.is_contiguous()
    pitch = lambda t: int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))   # noqa: E731
    N.check(N.lib().dcl_linear_dma_fwd(N.ptr(x), C.c_int64(pitch(x)), N.ptr(Wt), C.c_int64(pitch(Wt)), N.ptr(bias), N.ptr(out),
                                       C.c_int64(pitch(out)), int(M), int(n), int(K), int(bool(relu)), N.stream()), "linear_dma_fwd")
    return out


GEMM_SPLIT = True              # big launches of prepared layers run on the split-bf16 core (False: the fp32-MFMA core everywhere)
SPLIT_MIN_TILES = 192          # ... from this many 256 x 128 tiles on (fewer cannot fill the chip: the fp32 core's smaller tiles win)
_PREPARED = {}                 # id(Wt) -> (weakref to Wt, SplitWeight)


def prepare_linear(Wt):
    """Prepare a layer's (K, N) weight for the split-bf16 GEMM core: its three bf16 pieces in tile order, kept beside the weight
    for as long as the weight tensor lives.  The weight must not be modified in place afterwards (the models' folded weights are
    rebuilt, never modified).  Layers of at most 64 output columns or with K no multiple of 16 stay on the fp32 core."""
    K, n = Wt.shape
    if not Wt.is_cuda or n <= 64 or K % 16 or K < 16:
        return None
    ent = _PREPARED.get(id(Wt))
    if ent is not None and ent[0]() is Wt:
        return ent[1]
    sw = SplitWeight(Wt)
    sw.Wt = None                                            # (the registry must not keep the weight alive)
    key = id(Wt)
    _PREPARED[key] = (weakref.ref(Wt, lambda _r, key=key: _PREPARED.pop(key, None)), sw)
    return sw


def prepared_linear(Wt, x):
    """the SplitWeight of a prepared layer if this launch should run on the split-bf16 core, else None"""
    if not GEMM_SPLIT:
        return None
    ent = _PREPARED.get(id(Wt))
    if ent is None or ent[0]() is not Wt:
        return None
    sw = ent[1]
    M = x.shape[0]
    if ((M + 255) // 256) * ((sw.n + 127) // 128) < SPLIT_MIN_TILES or not linear_split_ok(x, sw.K):
        return None
    return sw


LINEAR_POOL_TILE = 128         # rows per partial of linear_pool (the GEMM's row tile)


def linear_pool(x, Wt, bias, roww, relu=True, part=None, rows_per_crop=None, w_stride=0):
    """The last fuser layer with the confidence-weighted pooling as its epilogue (csrc/linear_dma.hip, EPI = 1):
    part[t] = sum over rows j of row tile t (128 rows) of w_j * act(x[j] @ Wt + bias) -- (ceil(M/128), N); the (M, N)
    activation is never stored.  w_j = roww[j] by default; with rows_per_crop / w_stride, row j = (crop, point) weighs
    roww[crop * w_stride + point] (a direction's block of conf_softmax's (b, n1 + n2) weights).  With every crop a whole
    number of tiles, a crop's pooled feature is the sum of its tiles' partials (pool_finish2 adds them in tile order)."""
    N.need_cuda(x, Wt, roww)
    sw = prepared_linear(Wt, x)
    if sw is not None:
        return linear_split_pool(x, sw, bias, roww, relu, part, rows_per_crop, w_stride)
    M, K = x.shape
    n = Wt.shape[1]
    if rows_per_crop is None:
        rows_per_crop, w_stride = M, 0
        assert roww.is_contiguous() and roww.numel() == M
    assert roww.dtype == torch.float32 and M % rows_per_crop == 0
    tiles = (M + LINEAR_POOL_TILE - 1) // LINEAR_POOL_TILE
    if part is None:
        part = torch.empty((tiles, n), dtype=torch.float32, device=x.device)
    assert part.shape == (tiles, n) and part.stride(1) == 1
    pitch = lambda t: int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))   # noqa: E731
    N.check(N.lib().dcl_linear_pool_fwd(N.ptr(x), C.c_int64(pitch(x)), N.ptr(Wt), C.c_int64(pitch(Wt)), N.ptr(bias), N.ptr(roww),
                                        int(rows_per_crop), C.c_int64(int(w_stride)), N.ptr(part), C.c_int64(pitch(part)), int(M),
                                        int(n), int(K), int(bool(relu)), N.stream()), "linear_pool_fwd")
    return part


def linear_rowdot(x, Wt, bias, w3, b3, out=None):
    """out[m] = w3 . relu(x[m] @ Wt + bias) + b3: the last two layers of a one-output head (regressor_conf: 128 -> 128 -> 1) as ONE
    GEMM of the own core with the row dot as its epilogue (csrc/linear_dma.hip, EPI = 2) -- the hidden columns are never stored.
    x (M,K), Wt (K,N <= 128), bias (N,), w3 (N,1) (rows may be padded: pad_linear_weight), b3 (1,) -> (M,1)."""
    N.need_cuda(x, Wt, w3, b3)
    sw = prepared_linear(Wt, x)
    if sw is not None:
        return linear_split_rowdot(x, sw, bias, w3, b3, out)
    M, K = x.shape
    n = Wt.shape[1]
    assert n <= 128 and w3.shape == (n, 1) and b3.numel() == 1 and bias.numel() == n
    if out is None:
        out = torch.empty((M, 1), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.numel() == M
    pitch = lambda t: int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))   # noqa: E731
    N.check(N.lib().dcl_linear_rowdot_fwd(N.ptr(x), C.c_int64(pitch(x)), N.ptr(Wt), C.c_int64(pitch(Wt)), N.ptr(bias), N.ptr(w3),
                                          C.c_int64(int(w3.stride(0))), N.ptr(b3), N.ptr(out), int(M), int(n), int(K), N.stream()),
            "linear_rowdot_fwd")
    return out


class SplitWeight:
    """A layer's weight prepared for the split-bf16 GEMM core (csrc/linear_split.hip): its three bf16 pieces in the kernel's
    tile order (`planes`), beside the (K, N) fp32 weight they were made from."""
    __slots__ = ("Wt", "planes", "K", "n")

    def __init__(self, Wt):
        N.need_cuda(Wt)
        assert Wt.dim() == 2 and Wt.dtype == torch.float32 and (Wt.stride(1) == 1 or Wt.shape[1] == 1)
        self.Wt, (self.K, self.n) = Wt, Wt.shape
        lib = N.lib()
        lib.dcl_linear_split_weight_bytes.restype = C.c_int64
        nbytes = int(lib.dcl_linear_split_weight_bytes(int(self.K), int(self.n)))
        assert nbytes > 0, "split-bf16 GEMM core: K must be a multiple of 16"
        self.planes = torch.empty(nbytes, dtype=torch.uint8, device=Wt.device)
        pw = int(Wt.stride(0)) if self.K > 1 else self.n
        N.check(lib.dcl_linear_split_weight(N.ptr(Wt), C.c_int64(pw), int(self.K), int(self.n), N.ptr(self.planes), N.stream()),
                "linear_split_weight")


def linear_split_ok(x, K):
    """can the split-bf16 core take this operand?  (K in whole 16-chunks, 16-byte aligned rows)"""
    px = int(x.stride(0)) if x.shape[0] > 1 else K
    return K >= 16 and K % 16 == 0 and px % 4 == 0 and x.data_ptr() % 16 == 0


def linear_split(x, sw, bias=None, relu=False, out=None):
    """linear() on the split-bf16 GEMM core (csrc/linear_split.hip): act(x @ sw.Wt + bias) with fp32-sized errors at the bf16
    matrix pipe's rate (three bf16 pieces per operand, six piece products per product, fp32 accumulation).  sw: SplitWeight."""
    N.need_cuda(x)
    assert x.dim() == 2 and x.shape[1] == sw.K and x.dtype == torch.float32 and (x.stride(1) == 1 or x.shape[1] == 1)
    M, K, n = x.shape[0], sw.K, sw.n
    if out is None:
        out = torch.empty((M, n), dtype=torch.float32, device=x.device)
    assert out.shape == (M, n) and out.dtype == torch.float32 and out.is_cuda and (out.stride(1) == 1 or n == 1)
    if bias is not None:
        assert bias.is_cuda and bias.dtype == torch.float32 and bias.numel() == n and bias.is_contiguous()
    pitch = lambda t: int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))   # noqa: E731
    N.check(N.lib().dcl_linear_split_fwd(N.ptr(x), C.c_int64(pitch(x)), N.ptr(sw.planes), N.ptr(bias), N.ptr(out), C.c_int64(pitch(out)),
                                         int(M), int(n), int(K), int(bool(relu)), N.stream()), "linear_split_fwd")
    return out


def linear_split_pool(x, sw, bias, roww, relu=True, part=None, rows_per_crop=None, w_stride=0):
    """linear_pool() on the split-bf16 core: same partials layout (one row of `part` per 128 rows of x)"""
    N.need_cuda(x, roww)
    M, K, n = x.shape[0], sw.K, sw.n
    assert x.shape[1] == K
    if rows_per_crop is None:
        rows_per_crop, w_stride = M, 0
        assert roww.is_contiguous() and roww.numel() == M
    assert roww.dtype == torch.float32 and M % rows_per_crop == 0
    tiles = (M + LINEAR_POOL_TILE - 1) // LINEAR_POOL_TILE
    if part is None:
        part = torch.empty((tiles, n), dtype=torch.float32, device=x.device)
    assert part.shape == (tiles, n) and part.stride(1) == 1
    pitch = lambda t: int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))   # noqa: E731
    N.check(N.lib().dcl_linear_split_pool_fwd(N.ptr(x), C.c_int64(pitch(x)), N.ptr(sw.planes), N.ptr(bias), N.ptr(roww), int(rows_per_crop),
                                              C.c_int64(int(w_stride)), N.ptr(part), C.c_int64(pitch(part)), int(M), int(n), int(K),
                                              int(bool(relu)), N.stream()), "linear_split_pool_fwd")
    return part


def linear_split_vpieces(x, sw, bias, vplanes, rows_per_crop, relu=True):
    """act(x @ sw.Wt + bias) written AS the attention's V pieces into `vplanes` (attention_planes' scratch of the call that consumes
    them with V1 = None): csrc/linear_split.hip, EPI = 3 -- the fp32 activation is never stored.  Rows = keys; every crop
    rows_per_crop rows (a multiple of 256)."""
    N.need_cuda(x, vplanes)
    M, K, n = x.shape[0], sw.K, sw.n
    assert x.shape[1] == K and n <= 320 and n % 32 == 0 and rows_per_crop % 256 == 0 and M % rows_per_crop == 0
    pitch = lambda t: int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))   # noqa: E731
    N.check(N.lib().dcl_linear_split_vpieces_fwd(N.ptr(x), C.c_int64(pitch(x)), N.ptr(sw.planes), N.ptr(bias), N.ptr(vplanes),
                                                 int(rows_per_crop), int(M), int(n), int(K), int(bool(relu)), N.stream()),
            "linear_split_vpieces_fwd")
    return vplanes


def linear_split_rowdot(x, sw, bias, w3, b3, out=None):
    """linear_rowdot() on the split-bf16 core"""
    N.need_cuda(x, w3, b3)
    M, K, n = x.shape[0], sw.K, sw.n
    assert x.shape[1] == K and n <= 128 and w3.shape == (n, 1) and b3.numel() == 1 and bias.numel() == n
    if out is None:
        out = torch.empty((M, 1), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.numel() == M
    pitch = lambda t: int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))   # noqa: E731
    N.check(N.lib().dcl_linear_split_rowdot_fwd(N.ptr(x), C.c_int64(pitch(x)), N.ptr(sw.planes), N.ptr(bias), N.ptr(w3),
                                                C.c_int64(int(w3.stride(0))), N.ptr(b3), N.ptr(out), int(M), int(n), int(K), N.stream()),
            "linear_split_rowdot_fwd")
    return out


def conf_softmax(b, logit1, logit2):
    """the softmax half of conf_pool alone: logits (b*n1,), (b*n2,) -> conf (b, n1+n2) = sigmoid, w (b, n1+n2) = softmax(conf)
    per crop, wsum (b, 2) (models/DCL_Net.py:217-222)"""
    N.need_cuda(logit1, logit2)
    assert logit1.is_contiguous() and logit2.is_contiguous() and logit1.numel() % b == 0 and logit2.numel() % b == 0
    n1, n2 = logit1.numel() // b, logit2.numel() // b
    dev = logit1.device
    conf = torch.empty((b, n1 + n2), dtype=torch.float32, device=dev)
    w = torch.empty((b, n1 + n2), dtype=torch.float32, device=dev)
    ws = torch.empty((b, 2), dtype=torch.float32, device=dev)
    N.check(N.lib().dcl_conf_softmax(b, n1, n2, N.ptr(logit1), N.ptr(logit2), N.ptr(conf), N.ptr(w), N.ptr(ws), N.stream()),
            "conf_softmax")
    return conf, w, ws


def pool_finish2(part1, part2, ws, affine):
    """pooled (b, c) = ((s1*P1 + t1*wsum1) + s2*P2) + t2*wsum2 with P = a crop's tile partials added in tile order: part1
    (b*ns1, c), part2 (b*ns2, c) as linear_pool leaves them"""
    b, c = ws.shape[0], part1.shape[1]
    s1, t1, s2, t2 = affine
    assert part1.is_contiguous() and part2.is_contiguous() and part1.shape[0] % b == 0 and part2.shape[0] % b == 0
    out = torch.empty((b, c), dtype=torch.float32, device=part1.device)
    N.check(N.lib().dcl_pool_finish2(b, c, part1.shape[0] // b, part2.shape[0] // b, N.ptr(part1), N.ptr(part2), N.ptr(ws),
                                     N.ptr(s1), N.ptr(t1), N.ptr(s2), N.ptr(t2), N.ptr(out), N.stream()), "pool_finish2")
    return out


class _LinearJob(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int64), ("Wt", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p),
                ("y", C.c_void_p), ("ldy", C.c_int64), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("relu", C.c_int32)]


LINEAR_GROUP_MAX = 8


def linear_group(jobs):
    """Independent per-point linear layers in ONE launch (csrc/linear_group.hip; calls of a handful of crops, where a forward
    costs its number of dependent launches).  jobs: list of (x, Wt, bias, relu, out) with linear()'s conventions -- x (M,K),
    Wt (K,N), out (M,N) row-major with free row pitches, out=None allocates; K % 32 == 0; a Wt whose N is not a multiple of 4
    must be a column block of a buffer with rows padded to 4 floats (pad_linear_weight).  Returns the outputs."""
    assert 1 <= len(jobs) <= LINEAR_GROUP_MAX
    arr = (_LinearJob * len(jobs))()
    outs, keep = [], []
    pitch = lambda t: int(t.stride(0)) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))   # noqa: E731
    for q, (x, Wt, bias, relu, out) in zip(arr, jobs):
        N.need_cuda(x, Wt)
        assert x.dim() == 2 and Wt.dim() == 2 and x.shape[1] == Wt.shape[0] and x.dtype == Wt.dtype == torch.float32
        M, K = x.shape
        n = Wt.shape[1]
        if out is None:
            out = torch.empty((M, n), dtype=torch.float32, device=x.device)
        assert out.shape == (M, n) and out.dtype == torch.float32 and out.is_cuda
        for t in (x, Wt, out):
            assert t.stride(1) == 1 or t.shape[1] == 1, "rows must be dense"
        if bias is not None:
            assert bias.is_cuda and bias.dtype == torch.float32 and bias.numel() == n and bias.is_contiguous()
        q.x, q.ldx, q.Wt, q.ldw, q.bias = x.data_ptr(), pitch(x), Wt.data_ptr(), pitch(Wt), (None if bias is None else bias.data_ptr())
        q.y, q.ldy, q.M, q.N, q.K, q.relu = out.data_ptr(), pitch(out), int(M), int(n), int(K), int(bool(relu))
        outs.append(out)
        keep.append((x, Wt, bias))
    N.check(N.lib().dcl_linear_group_fwd(arr, len(jobs), N.stream()), "linear_group_fwd")
    return outs


def pad_linear_weight(Wt):
    """(K,N) weight as a column block of a (K, N rounded up to 4) zero-padded buffer: what linear_group wants for N % 4 != 0"""
    K, n = Wt.shape
    if n % 4 == 0:
        return Wt.contiguous()
    buf = torch.zeros((K, (n + 3) // 4 * 4), dtype=Wt.dtype, device=Wt.device)
    buf[:, :n] = Wt
    return buf[:, :n]


def mlp128_to1(x, layers, out=None):
    """The confidence regressor (models/DCL_Net.py:115-126: Head_MultiLayerPerceptron [128, 128, 128, 1], ReLU, ReLU, none) on
    point rows in ONE launch (csrc/dense.hip: k_mlp128_to1): x (M, 128) rows of pitch >= 128 (a column block of a wider buffer
    is fine), layers = [(W1t (128,128), b1), (W2t (128,128), b2), (W3t (128,1), b3)] as Network._fold() keeps them.
    Returns (M, 1) logits."""
    (W1t, b1), (W2t, b2), (W3t, b3) = layers
    N.need_cuda(x, W1t, W2t, W3t)
    M = x.shape[0]
    assert x.dim() == 2 and x.shape[1] == 128 and x.dtype == torch.float32 and (x.stride(1) == 1) and x.stride(0) % 4 == 0
    for W in (W1t, W2t):
        assert W.shape == (128, 128) and W.is_contiguous() and W.dtype == torch.float32
    assert W3t.shape == (128, 1) and b1.numel() == 128 and b2.numel() == 128 and b3.numel() == 1
    if out is None:
        out = torch.empty((M, 1), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.numel() == M
    N.check(N.lib().dcl_mlp128_to1(N.ptr(x), C.c_int64(int(x.stride(0)) if M > 1 else 128), int(M), N.ptr(W1t), N.ptr(b1), N.ptr(W2t),
                                   N.ptr(b2), N.ptr(W3t), C.c_int64(int(W3t.stride(0))), N.ptr(b3), N.ptr(out), N.stream()), "mlp128_to1")
    return out


def ortho9d_to_matrix(o9):
    """ortho9d2matrix (models/DCL_Net.py:15-36): (b,9) -> (b,3,3)."""
    N.need_cuda(o9)
    o9 = o9.contiguous()
    b = o9.shape[0]
    R = torch.empty((b, 3, 3), dtype=torch.float32, device=o9.device)
    N.check(N.lib().dcl_ortho9d_to_matrix(b, N.ptr(o9), N.ptr(R), N.stream()), "ortho9d_to_matrix")
    return R


def add_s(cld, R_pred, t_pred, R_gt, t_gt, cls=None, sym_flag=None, mode="adds"):
    """ADD-S per object (tools/test_YCBV_stage1.py:186-189) without the (b,P,P,3) intermediate.
    cld (n_clouds,P,3) f32; cls (b,) int32 picks each object's cloud (None: cloud o for object o) -> (b,) f32.
    mode="add": ADD (corresponding points, tools/test_LM.py:123); sym_flag (b,) given: ADD where it is 0, ADD-S elsewhere
    (the LineMOD rule, tools/test_LM.py:130-135)."""
    N.need_cuda(cld, R_pred, t_pred, R_gt, t_gt, cls, sym_flag)
    cld, R_pred, t_pred = N.f32c(cld), N.f32c(R_pred), N.f32c(t_pred)
    R_gt, t_gt = N.f32c(R_gt), N.f32c(t_gt)
    b, P = R_pred.shape[0], cld.shape[1]
    if cls is not None:
        cls = N.i32c(cls)
    else:
        assert cld.shape[0] == b
    part = torch.empty((b, (P + 255) // 256), dtype=torch.float32, device=cld.device)
    out = torch.empty(b, dtype=torch.float32, device=cld.device)
    if sym_flag is not None:
        sym = N.i32c(sym_flag)
        assert sym.numel() == b
        N.check(N.lib().dcl_add_by_symmetry(b, P, N.ptr(cld), N.ptr(cls), N.ptr(sym), N.ptr(R_pred), N.ptr(t_pred),
                                            N.ptr(R_gt), N.ptr(t_gt), N.ptr(part), N.ptr(out), N.stream()), "add_by_symmetry")
        return out
    fn = {"adds": N.lib().dcl_add_s, "add": N.lib().dcl_add}[mode]
    N.check(fn(b, P, N.ptr(cld), N.ptr(cls), N.ptr(R_pred), N.ptr(t_pred), N.ptr(R_gt), N.ptr(t_gt),
               N.ptr(part), N.ptr(out), N.stream()), "add_s")
    return out


def add(cld, R_pred, t_pred, R_gt, t_gt, cls=None):
    """ADD per object: mean_i |pred_i - gt_i| (tools/test_LM.py:123 `l2_dis`)."""
    return add_s(cld, R_pred, t_pred, R_gt, t_gt, cls, mode="add")


# ------------------------------------------------------------------------------------ crop builder
def legacy_choice_heads(ms, n):
    """[np.random.choice(m, n, replace=False) for m in ms] on the GLOBAL legacy generator, bit for bit and with the same
    advance of its state, ~3x faster (csrc/legacy_rng.cpp: numpy's Fisher-Yates walk as a tight branch-free loop).  Host
    code; every m must be >= n.  -> int64 array (len(ms), n).  (The loaders' sampling draws: YCBV/dataloader_test_YCBV.py:166-169.)"""
    import numpy as np
    k = len(ms)
    out = np.empty((k, int(n)), np.int64)
    if k == 0:
        return out
    st = np.random.get_state()
    assert st[0] == "MT19937"
    key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
    pos = C.c_int32(int(st[2]))
    m = np.ascontiguousarray(np.asarray(ms, np.int32))
    scratch = np.empty(int(m.max()), np.int32)
    N.check(N.lib().dcl_legacy_permutation_heads(key.ctypes.data_as(C.c_void_p), C.byref(pos), m.ctypes.data_as(C.c_void_p), k, int(n),
                                                 out.ctypes.data_as(C.c_void_p), scratch.ctypes.data_as(C.c_void_p)),
            "legacy_permutation_heads")
    np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
    return out


def crop_points(depth, label, rgb, boxes, obj_ids, cam, rgb_mean, half_extent, min_valid=32, always_filter=False, cap=None):
    """Masked back-projection + centring + grid filter of every object instance of one image
    (YCBV/dataloader_test_YCBV.py:124-165).  depth (H,W) u16 viewed as int16 storage, label (H,W) i32, rgb (H,W,C) u8,
    boxes (n,4) i32 [rmin,rmax,cmin,cmax], obj_ids (n) i32 -- all CUDA.  cam = (cx,cy,fx,fy,scale[,post_div]).
    -> xyz (n,cap,3), rgb (n,cap,3), centroid (n,3), counts (n,3) i32 [masked, inside grid, rows]."""
    N.need_cuda(depth, label, rgb, boxes, obj_ids)
    assert depth.dtype in (torch.int16, torch.uint16) and label.dtype == torch.int32 and rgb.dtype == torch.uint8
    assert depth.is_contiguous() and label.is_contiguous() and rgb.is_contiguous()
    assert boxes.dtype == torch.int32 and obj_ids.dtype == torch.int32 and boxes.is_contiguous()
    H, W = depth.shape
    n = boxes.shape[0]
    dev = depth.device
    if cap is None:                                            # (a caller that made the boxes on the host passes the largest area)
        bx = boxes.cpu()
        cap = int(max(1, ((bx[:, 1] - bx[:, 0]).clamp(min=0) * (bx[:, 3] - bx[:, 2]).clamp(min=0)).max().item())) if n else 1
    raw_xyz = torch.empty((n, cap, 3), dtype=torch.float32, device=dev)
    raw_rgb = torch.empty((n, cap, 3), dtype=torch.float32, device=dev)
    xyz = torch.empty((n, cap, 3), dtype=torch.float32, device=dev)
    col = torch.empty((n, cap, 3), dtype=torch.float32, device=dev)
    centroid = torch.empty((n, 3), dtype=torch.float32, device=dev)
    counts = torch.zeros((n, 3), dtype=torch.int32, device=dev)
    ws_ints = C.c_int64(0)
    N.check(N.lib().dcl_crop_points_ws_ints(n, int(cap), C.byref(ws_ints)), "crop_points_ws_ints")
    ws = torch.empty(max(int(ws_ints.value), 1), dtype=torch.int32, device=dev)
    cam_a = (C.c_float * 6)(*([float(v) for v in cam] + [1.0])[:6])
    mean_a = (C.c_double * 3)(*[float(v) for v in rgb_mean])
    he_a = (C.c_float * 3)(*[float(v) for v in half_extent])
    N.check(N.lib().dcl_crop_points(N.ptr(depth), N.ptr(label), N.ptr(rgb), H, W, rgb.shape[2], n, N.ptr(boxes),
                                    N.ptr(obj_ids), cam_a, mean_a, he_a, int(min_valid), int(bool(always_filter)), cap,
                                    N.ptr(raw_xyz),
                                    N.ptr(raw_rgb), N.ptr(xyz), N.ptr(col), N.ptr(centroid), N.ptr(counts), N.ptr(ws), N.stream()),
            "crop_points")
    return xyz, col, centroid, counts


def crop_sample(xyz, rgb, sample_idx, counts, half_extent0, unit, voxel_limit, npoint=None, min_valid=32):
    """Sampled points -> feats (n*npoint,7) [1,rgb,xyz] and voxelize_idx input rows (n*npoint,4) i64 [instance,ix,iy,iz]
    (dataloader_test_YCBV.py:166-177,186-190).  sample_idx (n,npoint) i64 CUDA or None (identity, template clouds)."""
    N.need_cuda(xyz, rgb, sample_idx, counts)
    assert xyz.is_contiguous() and rgb.is_contiguous() and xyz.dtype == torch.float32 and rgb.dtype == torch.float32
    n, cap = xyz.shape[0], xyz.shape[1]
    if sample_idx is not None:
        assert sample_idx.dtype == torch.int64 and sample_idx.is_contiguous() and sample_idx.shape[0] == n
        npoint = sample_idx.shape[1]
    else:
        npoint = cap if npoint is None else npoint
    dev = xyz.device
    feats = torch.empty((n * npoint, 7), dtype=torch.float32, device=dev)
    coords = torch.empty((n * npoint, 4), dtype=torch.int64, device=dev)
    unit_a = (C.c_float * 3)(*[float(v) for v in unit])
    N.check(N.lib().dcl_crop_sample(n, npoint, cap, N.ptr(xyz), N.ptr(rgb), N.ptr(sample_idx), N.ptr(counts),
                                    int(min_valid), _c_float(half_extent0), unit_a, int(voxel_limit), N.ptr(feats),
                                    N.ptr(coords), N.stream()), "crop_sample")
    return feats, coords


# ------------------------------------------------------------------------------------ training-side kernels
def rulebook_transpose(nbr, n_out, cap_in):
    """inv[k][i] = o for nbr[k][o] = i (csrc/backward.hip); nbr (kvol, cap_out) i32 -> (kvol, cap_in) i32."""
    N.need_cuda(nbr)
    kvol, cap_out = nbr.shape
    inv = torch.empty((kvol, max(cap_in, 1)), dtype=torch.int32, device=nbr.device)
    N.check(N.lib().dcl_rulebook_transpose(N.ptr(nbr), cap_out, N.vp(0), int(n_out), kvol, N.ptr(inv), max(cap_in, 1),
                                           N.stream()), "rulebook_transpose")
    return inv


def sparse_conv_backward(feat, W, dout, nbr, n_out, subm, need_dx=True):
    """indice_conv_backward_fp32 (spconv_ops.h:351-438): -> (d_feat (V_in,Cin) or None, dW (kvol,Cin,Cout))."""
    N.need_cuda(feat, W, dout, nbr)
    kvol, cap = nbr.shape
    cin, cout = W.shape[-2], W.shape[-1]
    n_in = feat.shape[0]
    dev = feat.device
    dout = dout.contiguous()
    dW = torch.zeros((kvol, cin, cout), dtype=torch.float32, device=dev)
    dx = None
    if n_out > 0:
        splits = C.c_int32(0)
        N.check(N.lib().dcl_sparse_conv_wgrad_splits(int(n_out), C.byref(splits)), "wgrad_splits")
        partial = torch.empty(splits.value * kvol * cin * cout, dtype=torch.float32, device=dev)
        N.check(N.lib().dcl_sparse_conv_wgrad(N.ptr(feat), N.ptr(nbr), cap, int(n_out), N.ptr(dout), cin, cout, kvol,
                                              N.ptr(partial), N.ptr(dW), N.stream()), "sparse_conv_wgrad")
    if need_dx:
        if n_in == 0 or n_out == 0:
            dx = torch.zeros((n_in, cin), dtype=torch.float32, device=dev)
        else:
            inv = rulebook_transpose(nbr, n_out, n_in)
            Wt = W.reshape(kvol, cin, cout).transpose(1, 2).contiguous()          # per-offset W^T: (kvol, Cout, Cin)
            dx = sparse_conv(dout, inv, n_in, Wt, subm)
    return dx, dW


def sparse_avgpool_backward(dout, nbr, n_out, n_in, rf):
    """indice_avgpool_backward_fp32 (avgpool.cu:178-206) -> d_feat (V_in, C)."""
    N.need_cuda(dout, nbr, rf)
    c = dout.shape[1]
    din = torch.zeros((n_in, c), dtype=torch.float32, device=dout.device)
    if n_in == 0 or n_out == 0:
        return din
    inv = rulebook_transpose(nbr, n_out, n_in)
    N.check(N.lib().dcl_sparse_avgpool_bwd(N.ptr(dout.contiguous()), N.ptr(inv), inv.shape[1], int(n_in), N.ptr(rf), c,
                                           nbr.shape[0], N.ptr(din), N.stream()), "sparse_avgpool_bwd")
    return din


def three_interpolate_grad_sp(grad_out, idx, weight, m):
    """three_interpolate_grad_wrapper of libs/pointnet_sp -> grad_points (m, C)."""
    N.need_cuda(grad_out, idx, weight)
    n, c = grad_out.shape
    gp = torch.zeros((m, c), dtype=torch.float32, device=grad_out.device)
    N.check(N.lib().dcl_three_interpolate_grad_sp(c, n, m, N.ptr(grad_out.contiguous()), N.ptr(idx), N.ptr(weight), N.ptr(gp),
                                                  N.stream()), "three_interpolate_grad_sp")
    return gp


def voxelize_bp(d_out, map_rule, n_points, mode=4):
    """PG_OP.voxelize_bp (pointgroup_ops.py:65-73) -> d_feats (N, C)."""
    N.need_cuda(d_out, map_rule)
    M, c = d_out.shape
    d_feats = torch.zeros((n_points, c), dtype=torch.float32, device=d_out.device)
    N.check(N.lib().dcl_voxelize_bp(N.ptr(d_out.contiguous()), N.ptr(map_rule), N.ptr(d_feats), M, map_rule.shape[1] - 1, c,
                                    int(mode == 4), N.stream()), "voxelize_bp")
    return d_feats

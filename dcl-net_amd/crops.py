"""Device-side crop builder: what `YCBDataset.__getitem__` does per image (YCBV/dataloader_test_YCBV.py:99-258), with
the per-pixel work on the GPU (csrc/crops.hip) and the result already resident in HBM for `Network.forward`.

    builder = CropBuilder(cfg, cad_points_mm, cad_colors)            # cfg: input_size, tmp_size, unit_voxel_extent, ...
    data = builder.build(img_u8, depth_u16, label, rois, gt_obj, poses)     # the loader's dict, CUDA tensors

The sampling draws stay `np.random.choice` on the host, made in the reference's order with the reference's arguments, so
a seeded run consumes the global numpy RNG stream exactly like the original loader and produces the same crops.  That
needs the per-instance point counts on the host: the one read-back of the builder (plus the {V, maxActive} read-back of
each voxelize_idx).  No CPU fallback: the arrays are uploaded and everything else happens on the device.
"""
import time

import numpy as np
import torch

from . import ops

RGB_MEAN = (0.485, 0.456, 0.406)                                     # dataloader_test_YCBV.py:57,145
YCBV_CAMERA = (312.9869, 241.3109, 1066.778, 1067.487, 10000.0, 1.0)  # cx, cy, fx, fy, depth scale (:77-81), post-division
LM_CAMERA = (325.26110, 242.04899, 572.41140, 573.57043, 1.0, 1000.0)  # LM/dataloader_test_LM.py:104-107,153-160
MIN_VALID = 32                                                        # :163
LM_MIN_VALID = 128                                                    # LM/dataloader_test_LM.py:197


def snap_box(rois, row, img_h=480, img_w=640):
    """`get_bbox` (dataloader_test_YCBV.py:266-303): the detection's box grown to sides that are multiples of 40 px
    (0 -> 40), kept centred and pushed back inside the image.  Returns (rmin, rmax, cmin, cmax)."""
    r0, r1 = max(int(rois[row][3]) + 1, 0), min(int(rois[row][5]) - 1, img_h)
    c0, c1 = max(int(rois[row][2]) + 1, 0), min(int(rois[row][4]) - 1, img_w)

    def grow(v):                       # open intervals between the borders -1, 40, 80, ... 680 round up to the next border
        return (v // 40 + 1) * 40 if -1 < v < 680 and (v % 40 != 0 or v == 0) else v
    hr, hc = int(grow(r1 - r0) / 2), int(grow(c1 - c0) / 2)
    mr, mc = int((r0 + r1) / 2), int((c0 + c1) / 2)
    r0, r1, c0, c1 = mr - hr, mr + hr, mc - hc, mc + hc
    if r0 < 0:
        r0, r1 = 0, r1 - r0
    if c0 < 0:
        c0, c1 = 0, c1 - c0
    if r1 > img_h:
        r0, r1 = r0 - (r1 - img_h), img_h
    if c1 > img_w:
        c0, c1 = c0 - (c1 - img_w), img_w
    return r0, r1, c0, c1


def lm_box(obj_bb, img_h=480, img_w=640):
    """`get_bbox` of the LineMOD loader (LM/dataloader_test_LM.py:287-333): [x, y, w, h] -> (rmin, rmax, cmin, cmax) with
    sides grown to multiples of 40 px like `snap_box`."""
    r0, r1, c0, c1 = obj_bb[1], obj_bb[1] + obj_bb[3], obj_bb[0], obj_bb[0] + obj_bb[2]
    r0, c0 = max(r0, 0), max(c0, 0)
    r1 = img_h - 1 if r1 >= img_h else r1
    c1 = img_w - 1 if c1 >= img_w else c1

    def grow(v):
        return (v // 40 + 1) * 40 if -1 < v < 680 and (v % 40 != 0 or v == 0) else v
    hr, hc = int(grow(r1 - r0) / 2), int(grow(c1 - c0) / 2)
    mr, mc = int((r0 + r1) / 2), int((c0 + c1) / 2)
    r0, r1, c0, c1 = mr - hr, mr + hr, mc - hc, mc + hc
    if r0 < 0:
        r0, r1 = 0, r1 - r0
    if c0 < 0:
        c0, c1 = 0, c1 - c0
    if r1 > img_h:
        r0, r1 = r0 - (r1 - img_h), img_h
    if c1 > img_w:
        c0, c1 = c0 - (c1 - img_w), img_w
    return r0, r1, c0, c1


def _upload(a, dev):
    """small host array -> device WITHOUT stalling the host: through pinned memory (torch's caching host allocator hands the
    block out again only after the copy has run) and a non-blocking copy.  A pageable .to(dev) is stream-ordered AND blocks
    the host: in the middle of build() it waited for the crop kernel, at its start for the previous frame's forward."""
    t = torch.from_numpy(np.ascontiguousarray(a)) if not torch.is_tensor(a) else a
    return t.pin_memory().to(dev, non_blocking=True)


MARKS = None          # tools/builder_trace.py sets a list: (label, perf_counter) host marks of build()


def _mark(label):
    if MARKS is not None:
        MARKS.append((label, time.perf_counter()))


class CropBuilder(object):
    def __init__(self, cfg, cad_points_mm, cad_colors, camera=YCBV_CAMERA, device="cuda", capacity=False, v2p_pitch=33):
        """cfg: mapping with input_size, tmp_size, unit_voxel_extent, voxel_num_limit, voxelization_mode (the `test`
        block of configs/config_YCBV_bs32.yaml).  cad_points_mm / cad_colors: {class id: (tmp_size,3) float64}, the
        loader's list_pc_CAD / list_rgb_CAD (millimetres; colours already mean-subtracted, :57-58).
        capacity=True: the observed side's voxelisation stays in CAPACITY form -- occupied_voxels (b*n, 4) and v2p_maps
        (b*n, v2p_pitch) with the live row count on the device (data["inp"]["v0_dev"]) -- so build() makes ONE host
        read-back per frame (the per-instance point counts the loader's random draws need) instead of two; Network.forward
        takes that form on its graph path (exact_form(data) converts, with the read-back).  v2p_pitch: 1 + the most points of
        a crop one voxel may hold.  Exact mode repeats an image whose crops exceed it with the general op; in capacity mode
        the device-side flag data["inp"]["vi_info"][2] says so (the caller reads it with its results; exact_form raises)."""
        self.capacity, self.v2p_pitch = bool(capacity), int(v2p_pitch)
        self.n_inp, self.n_tmp = int(cfg["input_size"]), int(cfg["tmp_size"])
        self.unit = np.array(cfg["unit_voxel_extent"]).astype(float)
        self.limit = np.array(cfg["voxel_num_limit"]).astype(float)
        self.extent = self.limit * self.unit
        self.mode = int(cfg["voxelization_mode"])
        self.camera = tuple(camera)
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("dcl-net_amd.CropBuilder runs on the GPU only")
        # template side (:179-183): constant per class -> feats rows and voxel coordinates once, on the device
        self.cls_ids = sorted(cad_points_mm.keys())
        self.cls_row = {c: i for i, c in enumerate(self.cls_ids)}
        pts = torch.stack([torch.FloatTensor(np.asarray(cad_points_mm[c]) / 1000.0) for c in self.cls_ids]).to(self.dev)
        col = torch.stack([torch.FloatTensor(np.asarray(cad_colors[c])) for c in self.cls_ids]).to(self.dev)
        assert pts.shape[1] == self.n_tmp
        feats, coords = ops.crop_sample(pts.contiguous(), col.contiguous(), None, None, self.extent[0] * 0.5, self.unit,
                                        int(self.limit[0]))
        self.tmp_feats = feats.view(len(self.cls_ids), self.n_tmp, 7)
        self.tmp_vox = coords.view(len(self.cls_ids), self.n_tmp, 4)[:, :, 1:].contiguous()
        self.draw_seconds = 0.0                  # host time spent in the loader's np.random.choice draws (accumulated)
        # The template side of a batch is a function of the crops' CLASSES alone (:179-183,223): voxelise every class once
        # (device voxelize_idx, crop id 0) and keep the three maps on the host; a frame's maps are those tables put side by
        # side with the crop ids / row offsets applied (_template_side) -- no kernel and no read-back per frame.
        S = int(self.limit[0])
        self._tmp_tab, self._tmp_side_cache = [], {}
        zero = torch.zeros((self.n_tmp, 1), dtype=torch.int64, device=self.dev)
        for r in range(len(self.cls_ids)):
            occ, p2v, v2p = ops.voxelize_idx_gpu(torch.cat([zero, self.tmp_vox[r]], 1).contiguous(), 1, S, self.mode)
            v2p = v2p.cpu().numpy()
            ids = (np.arange(v2p.shape[1])[None, :] >= 1) & (np.arange(v2p.shape[1])[None, :] <= v2p[:, :1])
            self._tmp_tab.append((occ.cpu().numpy(), p2v.cpu().numpy(), v2p, ids))

    def _template_side(self, rows):
        """occupied_voxels / p2v_maps / v2p_maps of the template clouds of the classes `rows` (one per crop), exactly what
        voxelize_idx returns for the batch's (b * n_tmp, 4) coordinate rows (first-encounter voxel ids crop by crop, point
        ids offset by the crop's first point, maxActive = the batch's maximum).  Assembled on the host from the per-class
        tables and cached per class tuple (the objects of a video sequence repeat frame after frame)."""
        key = tuple(int(r) for r in rows)
        hit = self._tmp_side_cache.get(key)
        if hit is not None:
            # the entry was made on whatever stream built it first (pageable uploads + a cat kernel); a hit from another
            # stream or thread orders itself behind that work.  The tensors are shared by every frame with this class tuple:
            # READ-ONLY for consumers.
            torch.cuda.current_stream(self.dev).wait_event(hit[1])
            return hit[0]
        tabs = [self._tmp_tab[r] for r in key]
        ma = max(t[2].shape[1] for t in tabs) - 1
        occ, p2v, v2p, voff = [], [], [], 0
        for j, (o, p, v, ids) in enumerate(tabs):
            oj = o.copy()
            oj[:, 0] = j
            occ.append(oj)
            p2v.append(p + voff)
            vj = np.zeros((v.shape[0], ma + 1), np.int32)
            vj[:, :v.shape[1]] = v + (j * self.n_tmp) * ids
            v2p.append(vj)
            voff += o.shape[0]
        out = tuple(torch.from_numpy(np.ascontiguousarray(np.concatenate(a, 0))).to(self.dev) for a in (occ, p2v, v2p))
        b = len(key)
        ids = torch.arange(b, device=self.dev).view(b, 1, 1).expand(b, self.n_tmp, 1)
        rows_t = torch.tensor(key, device=self.dev)
        out = out + (torch.cat([ids, self.tmp_vox[rows_t]], 2).reshape(b * self.n_tmp, 4).contiguous(),)   # voxelize_idx's input rows
        if len(self._tmp_side_cache) >= 64:
            self._tmp_side_cache.pop(next(iter(self._tmp_side_cache)))
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.dev))
        self._tmp_side_cache[key] = (out, ready)
        return out

    @staticmethod
    def resident(img, depth, label, device="cuda"):
        """upload one frame's arrays in the layout build() wants (u8 colour, depth as 16-bit storage, i32 labels): a caller
        that decodes frames ahead of the network keeps them in HBM and passes these tensors instead of numpy arrays"""
        dev = torch.device(device)
        return (torch.from_numpy(np.ascontiguousarray(img)).to(dev),
                torch.from_numpy(np.ascontiguousarray(depth).astype(np.uint16).view(np.int16)).to(dev),
                torch.from_numpy(np.ascontiguousarray(label).astype(np.int32)).to(dev))

    def build(self, img, depth, label, rois, gt_obj, poses=None):
        """img (H,W,3|4) u8, depth (H,W) u16, label (H,W) integer, rois (k,>=6), gt_obj (n) class ids, poses (3,4,n) or
        None -- numpy arrays as the loader reads them, or the CUDA tensors of CropBuilder.resident().  Returns the loader's
        dict with CUDA tensors (instances without a detection or with an empty mask are dropped and flagged 0 in
        `all_flags`, :116,134).  Host synchronisations per frame: the per-instance point counts (the sampling draws are the
        loader's own np.random.choice calls) and {V, maxActive} of the observed side's voxelisation; nothing else."""
        _mark("start")
        H, W = depth.shape
        gt_obj = np.asarray(gt_obj).astype(np.int32)
        rois = np.asarray(rois)
        cand, boxes = [], []
        for i, cls in enumerate(gt_obj):
            hit = np.where(rois[:, 1] == cls)[0]
            if hit.size:
                r0, r1, c0, c1 = snap_box(rois, hit[0], H, W)
                cand.append(i)
                boxes.append((max(r0, 0), min(r1, H), max(c0, 0), min(c1, W)))      # numpy slice clipping (:132)
        dev = self.dev
        flags = np.zeros(len(gt_obj), np.int8)
        if not cand:
            raise ValueError("no object instance of this image has a detection")
        if torch.is_tensor(depth):
            i_t, d_t, l_t = img, depth, label                                       # resident frame (CropBuilder.resident)
        else:
            i_t, d_t, l_t = self.resident(img, depth, label, dev)
        bo = np.concatenate([np.asarray(boxes, np.int32), gt_obj[cand][:, None]], 1)   # boxes + class ids: one upload
        bo_t = _upload(bo, dev)
        b_t, o_t = bo_t[:, :4].contiguous(), bo_t[:, 4].contiguous()
        cap = max(1, max(max(r1 - r0, 0) * max(c1 - c0, 0) for r0, r1, c0, c1 in boxes))
        xyz, col, centroid, counts = ops.crop_points(d_t, l_t, i_t, b_t, o_t, self.camera, RGB_MEAN, self.extent * 0.5,
                                                     MIN_VALID, cap=cap)
        _mark("crop_points issued")
        # What does not depend on the instances' point counts is issued NOW, while the GPU runs k_crop_points (~0.17 ms) and
        # before the host waits for the counts: the template side (class rows, feats, voxel tables), the gt labels (the
        # centroids are read on the device, in stream order) and the dict's host tensors -- for ALL candidates; the rare image
        # that then drops an instance (an empty mask) redoes this part for the kept ones.
        def count_free_part(keep):
            rows = [self.cls_row[int(gt_obj[cand[k]])] for k in keep]
            cls_rows = _upload(np.asarray(rows, np.int64), dev)
            part = {"rows": rows, "feats_tmp": self.tmp_feats[cls_rows].reshape(len(keep) * self.n_tmp, 7),
                    "tmp_side": self._template_side(rows), "labels": {}}
            cen = centroid if len(keep) == len(cand) else centroid[_upload(np.asarray(keep, np.int64), dev)]
            if poses is not None:
                # rot_gt = poses[:, 0:3], trans_gt = poses[:, 3] - centroid (:226-240: float64 difference, rounded to float32
                # once) -- formed on the device from one small upload, so the centroids never come back to the host
                P = _upload(np.asarray(poses, np.float64)[:, :, [cand[k] for k in keep]], dev)
                part["labels"] = {"rot_gt": P[:, 0:3, :].permute(2, 0, 1).float().contiguous(),
                                  "trans_gt": (P[:, 3, :].t() - cen.double()).float().contiguous()}
            part["centroid"] = cen
            part["host"] = {"batch_offsets": (torch.arange(len(keep) + 1) * 1024).int(), "voxel_num_limit": torch.tensor(self.limit),
                            "obj_idx": torch.IntTensor(gt_obj - 1), "flags": torch.IntTensor([-1])}
            return part
        part = count_free_part(list(range(len(cand))))
        _mark("count-free part issued")
        cnt = counts.cpu().numpy()                                                  # the builder's host read-back
        _mark("counts read back")
        keep = [k for k in range(len(cand)) if cnt[k, 0] > 0]
        if not keep:
            raise ValueError("every object mask of this image is empty")
        picks = []
        t_draw = time.perf_counter()
        # :166-169, the loader's RNG calls, in the loader's order on the global legacy generator: choice(m, n, replace=False) --
        # numpy shuffles all m masked points for it, 14.5 ns each -- runs of such objects go through ops.legacy_choice_heads (the
        # same walk on the same generator state, bit for bit, ~3x faster: 0.44 -> 0.14 ms per 6-object frame); the rare
        # choice(m, n) WITH replacement (m <= n points) stays numpy's.  A seeded run consumes the stream like the original.
        run = []
        def flush():
            if run:
                picks.extend(ops.legacy_choice_heads([int(cnt[k, 2]) for k in run], self.n_inp))
                del run[:]
        for k in keep:
            m = int(cnt[k, 2])
            if m > self.n_inp:
                run.append(k)
            else:
                flush()
                picks.append(np.random.choice(m, self.n_inp))
            flags[cand[k]] = 1
        flush()
        self.draw_seconds += time.perf_counter() - t_draw
        _mark("draws done")
        if len(keep) != len(cand):
            kt = _upload(np.asarray(keep, np.int64), dev)
            xyz, col, counts = xyz[kt].contiguous(), col[kt].contiguous(), counts[kt].contiguous()
            part = count_free_part(keep)
        pick_t = _upload(np.stack(picks).astype(np.int64), dev)
        feats_inp, coords_inp = ops.crop_sample(xyz, col, pick_t, counts, self.extent[0] * 0.5, self.unit,
                                                int(self.limit[0]), min_valid=MIN_VALID)
        b = len(keep)
        rows, feats_tmp = part["rows"], part["feats_tmp"]
        data = dict(part["host"], all_flags=torch.IntTensor(flags), all_centroids=part["centroid"], labels=part["labels"],
                    counts=cnt[keep])
        _mark("sample + labels issued")
        S = int(self.limit[0])
        if (b <= ops.VI_CROPS_MAX_BATCH and self.n_inp <= ops.VI_CROPS_MAX_POINTS and S == ops.VI_CROPS_S and
                self.mode in (3, 4)):
            # the image's crops are voxelised in ONE launch (one workgroup per crop) into capacity-shaped tensors
            occ, p2v, v2p, info = ops.voxelize_idx_crops(coords_inp, b, self.n_inp, S, self.mode, pitch=self.v2p_pitch)
            if self.capacity:
                # capacity form: nothing comes back to the host -- occupied_voxels / v2p_maps keep their b*n rows and `pitch`
                # columns, the live row count stays on the device (v0_dev; Network.forward's graph path takes it as it is)
                data["inp"] = {"feats": feats_inp, "coords": coords_inp, "occupied_voxels": occ, "p2v_maps": p2v, "v2p_maps": v2p,
                               "v0_dev": info[0:1], "vi_info": info}
            else:
                V, ma, err = info.cpu().tolist()                                    # the builder's second host read-back
                if err:      # a voxel with more points than the pitch holds (a tiny object sampled with replacement): general op
                    occ, p2v, v2p = ops.voxelize_idx_gpu(coords_inp, b, S, self.mode)
                    data["inp"] = {"feats": feats_inp, "coords": coords_inp, "occupied_voxels": occ, "p2v_maps": p2v, "v2p_maps": v2p}
                else:
                    data["inp"] = {"feats": feats_inp, "coords": coords_inp, "occupied_voxels": occ[:V], "p2v_maps": p2v,
                                   "v2p_maps": v2p[:V, :max(ma, 1) + 1].contiguous()}
        else:
            occ, p2v, v2p = ops.voxelize_idx_gpu(coords_inp, b, S, self.mode)
            data["inp"] = {"feats": feats_inp, "coords": coords_inp, "occupied_voxels": occ, "p2v_maps": p2v, "v2p_maps": v2p}
        _mark("voxelisation issued")
        occ, p2v, v2p, coords_tmp = part["tmp_side"]                                # tables: no kernel, no read-back
        data["tmp"] = {"feats": feats_tmp, "coords": coords_tmp, "occupied_voxels": occ, "p2v_maps": p2v, "v2p_maps": v2p}
        # a Network(async_inputs=True) lets its side streams wait for exactly this point instead of the whole stream
        data["ready_event"] = torch.cuda.Event()
        data["ready_event"].record(torch.cuda.current_stream(dev))
        _mark("end")
        return data

    def build_lm(self, img, depth, mask_label, obj_bb, obj, eval_mode=False):
        """One LineMOD sample (`PoseDataset.__getitem__`, LM/dataloader_test_LM.py:116-214, test / eval modes): img (H,W,3+) u8,
        depth (H,W) u16 in millimetres, mask_label (H,W) bool object mask (the loader's `mask_label`), obj_bb [x,y,w,h],
        obj = class id.  The builder must have been made with camera=LM_CAMERA.  Returns (feat_inp (N,7), voxel_inp (N,3) i64,
        feat_tmp (M,7), voxel_tmp (M,3) i64, centroid (3,)) as CUDA tensors, or None where the loader returns its all-zero
        dummy sample (empty mask; in test mode also when at most 128 points fall inside the voxel grid)."""
        H, W = depth.shape
        r0, r1, c0, c1 = lm_box(obj_bb, H, W)
        dev = self.dev
        d_t = torch.from_numpy(np.ascontiguousarray(depth).astype(np.uint16).view(np.int16)).to(dev)
        l_t = torch.from_numpy(np.ascontiguousarray(mask_label).astype(np.int32)).to(dev)
        i_t = torch.from_numpy(np.ascontiguousarray(img)).to(dev)
        b_t = torch.tensor([[max(r0, 0), min(r1, H), max(c0, 0), min(c1, W)]], dtype=torch.int32, device=dev)
        o_t = torch.ones(1, dtype=torch.int32, device=dev)
        xyz, col, centroid, counts = ops.crop_points(d_t, l_t, i_t, b_t, o_t, self.camera, RGB_MEAN, self.extent * 0.5,
                                                     LM_MIN_VALID, always_filter=eval_mode)
        n_mask, n_valid, m = counts.cpu().numpy()[0]
        if n_mask == 0 or not (n_valid > LM_MIN_VALID or eval_mode):
            return None
        pick = ops.legacy_choice_heads([int(m)], self.n_inp)[0] if m > self.n_inp else np.random.choice(m, self.n_inp)
        pick_t = torch.from_numpy(pick.astype(np.int64)).view(1, -1).to(dev)
        feats, coords = ops.crop_sample(xyz, col, pick_t, None, self.extent[0] * 0.5, self.unit, int(self.limit[0]))
        row = self.cls_row[int(obj)]
        return feats, coords[:, 1:].contiguous(), self.tmp_feats[row], self.tmp_vox[row], centroid[0]


class CropPrefetcher(object):
    """The role of the reference's loader workers (torch.utils.data.DataLoader(num_workers=...), tools/test_YCBV_stage1.py:133-
    137, tools/test_LM.py:88-92: frames are prepared ahead of the network): ONE host thread builds the crops of up to `depth`
    frames ahead on its own stream while the caller's thread runs the network.  `frames` yields the argument tuples of
    CropBuilder.build (a trailing dict = keyword arguments).  The frames are built strictly in order by that one thread, so
    a seeded run consumes the global np.random stream exactly like a serial loop; the caller must not draw from np.random
    while iterating.  Iterating yields build()'s dicts: the caller's current stream already waits for the builder's
    ready_event, and every tensor is marked as used on that stream (allocator hand-over between streams).  priority: of the
    builder's stream (0 = default; a high-priority builder stream (-1) was measured slower on 6-object frames: its kernels then
    cut into the network's); stream: an existing stream to build on (a long-lived process that has made many streams may
    want to reuse one it knows not to share a hardware queue with the network's)."""
    _END = object()

    def __init__(self, builder, frames, depth=2, priority=0, stream=None):
        import queue
        import threading
        self.builder, self.dev = builder, builder.dev
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._priority = int(priority)
        self._stream = stream                         # the builder's stream (None: a new one of the given priority)
        self._thread = threading.Thread(target=self._work, args=(iter(frames),), name="dcl-crop-prefetch", daemon=True)
        self._thread.start()

    def _put(self, item):
        import queue
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def _work(self, frames):
        try:
            with torch.cuda.device(self.dev):
                stream = self._stream if self._stream is not None else torch.cuda.Stream(self.dev, priority=self._priority)
                with torch.cuda.stream(stream):
                    for args in frames:
                        if self._stop.is_set():
                            return
                        args = tuple(args)
                        kw = args[-1] if args and isinstance(args[-1], dict) else {}
                        if kw:
                            args = args[:-1]
                        if not self._put(self.builder.build(*args, **kw)):
                            return
            self._put(self._END)
        except BaseException as e:                      # handed to the consumer, raised from its next()
            self._put(e)

    def __iter__(self):
        return self

    def __next__(self):
        item = self._q.get()
        if item is self._END:
            self._q.put(item)                           # a second next() ends as well
            raise StopIteration
        if isinstance(item, BaseException):
            self._q.put(self._END)
            raise item
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(item["ready_event"])

        def mark(v):
            if torch.is_tensor(v):
                if v.is_cuda:
                    v.record_stream(cur)
            elif isinstance(v, dict):
                for x in v.values():
                    mark(x)
        mark(item)
        return item

    def close(self):
        """stop the builder thread (frames already built are dropped)"""
        import queue
        self._stop.set()
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=5.0)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def exact_form(data):
    """a capacity-form data dict of CropBuilder(capacity=True) -> the loader's exact form (occupied_voxels (V,4), v2p_maps
    (V, 1+maxActive)): ONE host read-back of {V, maxActive, error}.  A dict that is exact already is returned as it is."""
    side = data["inp"]
    if "v0_dev" not in side:
        return data
    V, ma, err = side["vi_info"].cpu().tolist()
    if err:
        raise RuntimeError("CropBuilder: a voxel holds more points than v2p_pitch - 1, or a point lies outside its grid")
    out = dict(data)
    out["inp"] = {k: v for k, v in side.items() if k not in ("v0_dev", "vi_info")}
    out["inp"]["occupied_voxels"] = side["occupied_voxels"][:V]
    out["inp"]["v2p_maps"] = side["v2p_maps"][:V, :max(ma, 1) + 1].contiguous()
    return out

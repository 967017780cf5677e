"""Real-data half of the training mix: anakin/datasets/hodata.py:315-450 (HOdata.__getitem__) and
anakin/artiboost/mixed_dataset.py.

The reference decodes a frame with PIL in a DataLoader worker, assembles the ground truth in numpy and runs the
augmentation chain (flip, GaussianBlur, colour jitter, affine crop) in PIL on the CPU -- 62.5 % of every training batch.
Here the decoded frames of a batch are uploaded once and the whole chain runs in `ab_augment_batch` (the same kernels as
the tail of the synthetic render); the ground-truth geometry stays on the host in the reference's arithmetic.  The datasets
themselves (HO3D / DexYCB) are downloads: a source only has to provide the getters of `HOdataSource`."""
import os

import numpy as np
import torch

from . import _lib as L
from .registry import IMAGE_PLANE_KEY, PlaneTag, Queries, SynthQueries, tag_image_plane
from .synth import get_affine_transform, gt_core_batch


class HOdataSource:
    """The per-sample getters HOdata subclasses implement (hodata.py:204-296), as used by __getitem__."""
    raw_size = (640, 480)       # (W, H) of the frames (ho3d.py:40)
    sides = "right"             # CONST.SIDE

    def __len__(self):
        raise NotImplementedError

    def get_image(self, idx):
        """uint8 [H, W, 3] RGB (Image.open(path).convert("RGB"), ho3d.py:228-231)."""
        raise NotImplementedError

    def get_image_bytes(self, idx):
        """Optional: the bytes of the frame's .jpg file.  When a source provides them, RealBatcher uploads the FILES and decodes them on the
        device (jpeg.JpegDecoder: bit-identical to get_image's Pillow decode); None (the default) or a file the device decoder does not
        cover -> get_image."""
        return None

    def get_annots(self, idx):
        """dict: cam_intr (3,3), joints_3d (21,3), joints_2d (21,2), corners_3d (8,3), corners_2d (8,2), corners_can (8,3),
        obj_transf (4,4), obj_idx int, side str, bbox_center (2,), bbox_scale float (get_center_scale_wrt_bbox)."""
        raise NotImplementedError


def annot_center_scale(pts2d):
    """HOdata.get_annot_center / get_annot_scale (hodata.py:162-186): int-truncated centre, max span."""
    mn, mx = pts2d.min(0), pts2d.max(0)
    return np.asarray([int((mx[0] + mn[0]) / 2), int((mx[1] + mn[1]) / 2)]), max(mx[0] - mn[0], mx[1] - mn[1])


def assemble_real_gt(ann, image_size, raw_size, draws, center_idx=0, bbox_expand=1.2, center_jit=0.1, scale_jit=0.1,
                     sides="right", train_split=True):
    """HOdata.__getitem__ geometry (hodata.py:315-433).  draws: None (no augmentation) or dict(center (2,) in U(-1,1),
    scale ~ N(0, scale_jit/3), rot radians).  Returns the sample dict (+ "affine" 3x3 and "flip")."""
    flip = ann["side"] != sides
    center, scale = np.asarray(ann["bbox_center"]).copy(), float(ann["bbox_scale"]) * bbox_expand
    K = np.asarray(ann["cam_intr"])
    j3, j2 = np.asarray(ann["joints_3d"]), np.asarray(ann["joints_2d"])
    c3, c2 = np.asarray(ann["corners_3d"]), np.asarray(ann["corners_2d"])
    raw_j2, raw_c2 = j2, c2                                        # get_joints_vis / get_corners_vis re-read the raw annotation
    if flip:                                                       # hodata.py:336-343
        center[0] = raw_size[0] - center[0]
        j3, c3 = j3 * np.array([-1, 1, 1]), c3 * np.array([-1, 1, 1])
        j2, c2 = j2.copy(), c2.copy()
        j2[:, 0] = raw_size[0] - j2[:, 0]
        c2[:, 0] = raw_size[0] - c2[:, 0]
    rot = 0.0
    if draws is not None:                                          # hodata.py:346-359
        center = center + (center_jit * scale * np.asarray(draws["center"])).astype(int)
        scale = scale * np.clip(draws["scale"] + 1.0, 1 - scale_jit, 1 + scale_jit)
        rot = draws["rot"]
    rm = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1]]).astype(np.float32)
    aff, post = get_affine_transform(center, scale, [K[0, 2], K[1, 2]], image_size, rot)
    out = {"affine": aff, "flip": bool(flip), Queries.CAM_INTR: post.dot(K).astype(np.float32)}
    j3 = rm.dot(j3.transpose(1, 0)).transpose()
    root = j3[center_idx]
    c3 = rm.dot(c3.transpose(1, 0)).transpose()
    out[Queries.ROOT_JOINT], out[Queries.JOINTS_3D], out[Queries.CORNERS_3D] = root, j3 - root, c3 - root
    hom = lambda p: aff.dot(np.concatenate([p, np.ones((p.shape[0], 1))], 1).T).T[:, :2]   # noqa: E731  transform_coords
    j2a, c2a = hom(j2).astype(np.float32), hom(c2).astype(np.float32)
    out[Queries.JOINTS_2D], out[Queries.CORNERS_2D] = j2a, c2a

    def vis(raw2d, aug2d, n):                                      # hodata.py:296-313,383-396,418-431
        if not train_split:
            return np.ones(n, np.float32)
        v = (raw2d[:, 0] >= 0) & (raw2d[:, 0] < raw_size[0]) & (raw2d[:, 1] >= 0) & (raw2d[:, 1] < raw_size[1])
        if v.sum() < n * 0.4:
            return np.zeros(n, np.float32)
        va = ((aug2d[:, 0] >= 0) & (aug2d[:, 0] < image_size[0]) & (aug2d[:, 1] >= 0) & (aug2d[:, 1] < image_size[1])).astype(np.float32)
        return np.zeros(n, np.float32) if va.sum() < n * 0.4 else va

    out[Queries.JOINTS_VIS], out[Queries.CORNERS_VIS] = vis(raw_j2, j2a, 21), vis(raw_c2, c2a, 8)
    out[Queries.CORNERS_CAN] = np.asarray(ann["corners_can"])
    out[Queries.OBJ_IDX] = int(ann["obj_idx"])
    base = np.asarray(ann["obj_transf"]).astype(np.float32)
    T = np.concatenate([np.concatenate([rm @ base[:3, :3], rm.dot(base[:3, 3:])], 1), np.array([[0.0, 0.0, 0.0, 1.0]])], 0)
    out[Queries.OBJ_TRANSF] = T.astype(np.float32)
    return out


def assemble_real_gt_batch(anns, image_size, raw_size, draws, center_idx=0, bbox_expand=1.2, center_jit=0.1, scale_jit=0.1,
                           sides="right", train_split=True):
    """assemble_real_gt for a list of annotation dicts at once (numpy-vectorised; same arithmetic)."""
    st = lambda k: np.stack([np.asarray(a[k], np.float64) for a in anns])      # noqa: E731
    S = len(anns)
    flip = np.array([a["side"] != sides for a in anns])
    center, scale = st("bbox_center"), np.array([float(a["bbox_scale"]) for a in anns]) * bbox_expand
    j3, j2, c3, c2 = st("joints_3d"), st("joints_2d"), st("corners_3d"), st("corners_2d")
    raw_j2, raw_c2 = j2.copy(), c2.copy()
    center[flip, 0] = raw_size[0] - center[flip, 0]                              # hodata.py:336-343
    j3[flip, :, 0] *= -1
    c3[flip, :, 0] *= -1
    j2[flip, :, 0] = raw_size[0] - j2[flip, :, 0]
    c2[flip, :, 0] = raw_size[0] - c2[flip, :, 0]
    rot = np.zeros(S)
    if draws is not None:                                                        # hodata.py:346-359
        center = center + (center_jit * scale[:, None] * np.asarray(draws["center"])).astype(int)
        scale = scale * np.clip(np.asarray(draws["scale"]) + 1.0, 1 - scale_jit, 1 + scale_jit)
        rot = np.asarray(draws["rot"], np.float64)
    out = gt_core_batch(st("cam_intr"), j3, j2, c3, c2, st("corners_can"), st("obj_transf"), center, scale, rot, image_size, raw_size,
                        center_idx, raw_j2d=raw_j2, raw_c2d=raw_c2, train_split=train_split)
    out["flip"] = flip
    out[Queries.OBJ_IDX] = np.array([int(a["obj_idx"]) for a in anns], np.int64)
    return out


class RealBatcher:
    """Batches of real samples on the device: host GT assembly + one upload of the decoded frames + ab_augment_batch."""
    GT_KEYS = (Queries.CAM_INTR, Queries.ROOT_JOINT, Queries.JOINTS_3D, Queries.JOINTS_2D, Queries.JOINTS_VIS, Queries.CORNERS_3D,
               Queries.CORNERS_2D, Queries.CORNERS_VIS, Queries.CORNERS_CAN, Queries.OBJ_TRANSF)

    def __init__(self, source: HOdataSource, cfg_preset, aug=True, aug_param=None, device="cuda", compute_dtype=torch.bfloat16, seed=1,
                 num_workers=None):
        # compute_dtype "u8n": the padded frames as ONE bf16 plane of the odd integers 2 v - 255 (AB_DT_U8N; see synth.ArtiBoostLoader)
        self.image_plane = "u8n" if (isinstance(compute_dtype, str) and compute_dtype == "u8n") else "f32"
        if self.image_plane == "u8n":
            compute_dtype = torch.bfloat16
        self.src, self.dev, self.dtype, self.aug = source, torch.device(device), compute_dtype, aug
        # host decode threads (the reference's DataLoader num_workers, anakin/opt.py:16 / train_artiboost.py:175-190): zlib inflates of the
        # PNG path and the Pillow decodes of sources without file bytes run on this many threads (None: AB_DECODE_WORKERS or min(32, cores))
        self.num_workers = num_workers
        self.image_size = list(cfg_preset["IMAGE_SIZE"])
        self.center_idx = int(cfg_preset.get("CENTER_IDX", 9))
        self.bbox_expand = float(cfg_preset.get("BBOX_EXPAND_RATIO", 1.2))
        ap = aug_param or {"SCALE_JIT": 0.1, "CENTER_JIT": 0.1, "MAX_ROT": 0.2}
        self.scale_jit, self.center_jit, self.max_rot = ap["SCALE_JIT"], ap["CENTER_JIT"], ap["MAX_ROT"] * np.pi
        self.rng = np.random.default_rng(seed)
        self._ws = None
        self._pin, self._pin_i = None, 0
        self._jpeg = None               # jpeg.JpegDecoder, created with the first batch of file bytes
        self._jpeg_side = None          # a SECOND decoder (its own device blob and workspace) for predecode(side=True): see there
        self._png = self._png_side = None      # png.PngDecoder (host inflate pool + ab_png_unfilter_batch), same roles
        self._jpeg_cache, self._jpeg_tables, self._png_cache = {}, {}, {}
        self._predecoded = {}           # frames of upcoming batches decoded together: predecode()
        self._jobs = {}                 # inflate jobs of a group started ahead of its predecode(): prefetch_files()

    def predecode(self, idx_lists, side=False):
        """Decode the .jpg files of SEVERAL upcoming batches in one ab_jpeg_decode_batch call (its time is set by the longest Huffman chain,
        nearly independent of the number of frames up to a few hundred: DESIGN 12.4); augment() then takes a batch's frames from here.
        Does nothing when the source has no file bytes or a file is not covered (those batches decode as before).
        side=True: the caller runs this on a stream of its own next to augment() calls on the main stream -- the call then uses a SECOND
        JpegDecoder, because augment()'s per-batch decode (the path a group takes when ITS predecode bailed out) shares nothing with it:
        one decoder's device blob and workspace must never be written from two streams at once."""
        if getattr(self.src, "get_image_bytes", None) is None:
            return
        flat = [int(i) for idxs in idx_lists for i in idxs]
        files = [self.src.get_image_bytes(i) for i in flat]
        W, H = self.src.raw_size
        kind, infos = self._parse_files(flat, files)
        if kind is None:
            return
        dec = self._decoder(kind, side)
        frames = torch.empty((len(flat), H, W, 4), dtype=torch.uint8, device=self.dev)
        job = self._jobs.pop(tuple(flat), None)
        from .jpeg import JpegUnsupported
        from .png import PngUnsupported
        try:
            if job is not None and kind == "png":
                dec.complete(job, out=frames)          # the inflates were started by prefetch_files() a group earlier
            else:
                dec.decode(files, out=frames, infos=infos)
        except (JpegUnsupported, PngUnsupported):      # a stream that only fails while decoding (corrupt / short IDAT, bad filter byte):
            return                                     # nothing is kept; each batch of the group decodes on its own (augment() falls back to Pillow)
        o = 0
        for idxs in idx_lists:
            self._predecoded[tuple(int(i) for i in idxs)] = frames[o:o + len(idxs)]
            o += len(idxs)

    def prefetch_files(self, idx_lists, side=False):
        """Start the HOST share of a later predecode(idx_lists, side) now and return at once: for .png sources the zlib inflates of the group
        are handed to the decode pool (they take a few milliseconds per frame: started here, a group ahead, they are off the critical path
        of the step loop).  Nothing to do for .jpg sources (their host share is a header walk) or sources without file bytes."""
        if getattr(self.src, "get_image_bytes", None) is None:
            return
        flat = [int(i) for idxs in idx_lists for i in idxs]
        files = [self.src.get_image_bytes(i) for i in flat]
        kind, infos = self._parse_files(flat, files)
        if kind == "png" and tuple(flat) not in self._jobs:
            self.drop_jobs()                       # a job nobody collected (an epoch cut short): its staging buffer is needed again
            self._jobs[tuple(flat)] = self._decoder("png", side).submit(files, infos)

    def drop_jobs(self):
        """Await and forget inflate jobs that were started ahead and never collected."""
        for j in self._jobs.values():
            for f in j["futs"]:
                f.exception()
        self._jobs.clear()

    def _decoder(self, kind, side=False):
        """The device decoder of file kind "jpeg" / "png"; side=True: the second instance, for calls on another stream (predecode)."""
        name = "_" + kind + ("_side" if side else "")
        dec = getattr(self, name)
        if dec is None:
            if kind == "jpeg":
                from .jpeg import JpegDecoder
                dec = JpegDecoder(self.dev)
            else:
                from .png import PngDecoder
                dec = PngDecoder(self.dev, workers=self.num_workers)
            setattr(self, name, dec)
        return dec

    def _parse_files(self, idxs, files):
        """-> ("jpeg" | "png", infos) when every file of the batch is covered by ONE device decoder and has the dataset's frame size
        (header walks only, cached per frame index), else (None, None): the batch then decodes through get_image."""
        if any(f is None for f in files):
            return None, None
        from .jpeg import JpegUnsupported, parse as jparse
        from .png import SIGNATURE, PngUnsupported, parse as pparse
        W, H = self.src.raw_size
        try:
            if all(bytes(memoryview(f)[:8]) == SIGNATURE for f in files):
                kind, infos = "png", [self._png_info(i, f, pparse) for i, f in zip(idxs, files)]
            else:
                kind, infos = "jpeg", [self._jpeg_info(i, f, jparse) for i, f in zip(idxs, files)]
        except (JpegUnsupported, PngUnsupported):
            return None, None
        if any((it.width, it.height) != (W, H) for it in infos):
            return None, None
        return kind, infos

    def _png_info(self, idx, data, parse):
        hit = self._png_cache.get(idx)
        if hit is not None and hit[0] == len(data):
            return hit[1]
        it = parse(data)
        self._png_cache[idx] = (len(data), it)
        return it

    def _jpeg_info(self, idx, data, parse):
        """parse(data), remembered per frame index (a dataset re-reads the same files every epoch); the quantisation / Huffman tables of
        the entries are shared between frames that carry the same ones."""
        hit = self._jpeg_cache.get(idx)
        if hit is not None and hit[0] == len(data):
            return hit[1]
        it = parse(data)
        it.qt = self._jpeg_tables.setdefault(it.qt.tobytes(), it.qt)
        it.ht = self._jpeg_tables.setdefault(it.ht.tobytes(), it.ht)
        self._jpeg_cache[idx] = (len(data), it)
        return it

    def draw(self, n):
        """One set of augmentation draws per sample (hodata.py:346-359,435-442: ranges hard-coded at hodata.py:107-112)."""
        r = self.rng
        order = np.stack([r.permutation(4) for _ in range(n)]).astype(np.int32)
        fac_of = np.stack([r.uniform(0.9, 1.1, n), r.uniform(0.9, 1.1, n), r.uniform(-0.075, 0.075, n), r.uniform(0.9, 1.1, n)], 1)
        return dict(center=r.uniform(-1, 1, (n, 2)), scale=r.normal(0, self.scale_jit / 3.0, n), rot=r.uniform(-self.max_rot, self.max_rot, n),
                    blur=(0.1 * r.uniform(0, 1, n)).astype(np.float32), order=order,
                    factor=np.take_along_axis(fac_of, order, 1).astype(np.float32))

    def assemble(self, idxs, draws=None):
        """Host side of a batch: frames (uint8 RGBX), GT arrays, inverse affines, flips and jitter draws."""
        n = len(idxs)
        if draws is None and self.aug:
            draws = self.draw(n)
        W, H = self.src.raw_size
        # decoded frames go, RGB and contiguous, into one of two pinned staging buffers (a strided RGB -> RGBX scatter on the
        # host costs 1 ms per 640x480 frame; the X byte is added on the device) and are uploaded with one asynchronous copy
        files = infos = None
        predecoded = tuple(int(i) for i in idxs) in self._predecoded      # predecode() read, parsed and decoded these files already
        kind = None
        if not predecoded and getattr(self.src, "get_image_bytes", None) is not None:
            files = [self.src.get_image_bytes(idx) for idx in idxs]      # header walk only (15 - 50 us per file, cached); the decode runs in augment()
            kind, infos = self._parse_files([int(i) for i in idxs], files)
        if infos is not None or predecoded:
            stage = None
        else:
            files = None
            if self._pin is None or self._pin[0].shape[0] < n or tuple(self._pin[0].shape[1:3]) != (H, W):
                self._pin = [torch.empty((n, H, W, 3), dtype=torch.uint8).pin_memory() if torch.cuda.is_available()
                             else torch.empty((n, H, W, 3), dtype=torch.uint8) for _ in range(2)]
            self._pin_i ^= 1
            stage = self._pin[self._pin_i][:n]
            frames = stage.numpy()

            def one(a):                     # Pillow's decoders release the GIL: the frames of a batch decode side by side
                frames[a[0]] = self.src.get_image(a[1])
            from .png import pool
            list(pool(self.num_workers).map(one, enumerate(idxs)))
        # splits other than train / trainval: JOINTS_VIS / CORNERS_VIS are forced to ones (hodata.py:296-313, 389-391)
        r = assemble_real_gt_batch([self.src.get_annots(idx) for idx in idxs], self.image_size, self.src.raw_size, draws, self.center_idx,
                                   self.bbox_expand, self.center_jit, self.scale_jit, self.src.sides,
                                   train_split=getattr(self.src, "data_split", "train") in ("train", "trainval"))
        gt = {k: r[k] for k in self.GT_KEYS}
        full = np.tile(np.eye(3), (n, 1, 1))
        full[:, :2] = r["affine"][:, :2].astype(np.float64)
        inv = np.linalg.inv(full)[:, :2].reshape(n, 6).astype(np.float32)
        flip, obj_idx = r["flip"].astype(np.uint8), r[Queries.OBJ_IDX]
        if draws is None:                                          # no augmentation: identity jitter, no blur (hodata.py:113-121)
            order = np.tile(np.arange(4, dtype=np.int32), (n, 1))
            factor = np.tile(np.array([1, 1, 0, 1], np.float32), (n, 1))
            blur = None
        else:
            order, factor, blur = draws["order"], draws["factor"], draws["blur"]
        return dict(frames=stage, files=files, file_kind=kind, file_infos=infos, gt={k: np.asarray(v, np.float32) for k, v in gt.items()}, inv=inv, flip=flip, obj_idx=obj_idx,
                    order=order, factor=factor, blur=blur, idxs=np.asarray(idxs, np.int64))

    def _upload(self, arrays):
        """{name: host array} -> {name: device tensor} through ONE pinned blob and ONE asynchronous copy.  The small per-batch arrays
        (ground truth, jitter draws, inverse affines: 15 of them) used to go up one `.to(device)` each from pageable memory -- every one a
        blocking staged copy queued behind the training step on the same stream, i.e. the host could not run ahead of the device and the
        batch assembly ended up serialised behind each step (round 4: the mixed step 10.9 -> see DESIGN 13.8)."""
        items = [(k, np.ascontiguousarray(v)) for k, v in arrays.items() if v is not None]
        offs, o = [], 0
        for _, a in items:
            offs.append(o)
            o += (a.nbytes + 15) & ~15
        k = self._up_k = getattr(self, "_up_k", 0) ^ 1
        pins = self.__dict__.setdefault("_up_pins", [None, None])
        evs = self.__dict__.setdefault("_up_evs", [None, None])
        if pins[k] is None or pins[k].numel() < o:
            pins[k] = torch.empty(max(2 * o, 1 << 16), dtype=torch.uint8)
            if self.dev.type == "cuda":
                pins[k] = pins[k].pin_memory()
            evs[k] = None
        if evs[k] is not None:
            evs[k].synchronize()                   # the copy that last read this staging buffer (two uploads ago)
        stage = pins[k].numpy()
        for (_, a), of in zip(items, offs):
            stage[of:of + a.nbytes] = a.view(np.uint8).reshape(-1)
        blob = torch.empty(max(o, 16), dtype=torch.uint8, device=self.dev)
        blob[:o].copy_(pins[k][:o], non_blocking=True)
        if self.dev.type == "cuda":
            evs[k] = torch.cuda.Event()
            evs[k].record()
        out = {}
        for (name, a), of in zip(items, offs):
            tdt = torch.from_numpy(a[:0].reshape(-1)).dtype if a.dtype != np.bool_ else torch.bool
            out[name] = blob[of:of + a.nbytes].view(tdt).view(a.shape)
        return out

    def augment(self, host, out_pad=None, out_chw=None, dev=None):
        """Upload + ab_augment_batch.  out_pad: zero-bordered NHWC4 rows to fill (a slice of the training batch).  dev: the batch's small
        arrays already on the device (batch() uploads them together with the ground truth)."""
        n = len(host["idxs"])
        W, H = self.src.raw_size
        ow, oh = self.image_size
        if dev is None:
            dev = self._upload(dict(order=host["order"], factor=host["factor"], inv=host["inv"], flip=host["flip"], blur=host["blur"]))
        lib = L.lib()
        need = lib.ab_augment_workspace_bytes(L.i(n), L.i(W), L.i(H))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        if out_pad is None and out_chw is None:
            out_chw = torch.empty((n, 3, oh, ow), dtype=torch.float32, device=self.dev)
        pre = self._predecoded.pop(tuple(int(i) for i in host["idxs"]), None) if self._predecoded else None
        if pre is not None:
            frames = pre
        elif host.get("files") is not None:                        # the .jpg / .png files themselves: decoded to RGBX on the device
            frames = torch.empty((n, H, W, 4), dtype=torch.uint8, device=self.dev)
            from .jpeg import JpegUnsupported
            from .png import PngUnsupported
            try:
                self._decoder(host["file_kind"]).decode(host["files"], out=frames, infos=host["file_infos"])
            except (JpegUnsupported, PngUnsupported):      # fails only while decoding: this batch goes through Pillow (the reference's get_image)
                rgb = torch.from_numpy(np.stack([np.asarray(self.src.get_image(i))[..., :3] for i in host["idxs"]])).to(self.dev)
                frames.zero_()
                frames[..., :3].copy_(rgb)
        else:
            rgb = host["frames"].to(self.dev, non_blocking=True)
            frames = torch.zeros(rgb.shape[:3] + (4,), dtype=torch.uint8, device=self.dev)
            frames[..., :3].copy_(rgb)                              # RGBX: the kernels fetch a pixel as one aligned dword
        order, factor, inv, flip, blur = dev["order"], dev["factor"], dev["inv"], dev["flip"], dev.get("blur")
        dt = L.dt(out_pad) if out_pad is not None else 0
        if self.image_plane == "u8n" and out_pad is not None:
            if out_pad.dtype != torch.bfloat16:      # (as DeviceRenderer.render: bf16 integers written into another dtype's buffer otherwise)
                raise TypeError(f"the integer image plane (compute_dtype \"u8n\") is written as bfloat16; out_pad is {out_pad.dtype}")
            dt = 2                         # AB_DT_U8N
        L.check(lib.ab_augment_batch(L.ptr(frames), L.i(n), L.i(W), L.i(H), L.ptr(order), L.ptr(factor), L.ptr(inv), L.ptr(blur),
                                     L.ptr(flip), L.i(ow), L.i(oh), L.i(dt), _ptr(out_pad), L.ptr(out_chw), L.ptr(self._ws), L.stream()),
                "ab_augment_batch")
        return out_chw

    def _sample_idx_of(self, idxs):
        """SAMPLE_IDX as the reference reports it: the dataset index get_sample_idxs()[position] (hodata.py __getitem__), not the position."""
        gs = getattr(self.src, "get_sample_idxs", None)
        if gs is None:
            return idxs
        m = gs()
        return np.asarray([m[int(i)] for i in idxs], np.int64)

    def batch(self, idxs, draws=None, out_pad=None, want_chw=True, out_chw=None):
        """Device batch dict with the reference's keys (hodata.py:315-450; IS_SYNTH false, CCV ids -1).  out_pad / out_chw: rows of a larger
        batch's tensors to write the frames into."""
        host = self.assemble(idxs, draws)
        n = len(idxs)
        ow, oh = self.image_size
        chw = out_chw if out_chw is not None else (torch.empty((n, 3, oh, ow), dtype=torch.float32, device=self.dev) if want_chw else None)
        small = dict(host["gt"], __order=host["order"], __factor=host["factor"], __inv=host["inv"], __flip=host["flip"], __blur=host["blur"],
                     __obj_idx=np.asarray(host["obj_idx"], np.int64), __idxs=self._sample_idx_of(host["idxs"]),
                     __is_synth=np.zeros(n, np.bool_), __minus1=np.full(n, -1, np.int64))
        up = self._upload(small)                   # every small array of the batch: one pinned blob, one asynchronous copy
        self.augment(host, out_pad=out_pad, out_chw=chw,
                     dev=dict(order=up["__order"], factor=up["__factor"], inv=up["__inv"], flip=up["__flip"], blur=up.get("__blur")))
        b = {k: up[k] for k in host["gt"]}
        b[Queries.OBJ_IDX] = up["__obj_idx"]
        b[Queries.SAMPLE_IDX] = up["__idxs"]
        b[SynthQueries.IS_SYNTH] = up["__is_synth"]
        for k in (SynthQueries.OBJ_ID, SynthQueries.PERSP_ID, SynthQueries.GRASP_ID):
            b[k] = up["__minus1"]                  # one read-only tensor for the three CCV ids of a real sample
        if chw is not None:
            b[Queries.IMAGE] = chw
        return b


def _ptr(t):
    return L.view_ptr(t)


class MixedLoader:
    """MixedDataset (mixed_dataset.py:5-37) + the shuffling DataLoader of train_artiboost.py, with a STATIC per-batch split:
    every batch holds n_real = round(B * real_len / (real_len + synth_len)) real samples (a random permutation of the real
    set over the epoch) followed by B - n_real synthetic ones (the epoch's CCV draws, already i.i.d.), instead of a
    hypergeometric count per batch -- fixed shapes keep the step replayable as a hipGraph.  After
    ArtiBoostLoader.synth_shutdown() + update(): real samples only (MixedDataset.remove_synth).  Under data parallelism the
    real set is sharded like a DistributedSampler (shared permutation, rank r takes perm[r::world]); the synthetic loader
    shards its own epoch the same way."""

    def __init__(self, real: RealBatcher, synth_loader, batch_size, seed=1, rank=0, world_size=1, decode_group=4, decode_ahead=None,
                 want_chw=True, reuse_buffers=0):
        self.real, self.synth, self.B = real, synth_loader, batch_size
        # want_chw: also the reference-shaped float CHW `image` (hodata.py:446) next to the zero-bordered NHWC4 tensor the HIP model reads
        # (50 MB written, concatenated and copied per batch of 64 that a loop feeding TrainStep never looks at).
        # reuse_buffers = n > 0: the image tensors of a batch come from a ring of n zero-bordered buffers (no 70 MB fill per batch) and are
        # overwritten n batches later -- for loops that consume a batch before asking for the next but n - 1 (TrainStep copies it into its
        # static inputs; StreamPrefetcher runs one ahead: n >= 3).  0: fresh tensors per batch, like a DataLoader.
        self.want_chw, self.reuse_buffers = bool(want_chw), int(reuse_buffers)
        self._ring, self._ring_i = [], 0
        # the .jpg frames of `decode_group` consecutive batches are decoded in one call (sources that serve file bytes): the call's time is the
        # longest Huffman chain, not the frame count -- 11.55 -> 11.24 ms per mixed step at 4 (tools/bench_mixed.py); 1: per batch
        self.decode_group = max(1, int(decode_group))
        # ... and the NEXT group on a side stream while this group's steps run: the decode is a latency-bound chain of small launches, the
        # one two-stream schedule measured to pay here (10.79 -> 10.49 ms per mixed step).  AB_JPEG_SIDE_STREAM=0 / decode_ahead=False: off
        self.decode_ahead = (os.environ.get("AB_JPEG_SIDE_STREAM", "1") != "0") if decode_ahead is None else bool(decode_ahead)
        self.rank, self.world = rank, world_size      # DistributedSampler semantics: one shared permutation, rank r keeps perm[r::world]
        self.rng = np.random.default_rng(seed)
        self.update()

    def update(self):
        self.real_len = len(self.real.src)
        # the synthetic loader's NOMINAL epoch length (all ranks, before trimming to whole batches): the number n_synth_for() was given
        self.synth_len = self.synth.synth_len if (self.synth is not None and self.synth.use_synth and self.synth.epoch is not None) else 0
        tot = self.real_len + self.synth_len
        self.n_real = self.B if self.synth_len == 0 else int(round(self.B * self.real_len / tot))
        self.n_synth = self.B - self.n_real
        assert self.synth is None or self.n_synth == 0 or self.synth.batch_size == self.n_synth, \
            "construct the ArtiBoostLoader with batch_size == MixedLoader.n_synth_for(...)"
        if self.synth is not None and self.n_synth and getattr(self.real, "image_plane", "f32") != getattr(self.synth, "image_plane", "f32"):
            raise ValueError("both halves of a mixed batch write ONE image tensor: build RealBatcher and ArtiBoostLoader with the same compute_dtype "
                             f"(real: {getattr(self.real, 'image_plane', 'f32')}, synthetic: {getattr(self.synth, 'image_plane', 'f32')})")

    @staticmethod
    def n_synth_for(batch_size, real_len, synth_len):
        return batch_size - int(round(batch_size * real_len / (real_len + synth_len)))

    def __len__(self):
        # RANK-INDEPENDENT (drop_last over ranks, as plan_epoch does for the synthetic share): every rank keeps real_len // world samples.
        # len(range(rank, real_len, world)) differs by one between ranks when real_len % world != 0, and a rank with one more batch
        # than the others would wait forever in the gradient all-reduce.
        n = (self.real_len // self.world) // self.n_real if self.n_real else 0
        return min(n, len(self.synth)) if self.n_synth else n

    def _epoch_perm(self):
        """This rank's real sample indices of the next epoch: the same seed on every rank -> one shared permutation, cut to
        world * (real_len // world) first so that the rank slices are disjoint AND equally long."""
        return self.rng.permutation(self.real_len)[:self.world * (self.real_len // self.world)][self.rank::self.world]

    def __iter__(self):
        perm = self._epoch_perm()
        self.real._predecoded.clear()      # frames decoded ahead for an epoch that was not finished
        self.real.drop_jobs()
        if getattr(self, "_dec_stream", None) is not None:      # ... and a decode of that epoch possibly still in flight on the side stream
            torch.cuda.current_stream(self.real.dev).wait_stream(self._dec_stream)
        W, H = self.real.image_size
        static = self.synth.new_static_batch() if self.n_synth else None
        for bi in range(len(self)):
            if self.decode_group > 1 and bi % self.decode_group == 0:
                G = self.decode_group
                group = lambda g0: [perm[b * self.n_real:(b + 1) * self.n_real] for b in range(g0, min(g0 + G, len(self)))]      # noqa: E731
                if not self.decode_ahead:
                    self.real.predecode(group(bi))
                else:      # group g + 1 is decoded on its own stream while the steps of group g run
                    cur = torch.cuda.current_stream(self.real.dev)
                    if bi == 0:
                        if getattr(self, "_dec_stream", None) is None:
                            self._dec_stream = torch.cuda.Stream(device=self.real.dev)
                        self.real.predecode(group(0))
                    else:
                        cur.wait_event(self._dec_event)
                    if bi + G < len(self):
                        self._dec_stream.wait_stream(cur)
                        with torch.cuda.stream(self._dec_stream):
                            self.real.predecode(group(bi + G), side=True)
                        self._dec_event = torch.cuda.Event()
                        self._dec_event.record(self._dec_stream)
                        if bi + 2 * G < len(self):      # ... and the host share (PNG inflates) of the group after that one
                            self.real.prefetch_files(group(bi + 2 * G), side=True)
                        for fr in self.real._predecoded.values():
                            fr.record_stream(cur)
            pad, chw = self._image_buffers(H, W)
            rb = self.real.batch(perm[bi * self.n_real:(bi + 1) * self.n_real], out_pad=pad[:self.n_real], want_chw=self.want_chw,
                                 out_chw=None if chw is None else chw[:self.n_real])
            if not self.n_synth:
                rb["image_nhwc4_padded"] = pad
                rb[IMAGE_PLANE_KEY] = PlaneTag(self.real.image_plane)
                yield rb
                continue
            # both halves write their frames straight into the batch's tensors: the renderer into rows n_real.. (no copy, no concatenation)
            self.synth.load_batch(static, bi)
            self.synth.render_into(static, want_chw=self.want_chw, out_pad=pad[self.n_real:], out_chw=None if chw is None else chw[self.n_real:])
            out = {"image_nhwc4_padded": pad, IMAGE_PLANE_KEY: PlaneTag(self.real.image_plane)}
            if chw is not None:
                out[Queries.IMAGE] = chw
            for k, v in rb.items():
                if k == Queries.IMAGE:
                    continue
                sv = static[k]
                out[k] = torch.cat([v, sv.to(v.dtype) if sv.dtype != v.dtype else sv])
            yield out

    def _image_buffers(self, H, W):
        """(zero-bordered NHWC4 [B, H + 6, W + 8, 4], float CHW [B, 3, H, W] or None) of the next batch: fresh, or the next of the ring."""
        dev, dt = self.real.dev, self.real.dtype
        fresh = lambda: (tag_image_plane(torch.zeros((self.B, H + 6, W + 8, 4), dtype=dt, device=dev), self.real.image_plane),      # noqa: E731
                         torch.empty((self.B, 3, H, W), dtype=torch.float32, device=dev) if self.want_chw else None)
        if self.reuse_buffers <= 0:
            return fresh()
        if len(self._ring) < self.reuse_buffers or self._ring[0][0].shape[1:3] != (H + 6, W + 8):
            if self._ring and self._ring[0][0].shape[1:3] != (H + 6, W + 8):
                self._ring = []
            self._ring.append(fresh())      # the borders are zeroed once: neither half ever writes them
            self._ring_i = len(self._ring) - 1
            return self._ring[-1]
        self._ring_i = (self._ring_i + 1) % self.reuse_buffers
        return self._ring[self._ring_i]


class StreamPrefetcher:
    """Iterates a device-batch loader (MixedLoader, or anything that yields dicts of device tensors) ONE BATCH AHEAD on its own HIP stream:
    while the training step of batch i runs on the caller's stream, batch i + 1 is decoded (ab_jpeg_decode_batch), augmented, rendered and
    concatenated on the side stream -- the role of the reference's DataLoader worker processes (train_artiboost.py: num_workers), without
    processes.  The batches are the loader's own, in its order (same draws, same bytes); a batch is handed over with a stream wait and
    `record_stream`, so the caching allocator does not recycle its memory while the consumer reads it."""

    def __init__(self, loader, device=None):
        self.loader = loader
        self.dev = torch.device(device) if device is not None else None

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        it = iter(self.loader)
        dev = self.dev
        side = None
        nxt = None

        def produce():
            nonlocal side, dev
            if side is None:
                dev = dev or torch.device("cuda", torch.cuda.current_device())
                side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))      # the previous batch's buffers may be freed by the consumer's stream
            with torch.cuda.stream(side):
                return next(it, None)

        nxt = produce()
        while nxt is not None:
            cur = torch.cuda.current_stream(dev)
            cur.wait_stream(side)
            batch = nxt
            for v in batch.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)
            nxt = produce()
            yield batch


class ThreadedPrefetcher:
    """StreamPrefetcher with the HOST side of the batch assembly off the training thread too: a worker thread iterates the loader up to `depth`
    batches ahead on its own HIP stream (ground-truth assembly in numpy, file parsing, decode planning, ~40 launches per mixed batch) while the
    training thread replays the step's graphs -- the role of the reference's DataLoader worker PROCESSES (train_artiboost.py: num_workers),
    as one thread.  Round 6: under the graph-replayed bf16x3 step (8.5 ms of device work) the mixed real + synthetic loop was bound by this
    host work -- the main queue idled 1.1 ms per step (2.6 under rocprofv3) with everything on one thread.

    Same batches, same order, same bytes as the loader's own iteration.  Hand-over: the worker records an event behind a batch's last launch,
    the consumer's stream waits on it and `record_stream`s the tensors.  Buffer reuse (MixedLoader's ring of `reuse_buffers` image tensors): when
    the consumer asks for the next batch, an event is recorded on ITS stream behind everything it enqueued for the previous one; the worker's
    stream waits on the event of batch j - n before it produces batch j into the same ring slot -- which exists by then because the queue
    keeps the worker at most depth + 1 batches ahead of the consumer: depth <= reuse_buffers - 2 (checked)."""

    def __init__(self, loader, depth=2, device=None, inline_first=0):
        self.loader, self.depth = loader, int(depth)
        self.inline_first = int(inline_first)      # the first batches are produced on the consumer's own thread and stream (see __iter__)
        self.dev = torch.device(device) if device is not None else None
        ring = int(getattr(loader, "reuse_buffers", 0))
        if ring > 0 and self.depth > ring - 2:
            raise ValueError(f"ThreadedPrefetcher(depth={self.depth}) over a ring of {ring} image buffers: needs depth <= reuse_buffers - 2")
        self.ring = ring

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        import queue
        import threading
        dev = self.dev or torch.device("cuda", torch.cuda.current_device())
        side = torch.cuda.Stream(device=dev)
        q = queue.Queue(maxsize=max(self.depth, 1))
        stop = threading.Event()
        done = {}                          # batch index -> event on the consumer's stream behind its use of that batch
        ring = self.ring
        it = iter(self.loader)
        k = 0
        for _ in range(self.inline_first):
            batch = next(it, None)
            if batch is None:
                return
            yield batch
            e = torch.cuda.Event()
            e.record(torch.cuda.current_stream(dev))
            done[k] = e
            k += 1
        start = torch.cuda.Event()
        start.record(torch.cuda.current_stream(dev))      # the worker's first launches follow whatever the consumer enqueued so far
        j0 = k

        def work():
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(side):
                    side.wait_event(start)
                    j = j0
                    while not stop.is_set():
                        if ring > 0 and j - ring in done:
                            side.wait_event(done.pop(j - ring))      # the ring slot batch j is written into was last read by batch j - ring
                        batch = next(it, None)
                        if batch is None:
                            break
                        ev = torch.cuda.Event()
                        ev.record(side)
                        while not stop.is_set():
                            try:
                                q.put((batch, ev), timeout=0.1)
                                break
                            except queue.Full:
                                pass
                        j += 1
                q.put(None)
            except BaseException as e:      # noqa: BLE001 -- re-raised in the consumer
                q.put(e)

        th = threading.Thread(target=work, name="artiboost-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                batch, ev = item
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                for v in batch.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
                yield batch
                e = torch.cuda.Event()      # (the consumer is back: everything it enqueued for batch k is on its stream by now)
                e.record(torch.cuda.current_stream(dev))
                done[k] = e
                k += 1
        finally:
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)
            torch.cuda.current_stream(dev).wait_stream(side)

"""Metric plugin classes used on the hot path (anakin/metrics/evaluator.py:12-85, meanepe.py:13-101,
val_metric.py:28-143, lossesmetric.py): Mean3DEPE (MPJPE/MPCPE in mm) and ValMetricMean3DEPE2, whose per-(object,
view, grasp) errors drive the CCV re-weighting (ArtiBoostLoader.step_eval)."""
from typing import Dict, List

import numpy as np
import torch

from .registry import CONST, METRIC, Queries, SynthQueries


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.avg, self.sum, self.count = 0.0, 0.0, 0

    def update(self, val, n=1):
        self.sum += val
        self.count += n
        self.avg = self.sum / self.count if self.count else 0.0


class Metric:
    def reset(self):
        pass

    def feed(self, preds, targs, **kwargs):
        pass

    def get_measures(self, **kwargs):
        return {}

    # Two-phase feeding (Evaluator.feed_all): `feed_device` enqueues the metric's device arithmetic and returns the small tensors its
    # host bookkeeping needs (None: not deferrable -- the evaluator calls `feed`, which reads the device back at once);
    # `feed_host` receives them as numpy arrays, one step later.  A metric's `feed` is exactly feed_device + feed_host.
    def feed_device(self, preds, targs, **kwargs):
        return None

    def feed_host(self, arrays, **kwargs):
        pass


def _epe_mm(preds, targs, key, mm, memo=None):
    """(B,) mean end-point error of `key`.  memo: per-feed_all cache (several metrics ask for the same key)."""
    if memo is not None:
        mk = ("epe", key, bool(mm))
        if mk not in memo:
            memo[mk] = _epe_mm(preds, targs, key, mm)
        return memo[mk]
    pred = preds[key]
    if "_abs" in key:
        val = targs[key.replace("_abs", "")].to(pred.device) + targs[Queries.ROOT_JOINT].to(pred.device).unsqueeze(1)
    else:
        val = targs[key].to(pred.device)
    diff = pred.detach() - val
    if mm:
        diff = diff * 1000.0
    return torch.norm(diff, p="fro", dim=2).mean(dim=1)       # (B,)


@METRIC.register_module
class Mean3DEPE(Metric):
    def __init__(self, **cfg):
        self.val_keys_list: List[str] = cfg["VAL_KEYS"]
        self.avg_meters = {k: AverageMeter() for k in self.val_keys_list}
        self.to_millimeters = cfg.get("MILLIMETERS", False)
        # meanepe.py:28-31,62-66: --filter_unseen_obj_idxs (through builder.build_evaluator_metric_list(..., arg=arg)) drops the
        # samples of those object classes from the CORNER errors (HO3D's unseen pitcher, README evaluation commands)
        arg = cfg.get("arg", cfg.get("ARG"))
        self.filter_unseen_obj_idxs = list(getattr(arg, "filter_unseen_obj_idxs", []) or []) if arg is not None else []

    def reset(self):
        for m in self.avg_meters.values():
            m.reset()

    def feed_device(self, preds, targs, memo=None, **kwargs):
        """Per key: [sum of the kept samples' errors, number kept] as one (2,) fp32 tensor (no boolean indexing: no host read)."""
        out = []
        for key in self.val_keys_list:
            d = _epe_mm(preds, targs, key, self.to_millimeters, memo)
            n = torch.full((), float(d.shape[0]), dtype=torch.float32, device=d.device)
            if "corners" in key and self.filter_unseen_obj_idxs:
                oi = targs[Queries.OBJ_IDX].to(d.device)
                keep = torch.ones_like(oi, dtype=torch.bool)
                for idx in self.filter_unseen_obj_idxs:
                    keep &= oi != idx
                d = torch.where(keep, d, torch.zeros_like(d))
                n = keep.sum().to(torch.float32)
            out.append(torch.stack([d.sum().to(torch.float32), n]))
        return out

    def feed_host(self, arrays, **kwargs):
        for key, a in zip(self.val_keys_list, arrays):
            self.avg_meters[key].update(float(a[0]), n=int(a[1]))

    def feed(self, preds, targs, **kwargs):
        self.feed_host([t.cpu().numpy() for t in self.feed_device(preds, targs)])

    def get_measures(self, **kwargs):
        return {f"{k}_mepe": m.avg for k, m in self.avg_meters.items()}

    def __str__(self):
        return " | ".join(f"{k}_mepe: {m.avg:6.4f}" for k, m in self.avg_meters.items())


@METRIC.register_module
class ValMetricMean3DEPE2(Metric):
    def __init__(self, **cfg):
        self.val_keys_list: List[str] = cfg["VAL_KEYS"]
        self.storage = {k: {} for k in self.val_keys_list}
        self.to_millimeters = cfg.get("MILLIMETERS", False)

    def reset(self):
        for k in self.storage:
            self.storage[k] = {}

    def feed_device(self, preds, targs, memo=None, **kwargs):
        """[(B, 4) int64 (obj, persp, grasp, is_synth)] + one (B,) error vector per key."""
        d = [_epe_mm(preds, targs, key, self.to_millimeters, memo).to(torch.float32) for key in self.val_keys_list]
        dev = d[0].device if d else None
        ids = torch.stack([torch.as_tensor(targs[k]).to(device=dev, dtype=torch.int64) for k in
                           (SynthQueries.OBJ_ID, SynthQueries.PERSP_ID, SynthQueries.GRASP_ID, SynthQueries.IS_SYNTH)], 1)
        return [ids] + d

    def feed_host(self, arrays, **kwargs):
        ids = arrays[0].tolist()
        for key, d in zip(self.val_keys_list, arrays[1:]):
            st = self.storage[key]
            for i, t in enumerate(ids):
                if t[3]:
                    st[(t[0], t[1], t[2])] = d[i]                          # last write wins (val_metric.py:51-52)

    def feed(self, preds, targs, **kwargs):
        self.feed_host([t.cpu().numpy() for t in self.feed_device(preds, targs)])

    def get_measures(self, **kwargs):
        return dict(self.storage)

    def get_measures_averaged(self, **kwargs) -> Dict:
        stores = [self.storage[k] for k in self.val_keys_list]
        return {k: sum(s[k] for s in stores) / len(stores) for k in stores[0].keys()}

    def __str__(self):
        return ""


@METRIC.register_module
class LossesMetric(Metric):
    def __init__(self, **cfg):
        self.meters = {}

    def reset(self):
        self.meters = {}

    def feed_device(self, preds, targs, losses=None, **kwargs):
        """The step's loss scalars stacked into ONE fp32 vector (the reference reads each with .item(): a host round trip per loss);
        the key order travels on the host side.  Host numbers (python floats) ride along as constants."""
        keys, vals, dev = [], [], None
        for k, v in (losses or {}).items():
            if v is None:
                continue
            keys.append(k)
            vals.append(v)
            if torch.is_tensor(v) and dev is None:
                dev = v.device
        if dev is None:                      # nothing on a device: plain floats
            return [np.asarray([float(v) for v in vals], np.float32)], keys
        return [torch.stack([(v.detach() if torch.is_tensor(v) else torch.as_tensor(float(v))).to(device=dev, dtype=torch.float32).reshape(())
                             for v in vals])], keys

    def feed_host(self, arrays, extra=None, **kwargs):
        for k, v in zip(extra or [], arrays[0].tolist()):
            self.meters.setdefault(k, AverageMeter()).update(v, 1)

    def feed(self, preds, targs, losses=None, **kwargs):
        arrays, keys = self.feed_device(preds, targs, losses=losses)
        self.feed_host([a.cpu().numpy() if torch.is_tensor(a) else a for a in arrays], extra=keys)

    def get_measures(self, **kwargs):
        return {k: m.avg for k, m in self.meters.items()}

    def __str__(self):
        m = self.meters.get("final_loss")
        return f"final_loss: {m.avg:.4e}" if m else ""


class Evaluator:
    """anakin/metrics/evaluator.py:12-85, without a device synchronisation per step.

    The reference's `feed_all` (called after every batch, train_artiboost.py:96-98) moves tensors to the host inside every metric
    (`.item()` per loss scalar, `.cpu()` per error vector): on a GPU that runs ahead of the host each of them stalls the loop until the
    device has drained.  Here a metric with the two-phase API (Metric.feed_device / feed_host: Mean3DEPE, ValMetricMean3DEPE2,
    LossesMetric) only ENQUEUES its arithmetic; the few hundred bytes it needs on the host are packed into one pinned buffer with one
    asynchronous copy + an event, and applied when a later feed_all finds the event complete -- at most `max_lag` steps late (1: the
    progress string of train_artiboost.py:105 shows the numbers through the previous step).  Every read of the measures
    (`metrics_list`, `get_measures_all*`, `dump_images`, `reset_all`) applies what is still in flight first, so at those points the
    state equals per-step blocking feeds exactly.  Metrics without the API (PCK, AR, Vis*) are fed at once, as before.
    AB_EVAL_BLOCKING=1 (or max_lag=0): every feed applied before feed_all returns."""

    def __init__(self, cfg, metrics_list, max_lag=None):
        import os
        self._metrics_list = metrics_list
        if max_lag is None:
            max_lag = 0 if os.environ.get("AB_EVAL_BLOCKING") == "1" else 1
        self.max_lag = int(max_lag)
        self._inflight = []        # FIFO of (event, pinned buffer, [(metric, [(offset, nbytes, dtype, shape)], extra)])
        self._free = []            # pinned buffers to reuse

    @property
    def metrics_list(self):
        self.flush()               # a reader of the metrics (ArtiBoostLoader.step_eval, the recorder) sees every fed batch
        return self._metrics_list

    def reset_all(self):
        self.flush()
        for m in self._metrics_list:
            m.reset()

    # ---- deferred read-back
    def _apply(self, entry):
        ev, pin, plan = entry
        ev.synchronize()
        host = pin.numpy()
        for m, lay, extra in plan:
            arrays = [host[o:o + nb].view(dt).reshape(shp).copy() for o, nb, dt, shp in lay]
            m.feed_host(arrays, extra=extra)
        self._free.append(pin)

    def _drain(self, keep):
        while len(self._inflight) > keep:
            self._apply(self._inflight.pop(0))
        while self._inflight and self._inflight[0][0].query():      # already on the host: apply (keeps the progress string fresh)
            self._apply(self._inflight.pop(0))

    def flush(self):
        """Apply every feed still in flight (blocks until the device has produced them)."""
        self._drain(0)

    def feed_all(self, preds, targs, losses=None, **kwargs):
        memo, plan, chunks, off = {}, [], [], 0
        dev = None
        for m in self._metrics_list:
            res = m.feed_device(preds, targs, losses=losses, memo=memo) if not isinstance(m, VisMetric) else None
            if res is None:
                m.feed(preds, targs, losses=losses) if isinstance(m, LossesMetric) else m.feed(preds, targs)
                continue
            tensors, extra = res if isinstance(res, tuple) else (res, None)
            if not all(torch.is_tensor(t) and t.is_cuda for t in tensors):      # host tensors / numbers: nothing to wait for
                m.feed_host([t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t) for t in tensors], extra=extra)
                continue
            lay = []
            for t in tensors:
                t = t.detach().contiguous()
                dev = t.device
                nb = t.numel() * t.element_size()
                lay.append((off, nb, torch.empty(0, dtype=t.dtype).numpy().dtype, tuple(t.shape)))
                chunks.append(t.view(torch.uint8).reshape(-1) if t.dim() else t.reshape(1).view(torch.uint8))
                pad = (-nb) % 8
                if pad:
                    chunks.append(torch.zeros(pad, dtype=torch.uint8, device=dev))
                off += nb + pad
            plan.append((m, lay, extra))
        if plan:
            blob = torch.cat(chunks) if len(chunks) > 1 else chunks[0]
            pin = None
            for i, b in enumerate(self._free):
                if b.numel() >= off:
                    pin = self._free.pop(i)
                    break
            if pin is None:
                pin = torch.empty(max(off, 4096), dtype=torch.uint8).pin_memory()
            pin[:off].copy_(blob, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            self._inflight.append((ev, pin, plan))
        self._drain(self.max_lag)

    def get_measures_all(self):
        self.flush()
        out = {}
        for m in self._metrics_list:
            if isinstance(m, VisMetric):
                continue
            out.update(m.get_measures())
        return out

    def get_measures_all_striped(self, return_losses=True):
        """evaluator.py:58-74: {metric class name: {measure: float}} (scalars only) for the recorder / summarizer."""
        self.flush()
        out = {}
        for m in self._metrics_list:
            if isinstance(m, VisMetric) or (not return_losses and isinstance(m, LossesMetric)):
                continue
            out[type(m).__name__] = {k: float(v) for k, v in m.get_measures().items()
                                     if isinstance(v, (float, int)) or (hasattr(v, "ndim") and getattr(v, "ndim") == 0)}
        return out

    def dump_images(self):
        """evaluator.py:76-82: {metric class name: image} of the Vis* metrics."""
        return {type(m).__name__: m.image for m in self._metrics_list if isinstance(m, VisMetric)}

    def __str__(self):
        """The progress string (train_artiboost.py:105): never waits for the device -- numbers through the last feed already on the host."""
        self._drain(self.max_lag)
        return " | ".join(s for s in (str(m) for m in self._metrics_list if not isinstance(m, VisMetric)) if s)


class VisMetric(Metric):
    """anakin/metrics/vismetric.py:18-68: a metric whose product is an image (skipped by the evaluator's measure tables)."""

    def __init__(self, **cfg):
        self.image, self.count = None, 0

    def reset(self):
        self.count = 0

    def get_measures(self, **kwargs):
        raise NotImplementedError()

    def __str__(self):
        return ""


@METRIC.register_module
class Vis2DMetric(VisMetric):
    """anakin/metrics/vismetric.py:71-200: the first batch fed after a reset is drawn as an NROW x NCOL grid of crops with the
    hand skeleton, the object's box edges, the ground-truth root (star) and box corners 0 / 7 (triangles) -- predictions on the
    left half, ground truth on the right.  `image`: uint8 BGR [H, 2W, 3] like the reference's.  Drawn with PIL (the reference
    uses matplotlib); later batches of the epoch only count."""
    FINGER_COLORS = [(255, 0, 0), (255, 0, 255), (0, 0, 255), (0, 255, 255), (0, 255, 0)]       # thumb .. little (RGB)
    BOX_EDGES = [(0, 1), (1, 3), (3, 2), (2, 0), (4, 5), (5, 7), (7, 6), (6, 4), (1, 5), (2, 6), (3, 7), (0, 4)]

    def __init__(self, **cfg):
        super().__init__(**cfg)
        self.inp_res = cfg["DATA_PRESET"]["IMAGE_SIZE"]
        self.ncol, self.nrow = int(cfg.get("NCOL", 3)), int(cfg.get("NROW", 3))
        self.corner_link_order = list(cfg.get("CORNER_LINK_ORDER", range(8)))

    @staticmethod
    def _images(targs):
        img = targs.get(Queries.IMAGE)
        if img is not None:
            return (img.detach().float().cpu().permute(0, 2, 3, 1) + 0.5).clamp(0, 1).numpy()
        pad = targs["image_nhwc4_padded"]                    # the HIP loader's batch: zero-bordered NHWC4 (fp32 / bf16 or its planes)
        if pad.dim() == 5:
            pad = pad[0].float() + pad[1].float()
        from .registry import image_plane_of
        u8n = image_plane_of(targs, pad) == "u8n"            # the loaders' tag: the integer plane 2 v - 255 (AB_DT_U8N)
        pad = pad.detach().float().cpu()
        if u8n:
            pad = pad / 510.0
        return (pad[:, 3:-3, 3:-5, :3] + 0.5).clamp(0, 1).numpy()

    def _draw(self, images, joints, corners, root, jvis, cvis):
        from PIL import Image, ImageDraw
        W, H = int(self.inp_res[0]), int(self.inp_res[1])
        grid = Image.new("RGB", (self.ncol * W, self.nrow * H))
        for i in range(min(self.ncol * self.nrow, images.shape[0])):
            tile = Image.fromarray((images[i] * 255.0 + 0.5).astype(np.uint8)).resize((W, H))
            d = ImageDraw.Draw(tile)
            j, c = joints[i], corners[i]
            for k in range(1, CONST.NUM_JOINTS):
                par = CONST.JOINTS_IDX_PARENTS[k]
                d.line([tuple(j[par]), tuple(j[k])], fill=self.FINGER_COLORS[(k - 1) // 4], width=2)
            for k in range(CONST.NUM_JOINTS):
                r = 2 if jvis is None or jvis[i][k] > 0 else 1
                d.ellipse([j[k][0] - r, j[k][1] - r, j[k][0] + r, j[k][1] + r], outline=(255, 255, 255))
            for a, b in self.BOX_EDGES:
                a, b = self.corner_link_order[a], self.corner_link_order[b]
                d.line([tuple(c[a]), tuple(c[b])], fill=(64, 224, 208), width=2)
            d.regular_polygon((float(root[i][0]), float(root[i][1]), 5), 5, fill=(138, 43, 226))
            d.regular_polygon((float(c[0][0]), float(c[0][1]), 4), 3, fill=(255, 0, 0))
            d.regular_polygon((float(c[7][0]), float(c[7][1]), 4), 3, fill=(255, 255, 0))
            grid.paste(tile, ((i % self.ncol) * W, (i // self.ncol) * H))
        return np.asarray(grid)[:, :, ::-1]

    def feed(self, preds, targs, **kwargs):
        bs = int(targs[Queries.JOINTS_2D].shape[0])
        if self.count > 0:
            self.count += bs
            return
        res = np.asarray(self.inp_res, dtype=np.float32)
        if "2d_uvd" in preds:
            uvd = preds["2d_uvd"].detach().float().cpu().numpy()
            pj, pc = uvd[:, :CONST.NUM_JOINTS, :2] * res, uvd[:, CONST.NUM_JOINTS:CONST.NUM_JOINTS + 8, :2] * res
        else:
            pj, pc = preds["joints_2d"].detach().float().cpu().numpy(), preds["corners_2d"].detach().float().cpu().numpy()
        gj = targs[Queries.JOINTS_2D][:, :CONST.NUM_JOINTS].detach().float().cpu().numpy()
        gc = targs[Queries.CORNERS_2D].detach().float().cpu().numpy()
        jv = targs[Queries.JOINTS_VIS].detach().float().cpu().numpy() if Queries.JOINTS_VIS in targs else None
        cv = targs[Queries.CORNERS_VIS].detach().float().cpu().numpy() if Queries.CORNERS_VIS in targs else None
        images = self._images(targs)
        self.image = np.concatenate([self._draw(images, pj, pc, gj[:, 0], jv, cv), self._draw(images, gj, gc, gj[:, 0], jv, cv)], axis=1)
        self.count += bs


@METRIC.register_module
class VisHand2DMetric(Vis2DMetric):
    """vismetric.py:361: the hand-only variant shares the drawing."""


# ============================================================================ eval / submit path
@METRIC.register_module
class Mean2DEPE(Mean3DEPE):
    """anakin/metrics/meanepe.py:97-101: pixel errors, never scaled to millimetres."""

    def __init__(self, **cfg):
        super().__init__(**cfg)
        self.to_millimeters = False


class PCKMetric(Metric):
    """anakin/metrics/pckmetric.py:12-143: per-keypoint Euclidean errors of the visible keypoints; mean EPE, PCK curve
    over STEPS thresholds in [VAL_MIN, VAL_MAX], AUC by the trapezoid rule normalised by the area under 1."""
    num_kp = 0
    keys = ("", "", "")

    def __init__(self, **cfg):
        self.val_min, self.val_max, self.steps = cfg["VAL_MIN"], cfg["VAL_MAX"], cfg["STEPS"]
        self.reset()

    def reset(self):
        self.data = [[] for _ in range(self.num_kp)]
        self.count = 0

    def feed(self, preds, targs, **kwargs):
        kp, kt, kv = self.keys
        p, t = preds[kp].detach(), targs[kt].to(preds[kp].device)
        dist = torch.sqrt(torch.sum((p - t) ** 2, dim=-1)).cpu().numpy()           # (B, N)
        vis = np.asarray(targs[kv].detach().cpu()).astype(bool)
        assert dist.ndim == 2 and vis.shape == dist.shape
        for i in range(self.num_kp):
            self.data[i].extend(dist[vis[:, i], i].tolist())
        self.count += dist.shape[0]

    def _get_pck(self, kp_id, threshold):
        if not self.data[kp_id]:
            return None
        return float(np.mean((np.array(self.data[kp_id]) <= threshold).astype("float")))

    def get_pck_all(self, threshold):
        vals = [v for v in (self._get_pck(i, threshold) for i in range(self.num_kp)) if v is not None]
        return float(np.mean(np.array(vals))) if vals else float("nan")

    def get_measures(self, **kwargs):
        thresholds = np.array(np.linspace(self.val_min, self.val_max, self.steps))
        area_under_one = np.trapezoid(np.ones_like(thresholds), thresholds)
        epe, auc, curves = [], [], []
        for i in range(self.num_kp):
            if not self.data[i]:
                continue
            d = np.array(self.data[i])
            epe.append(np.mean(d))
            curve = np.array([np.mean((d <= t).astype("float")) for t in thresholds])
            curves.append(curve)
            auc.append(np.trapezoid(curve, thresholds) / area_under_one)
        return {"epe_mean_per_kp": np.array(epe), "pck_curve_per_kp": np.array(curves), "auc_per_kp": np.array(auc),
                "epe_mean_all": np.mean(np.array(epe)), "auc_all": np.mean(np.array(auc)), "thresholds": thresholds}


def _pck(name, n, keys, label=None):
    def __str__(self):
        return f"{label}: {self.get_pck_all(0.02):6.4f}" if label else ""
    return METRIC.register_module(type(name, (PCKMetric,), {"num_kp": n, "keys": keys, "__str__": __str__,
                                                          "__doc__": "anakin/metrics/pckmetric.py:146-197"}))


Hand3DPCKMetric = _pck("Hand3DPCKMetric", 21, ("joints_3d", "joints_3d", "joints_vis"), "hand3d pck")
Hand2DPCKMetric = _pck("Hand2DPCKMetric", 21, ("joints_2d", "joints_2d", "joints_vis"))
Obj3DPCKMetric = _pck("Obj3DPCKMetric", 8, ("corners_3d", "corners_3d", "corners_vis"), "obj3d pck")
Obj2DPCKMetric = _pck("Obj2DPCKMetric", 8, ("corners_2d", "corners_2d", "corners_vis"))


class _MSSDBase:
    """Maximum symmetry-aware surface distance (anakin/metrics/bopAR.py:74-195, val_metric.py:235-327): per sample
    min over the object's symmetry set of max over the model points of ||sym(gt) - pred||.  The reference loops over the
    object classes with boolean masks (a device synchronisation per class); here the symmetry sets are padded to one
    length with identities (the identity is in every set, so the minimum is unchanged) and the whole batch is one product."""

    def __init__(self, **cfg):
        import json
        from .criterions import get_symmetry_transformations
        info = cfg.get("MODEL_INFO") or json.load(open(cfg["MODEL_INFO_PATH"], "r"))
        step = cfg.get("MAX_SYM_DISC_STEP", 0.01)
        self.n_obj = len(info)
        self.mssd_use_corners = cfg.get("MSSD_USE_CORNERS", False)
        self.use_ho3d_ycb = cfg.get("USE_HO3D_YCB", False)
        self.center_idx = cfg["DATA_PRESET"]["CENTER_IDX"] if cfg.get("MSSD_USE_CENTER_IDX", False) else None
        syms = [get_symmetry_transformations(info[str(i)], step) for i in range(1, self.n_obj + 1)]
        kmax = max(len(x) for x in syms)
        R = np.tile(np.eye(3), (self.n_obj, kmax, 1, 1))
        t = np.zeros((self.n_obj, kmax, 3, 1))
        for i, tr in enumerate(syms):
            for k, x in enumerate(tr):
                R[i, k], t[i, k] = x["R"], x["t"]
        self.R, self.t = torch.Tensor(R), torch.Tensor(t) / 1000.0                 # mm -> m
        self._dev = None

    def values(self, preds, targs):
        """-> (obj_idx [B] int64 1-based, mssd [B] in metres), both on the predictions' device."""
        dev = preds["box_rot_rotmat"].device
        if self._dev != dev:
            self.R, self.t, self._dev = self.R.to(dev), self.t.to(dev), dev
        can = targs[Queries.CORNERS_CAN if self.mssd_use_corners else "obj_verts_can"].to(dev)
        transf = targs[Queries.OBJ_TRANSF].to(dev)
        obj_idx = targs[Queries.OBJ_IDX].to(dev).long()
        sym_R, sym_t = self.R[obj_idx - 1], self.t[obj_idx - 1]                      # [B,K,3,3], [B,K,3,1]
        if not self.use_ho3d_ycb:
            sym_can = (torch.einsum("bkmn,bvn->bkmv", sym_R, can) + sym_t).transpose(-2, -1)
        else:
            ext = torch.tensor([[1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0]], dtype=torch.float32, device=dev)
            sym_can = (ext @ (torch.einsum("bkmn,bnv->bkmv", sym_R, ext @ can.transpose(-2, -1)) + sym_t)).transpose(-2, -1)
        sym_abs = (torch.einsum("bij,bklj->bkil", transf[:, :3, :3], sym_can) + transf[:, None, :3, 3:]).transpose(-2, -1)
        if self.mssd_use_corners:
            pred_abs = preds["corners_3d_abs"]
        else:
            pred_abs = (preds["box_rot_rotmat"] @ can.transpose(-2, -1)).transpose(-2, -1) + preds["boxroot_3d_abs"]
        if self.center_idx is None:
            d = sym_abs - pred_abs.unsqueeze(1)
        else:
            d = ((sym_abs - targs[Queries.ROOT_JOINT].to(dev)[:, None, None, :]) -
                 (pred_abs - preds["joints_3d_abs"][:, [self.center_idx]]).unsqueeze(1))
        return obj_idx, torch.norm(d, dim=-1).max(-1)[0].min(-1)[0].detach()


@METRIC.register_module
class AR(Metric):
    """anakin/metrics/bopAR.py:15-61 with USE_MSSD (VSD / MSPD raise NotImplementedError in the reference too)."""

    def __init__(self, **cfg):
        if cfg.get("USE_VSD", False) or cfg.get("USE_MSPD", False):
            raise NotImplementedError()
        self.mssd = _MSSDBase(**cfg) if cfg.get("USE_MSSD", False) else None
        self.reset()

    def reset(self):
        self._sum = self._cnt = None

    def feed(self, preds, targs, **kwargs):
        if self.mssd is None:
            return
        obj_idx, v = self.mssd.values(preds, targs)
        if self._sum is None:
            self._sum = torch.zeros(self.mssd.n_obj, dtype=torch.float64, device=v.device)
            self._cnt = torch.zeros(self.mssd.n_obj, dtype=torch.float64, device=v.device)
        self._sum.index_add_(0, obj_idx - 1, v.double())                 # per-object sums stay on the device: no sync per step
        self._cnt.index_add_(0, obj_idx - 1, torch.ones_like(v, dtype=torch.float64))

    @property
    def objs_error(self):
        out = {i + 1: AverageMeter() for i in range(self.mssd.n_obj)}
        if self._sum is not None:
            for i, (s_, c_) in enumerate(zip(self._sum.cpu().tolist(), self._cnt.cpu().tolist())):
                if c_:
                    out[i + 1].update(s_, n=int(c_))
        return out

    @property
    def avg(self):
        if self._sum is None:
            return float("nan")
        c = float(self._cnt.sum())
        return float(self._sum.sum()) / c * 1000.0 if c else float("nan")

    def get_measures(self, **kwargs):
        if self.mssd is None:
            return {}
        tag = ".corner" if self.mssd.mssd_use_corners else ""
        out = {"MSSD": self.avg}
        out.update({f"{i}{tag}.mssd": m.avg * 1000.0 for i, m in self.objs_error.items()})
        return out

    def __str__(self):
        return f"mssd: {self.avg:6.4f}" if self.mssd is not None else ""


@METRIC.register_module
class ValMetricAR2(Metric):
    """anakin/metrics/val_metric.py:145-222: per-(object, view, grasp) MSSD in mm of the synthetic samples (last write
    wins), the second mining signal ArtiBoostLoader.get_evaluator_result accepts (artiboost_loader.py:301-327)."""

    def __init__(self, **cfg):
        if cfg.get("USE_VSD", False) or cfg.get("USE_MSPD", False):
            raise NotImplementedError()
        self.mssd = _MSSDBase(**cfg) if cfg.get("USE_MSSD", False) else None
        self.reset()

    def reset(self):
        self.storage = {}

    def feed(self, preds, targs, **kwargs):
        if self.mssd is None:
            return
        obj_idx, v = self.mssd.values(preds, targs)
        order = torch.argsort(obj_idx, stable=True).cpu().numpy()        # the reference visits the object classes in order:
        vals = (v * 1000.0).cpu().numpy()                                # later classes overwrite earlier ones on a repeated triplet
        flags = np.asarray(targs[SynthQueries.IS_SYNTH].cpu()).astype(bool)
        ids = np.stack([np.asarray(targs[k].cpu()) for k in (SynthQueries.OBJ_ID, SynthQueries.PERSP_ID, SynthQueries.GRASP_ID)], 1)
        for i in order:
            if flags[i]:
                self.storage[tuple(int(x) for x in ids[i])] = vals[i]

    def get_measures(self, **kwargs):
        return {"mssd": self.storage} if self.mssd is not None else {}

    def get_measures_averaged(self, **kwargs):
        return dict(self.storage)

    def __str__(self):
        return ""

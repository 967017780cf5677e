"""Metric plugin classes used on the hot path (anakin/metrics/evaluator.py:12-85, meanepe.py:13-101,
val_metric.py:28-143, lossesmetric.py): Mean3DEPE (MPJPE/MPCPE in mm) and ValMetricMean3DEPE2, whose per-(object,
view, grasp) errors drive the CCV re-weighting (ArtiBoostLoader.step_eval)."""
from typing import Dict, List

import numpy as np
import torch

from .registry import METRIC, Queries, SynthQueries


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


def _epe_mm(preds, targs, key, mm):
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

    def reset(self):
        for m in self.avg_meters.values():
            m.reset()

    def feed(self, preds, targs, **kwargs):
        for key in self.val_keys_list:
            d = _epe_mm(preds, targs, key, self.to_millimeters)
            self.avg_meters[key].update(float(d.sum()), n=d.shape[0])

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

    def feed(self, preds, targs, **kwargs):
        synth = np.asarray(targs[SynthQueries.IS_SYNTH].cpu()).astype(bool)
        ids = list(zip(*(np.asarray(targs[k].cpu()).tolist() for k in
                         (SynthQueries.OBJ_ID, SynthQueries.PERSP_ID, SynthQueries.GRASP_ID))))
        for key in self.val_keys_list:
            d = _epe_mm(preds, targs, key, self.to_millimeters).cpu().numpy()
            for i, t in enumerate(ids):
                if synth[i]:
                    self.storage[key][tuple(int(x) for x in t)] = d[i]     # last write wins (val_metric.py:51-52)

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

    def feed(self, preds, targs, losses=None, **kwargs):
        for k, v in (losses or {}).items():
            if v is None:
                continue
            self.meters.setdefault(k, AverageMeter()).update(float(v), 1)

    def get_measures(self, **kwargs):
        return {k: m.avg for k, m in self.meters.items()}

    def __str__(self):
        m = self.meters.get("final_loss")
        return f"final_loss: {m.avg:.4e}" if m else ""


class Evaluator:
    """anakin/metrics/evaluator.py:12-85."""

    def __init__(self, cfg, metrics_list):
        self._metrics_list = metrics_list

    @property
    def metrics_list(self):
        return self._metrics_list

    def reset_all(self):
        for m in self._metrics_list:
            m.reset()

    def feed_all(self, preds, targs, losses=None, **kwargs):
        for m in self._metrics_list:
            if isinstance(m, LossesMetric):
                m.feed(preds, targs, losses=losses)
            else:
                m.feed(preds, targs)

    def get_measures_all(self):
        out = {}
        for m in self._metrics_list:
            out.update(m.get_measures())
        return out

    def __str__(self):
        return " | ".join(s for s in (str(m) for m in self._metrics_list) if s)

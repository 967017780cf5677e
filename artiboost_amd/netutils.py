"""build_optimizer / build_scheduler with the reference's signatures (anakin/utils/netutils.py:26-63), as called by
train/train_artiboost.py:133-137 with **cfg["TRAIN"].

Adam over the flat parameter of the HIP model returns `FusedClipAdam(max_norm=None)`: the same update as torch.optim.Adam
(weight_decay 0) as one fused pass that also refreshes the compute-precision weight copies; the caller's own
`clip_grad_norm_(arch_model.parameters(), grad_clip)` keeps working because the model exposes exactly that one parameter.
Anything else (other parameters, weight decay, SGD) gets the torch optimizer the reference would build."""
from typing import Iterable

import torch
from torch.optim import Optimizer


def _flat_hip_owner(groups):
    """The HybridBaseline whose single flat parameter is the only entry of `groups`, else None."""
    ps = [p for g in groups for p in g["params"]]
    if len(ps) != 1:
        return None
    owner = getattr(ps[0], "_ab_owner", None)
    return owner if owner is not None and owner.flat_param is ps[0] else None


def build_optimizer(params: Iterable, **cfg):
    groups = [dict(g, params=list(g["params"])) if isinstance(g, dict) else {"params": [g]} for g in params]
    name = cfg["OPTIMIZER"]
    wd = float(cfg.get("WEIGHT_DECAY", 0.0))
    owner = _flat_hip_owner(groups)
    if owner is not None and getattr(owner.store, "frozen_bn", False) and (wd != 0.0 or name not in ("Adam", "adam")):
        # FrozenBatchNorm2d's weight / bias are BUFFERS in the reference (resnet.py:33-69): no optimizer ever touches them.  Here they
        # live in the one flat parameter and are protected by a zero gradient -- which only the fused Adam (no decay) honours: a torch
        # optimizer with weight decay, or SGD with momentum state, would move them.
        raise NotImplementedError("BACKBONE.FREEZE_BATCHNORM with WEIGHT_DECAY != 0 or OPTIMIZER != Adam: the frozen affine terms share the "
                                  "flat parameter and would be decayed; not built (no shipped config combines them)")
    if name in ("Adam", "adam"):
        if owner is not None and wd == 0.0 and owner.flat_param.is_cuda:
            from .optim import FusedClipAdam
            return FusedClipAdam(groups, lr=cfg["LR"], max_norm=None, model=owner)
        return torch.optim.Adam(groups, lr=cfg["LR"], weight_decay=wd)
    if name in ("SGD", "sgd"):
        return torch.optim.SGD(groups, lr=cfg["LR"], momentum=float(cfg.get("MOMENTUM", 0.0)), weight_decay=wd)
    raise NotImplementedError(f"{name} not yet be implemented")


def build_scheduler(optimizer: Optimizer, **cfg):
    scheduler = cfg.get("SCHEDULER", "StepLR")
    if scheduler == "StepLR":
        return torch.optim.lr_scheduler.StepLR(optimizer, cfg["LR_DECAY_STEP"], gamma=cfg["LR_DECAY_GAMMA"])
    if scheduler in ("constant_warmup", "cosine_warmup", "linear_warmup"):
        import transformers      # the reference's dependency for the warm-up schedules (requirements.txt)
        if scheduler == "constant_warmup":
            return transformers.get_constant_schedule_with_warmup(optimizer, num_warmup_steps=cfg["NUM_WARMUP_STEPS"])
        fn = transformers.get_cosine_schedule_with_warmup if scheduler == "cosine_warmup" else transformers.get_linear_schedule_with_warmup
        return fn(optimizer, num_warmup_steps=cfg["NUM_WARMUP_STEPS"], num_training_steps=cfg["NUM_TRAINING_STEPS"])
    raise NotImplementedError(f"{scheduler} not yet be implemented")

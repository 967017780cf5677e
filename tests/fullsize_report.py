"""TEST INFRASTRUCTURE (run by hand on the GPU box: B=64 DT=bf16x3 python tests/fullsize_report.py): two full-size training steps
next to the torch-CPU oracle on the same rendered batches -- losses, per-tensor gradient and update deviations."""
import os, random, sys
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")): sys.path.insert(0, p)
import learner_oracle as lo
from artiboost_amd import registry as R
from artiboost_amd.assets import SceneAssets
from artiboost_amd.criterions import Criterion
from artiboost_amd.models import Arch
from artiboost_amd.optim import FusedClipAdam
from artiboost_amd.synth import ArtiBoostLoader
from artiboost_amd.train import TrainStep
B, size, lr, clip = int(os.environ.get("B", 64)), 256, 5e-5, 0.001
cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [size, size], [size // 8, size // 8]
arch = dict(cfg["ARCH"], COMPUTE_DTYPE=os.environ.get("DT", "bf16x3"), INIT_SEED=3)
model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
hb = model.model_list[0]
opt = FusedClipAdam(model.models_params, lr=lr, max_norm=clip, model=hb)
loader = ArtiBoostLoader.from_assets(SceneAssets("HO3D", seed=1), cfg["MANAGER"], cfg["DATA_PRESET"], B, 2 * B, compute_dtype=torch.float32, random_seed=3)
loader.prepare()
params0 = {k: v.clone() for k, v in hb.state_dict().items()}
static = loader.new_static_batch(); loader.load_batch(static, 0); model.train()
ts = TrainStep(model, crit, opt, static, use_graph=os.environ.get("EAGER") is None, renderer=loader); ts.static = static
leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in params0.items()}
names = ms = vs = None
for step in range(2):
    loader.load_batch(static, step)
    random.seed(100 + step); torch.manual_seed(100 + step)
    _, losses, _ = ts()
    got = {k: float(v) for k, v in zip(ts.fused.LOSS_KEYS, losses.float().cpu())}
    g_gpu = hb.store.reference_state_dict(grads=True)
    g_gpu = {k: v.cpu().clone() for k, v in g_gpu.items()}
    xpad = static["image_nhwc4_padded"].float().cpu()
    batch = {"image": xpad[:, 3:3 + size, 3:3 + size, :3].permute(0, 3, 1, 2).contiguous()}
    for k in ("root_joint", "cam_intr", "corners_can", "joints_3d", "corners_3d", "joints_vis", "corners_vis"):
        batch[k] = static[k].float().cpu()
    for v in leaf.values():
        if getattr(v, "grad", None) is not None: v.grad = None
    random.seed(100 + step); torch.manual_seed(100 + step)
    preds = lo.hybrid_forward(leaf, batch, [size, size], 22, 28, 0, training=True)
    total, ref, _ = lo.criterion(preds, batch)
    total.backward()
    print("step", step, {k: (round(got[k], 7), round(float(ref[k]), 7)) for k in got})
    worst = sorted(((float((g_gpu[k] - leaf[k].grad).norm() / (leaf[k].grad.norm() + 1e-30)), k) for k in leaf if getattr(leaf[k], "grad", None) is not None), reverse=True)[:6]
    print("  worst grad rel diffs:", [(round(a, 5), k) for a, k in worst])
    if names is None:
        names = [k for k, v in leaf.items() if v.dtype.is_floating_point and getattr(v, "grad", None) is not None]
        ms = [torch.zeros_like(leaf[k]) for k in names]; vs = [torch.zeros_like(leaf[k]) for k in names]
    rn = float(lo.clip_and_adam([leaf[k].detach() for k in names], [leaf[k].grad for k in names], ms, vs, step + 1, lr=lr, max_norm=clip))
    print("  grad norm gpu/ref:", float(opt.total_norm.cpu()), rn)
    sd = hb.state_dict()
    worst = sorted(((float(((sd[k] - params0[k]) - (leaf[k].detach() - params0[k])).norm() / ((leaf[k].detach() - params0[k]).norm() + 1e-30)), k) for k in names), reverse=True)[:8]
    print("  worst update rel diffs:", [(round(a, 4), k) for a, k in worst])
    for k in ("backbone.bn1.running_mean", "backbone.layer4.2.bn2.running_var"):
        print("  ", k, float((sd[k] - leaf[k]).abs().max()), float(leaf[k].abs().max()))

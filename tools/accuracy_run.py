#!/usr/bin/env python
"""The fixed-sample precision comparison -- the stand-in this environment allows for the north star's "HO3Dv2 MPJPE within 0.5 mm of the
reference checkpoint" (README.md:71-81, anakin/metrics/meanepe.py:13-101: dataset and checkpoint are downloads).

Every precision trains on the IDENTICAL sample sequence: the CCV mining update is frozen (sampling weights stay all-ones, step_eval is never
called), loader, criterion and weight seeds are the same, so the only difference between the runs is the arithmetic of the learner:
    f32     exact-f32 MFMA (the reference's own precision)
    bf16x3  split-bf16 x3 on the integer image plane "u8n" -- bench.py's headline configuration
    bf16    reduced precision (reported beside them; not a parity configuration)
At the listed steps the model is put in eval mode (running BatchNorm statistics) and measured on a HELD-OUT validation set: `--val` synthetic
samples of the val-mode CCV sampler (OVGSet.val(), ovg_set.py:108-118: uniform, no replacement), another seed, rendered ONCE and shown to every
precision as the same pixels.  Reported: Mean3DEPE of joints (MPJPE) and corners (MPCPE) in mm (meanepe.py:13-101).

    python tools/accuracy_run.py [--steps 3000] [--at 500,1000,2000,3000] [--val 2048] [--dtypes f32,bf16x3,bf16] [--out profiles/round6_accuracy.txt]"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def _val_set(cfg, bs, n, dev):
    """n held-out samples as (pixels uint8 [n, H, W, 3], ground-truth dict) -- rendered once, on the fp32 image path."""
    import torch
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.synth import ArtiBoostLoader
    vl = ArtiBoostLoader.from_assets(SceneAssets("HO3D", seed=1), dict(cfg["MANAGER"], EPOCH=1), cfg["DATA_PRESET"], bs, n, device=dev,
                                     compute_dtype=torch.float32, random_seed=977)
    vl.prepare(is_train=False)
    st = vl.new_static_batch()
    keys = ("root_joint", "cam_intr", "corners_can", "joints_3d", "corners_3d", "joints_vis", "corners_vis", "obj_idx", "obj_transf")
    pix, gts = [], []
    for bi in range(len(vl)):
        vl.load_batch(st, bi)
        vl.render_into(st)
        x = st["image_nhwc4_padded"][:, 3:-3, 3:-5, :3]
        pix.append(torch.round((x + 0.5) * 255.0).to(torch.uint8))
        gts.append({k: st[k].clone() for k in keys})
    return torch.cat(pix), gts, bs


def _val_batches(pix, gts, bs, plane, dtype):
    """The validation pixels as the padded NHWC4 tensors a model of this precision consumes (same uint8 pixels for all)."""
    import torch
    from artiboost_amd.registry import IMAGE_PLANE_KEY, PlaneTag, tag_image_plane
    out = []
    for i, gt in enumerate(gts):
        v = pix[i * bs:(i + 1) * bs].float()
        B, H, W, _ = v.shape
        pad = torch.zeros((B, H + 6, W + 8, 4), dtype=dtype, device=v.device)
        pad[:, 3:-3, 3:-5, :3] = (2.0 * v - 255.0).to(dtype) if plane == "u8n" else (v / torch.full((), 255.0, device=v.device) - 0.5).to(dtype)
        b = dict(gt)
        b["image_nhwc4_padded"] = tag_image_plane(pad, plane)
        b[IMAGE_PLANE_KEY] = PlaneTag(plane)
        out.append(b)
    return out


def run(dtype, steps, at, val, cfg, bs=64, size=256, dev="cuda:0", log=print, per_epoch=500, replica=0):
    import numpy as np
    import torch
    from artiboost_amd import registry as R
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.metrics import Mean3DEPE
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.synth import ArtiBoostLoader
    from artiboost_amd.train import TrainStep
    seed = cfg["TRAIN"]["MANUAL_SEED"]
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    arch = dict(cfg["ARCH"], COMPUTE_DTYPE=dtype, DEVICE=dev, INIT_SEED=seed)
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=cfg["TRAIN"]["LR"], max_norm=cfg["TRAIN"]["GRAD_CLIP"], model=hb)
    cd = "u8n" if dtype == "bf16x3" else hb.net.dtype
    per_epoch = min(per_epoch, steps)
    loader = ArtiBoostLoader.from_assets(SceneAssets("HO3D", seed=1), dict(cfg["MANAGER"], EPOCH=1000), cfg["DATA_PRESET"], bs, per_epoch * bs,
                                         device=dev, compute_dtype=cd, random_seed=seed)
    vb = _val_batches(*val, plane=loader.image_plane, dtype=loader.dtype)
    # replica r > 0: the same weights, samples and pixels; only the host RNG streams behind the ordinal losses' random pair / view draws
    # (criterions.py: random.shuffle, torch.rand -- reseeded per run by the reference's set_all_seeds too) start elsewhere.  What two runs
    # of ONE precision differ by is the noise floor any cross-precision difference has to be read against.
    random.seed(seed + 7919 * replica); torch.manual_seed(seed + 7919 * replica)
    metric = Mean3DEPE(VAL_KEYS=["joints_3d_abs", "corners_3d_abs"], MILLIMETERS=True)
    ts, step, res, t_train = None, 0, {}, 0.0
    w0 = loader.sample_weight_map.clone()
    while step < steps:
        loader.prepare()                                   # mining frozen: the weights this draws from never change
        assert torch.equal(loader.sample_weight_map, w0)
        model.train()
        if ts is None:
            static = loader.new_static_batch()
            loader.load_batch(static, 0)
            ts = TrainStep(model, crit, opt, static, use_graph=True, renderer=loader)
        torch.cuda.synchronize()
        t0 = time.time()
        for bi in range(len(loader)):
            ts.stage(loader, bi)
            ts()
            step += 1
            if step in at or step == steps:
                torch.cuda.synchronize()
                t_train += time.time() - t0
                loss = float(ts.out[1][5]) if ts.fused is not None else float("nan")
                model.eval()
                metric.reset()
                with torch.no_grad():
                    for b in vb:
                        metric.feed(model(b)["HybridBaseline"], b)
                m = metric.get_measures()
                res[step] = {"mpjpe_mm": m["joints_3d_abs_mepe"], "mpcpe_mm": m["corners_3d_abs_mepe"], "train_final_loss": loss}
                log(f"  {dtype:7s} r{replica} step {step:5d}: val MPJPE {m['joints_3d_abs_mepe']:8.3f} mm  MPCPE {m['corners_3d_abs_mepe']:8.3f} mm   train final_loss {loss:.4e}")
                model.train()
                t0 = time.time()
            if step >= steps:
                break
        torch.cuda.synchronize()
        t_train += time.time() - t0
    return {"dtype": dtype, "replica": replica, "image_plane": loader.image_plane, "checkpoints": res, "train_samples_per_s": round(steps * bs / t_train, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--at", default="500,1000,2000,3000")
    ap.add_argument("--val", type=int, default=2048)
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtypes", default="f32,bf16x3,bf16")
    ap.add_argument("--replicas", type=int, default=3)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [a.size, a.size], [a.size // 8, a.size // 8]
    cfg["ARCH"]["BACKBONE"]["PRETRAINED"] = False
    cfg["MANAGER"].pop("REFINER", None)
    at = sorted(int(x) for x in a.at.split(",") if x)
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    log(f"# fixed-sample precision comparison: {a.steps} steps at B = {a.bs}, {a.size} x {a.size}; mining frozen (identical samples for every precision); "
        f"held-out validation set of {a.val} synthetic val-mode CCV samples, eval-mode forward; Mean3DEPE in mm")
    val = _val_set(cfg, a.bs, a.val, "cuda:0")
    out = {}
    for dt in a.dtypes.split(","):
        out[dt] = [run(dt, a.steps, at, val, cfg, bs=a.bs, size=a.size, log=log, replica=r) for r in range(a.replicas)]
        log(f"  {dt:7s} trained at {out[dt][0]['train_samples_per_s']} samples/s (image plane {out[dt][0]['image_plane']})")
    import numpy as np
    log("# mean +- sample standard deviation over the replicas (same weights / samples / pixels; only the losses' random draws differ)")
    stat = {}
    for dt, runs in out.items():
        for s_ in sorted(runs[0]["checkpoints"]):
            for k in ("mpjpe_mm", "mpcpe_mm"):
                v = np.array([r["checkpoints"][s_][k] for r in runs])
                stat[(dt, s_, k)] = (float(v.mean()), float(v.std(ddof=1)) if len(v) > 1 else float("nan"), len(v))
            (mj, sj, n), (mc, sc, _) = stat[(dt, s_, "mpjpe_mm")], stat[(dt, s_, "mpcpe_mm")]
            log(f"  {dt:7s} step {s_:5d}: MPJPE {mj:8.3f} +- {sj:6.3f} mm   MPCPE {mc:8.3f} +- {sc:6.3f} mm   (n = {n})")
    if "f32" in out:
        log("# difference of the means to f32, with the standard error of that difference (sqrt(sd_a^2 / n + sd_b^2 / n))")
        for dt in out:
            if dt == "f32":
                continue
            for s_ in sorted(out[dt][0]["checkpoints"]):
                parts = []
                for k, name in (("mpjpe_mm", "MPJPE"), ("mpcpe_mm", "MPCPE")):
                    (ma, sa, n), (mb, sb, _) = stat[(dt, s_, k)], stat[("f32", s_, k)]
                    se = float(np.sqrt(sa * sa / n + sb * sb / n)) if n > 1 else float("nan")
                    parts.append(f"{name} {ma - mb:+7.3f} +- {se:5.3f} mm")
                log(f"  {dt:7s} - f32 at step {s_:5d}: " + "   ".join(parts))
    log(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

"""Timing of the learner half at the benchmark geometry, on a seeded synthetic batch:
   default : forward + loss + backward + clip/Adam (graph replay)            -- BASELINE.json configs[2] without the render
   --eval  : eval-mode forward only (conv stack + soft-argmax + pose assembly) -- BASELINE.json configs[1]"""
import argparse
import os
import sys
import time

import torch
import yaml

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))      # gen_batch: seeded inputs in the reference's batch schema

ap = argparse.ArgumentParser()
ap.add_argument("--bs", type=int, default=64)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--eval", action="store_true")
a = ap.parse_args()

from gen_batch import make_batch
from artiboost_amd import registry as R
from artiboost_amd.criterions import Criterion
from artiboost_amd.models import Arch
from artiboost_amd.optim import FusedClipAdam
from artiboost_amd.train import CAPTURE_MODE, TrainStep

cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [a.size, a.size], [a.size // 8, a.size // 8]
arch = dict(cfg["ARCH"], COMPUTE_DTYPE=a.dtype, INIT_SEED=1)
model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
hb = model.model_list[0]
batch = {k: v.cuda() for k, v in make_batch(a.bs, a.size, 5).items()}

if a.eval:
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            out = model(batch)
        step = lambda: model(batch)           # noqa: E731
        if a.graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                out = model(batch)
            step = g.replay
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / a.steps
    print(f"eval forward: {dt * 1e3:.2f} ms  ({a.bs / dt:.0f} samples/s)  dtype={a.dtype} bs={a.bs} size={a.size} graph={a.graph}")
    sys.exit(0)

opt = FusedClipAdam(model.models_params, lr=5e-5, max_norm=0.001, model=hb)
model.train()
ts = TrainStep(model, crit, opt, batch, use_graph=bool(a.graph))
for _ in range(3):
    ts()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.steps):
    ts()
t_host = time.time() - t0
torch.cuda.synchronize()
dt = (time.time() - t0) / a.steps
print(f"learner step: {dt * 1e3:.2f} ms  ({a.bs / dt:.0f} samples/s)  host-issue {t_host / a.steps * 1e3:.2f} ms  dtype={a.dtype} bs={a.bs} size={a.size}")

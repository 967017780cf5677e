"""Quick timing of the learner half (fwd + loss + bwd + clip/Adam) at the benchmark geometry."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))      # gen_batch: seeded inputs
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))

ap = argparse.ArgumentParser()
ap.add_argument("--bs", type=int, default=64)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--graph", type=int, default=1)
a = ap.parse_args()

from test_gpu_learner import build
from gen_batch import make_batch
from artiboost_amd.optim import FusedClipAdam

model, crit, _ = build(a.size, a.size // 8, a.dtype, 1)
hb = model.model_list[0]
opt = FusedClipAdam(model.models_params, lr=5e-5, max_norm=0.001, model=hb)
batch = {k: v.cuda() for k, v in make_batch(a.bs, a.size, 5).items()}
model.train()


from artiboost_amd.train import TrainStep
ts = TrainStep(model, crit, opt, batch, use_graph=bool(a.graph))


def step():
    ts()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.steps):
    step()
t_host = time.time() - t0
torch.cuda.synchronize()
dt = (time.time() - t0) / a.steps
print(f"learner step: {dt*1e3:.2f} ms  ({a.bs/dt:.0f} samples/s)  host-issue {t_host/a.steps*1e3:.2f} ms  dtype={a.dtype} bs={a.bs} size={a.size}")

"""Which aten kernels does one training step launch (they end up as hipGraph nodes), and from which Python lines?
Runs the step eagerly under torch.profiler with stacks."""
import argparse
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench

a = argparse.Namespace(bs=64, size=256, dtype="bf16", dataset="HO3D", eager=True, pipeline=False, pipeline_opt=False, steps=4, warmup=1)
cfg, model, crit, opt, loader, ts, static = bench.build_everything(a, 0, 1, "cuda:0")
for i in range(2):
    ts.stage(loader, i)
    ts()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.stage(loader, 2)
    ts()
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if not e.name.startswith("aten::") or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    st = [f for f in (e.stack or []) if "artiboost_amd" in f or "bench.py" in f]
    key = (e.name, st[0] if st else "?")
    rows[key] = rows.get(key, 0) + 1
for (name, where), n in sorted(rows.items(), key=lambda kv: kv[0][1]):
    print(f"{n:3d}  {name:28s} {where}")

"""A/B of the mixed real + synthetic training loop with the batch assembly on the training thread vs on a worker thread (realdata.ThreadedPrefetcher):
ms per step, and the per-step final_loss of the first steps of both (the same batches in the same order -> the same numbers).
    python tools/mixed_thread_ab.py [jpeg|png]"""
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_mixed  # noqa: E402

cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
cfg["DATA_PRESET"]["IMAGE_SIZE"] = [256, 256]; cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [32, 32]
modes = ("same stream, frames of 4 batches decoded per call one group ahead on a side stream",
         "worker thread two batches ahead, frames of 4 batches decoded per call one group ahead on a side stream")
src = sys.argv[1] if len(sys.argv) > 1 else "jpeg"
for rep in range(2):
    bench_mixed.train_loop(cfg, steps=40, modes=modes, source=src)

"""A/B of the mixed step over .jpg / .png sources (tools/bench_mixed.train_loop): python tools/mixed_png_ab.py [source ...]
env: AB_DECODE_WORKERS, AB_SWITCH_INTERVAL (sys.setswitchinterval)"""
import os
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_mixed   # noqa: E402

if os.environ.get("AB_SWITCH_INTERVAL"):
    sys.setswitchinterval(float(os.environ["AB_SWITCH_INTERVAL"]))
cfg = yaml.safe_load(open(os.path.join(bench_mixed.ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
cfg["DATA_PRESET"]["IMAGE_SIZE"] = [256, 256]
ahead = ("same stream, frames of 4 batches decoded per call one group ahead on a side stream",)
for src in (sys.argv[1:] or ["jpeg", "png", "png-pillow", "png"]):
    r = bench_mixed.train_loop(cfg, steps=40, modes=ahead, source=src, quiet=True)
    print(f"{src:12s} {r[ahead[0]]:.2f} ms per step  (workers {os.environ.get('AB_DECODE_WORKERS', 'default')}, switch interval {sys.getswitchinterval()})")

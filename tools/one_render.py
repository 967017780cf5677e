"""A few batched renders at the benchmark geometry (for rocprofv3 / PMC runs)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd.assets import SceneAssets
from artiboost_amd.synth import ArtiBoostLoader
import yaml
root = os.path.join(os.path.dirname(__file__), "..")
cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
cfg["DATA_PRESET"]["IMAGE_SIZE"] = [256, 256]; cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [32, 32]
assets = SceneAssets("HO3D", seed=1)
mgr = dict(cfg["MANAGER"], EPOCH=1)
loader = ArtiBoostLoader.from_assets(assets, mgr, cfg["DATA_PRESET"], 64, 256, device="cuda", compute_dtype=torch.bfloat16, random_seed=1)
loader.prepare()
st = loader.new_static_batch()
for i in range(4):
    loader.load_batch(st, i)
    loader.render_into(st)
torch.cuda.synchronize()

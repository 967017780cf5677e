"""Per-batch cost of the real-data half (realdata.RealBatcher) at HO3D frame size, frames already decoded in host memory:
host GT assembly + upload + ab_augment_batch, and the mixed real/synthetic batch of MixedLoader."""
import os
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from artiboost_amd.assets import SceneAssets
from artiboost_amd.realdata import HOdataSource, MixedLoader, RealBatcher
from artiboost_amd.synth import ArtiBoostLoader


class MemorySource(HOdataSource):
    """Stand-in for a decoded HO3D split: n frames of 640x480 with seeded annotations (the dataset is a download)."""
    raw_size = (640, 480)

    def __init__(self, n=256, seed=0):
        rng = np.random.default_rng(seed)
        self.frames = rng.integers(0, 256, (16, 480, 640, 3), dtype=np.uint8)
        self.n = n
        K = np.array([[615.0, 0, 320.0], [0, 615.0, 240.0], [0, 0, 1.0]], np.float32)
        self.ann = []
        for i in range(n):
            j3 = (rng.uniform(-0.06, 0.06, (21, 3)) + [0.0, 0.0, 0.55]).astype(np.float32)
            can = (rng.uniform(0.03, 0.06, 3) * np.array([[a, b, c] for a in (-1, 1) for b in (-1, 1) for c in (-1, 1)])).astype(np.float32)
            T = np.eye(4, dtype=np.float32); T[:3, 3] = [0.02, -0.01, 0.57]
            c3 = can + T[:3, 3]
            proj = lambda p: ((K @ p.T).T[:, :2] / (K @ p.T).T[:, 2:]).astype(np.float32)     # noqa: E731
            j2, c2 = proj(j3), proj(c3)
            all2d = np.concatenate([j2[:1], c2])
            mn, mx = all2d.min(0), all2d.max(0)
            self.ann.append(dict(cam_intr=K, joints_3d=j3, joints_2d=j2, corners_3d=c3, corners_2d=c2, corners_can=can, obj_transf=T,
                                 obj_idx=1 + i % 4, side="right", bbox_center=np.array([int((mx[0] + mn[0]) / 2), int((mx[1] + mn[1]) / 2)]),
                                 bbox_scale=float(max(mx - mn))))

    def __len__(self):
        return self.n

    def get_image(self, idx):
        return self.frames[idx % 16]

    def get_annots(self, idx):
        return self.ann[idx]


def main():
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"] = [256, 256]
    src = MemorySource(n=1024)
    B = 64
    synth_len = int(0.6 * len(src))
    n_synth = MixedLoader.n_synth_for(B, len(src), synth_len)
    synth = ArtiBoostLoader.from_assets(SceneAssets("HO3D", seed=1), cfg["MANAGER"], cfg["DATA_PRESET"], n_synth, synth_len, compute_dtype=torch.bfloat16)
    synth.prepare()
    real = RealBatcher(src, cfg["DATA_PRESET"], compute_dtype=torch.bfloat16)
    ml = MixedLoader(real, synth, B)
    print(f"batch {B} = {ml.n_real} real + {ml.n_synth} synthetic")
    idxs = list(range(ml.n_real))
    for name, fn in (("host assemble (GT + frames into one array)", lambda: real.assemble(idxs)),
                     ("assemble + upload + ab_augment_batch", lambda: real.batch(idxs)),):
        fn(); torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(f"{name}: {(time.time() - t0) / 10 * 1e3:.2f} ms per batch of {ml.n_real}")
    it = iter(ml); next(it); torch.cuda.synchronize(); t0 = time.time(); n = 0
    for _ in it:
        n += 1
    torch.cuda.synchronize()
    print(f"MixedLoader: {(time.time() - t0) / max(n, 1) * 1e3:.2f} ms per mixed batch ({n} batches)")


if __name__ == "__main__":
    main()

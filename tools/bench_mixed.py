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


class JpegFileSource(MemorySource):
    """The same split as .jpg FILES in memory (what a dataset's disk cache holds): get_image decodes with Pillow on the host (the reference's
    path), get_image_bytes hands the file to the device decoder."""

    def __init__(self, n=256, seed=0, device_decode=True):
        import io
        from PIL import Image
        from bench_jpeg import _photo
        super().__init__(n, seed)
        self.files = []
        for i in range(16):
            b = io.BytesIO()
            Image.fromarray(_photo(640, 480, i)).save(b, "JPEG", quality=92, subsampling=2)
            self.files.append(b.getvalue())
        if not device_decode:
            self.get_image_bytes = None

    def get_image(self, idx):
        import io
        from PIL import Image
        return np.asarray(Image.open(io.BytesIO(self.files[idx % 16])).convert("RGB"))

    def get_image_bytes(self, idx):
        return self.files[idx % 16]


class PngFileSource(JpegFileSource):
    """The same frames as .png FILES (HO3D v2's own format, ho3d.py:181): zlib inflate on the host pool, scanline reconstruction on the device."""

    def __init__(self, n=256, seed=0, device_decode=True):
        import io
        from PIL import Image
        from bench_jpeg import _photo
        MemorySource.__init__(self, n, seed)
        self.files = []
        for i in range(16):
            b = io.BytesIO()
            Image.fromarray(_photo(640, 480, i)).save(b, "PNG")
            self.files.append(b.getvalue())
        if not device_decode:
            self.get_image_bytes = None


def jpeg_compare(cfg):
    """40 real frames per batch from .jpg files: decode on the device vs Pillow on this host (one thread, as one DataLoader worker)."""
    for dev_dec in (True, False):
        src = JpegFileSource(n=1024, device_decode=dev_dec)
        real = RealBatcher(src, cfg["DATA_PRESET"], compute_dtype=torch.bfloat16)
        idxs = list(range(40))
        real.batch(idxs); torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10):
            real.batch(idxs)
        torch.cuda.synchronize()
        print(f".jpg files, decode on the {'device (ab_jpeg_decode_batch)' if dev_dec else 'host (Pillow, 1 thread)'}: "
              f"{(time.time() - t0) / 10 * 1e3:.2f} ms per batch of 40 real frames (GT assembly + decode + ab_augment_batch)")


def train_loop(cfg, steps=30, modes=("same stream", "same stream, frames of 4 batches decoded per call",
                                     "same stream, frames of 4 batches decoded per call one group ahead on a side stream", "side stream, one batch ahead",
                                     "worker thread two batches ahead, frames of 4 batches decoded per call one group ahead on a side stream"),
               quiet=False, source="jpeg"):
    """The training step (hipGraph replay, bf16x3, B = 64, 256 x 256) over MixedLoader batches -- 40 real frames served as .jpg files and
    decoded on the device + 24 synthetic samples rendered per batch -- with the batch assembly on the step's own stream, and one batch
    ahead on a side stream (realdata.StreamPrefetcher)."""
    import random
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.realdata import StreamPrefetcher, ThreadedPrefetcher
    from artiboost_amd.train import TrainStep
    B = 64
    res = {}
    for mode in modes:
        random.seed(5); torch.manual_seed(5); np.random.seed(5)
        ahead = "one group ahead" in mode
        src = {"jpeg": JpegFileSource, "png": PngFileSource, "png-pillow": lambda n: PngFileSource(n, device_decode=False)}[source](n=4096)
        synth_len = int(0.6 * len(src))
        n_synth = MixedLoader.n_synth_for(B, len(src), synth_len)
        # both halves write the stem's integer image plane (AB_DT_U8N; AB_IMAGE_PLANE=f32: the fp32 image + split pass of round 4)
        cdt = "u8n" if os.environ.get("AB_IMAGE_PLANE", "u8n") == "u8n" else torch.float32
        synth = ArtiBoostLoader.from_assets(SceneAssets("HO3D", seed=1), cfg["MANAGER"], cfg["DATA_PRESET"], n_synth, synth_len, compute_dtype=cdt)
        synth.prepare()
        # as train/train_artiboost.py builds it: no float CHW copy of the frames (TrainStep reads the NHWC4 tensor), image tensors from a ring
        ml = MixedLoader(RealBatcher(src, cfg["DATA_PRESET"], compute_dtype=cdt), synth, B, decode_group=4 if "4 batches" in mode else 1,
                         decode_ahead=ahead, want_chw=os.environ.get("AB_MIXED_CHW", "0") == "1", reuse_buffers=int(os.environ.get("AB_MIXED_RING", "4")))
        arch = dict(cfg["ARCH"], COMPUTE_DTYPE="bf16x3", INIT_SEED=3)
        model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
        crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
        hb = model.model_list[0]
        opt = FusedClipAdam(model.models_params, lr=5e-5, max_norm=0.001, model=hb)
        model.train()
        loader = StreamPrefetcher(ml) if mode.startswith("side") else ThreadedPrefetcher(ml) if mode.startswith("worker thread") else ml
        it = iter(loader)
        first = next(it)
        ts = TrainStep(model, crit, opt, {k: v.clone() for k, v in first.items()}, use_graph=True, renderer=None)
        for _ in range(3):
            ts(next(it))
        torch.cuda.synchronize()
        t0 = time.time()
        n = 0
        trail = []
        for b in it:
            _, losses, _ = ts(b)
            n += 1
            if n <= 6:
                trail.append(losses[5].clone())
            if n == steps:
                break
        torch.cuda.synchronize()
        res[mode] = (time.time() - t0) / n * 1e3
        res["final_loss"] = float(losses[5])
        res["first_losses"] = [round(float(t), 8) for t in trail]
        if not quiet:
            print("   first losses:", res["first_losses"])
        if not quiet:
            print(f"training over mixed batches (40 real {source} frames + 24 synthetic, bf16x3), batch assembly on the {mode}: "
                  f"{res[mode]:.2f} ms per step ({n} steps), final loss {float(losses[5]):.5f}")
    return res


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
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    jpeg_compare(cfg)
    train_loop(cfg)
    ahead = ("same stream, frames of 4 batches decoded per call one group ahead on a side stream",)
    train_loop(cfg, modes=ahead, source="png")             # HO3D v2's own frame format: pooled inflate + device reconstruction
    train_loop(cfg, modes=ahead, source="png-pillow")      # ... and the same files through Pillow on the decode pool (no file bytes served)


if __name__ == "__main__":
    main()

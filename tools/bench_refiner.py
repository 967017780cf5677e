"""Timing of the grasp refiner at the reference's pose-generation batch (B = 256, 3 iterations, 10 000 object points):
whole forward, and the nearest-point kernel alone (point pairs / s; 8 flops per pair)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def random_state_dict(seed):
    """Seeded stand-in for GrabNet's refinenet.pt (a download): the _RefineNet key names and shapes (refiner.py:229-241)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def bn(p, n):
        sd.update({p + ".weight": 0.5 + torch.rand(n, generator=g), p + ".bias": 0.1 * torch.randn(n, generator=g),
                   p + ".running_mean": 0.05 * torch.randn(n, generator=g), p + ".running_var": 0.5 + torch.rand(n, generator=g)})

    def lin(p, n, k, gain=1.0):
        sd.update({p + ".weight": gain * torch.randn(n, k, generator=g) / k ** 0.5, p + ".bias": 0.01 * torch.randn(n, generator=g)})

    bn("bn1", 778)
    for name, fin in (("rb1", 877), ("rb2", 1389), ("rb3", 1389)):
        lin(name + ".fc1", 256, fin); bn(name + ".bn1", 256); lin(name + ".fc2", 512, 256); bn(name + ".bn2", 512); lin(name + ".fc3", 512, fin)
    lin("out_p", 96, 512, 0.002); lin("out_t", 3, 512, 0.002)
    return sd


def main():
    from artiboost_amd.assets import SceneAssets, resample_objects
    from artiboost_amd.refiner import Refiner, nearest_dist
    from artiboost_amd.synth import ManoLayerHIP
    B = int(os.environ.get("BS", 256))
    assets = SceneAssets("HO3D", seed=1)
    pts = resample_objects(assets, 10000, seed=7)
    ref = Refiner.build("hand_obj", {"PRETRAINED": "", "ITERS": 3, "ALLOW_RANDOM_INIT": True}, ManoLayerHIP(assets.hand))
    ref.load_state_dict(random_state_dict(4))
    ref.setup(pts)
    rng = np.random.default_rng(1)
    pose = np.clip(0.3 * rng.standard_normal((B, 48)), -1.2, 1.2).astype(np.float32)
    tsl = rng.uniform(-0.05, 0.05, (B, 3)).astype(np.float32)
    q = rng.standard_normal((B, 3, 3))
    rot = np.linalg.qr(q)[0].astype(np.float32)
    oi = rng.integers(0, assets.n_obj, B)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    inp = {"hand_pose": t(pose), "hand_tsl": t(tsl), "obj_rot": t(rot)}
    oid = t(oi.astype(np.int64))

    def timed(fn, n=10):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    ms = timed(lambda: ref(inp, oid))
    x = torch.randn(B, 778, 3, device="cuda") * 0.05
    ms_nn = timed(lambda: nearest_dist(x, ref.resampled, oid, inp["obj_rot"]))
    pairs = B * 778 * 10000
    print(f"refiner forward B={B}: {ms:.3f} ms ({B / ms * 1e3:.0f} grasps/s); nearest_dist: {ms_nn:.3f} ms, "
          f"{pairs / ms_nn / 1e6:.1f} Gpairs/s, {8 * pairs / ms_nn / 1e9:.2f} TFLOP/s fp32")


if __name__ == "__main__":
    main()

"""Timing of the grasp refiner at the reference's pose-generation batch (B = 256, 3 iterations, 10 000 object points):
whole forward, and the nearest-point kernel alone (point pairs / s; 8 flops per pair)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))


def main():
    import refiner_oracle as rfo
    from artiboost_amd.assets import SceneAssets, resample_objects
    from artiboost_amd.refiner import Refiner, nearest_dist
    from artiboost_amd.synth import ManoLayerHIP
    B = int(os.environ.get("BS", 256))
    assets = SceneAssets("HO3D", seed=1)
    pts = resample_objects(assets, 10000, seed=7)
    ref = Refiner.build("hand_obj", {"PRETRAINED": "", "ITERS": 3, "ALLOW_RANDOM_INIT": True}, ManoLayerHIP(assets.hand))
    ref.load_state_dict(rfo.fill_params(4))
    ref.setup(pts)
    pose, tsl, rot, oi = rfo.make_inputs(assets, B, 1)
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

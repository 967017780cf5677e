"""TEST INFRASTRUCTURE ONLY -- golden vectors of the SIGMOID heat-map head from the REAL reference (run in the build container;
output committed as tests/golden/head_sigmoid.npz): anakin/models/simplebaseline.py norm_heatmap("sigmoid") (:16-40), the confidence
/ renormalisation of IntegralDeconvHead.forward (:183-189) and integral_heatmap3d (:43-71), forward and the gradient wrt the logits
(through uvd only: the HIP head does not differentiate the sigmoid head's confidence)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import  # noqa: E402


def main(seed=4):
    ref_import.load()
    from anakin.models.simplebaseline import norm_heatmap, integral_heatmap3d
    out = {}
    for tag, (C, D, H, W) in {"g256": (22, 28, 32, 32), "tiny": (3, 4, 5, 6), "odd": (5, 9, 7, 3)}.items():
        g = torch.Generator().manual_seed(seed)
        B = 2
        logits = (3.0 * torch.randn(B, C * D, H, W, generator=g)).requires_grad_(True)
        x = logits.reshape(B, C, -1)
        x = norm_heatmap("sigmoid", x)
        conf = torch.max(x, dim=-1).values
        x = x / (x.sum(dim=-1, keepdim=True) + 1e-7)
        x = x.contiguous().view(B, C, D, H, W)
        uvd = integral_heatmap3d(x)
        gu = torch.randn(uvd.shape, generator=g)
        (uvd * gu).sum().backward()
        out[f"{tag}.seed"] = np.array([seed, B, C, D, H, W])
        out[f"{tag}.uvd"] = uvd.detach().numpy().copy()
        out[f"{tag}.conf"] = conf.detach().numpy().copy()
        out[f"{tag}.g_uvd"] = gu.numpy().copy()
        dl = logits.grad
        out[f"{tag}.dlogits.sample"] = dl.reshape(B, C, -1)[:, :, ::53].numpy().copy()
        out[f"{tag}.dlogits.abs_sum"] = dl.abs().sum().numpy().copy()
    path = os.path.join(ROOT, "tests", "golden", "head_sigmoid.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()

"""TEST INFRASTRUCTURE ONLY -- golden vectors of the reference's SimpleBaseline (anakin/models/simplebaseline.py:194-241: ResNet backbone +
IntegralDeconvHead with 29 heat-map classes, no box head) run in the build container (/root/reference via oracle/ref_import.py); output
committed as tests/golden/simplebaseline.npz.  Backbone ResNet18 (BasicBlock, resnet.py:236-241) so that the second BasicBlock stage
count is pinned too.  Weights: oracle/learner_oracle.fill_params on the reference's own state-dict key list (seeded), loaded strictly."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import learner_oracle as lo  # noqa: E402
import ref_import  # noqa: E402
from gen_batch import make_batch  # noqa: E402


def main(seed=6, size=128, B=2, depth=28):
    ref_import.load()
    import anakin.models as rm
    import anakin.models.resnet as rresnet
    import anakin.models.simplebaseline as rsb
    rm.ResNet18, rm.ResNet34, rm.IntegralDeconvHead, rm.SimpleBaseline = rresnet.ResNet18, rresnet.ResNet34, rsb.IntegralDeconvHead, rsb.SimpleBaseline
    heat = size // 8
    cfg = {"TYPE": "SimpleBaseline", "PRETRAINED": "", "BACKBONE": {"TYPE": "ResNet18", "PRETRAINED": False, "FREEZE_BATCHNORM": False},
           "HEAD": {"TYPE": "IntegralDeconvHead", "NCLASSES": 29, "DECONV_WITH_BIAS": False, "NORM_TYPE": "softmax", "INPUT_CHANNEL": 512,
                    "DEPTH_RESOLUTION": depth, "NUM_DECONV_LAYERS": 2, "NUM_DECONV_FILTERS": [256, 256], "NUM_DECONV_KERNELS": [4, 4],
                    "FINAL_CONV_KERNEL": 1},
           "DATA_PRESET": {"IMAGE_SIZE": [size, size], "HEATMAP_SIZE": [heat, heat], "CENTER_IDX": 0}}
    torch.manual_seed(seed)
    model = rsb.SimpleBaseline(**cfg)
    keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    want = lo.param_shapes(29, depth, layers=(2, 2, 2, 2), head_prefix="pose_head", box_head=False)
    assert keys == [(k, tuple(s)) for k, s in want], [a for a, b in zip(keys, want) if a != (b[0], tuple(b[1]))][:3]
    params = lo.fill_params(want, seed=seed)
    model.load_state_dict(params, strict=True)
    batch = make_batch(B, size, seed + 100)
    batch["corners_3d"] = 0.08 * torch.randn(B, 8, 3, generator=torch.Generator().manual_seed(seed))
    out = {"meta": np.array([size, heat, depth, B, seed])}
    model.eval()
    with torch.no_grad():
        pe = model(batch)
    for k, v in pe.items():
        out[f"eval.pred.{k}"] = v.numpy().copy()
    model.train()
    pt = model(batch)
    for k, v in pt.items():
        out[f"train.pred.{k}"] = v.detach().numpy().copy()
    # JointsLoss (jointloss.py:25-67) with lambda 1.0 / 0.2 on the absolute joints / corners, vis-masked
    tj = batch["joints_3d"] + batch["root_joint"][:, None]
    tc = batch["corners_3d"] + batch["root_joint"][:, None]
    lj = torch.nn.functional.mse_loss(pt["joints_3d_abs"] * batch["joints_vis"][..., None], tj * batch["joints_vis"][..., None])
    lc = torch.nn.functional.mse_loss(pt["corners_3d_abs"] * batch["corners_vis"][..., None], tc * batch["corners_vis"][..., None])
    total = 1.0 * lj + 0.2 * lc
    total.backward()
    out["loss.total"] = total.detach().numpy().copy()
    named = dict(model.named_parameters())
    names = sorted(named)
    out["grad.names"] = np.array(names)
    out["grad.norms"] = np.array([float(named[n].grad.norm()) if named[n].grad is not None else 0.0 for n in names])
    out["corners_3d"] = batch["corners_3d"].numpy().copy()
    path = os.path.join(ROOT, "tests", "golden", "simplebaseline.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()

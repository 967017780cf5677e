"""TEST INFRASTRUCTURE ONLY -- golden vectors of the reference's HybridBaseline on a Bottleneck backbone (ResNet50, anakin/models/resnet.py:
104-141,252-258; head INPUT_CHANNEL 2048, MLP_O LAYERS_N [2048, 256, 128]) run in the build container (/root/reference via oracle/ref_import.py);
output committed as tests/golden/resnet50_hybrid.npz.  Weights: lo.fill_params on the reference's own state-dict key list, loaded strictly."""
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


def main(seed=8, size=128, B=2, depth=28):
    ref_import.load()
    import anakin.models as rm
    import anakin.models.resnet as rresnet
    import anakin.models.simplebaseline as rsb
    import anakin.models.mlp as rmlp
    import anakin.models.hybridbaseline as rhb
    rm.ResNet50, rm.IntegralDeconvHead, rm.MLP_O = rresnet.ResNet50, rsb.IntegralDeconvHead, rmlp.MLP_O
    heat = size // 8
    cfg = {"TYPE": "HybridBaseline", "PRETRAINED": "", "BACKBONE": {"TYPE": "ResNet50", "PRETRAINED": False, "FREEZE_BATCHNORM": False},
           "HYBRID_HEAD": {"TYPE": "IntegralDeconvHead", "NCLASSES": 22, "DECONV_WITH_BIAS": False, "NORM_TYPE": "softmax", "INPUT_CHANNEL": 2048,
                           "DEPTH_RESOLUTION": depth, "NUM_DECONV_LAYERS": 2, "NUM_DECONV_FILTERS": [256, 256], "NUM_DECONV_KERNELS": [4, 4],
                           "FINAL_CONV_KERNEL": 1},
           "BOX_HEAD": {"TYPE": "MLP_O", "LAYERS_N": [2048, 256, 128], "OUT_CHANNEL": 6},
           "DATA_PRESET": {"IMAGE_SIZE": [size, size], "HEATMAP_SIZE": [heat, heat], "CENTER_IDX": 0}}
    torch.manual_seed(seed)
    model = rhb.HybridBaseline(**cfg)
    keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    want = lo.param_shapes(22, depth, bottleneck=True)
    assert keys == [(k, tuple(s)) for k, s in want], [a for a, b in zip(keys, want) if a != (b[0], tuple(b[1]))][:3]
    params = lo.fill_params(want, seed=seed)
    model.load_state_dict(params, strict=True)
    batch = make_batch(B, size, seed + 100)
    out = {"meta": np.array([size, heat, depth, B, seed])}
    model.eval()
    with torch.no_grad():
        pe = model(batch)
    for k in ("joints_3d_abs", "corners_3d_abs", "2d_uvd"):
        out[f"eval.pred.{k}"] = pe[k].numpy().copy()
    model.train()
    pt = model(batch)
    for k in ("joints_3d_abs", "corners_3d_abs", "2d_uvd", "box_rot_rotmat"):
        out[f"train.pred.{k}"] = pt[k].detach().numpy().copy()
    total, _ = lo.joints_loss(pt, batch)            # JointsLoss arithmetic (jointloss.py:25-67; the oracle's is pinned by learner_*.npz)
    total.backward()
    out["loss.total"] = total.detach().numpy().copy()
    named = dict(model.named_parameters())
    names = sorted(named)
    out["grad.names"] = np.array(names)
    out["grad.norms"] = np.array([float(named[n].grad.norm()) if named[n].grad is not None else 0.0 for n in names])
    path = os.path.join(ROOT, "tests", "golden", "resnet50_hybrid.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()

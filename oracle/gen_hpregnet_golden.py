"""TEST INFRASTRUCTURE ONLY -- golden vectors for the regression-based model of BASELINE.json configs[0] from the REAL reference
(/root/reference, imported via oracle/ref_import.py; run in the build container, output committed as tests/golden/hpregnet.npz).

What the reference can run here: its ResNet18 backbone (anakin/models/resnet.py:142-236), HOPRegNet.TransHead and
HOPRegNet.recover_object (anakin/models/hpregnet.py:51-70,112-147) -- plain torch.  Its MANO branch needs manotorch (absent):
the MANO arithmetic is pinned separately against the reference's in-tree MANO layer (tests/golden/mano.npz).

Weights: the build's own modules are initialised from a seed and their state_dict is loaded into the reference modules (same
keys), so the test can rebuild them without shipping weights."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import  # noqa: E402


def build_ours(seed):
    """Build this repo's ResNet18 + TransHead with seeded weights; returns their state_dicts (CPU)."""
    # the repo package is imported under its real name BEFORE the reference is put on sys.path
    sys.path.insert(0, ROOT)
    from artiboost_amd import hpregnet
    torch.manual_seed(seed)
    net = hpregnet.ResNet18(PRETRAINED=False, FREEZE_BATCHNORM=False)
    for m in net.modules():                              # non-trivial BatchNorm statistics (eval mode uses them)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    head = hpregnet.HOPRegNet.TransHead(512, 9)
    return net.state_dict(), head.state_dict()


def main():
    seed = 5
    sd_net, sd_head = build_ours(seed)
    import transformers  # noqa: F401  (anakin/utils/netutils.py imports it; must be resolved before torchvision is stubbed)
    ref_import.load_control_plane()
    import anakin.models.resnet as rresnet
    import anakin.models.hpregnet as rh
    net = rresnet.ResNet18(PRETRAINED=False, FREEZE_BATCHNORM=False)
    net.load_state_dict(sd_net, strict=True)
    head = rh.HOPRegNet.TransHead(inp_dim=512, out_dim=9)
    head.load_state_dict(sd_head, strict=True)
    net.eval(); head.eval()
    g = torch.Generator().manual_seed(seed + 1)
    B = 2
    image = torch.rand((B, 3, 96, 96), generator=g) - 0.5      # small on purpose: the fixture stores it
    samples = {"cam_intr": torch.tensor([[[617.0, 0, 112.0], [0, 617.0, 112.0], [0, 0, 1.0]]]).repeat(B, 1, 1),
               "root_joint": torch.tensor([[0.01, -0.02, 0.55], [-0.03, 0.04, 0.62]]),
               "corners_can": 0.05 * (torch.rand((B, 8, 3), generator=g) * 2 - 1)}
    with torch.no_grad():
        feats = net(image=image)
        self_like = types.SimpleNamespace(obj_transfhead=head, proj2d_func=rh.batch_persp_proj2d)
        obj = rh.HOPRegNet.recover_object(self_like, feats["res_layer4_mean"], samples)
    out = {"seed": np.int64(seed), "image": image.numpy(), "res_layer4_mean": feats["res_layer4_mean"].numpy(),
           "res_layer1_sample": feats["res_layer1"][:, ::8, ::5, ::5].numpy()}
    out.update({"sample." + k: v.numpy() for k, v in samples.items()})
    out.update({"obj." + k: v.numpy() for k, v in obj.items()})
    path = os.path.join(ROOT, "tests", "golden", "hpregnet.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


if __name__ == "__main__":
    main()

"""TEST INFRASTRUCTURE ONLY -- the state_dict KEY SET (names + shapes) of the reference's HOPRegNet, written to
tests/golden/hpregnet_keys.json (run in the build container; /root/reference imported via oracle/ref_import.py).

The reference's regbased checkpoint is loaded with strict=True (anakin/models/hpregnet.py:59-64), so the build's HOPRegNet must
own exactly these keys, apart from `mano_branch.mano_layer.*`: those are the asset buffers manotorch's ManoLayer persists
(manotorch is an absent, un-pinned git dependency -- requirements.txt:178).  The stand-in ManoLayer below registers the `th_*`
names the reference itself reads (mano.py:24 `_buffers["th_J_regressor"]`, mano.py:107 `th_faces`) plus the other MANO tables;
the test only requires that ANY key under that prefix is dropped on load, so the stand-in's exact list is not load-bearing."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_import  # noqa: E402


def main():
    import transformers  # noqa: F401
    ref_import.load_control_plane()

    class ManoLayer(torch.nn.Module):
        def __init__(self, ncomps=15, center_idx=None, side="right", mano_assets_root=None, use_pca=True, flat_hand_mean=False, **kw):
            super().__init__()
            for k, shp in (("th_betas", (1, 10)), ("th_shapedirs", (778, 3, 10)), ("th_posedirs", (778, 3, 135)), ("th_v_template", (1, 778, 3)),
                           ("th_J_regressor", (16, 778)), ("th_weights", (778, 16)), ("th_hands_mean", (1, 45)), ("th_comps", (45, 45)),
                           ("th_selected_comps", (ncomps, 45))):
                self.register_buffer(k, torch.zeros(shp))
            self.register_buffer("th_faces", torch.zeros((1538, 3), dtype=torch.long))

    sys.modules["manotorch.manolayer"].ManoLayer = ManoLayer
    import anakin.models.resnet  # noqa: F401  (registers ResNet18/34)
    import anakin.models.mano as rmano
    rmano.ManoLayer = ManoLayer
    import anakin.models.hpregnet as rh
    import anakin.models as rm
    rm.ResNet34 = anakin.models.resnet.ResNet34
    rm.ResNet18 = anakin.models.resnet.ResNet18
    rm.ManoBranch = rmano.ManoBranch
    out = {}
    for bb in ("ResNet18", "ResNet34"):
        cfg = {"TYPE": "HOPRegNet", "PRETRAINED": "", "BACKBONE": {"TYPE": bb, "PRETRAINED": False, "FREEZE_BATCHNORM": False},
               "HEAD": {"TYPE": "ManoBranch", "INPUT_DIM": 512, "NCOMPS": 15, "USE_PCA": True, "USE_SHAPE": True,
                        "MANO_ASSETS_ROOT": "assets/mano_v1_2"},
               "DATA_PRESET": {"IMAGE_SIZE": [224, 224], "CENTER_IDX": 9}}
        net = rh.HOPRegNet(**cfg)
        out[bb] = {k: list(v.shape) for k, v in net.state_dict().items()}
    path = os.path.join(ROOT, "tests", "golden", "hpregnet_keys.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()

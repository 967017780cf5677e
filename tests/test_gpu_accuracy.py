"""The precision clause of the north star ("HO3Dv2 MPJPE within 0.5 mm of the reference checkpoint") in the only form this environment allows
(dataset and checkpoint are downloads): bf16x3 on the integer image plane -- bench.py's configuration -- against the exact-f32 learner on the
IDENTICAL sample sequence (mining frozen), measured on a held-out synthetic validation set.  tools/accuracy_run.py is the 3 000-step version
(profiles/round6_accuracy.txt); this is its 300-step form."""
import os
import sys

import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16x3_validation_mpjpe_tracks_the_f32_learner():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import accuracy_run as A
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [256, 256], [32, 32]
    cfg["ARCH"]["BACKBONE"]["PRETRAINED"] = False
    cfg["MANAGER"].pop("REFINER", None)
    val = A._val_set(cfg, 64, 512, "cuda:0")
    res = {dt: A.run(dt, 300, [150, 300], val, cfg, per_epoch=300, log=lambda s: None) for dt in ("f32", "bf16x3")}
    assert res["bf16x3"]["image_plane"] == "u8n"
    f, x = res["f32"]["checkpoints"], res["bf16x3"]["checkpoints"]
    assert f[300]["mpjpe_mm"] < f[150]["mpjpe_mm"] * 1.02 and f[300]["mpjpe_mm"] < 130.0          # it learns (random init: ~100+ mm)
    for step in (150, 300):
        for k in ("mpjpe_mm", "mpcpe_mm"):
            assert abs(x[step][k] - f[step][k]) <= 1.0, (step, k, x[step][k], f[step][k])

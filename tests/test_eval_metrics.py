"""Eval / submit path.  Golden: tests/golden/eval_metrics.npz = measures of the reference's own
metric classes (pckmetric.py, meanepe.py, bopAR.py, val_metric.py) on seeded predictions (oracle/gen_golden.py
gen_eval_metrics).  The submit pass of the reference drags in the IK fitting unit and an OpenDR renderer and is not
importable here; its prediction-file format is checked against hodata_submit_epoch_pass.py:34-56,141-145 by construction."""
import json
import os
import zipfile

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "eval_metrics.npz")


def _data():
    g = np.load(GOLD)
    t = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("t.")}
    p = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("p.")}
    return g, p, t


def test_pck_and_mean2d_match_reference():
    from artiboost_amd import registry as R
    import artiboost_amd.metrics as M
    g, p, t = _data()
    for name, typ, cfg, thr in (("hand3d", "Hand3DPCKMetric", dict(VAL_MIN=0.0, VAL_MAX=0.05, STEPS=20), 0.02),
                                ("obj2d", "Obj2DPCKMetric", dict(VAL_MIN=0.0, VAL_MAX=30.0, STEPS=15), 10.0)):
        m = R.build_from_cfg(dict(cfg, TYPE=typ), R.METRIC)
        m.feed(p, t); m.feed(p, t)
        meas = m.get_measures()
        for k in ("epe_mean_per_kp", "pck_curve_per_kp", "auc_per_kp", "epe_mean_all", "auc_all", "thresholds"):
            np.testing.assert_allclose(meas[k], g[f"{name}.{k}"], rtol=1e-6, atol=1e-9, err_msg=f"{name}.{k}")
        np.testing.assert_allclose(m.get_pck_all(thr), g[f"{name}.pck_all"], rtol=1e-9)
    m3 = M.Hand3DPCKMetric(VAL_MIN=0, VAL_MAX=1, STEPS=2)
    m3.feed(p, t)
    assert str(m3).startswith("hand3d pck:")
    m = M.Mean2DEPE(VAL_KEYS=["joints_2d", "corners_2d"], MILLIMETERS=True)
    m.feed(p, t)
    np.testing.assert_allclose([m.get_measures()["joints_2d_mepe"], m.get_measures()["corners_2d_mepe"]], g["mean2d"], rtol=1e-6)


@pytest.mark.parametrize("tag,extra", [("ar", {}), ("ar_center", {"MSSD_USE_CENTER_IDX": True}), ("ar_ycb", {"USE_HO3D_YCB": True})])
def test_mssd_average_recall_matches_reference(tag, extra):
    import artiboost_amd.metrics as M
    g, p, t = _data()
    cfg = dict(USE_MSSD=True, MODEL_INFO=json.loads(str(g["model_info"])), MAX_SYM_DISC_STEP=0.25, MSSD_USE_CORNERS=True,
               DATA_PRESET={"CENTER_IDX": 0}, **extra)
    a = M.AR(**cfg)
    a.feed(p, t)
    meas = a.get_measures()
    assert sorted(meas) == [str(k) for k in g[f"{tag}.keys"]]
    np.testing.assert_allclose([meas[k] for k in sorted(meas)], g[f"{tag}.vals"], rtol=2e-5)
    assert str(a).startswith("mssd:")


def test_val_metric_ar2_matches_reference_and_feeds_the_mining_update():
    import artiboost_amd.metrics as M
    g, p, t = _data()
    v = M.ValMetricAR2(USE_MSSD=True, MODEL_INFO=json.loads(str(g["model_info"])), MAX_SYM_DISC_STEP=0.25, MSSD_USE_CORNERS=True)
    v.feed(p, t)
    avg = v.get_measures_averaged()
    assert [list(k) for k in sorted(avg)] == g["val_ar2.ids"].tolist()
    np.testing.assert_allclose([avg[k] for k in sorted(avg)], g["val_ar2.vals"], rtol=2e-5)
    assert set(avg) == {tuple(int(x) for x in r) for r, s in zip(torch.stack([t["obj_id"], t["persp_id"], t["grasp_id"]], 1).tolist(), t["is_synth"].tolist()) if s}
    with pytest.raises(NotImplementedError):
        M.AR(USE_VSD=True)


def test_submit_pass_prediction_file(tmp_path):
    """hodata_submit_epoch_pass.py:34-56,141-145: [joints, verts], 5 decimals, joints un-reordered with x negated then the
    whole vector negated; zero vertices when no mesh is fitted; a flat zip next to the json."""
    from artiboost_amd.submit import HOSubmitEpochPass
    from artiboost_amd.metrics import Evaluator, Mean3DEPE
    g, p, t = _data()

    class Model:
        def eval(self):
            self.evaled = True

        def __call__(self, batch):
            return {"HybridBaseline": {k: v.clone() for k, v in p.items()}}

    sp = HOSubmitEpochPass({"DUMP": True})
    ev = Evaluator({}, [Mean3DEPE(VAL_KEYS=["joints_3d_abs"], MILLIMETERS=True)])
    model = Model()
    path = str(tmp_path / "pred.json")
    sp(0, [t], model, None, ev, 0, path)
    assert model.evaled
    xyz, verts = json.load(open(path))
    B = p["joints_3d_abs"].shape[0]
    assert len(xyz) == B and len(verts) == B and np.asarray(verts).shape == (B, 778, 3) and not np.asarray(verts).any()
    reorder, unorder = HOSubmitEpochPass.get_order_idxs()
    assert reorder == [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20] and (np.asarray(reorder)[unorder] == np.arange(21)).all()
    want = p["joints_3d_abs"][:, unorder].numpy().copy()
    want[:, :, 0] *= -1
    want = -want
    np.testing.assert_allclose(np.asarray(xyz), np.round(want.astype(np.float64), 5), atol=1.1e-5)
    assert all(len(f"{v}".split(".")[-1]) <= 5 for v in np.asarray(xyz).reshape(-1)[:50])
    with zipfile.ZipFile(str(tmp_path / "pred.zip")) as z:
        assert z.namelist() == ["pred.json"]
    assert ev.get_measures_all()["joints_3d_abs_mepe"] > 0


def test_vis2d_metric_draws_first_batch_only():
    """Vis2DMetric (the reference's shipped YAML lists it under EVALUATOR): an image metric -- first batch after a reset drawn as a
    [prediction | ground truth] grid, skipped by the evaluator's measure tables, returned by dump_images()."""
    import torch
    from artiboost_amd import registry as R
    from artiboost_amd.metrics import Evaluator, Vis2DMetric
    preset = {"IMAGE_SIZE": [64, 48]}
    mets = R.build_evaluator_metric_list([{"TYPE": "Vis2DMetric", "NCOL": 3, "NROW": 2},
                                          {"TYPE": "Mean3DEPE", "VAL_KEYS": ["joints_3d_abs"], "MILLIMETERS": True}], preset_cfg=preset)
    ev = Evaluator({}, mets)
    g = torch.Generator().manual_seed(0)
    B = 4
    targs = {"image": torch.rand((B, 3, 48, 64), generator=g) - 0.5, "joints_2d": torch.rand((B, 21, 2), generator=g) * 40 + 4,
             "corners_2d": torch.rand((B, 8, 2), generator=g) * 40 + 4, "joints_vis": torch.ones(B, 21), "corners_vis": torch.ones(B, 8),
             "joints_3d": torch.zeros(B, 21, 3), "root_joint": torch.zeros(B, 3)}
    preds = {"2d_uvd": torch.rand((B, 30, 3), generator=g), "joints_3d_abs": torch.zeros(B, 21, 3)}
    ev.reset_all()
    ev.feed_all(preds, targs, {})
    vis = mets[0]
    assert isinstance(vis, Vis2DMetric) and vis.count == B
    img = vis.image
    assert img.shape == (2 * 48, 2 * 3 * 64, 3) and img.dtype == np.uint8
    assert img[:, :3 * 64].any() and not np.array_equal(img[:, :3 * 64], img[:, 3 * 64:])       # predictions drawn != ground truth drawn
    assert not img[48:, 64:3 * 64].any()                                                          # tiles 5, 6 of the grid stay empty (B = 4)
    first = img.copy()
    ev.feed_all({"2d_uvd": torch.rand((B, 30, 3), generator=g), "joints_3d_abs": torch.zeros(B, 21, 3)}, targs, {})
    assert vis.count == 2 * B and np.array_equal(vis.image, first)
    assert "Vis2DMetric" not in ev.get_measures_all_striped() and list(ev.dump_images()) == ["Vis2DMetric"]
    assert "Vis2DMetric" not in str(ev)

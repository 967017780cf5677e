"""CCV-space state and its on-disk formats (SURVEY.md sections 8a-R0 and 8f-2): the back-of-hand blacklist
(artiboost_loader.py:415-500), the per-sample pose cache (cache_recorder.py:22-45) and the mining-state files
(utils/recorder.py:177-226).  Golden data: tests/golden/blacklist.npz (output of the reference's own
_construct_blacklist_map) and tests/golden/ref_state/ (files written by the reference's own writers)."""
import os
import pickle

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _cfg():
    return yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))


def _host_loader(cfg=None, **kw):
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.synth import ArtiBoostLoader
    cfg = cfg or _cfg()
    return ArtiBoostLoader.from_assets(SceneAssets("HO3D", seed=1), cfg["MANAGER"], cfg["DATA_PRESET"], 4, 16, device="cpu", random_seed=3, **kw)


def test_blacklist_matches_reference_golden():
    from artiboost_amd import synth
    g = np.load(os.path.join(GOLD, "blacklist.npz"))
    got = synth.back_facing(g["grasps"], g["views"]).numpy()
    np.testing.assert_array_equal(got, g["blacklist"])
    assert 0.02 < got.mean() < 0.3
    # the md5-keyed cache file name (the golden run's engines were SimpleNamespace stand-ins, hence the type names)
    cfg = {"OBJ_ENGINE": {"OBJ": ["obj0", "obj1"]}}
    path = synth.blacklist_cache_path(cfg, 2, 7, int(g["u_bins"]), int(g["th_bins"]), True, obj_engine_type="SimpleNamespace",
                                      grasp_engine_type="SimpleNamespace")
    assert os.path.basename(path) == str(g["cache_name"])
    # vectorised view construction == the scalar restatement (itself pinned by misc.npz)
    rng = np.random.default_rng(0)
    ids, a, b = rng.integers(0, 288, 64), rng.uniform(-0.5, 0.5, 64), rng.uniform(-0.5, 0.5, 64)
    R = synth.perspective_rotmats(ids, a, b, 12, 24)
    R0 = np.stack([synth.perspective_rotmat(int(i), x, y, 12, 24) for i, x, y in zip(ids, a, b)])
    np.testing.assert_allclose(R, R0, atol=1e-13)


def test_loader_masks_blacklisted_triplets(tmp_path):
    cfg = _cfg()
    cfg["MANAGER"]["BLACKLIST_CACHE_ROOT"] = str(tmp_path / "CCV_blacklist")
    ld = _host_loader(cfg)
    bl = ld.blacklist_map
    assert bl.shape == (4, 288, 50) and bl.dtype == torch.bool and 0.02 < bl.float().mean() < 0.3
    assert (ld.sample_weight_map[bl] == 0).all() and (ld.sample_weight_map[~bl] == 1).all()
    o, v, g = ld._sample_ccv()                                  # blacklisted triplets are never drawn
    assert not bl[o, v, g].any()
    plan = ld.plan_epoch(is_train=False)                        # validation mode: without replacement, blacklist excluded
    trip = list(zip(plan["o"].tolist(), plan["v"].tolist(), plan["g"].tolist()))
    assert len(set(trip)) == len(trip) == 16 and not bl[plan["o"], plan["v"], plan["g"]].any()
    files = os.listdir(tmp_path / "CCV_blacklist")
    assert len(files) == 1
    # a second loader reads the cache instead of recomputing: plant a recognisable map
    planted = torch.zeros_like(bl)
    planted[1, 2, 3] = True
    with open(tmp_path / "CCV_blacklist" / files[0], "wb") as f:
        pickle.dump(planted, f)
    assert _host_loader(cfg).blacklist_map.sum() == 1
    cfg["MANAGER"]["FILTER"]["BACK"] = False
    assert _host_loader(cfg).blacklist_map.sum() == 0


def test_reads_state_files_written_by_the_reference():
    from artiboost_amd import ccv_cache
    root = os.path.join(GOLD, "ref_state")
    exp = np.load(os.path.join(root, "expected.npz"))
    c = ccv_cache.load_cache(os.path.join(root, "cache"))
    assert c["obj_name"] == ["021_bleach_cleanser", "010_potted_meat_can"]
    for k_file, k_exp in (("obj_id", "obj_id"), ("persp_id", "persp_id"), ("grasp_id", "grasp_id"), ("obj_pose", "final_obj_pose"),
                          ("hand_verts", "final_hand_verts"), ("hand_joints", "final_joints")):
        np.testing.assert_array_equal(c[k_file], exp[k_exp])
    ld = type("L", (), {})()
    ld.sample_weight_map, ld.occurence_map, ld.use_synth = torch.ones(2, 6, 5), torch.zeros(2, 6, 5, dtype=torch.bool), True
    ld.synth_shutdown = lambda: setattr(ld, "use_synth", False)
    ccv_cache.resume_artiboost_loader(ld, 5, os.path.join(root, "dump"))
    np.testing.assert_array_equal(ld.sample_weight_map.numpy(), exp["weight"])
    np.testing.assert_array_equal(ld.occurence_map.numpy(), exp["occ"])
    assert ld.use_synth is False                                 # the shutdown marker was honoured


def test_state_files_roundtrip_and_match_reference_bytes(tmp_path):
    """What this build writes is byte-identical to what the reference wrote for the same data."""
    from artiboost_amd import ccv_cache
    root = os.path.join(GOLD, "ref_state")
    exp = np.load(os.path.join(root, "expected.npz"))
    ld = type("L", (), {})()
    ld.sample_weight_map, ld.occurence_map, ld.use_synth = torch.from_numpy(exp["weight"]), torch.from_numpy(exp["occ"]), False
    ccv_cache.record_artiboost_loader(ld, 4, str(tmp_path))
    for rel in ("artiboost/sample_weight/004_train.pkl", "artiboost/occurence_map/004.pkl", "artiboost/shutdown"):
        assert open(tmp_path / rel, "rb").read() == open(os.path.join(root, "dump", rel), "rb").read(), rel
    rec = ccv_cache.CacheRecorder(str(tmp_path / "cache"))
    rec({"index": exp["index"], "obj_id": exp["obj_id"], "persp_id": exp["persp_id"], "grasp_id": exp["grasp_id"],
         "obj_name": ["021_bleach_cleanser", "010_potted_meat_can"], "final_obj_pose": torch.from_numpy(exp["final_obj_pose"]),
         "final_hand_verts": torch.from_numpy(exp["final_hand_verts"]), "final_joints": torch.from_numpy(exp["final_joints"])})
    for name in ("0003.pkl", "0017.pkl"):
        a, b = pickle.load(open(tmp_path / "cache" / name, "rb")), pickle.load(open(os.path.join(root, "cache", name), "rb"))
        assert a.keys() == b.keys() and a["obj_name"] == b["obj_name"] and a["obj_id"] == b["obj_id"]
        for k in ("obj_pose", "hand_verts", "hand_joints"):
            np.testing.assert_array_equal(a[k], b[k])


@pytest.mark.gpu
def test_epoch_through_the_reference_cache_format(tmp_path):
    """prepare() -> export_epoch (one pickle per sample, the reference's layout) -> load_cache -> prepare(cache=...) on a
    fresh loader with the same seed reproduces the epoch bit for bit (poses come from the files, the rest from the plan)."""
    from test_gpu_synth import _loader
    from artiboost_amd import ccv_cache
    _, a = _loader()
    a.prepare()
    n = ccv_cache.export_epoch(a, str(tmp_path / "cache"))
    assert n == a.epoch_len and len(os.listdir(tmp_path / "cache")) == n
    cache = ccv_cache.load_cache(str(tmp_path / "cache"))
    assert cache["obj_name"][0] in a.cfg["OBJ_ENGINE"]["OBJ"]
    _, b = _loader()
    b.prepare(cache=cache)
    assert set(a.epoch) == set(b.epoch)
    for k in a.epoch:
        assert torch.equal(a.epoch[k], b.epoch[k]), k
    for x, y in zip(a, b):
        assert torch.equal(x["image"], y["image"])
        break

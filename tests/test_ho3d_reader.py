"""artiboost_amd.datasets.HO3D (round 4: the reader of the HO3D v2 download, anakin/datasets/ho3d.py:28-560, SPLIT_MODE "paper") against
tests/golden/ho3d_reader.npz -- the getters of the REAL reference class run on the same miniature tree (tests/ho3d_fake_tree.py, seed 7;
oracle/gen_ho3d_reader_golden.py, stand-ins only for cv2.Rodrigues and trimesh.load) -- and its frames against Pillow."""
import os

import numpy as np
import pytest

import ho3d_fake_tree as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ho3d_reader.npz")
PRESET = {"USE_CACHE": True, "FILTER_NO_CONTACT": False, "FILTER_THRESH": 0.0, "BBOX_EXPAND_RATIO": 1.2, "FULL_IMAGE": False,
          "IMAGE_SIZE": [224, 224], "CENTER_IDX": 0}


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = tmp_path_factory.mktemp("ho3d")
    T.build(str(root), seed=7)
    return str(root)


def _ds(tree, split, crop, **kw):
    from artiboost_amd import datasets as D
    return D.HO3D(DATA_ROOT=tree, DATA_SPLIT=split, SPLIT_MODE="paper", AUG=split == "train", AUG_PARAM=None,
                  DATA_PRESET=dict(PRESET, CROP_MODEL=crop), **kw)


def test_reader_matches_the_reference_class_getter_by_getter(tree, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)                               # the annotation cache goes to ./common/cache/HO3D like the reference's
    g = np.load(GOLD, allow_pickle=False)
    for split in ("train", "test"):
        for crop in ("root_obj", "hand_obj"):
            ds = _ds(tree, split, crop)
            assert len(ds) == int(g[f"{split}.n"]) > 0 and ds.raw_size == (640, 480)
            for i in range(len(ds)):
                a = ds.get_annots(i)
                np.testing.assert_array_equal(a["bbox_center"], g[f"{split}.{crop}.{i}.center"])
                np.testing.assert_allclose(a["bbox_scale"], float(g[f"{split}.{crop}.{i}.scale"]), rtol=2e-6)
                if crop != "root_obj":
                    continue
                pre = f"{split}.{i}."
                for k in ("cam_intr", "joints_3d", "joints_2d", "corners_3d", "corners_2d", "corners_can", "obj_transf"):
                    np.testing.assert_allclose(a[k], g[pre + k], rtol=2e-6, atol=2e-6, err_msg=pre + k)
                assert a["obj_idx"] == int(g[pre + "obj_idx"]) and a["side"] == "right"
                assert os.path.relpath(ds.get_image_path(i), tree) == bytes(g[pre + "path"]).decode()
    assert os.path.isdir(os.path.join(str(tmp_path), "common", "cache", "HO3D"))
    again = _ds(tree, "train", "root_obj")                     # second construction: from the cache
    np.testing.assert_array_equal(again.get_annots(3)["joints_3d"], _ds(tree, "train", "root_obj", ).get_annots(3)["joints_3d"])


def test_frames_are_served_as_files_and_as_pixels(tree, tmp_path, monkeypatch):
    pytest.importorskip("PIL")
    from PIL import Image
    from artiboost_amd import png
    monkeypatch.chdir(tmp_path)
    ds = _ds(tree, "train", "root_obj")
    data = ds.get_image_bytes(2)
    it = png.parse(data)
    assert (it.width, it.height, it.bpp) == (640, 480, 3)
    np.testing.assert_array_equal(ds.get_image(2), np.asarray(Image.open(ds.get_image_path(2)).convert("RGB")))
    half = _ds(tree, "train", "root_obj", MINI_FACTOR=0.5)
    assert len(half) == int(0.5 * len(ds))


def test_unbuilt_modes_and_absent_roots(tree, tmp_path, monkeypatch):
    from artiboost_amd import datasets as D
    monkeypatch.chdir(tmp_path)
    assert len(D.HO3D(DATA_ROOT=str(tmp_path / "nowhere"), DATA_SPLIT="train", DATA_PRESET=PRESET)) == 0      # a download that is absent: empty set
    with pytest.raises(NotImplementedError):
        D.HO3D(DATA_ROOT=tree, DATA_SPLIT="train", SPLIT_MODE="v1", DATA_PRESET=PRESET)
    with pytest.raises(NotImplementedError):
        D.HO3D(DATA_ROOT=tree, DATA_SPLIT="train", SPLIT_MODE="paper", DATA_PRESET=dict(PRESET, FILTER_NO_CONTACT=True))


@pytest.mark.gpu
def test_real_batches_over_the_reader_decode_png_on_the_device(tree, tmp_path, monkeypatch):
    """RealBatcher over the reader: the .png frames go through the device path and give the batches of the Pillow path, bit for bit."""
    import torch
    from artiboost_amd.realdata import RealBatcher
    monkeypatch.chdir(tmp_path)
    a, b = _ds(tree, "train", "root_obj"), _ds(tree, "train", "root_obj")
    b.get_image_bytes = None
    cfg = dict(PRESET, IMAGE_SIZE=[128, 128])
    ra, rb = RealBatcher(a, cfg, compute_dtype=torch.float32, seed=3), RealBatcher(b, cfg, compute_dtype=torch.float32, seed=3)
    assert ra.assemble([0, 4])["file_kind"] == "png"
    ra.rng, rb.rng = np.random.default_rng(3), np.random.default_rng(3)
    ba, bb = ra.batch([0, 4, 8, 5]), rb.batch([0, 4, 8, 5])
    for k in ba:
        torch.testing.assert_close(ba[k], bb[k], rtol=0, atol=0, msg=k)
    assert ba["obj_idx"].tolist() == [12, 9, 5, 9]


def test_v3_reader_is_the_v2_reader_on_jpg_frames(tmp_path, monkeypatch):
    pytest.importorskip("PIL")
    from artiboost_amd import datasets as D
    from artiboost_amd import jpeg
    monkeypatch.chdir(tmp_path)
    root = tmp_path / "data"
    T.build(str(root), seed=9, version=3)
    g = np.load(GOLD, allow_pickle=False)
    ds = D.HO3DV3(DATA_ROOT=str(root), DATA_SPLIT="train", SPLIT_MODE="paper", AUG=True, DATA_PRESET=dict(PRESET, CROP_MODEL="root_obj"))
    assert len(ds) == int(g["v3.n"]) > 0
    for i in range(len(ds)):
        a = ds.get_annots(i)
        np.testing.assert_allclose(a["joints_3d"], g[f"v3.{i}.joints_3d"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(a["obj_transf"], g[f"v3.{i}.obj_transf"], rtol=2e-6, atol=2e-6)
        assert os.path.relpath(ds.get_image_path(i), str(root)) == bytes(g[f"v3.{i}.path"]).decode()
    it = jpeg.parse(ds.get_image_bytes(0))                    # baseline .jpg: the device decoder's header walk accepts it
    assert (it.width, it.height) == (640, 480)
    assert os.path.isdir(tmp_path / "common" / "cache" / "HO3D_v3")
    assert len(D.HO3DV3(DATA_ROOT=str(tmp_path / "none"), DATA_PRESET=PRESET)) == 0

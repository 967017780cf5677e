"""CPU: pin the colour-jitter / affine-crop half of oracle/render_oracle.c against the REAL Pillow (the library the
reference calls: anakin/utils/img_augment.py:6-80, rendered_dataset.py:256-270)."""
import numpy as np
import pytest
from PIL import Image, ImageEnhance

import render_oracle as ro


def _img(seed, h=96, w=80):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[:10] = rng.integers(0, 256, 3, dtype=np.uint8)     # flat patches: zero-saturation / grey branches
    a[10:20] = np.repeat(rng.integers(0, 256, (10, w, 1), dtype=np.uint8), 3, axis=2)
    return a


def _pil_hue(img, hue_factor):          # img_augment._adjust_hue (img_augment.py:151-190), verbatim semantics
    h, s, v = img.convert("HSV").split()
    np_h = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        np_h += np.array(int(hue_factor * 255)).astype(np.uint8)   # NumPy<2 semantics of np.uint8(negative float): C cast + wrap
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


PIL_OPS = {0: lambda im, f: ImageEnhance.Brightness(im).enhance(f), 1: lambda im, f: ImageEnhance.Color(im).enhance(f),
           2: _pil_hue, 3: lambda im, f: ImageEnhance.Contrast(im).enhance(f)}


@pytest.mark.parametrize("op", [0, 1, 2, 3])
@pytest.mark.parametrize("f", [0.9, 1.0, 1.1, 0.93, 1.07, 0.05, -0.075, 0.0749, 0.3])
def test_single_op_matches_pillow(op, f):
    if op == 2 and abs(f) > 0.5:
        pytest.skip("hue range")
    a = _img(int(abs(f) * 1000) + op)
    ref = np.asarray(PIL_OPS[op](Image.fromarray(a), f))
    rgbx = np.concatenate([a, np.zeros(a.shape[:2] + (1,), np.uint8)], 2)
    order = [op] + [o for o in (0, 1, 3) if o != op][:3]
    fac = [f, 1.0, 1.0, 1.0]     # identity factors for the remaining ops (blend with f == 1 is exact identity)
    if op != 2:
        order = [op, 0, 1, 3] if op not in (0, 1, 3) else [op] + [o for o in (0, 1, 3) if o != op] + [0]
    got = ro.color_jitter(rgbx, order[:4], fac)[:, :, :3]
    bad = (got.astype(int) - ref.astype(int))
    assert np.abs(bad).max() == 0, (op, f, np.abs(bad).max(), (bad != 0).mean())


def test_hsv_roundtrip_exhaustive_subset():
    """Every (r,g,b) on a 52^3 lattice + all greys through PIL's RGB->HSV->RGB vs the restatement (hue shift 0)."""
    v = np.arange(0, 256, 5, dtype=np.uint8)
    r, g, b = np.meshgrid(v, v, v, indexing="ij")
    a = np.stack([r, g, b], -1).reshape(-1, 52, 3)
    ref = np.asarray(_pil_hue(Image.fromarray(a), 0.0))
    rgbx = np.concatenate([a, np.zeros(a.shape[:2] + (1,), np.uint8)], 2)
    got = ro.color_jitter(rgbx, [2, 0, 1, 3], [0.0, 1.0, 1.0, 1.0])[:, :, :3]
    assert np.abs(got.astype(int) - ref.astype(int)).max() == 0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_full_chain_and_crop_match_reference_flow(seed):
    """apply_jitter in a shuffled order + transform_img + to_tensor - 0.5 (rendered_dataset.py:256-270)."""
    rng = np.random.default_rng(seed)
    a = _img(seed, 128, 128)
    order = rng.permutation(4).tolist()
    fac = {0: rng.uniform(0.9, 1.1), 1: rng.uniform(0.9, 1.1), 2: rng.uniform(-0.075, 0.075), 3: rng.uniform(0.9, 1.1)}
    im = Image.fromarray(a)
    for op in order:
        im = PIL_OPS[op](im, fac[op])
    th = rng.uniform(-0.6, 0.6)
    s = rng.uniform(0.4, 0.9)
    fwd = np.array([[s * np.cos(th), -s * np.sin(th), 20.0 + 5 * rng.uniform()], [s * np.sin(th), s * np.cos(th), 10.0], [0, 0, 1]], np.float32)
    inv = np.linalg.inv(fwd)
    res = (64, 48)
    warped = im.transform(res, Image.AFFINE, (inv[0, 0], inv[0, 1], inv[0, 2], inv[1, 0], inv[1, 1], inv[1, 2]))
    ref = np.asarray(warped, np.float32).transpose(2, 0, 1) / 255.0 - 0.5
    rgbx = np.concatenate([a, np.zeros(a.shape[:2] + (1,), np.uint8)], 2)
    jit = ro.color_jitter(rgbx, order, [fac[o] for o in order])
    got = ro.affine_crop(jit, inv[:2].reshape(-1), res[0], res[1])
    mism = np.abs(got - ref) > 1e-6
    assert mism.mean() < 2e-3, mism.mean()        # nearest-neighbour ties at exact .0 source coordinates only

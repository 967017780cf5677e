"""CPU: pin the colour-jitter / affine-crop half of oracle/render_oracle.c against the REAL Pillow (the library the
reference calls: anakin/utils/img_augment.py:6-80, rendered_dataset.py:256-270)."""
import numpy as np
import pytest
from PIL import Image, ImageEnhance, ImageFilter

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


@pytest.mark.parametrize("radius", [0.0, 1e-4, 0.01, 0.05, 0.0765, 0.0767, 0.09, 0.1, 0.3, 0.9, 1.2, 1.4, 2.0, 3.7])
def test_gaussian_blur_matches_pillow(radius):
    """rendered_dataset.py:257-258: img.filter(ImageFilter.GaussianBlur(radius)), radius = U(0,1) * 0.1; the restatement
    (three fractional box-blur passes per axis, libImaging/BoxBlur.c) is bit-exact also for larger radii."""
    rng = np.random.default_rng(int(radius * 1e4))
    a = rng.integers(0, 256, (70, 90, 3), dtype=np.uint8)
    a[:8] = 0
    a[8:16:2] = 255
    a[:, :3] = 255
    ref = np.asarray(Image.fromarray(a).filter(ImageFilter.GaussianBlur(radius)))
    rgbx = np.concatenate([a, np.full(a.shape[:2] + (1,), 9, np.uint8)], 2)
    got = ro.gaussian_blur(rgbx, radius)
    assert (got[:, :, 3] == 9).all()
    assert np.abs(got[:, :, :3].astype(int) - ref.astype(int)).max() == 0
    if radius <= 0.0765:
        assert (ref == a).all()          # below ~0.0766 every pass is the identity (used by the HIP path to skip samples)


def test_background_resize_is_bilinear_within_one_lsb():
    """renderer.py:125-136: the background is a random crop resized with cv2.resize (INTER_LINEAR).  cv2 is absent, so the
    fixed-point restatement in the oracle is unpinned; this checks it against a float half-pixel-centre bilinear
    interpolation (<= 1 LSB) and that an unscaled crop is copied bit for bit."""
    from artiboost_amd.assets import SceneAssets
    assets = SceneAssets("HO3D", seed=1)
    holder = ro.SceneHolder(assets)
    keys = np.full((512, 512), 0xFFFFFFFFFFFFFFFF, np.uint64)          # no geometry anywhere
    hv = np.zeros((778, 3), np.float32)
    for crop, x0, y0 in [(512, 100, 256), (768, 0, 0), (600, 168, 3), (513, 255, 255), (701, 11, 67)]:
        smp = np.zeros(1, ro.SAMPLE_DTYPE)
        smp["bg_id"], smp["bg_w"], smp["bg_h"], smp["bg_x0"], smp["bg_y0"], smp["light"] = 2, crop, crop, x0, y0, 1.0
        smp["obj_pose"][0, [0, 5, 10, 15]] = 1.0
        got = holder.shade(smp, hv, keys)[:, :, :3].astype(np.float64)
        src = assets.backgrounds[2, y0:y0 + crop, x0:x0 + crop].astype(np.float64)
        if crop == 512:
            assert (got == src).all()
            continue
        f = (np.arange(512) + 0.5) * (crop / 512.0) - 0.5
        s = np.clip(np.floor(f).astype(int), 0, crop - 2)
        w = np.clip(f - s, 0.0, 1.0)
        rows = src[s] * (1 - w)[:, None, None] + src[s + 1] * w[:, None, None]
        ref = rows[:, s] * (1 - w)[None, :, None] + rows[:, s + 1] * w[None, :, None]
        assert np.abs(got - ref).max() <= 1.0, (crop, np.abs(got - ref).max())

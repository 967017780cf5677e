"""GPU parity of the batched renderer against the CPU oracle (oracle/render_oracle.c): integer z-test / visibility
keys bit-exact, shaded RGBX bit-exact, jittered + cropped network input exact."""
import numpy as np
import pytest
import torch

import gen_scene
import render_oracle as ro

pytestmark = pytest.mark.gpu


def _run(dataset, B, seed, res, blur=True, want_u8n=False):
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.render import DeviceRenderer
    assets = SceneAssets(dataset)
    K = np.array([[435.0, 0, 256.0], [0, 435.0, 256.0], [0, 0, 1.0]])
    sc = gen_scene.make_samples(assets, B, seed, out_res=(res, res))
    holder = ro.SceneHolder(assets)
    out_ref, rgbx_ref, keys_ref = holder.render_batch(sc["samples"], sc["hand_verts"], sc["order"], sc["factor"],
                                                      sc["inv_affine"], res, res, blur=sc["blur"] if blur else None)
    # un-jittered render for the bit-exact shading check
    rgbx_plain = np.stack([holder.shade(sc["samples"][b:b + 1], sc["hand_verts"][b], keys_ref[b]) for b in range(B)])
    r = DeviceRenderer(assets, K)
    dev = r.dev
    smp = torch.from_numpy(sc["samples"].view(np.uint8).reshape(B, -1)).to(dev)
    out_pad = torch.zeros((B, res + 6, res + 8, 4), dtype=torch.float32, device=dev)
    out_chw = torch.empty((B, 3, res, res), dtype=torch.float32, device=dev)
    o = r.render(smp, torch.from_numpy(sc["hand_verts"]).to(dev), torch.from_numpy(sc["order"]).to(dev),
                 torch.from_numpy(sc["factor"]).to(dev), torch.from_numpy(sc["inv_affine"]).to(dev), res, res,
                 out_pad=out_pad, out_chw=out_chw, want_keys=True, want_rgbx=True,
                 blur=torch.from_numpy(sc["blur"]).to(dev) if blur else None)
    if want_u8n:      # the same batch once more into the integer plane (AB_DT_U8N: what bench.py's loader hands the bf16x3 stem)
        pad8 = torch.zeros((B, res + 6, res + 8, 4), dtype=torch.bfloat16, device=dev)
        r.render(smp, torch.from_numpy(sc["hand_verts"]).to(dev), torch.from_numpy(sc["order"]).to(dev),
                 torch.from_numpy(sc["factor"]).to(dev), torch.from_numpy(sc["inv_affine"]).to(dev), res, res, out_pad=pad8, pad_code=2,
                 blur=torch.from_numpy(sc["blur"]).to(dev) if blur else None)
        sc["u8n_plane"] = pad8.float().cpu().numpy()
    return sc, (out_ref, rgbx_plain, keys_ref), (out_chw.cpu().numpy(), out_pad.cpu().numpy(), o["rgbx"].cpu().numpy(),
                                                o["keys"].cpu().numpy().view(np.uint64))


@pytest.mark.parametrize("dataset,B,seed,res,blur", [("HO3D", 3, 0, 224, True), ("DexYCB", 2, 1, 256, True),
                                                     ("HO3D", 8, 2, 256, True), ("HO3D", 3, 3, 256, False)])
def test_render_bit_exact_vs_oracle(dataset, B, seed, res, blur):
    sc, (out_ref, rgbx_ref, keys_ref), (out, out_pad, rgbx, keys) = _run(dataset, B, seed, res, blur)
    covered = keys_ref != np.uint64(0xFFFFFFFFFFFFFFFF)
    assert covered.mean() > 0.01, "scene should contain geometry"
    np.testing.assert_array_equal(keys, keys_ref)                      # depth24<<32 | face id, every pixel
    np.testing.assert_array_equal(rgbx, rgbx_ref)                      # shaded + composited u8 image, every pixel
    np.testing.assert_array_equal(out, out_ref)                        # jitter + crop + normalise
    np.testing.assert_array_equal(out_pad[:, 3:-3, 3:-5, :3].transpose(0, 3, 1, 2), out_ref)
    assert np.abs(out_pad[:, :3]).max() == 0 and np.abs(out_pad[:, :, :3]).max() == 0 and np.abs(out_pad[..., 3]).max() == 0


@pytest.mark.parametrize("dataset,seed", [("HO3D", 7), ("DexYCB", 8)])
def test_render_bit_exact_vs_oracle_at_the_benchmark_batch(dataset, seed):
    """B = 64 at 256 x 256 -- bench.py's batch.  tile_order_kernel compacts the tiles of ALL samples behind two global cursors
    (render.hip:193-206), so the cross-sample schedule only shows at the full batch: keys, shaded RGBX and the jittered output of all 64
    samples against the C oracle, plus the integer plane the benchmark's stem reads (2 v - 255 of the same pixels, zero border)."""
    sc, (out_ref, rgbx_ref, keys_ref), (out, out_pad, rgbx, keys) = _run(dataset, 64, seed, 256, True, want_u8n=True)
    assert (keys_ref != np.uint64(0xFFFFFFFFFFFFFFFF)).reshape(64, -1).mean(1).min() > 0.005, "every sample should contain geometry"
    np.testing.assert_array_equal(keys, keys_ref)
    np.testing.assert_array_equal(rgbx, rgbx_ref)
    np.testing.assert_array_equal(out, out_ref)
    np.testing.assert_array_equal(out_pad[:, 3:-3, 3:-5, :3].transpose(0, 3, 1, 2), out_ref)
    plane = sc["u8n_plane"]
    v = np.rint((out_ref.astype(np.float64) + 0.5) * 255.0)                      # the uint8 pixels behind out_ref = v / 255 - 0.5
    np.testing.assert_array_equal((v / 255.0).astype(np.float32) - np.float32(0.5), out_ref)
    np.testing.assert_array_equal(plane[:, 3:-3, 3:-5, :3].transpose(0, 3, 1, 2), 2.0 * v - 255.0)
    assert np.abs(plane[:, :3]).max() == 0 and np.abs(plane[:, :, :3]).max() == 0 and np.abs(plane[..., 3]).max() == 0


def test_render_properties():
    """Size-independent properties: both hand and object are visible; every covered pixel's face id is valid; the
    depth of covered pixels lies inside the quantised range of the scene (0.3 m .. 0.8 m)."""
    sc, _, (out, out_pad, rgbx, keys) = _run("HO3D", 4, 5, 256)
    fid = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
    cov = keys != np.uint64(0xFFFFFFFFFFFFFFFF)
    assert (fid[cov] < 1538 + 4032).all()
    assert ((fid < 1538) & cov).any() and ((fid >= 1538) & cov).any()
    z = (keys[cov] >> np.uint64(32)).astype(np.float64) / 16777215.0
    Z = 1.0 / (20.0 + z * (0.01 - 20.0))
    assert Z.min() > 0.25 and Z.max() < 0.9
    assert np.isfinite(out).all() and out.min() >= -0.5 and out.max() <= 0.5


def test_gaussian_blur_vs_oracle():
    """ab_gaussian_blur == the CPU oracle (itself pinned against Pillow's ImageFilter.GaussianBlur) on noise images with
    flat and striped regions (worst case for the rounding), for radii over the reference's range and beyond, including
    the radii where every pass is the identity."""
    from artiboost_amd import _lib as L
    rng = np.random.default_rng(0)
    radii = np.array([0.0, 1e-4, 0.05, 0.0765, 0.0767, 0.08, 0.0999, 0.1, 0.3, 0.7, 1.2, 1.4], np.float32)
    B, H, W = len(radii), 96, 160
    img = rng.integers(0, 256, (B, H, W, 4), dtype=np.uint8)
    img[:, :8] = 0; img[:, 8:16:2] = 255; img[:, :, :4, :3] = 255; img[:, -3:] = 7
    ref = np.stack([ro.gaussian_blur(img[b], float(radii[b])) for b in range(B)])
    src = torch.from_numpy(img).cuda()
    out = torch.empty_like(src)
    rc = L.lib().ab_gaussian_blur(L.ptr(src), L.i(B), L.i(W), L.i(H), L.ptr(torch.from_numpy(radii).cuda()), L.ptr(out), L.stream())
    assert rc == 0
    got = out.cpu().numpy()
    for b in range(B):
        np.testing.assert_array_equal(got[b], ref[b], err_msg=f"radius {radii[b]}")
    assert (ref[2] == img[2]).all() and (ref[7] != img[7]).any()      # 0.05: identity; 0.1: acts


def test_color_jitter_all_rgb_values_vs_oracle():
    """ab_color_jitter over an image holding all 2^24 RGB triples == the CPU oracle (itself pinned against Pillow in
    tests/test_render_oracle.py), for hue / saturation / brightness / contrast in several orders and factor signs: the
    tabulated hsv->rgb terms, the fmod-free hue wrap and the trunc-based rounding are bit-exact."""
    import ctypes
    from artiboost_amd import _lib as L
    v = np.arange(1 << 24, dtype=np.uint32)
    rgbx = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255, np.full_like(v, 255)], axis=1).astype(np.uint8).reshape(4096, 4096, 4)
    cases = [([2, 0, 1, 3], [0.075, 1.1, 0.9, 1.05]), ([3, 2, 1, 0], [0.93, -0.075, 1.1, 0.95]),
             ([1, 3, 0, 2], [1.07, 1.1, 0.9, -0.031]), ([2, 2, 2, 2], [0.5, 0.013, -0.2, 0.33]),
             ([2, 2, 2, 2], [100.5 / 255, 200.5 / 255, 254.5 / 255, 0.0])]      # (all 256 hue shifts x all triples: tests/hue_exhaustive.py, 0 mismatches)
    src = torch.from_numpy(rgbx).cuda()
    ws = torch.empty(8, dtype=torch.uint8, device="cuda")
    for order, factor in cases:
        ref = ro.color_jitter(rgbx, order, factor)
        o = torch.tensor([order], dtype=torch.int32, device="cuda")
        f = torch.tensor([factor], dtype=torch.float32, device="cuda")
        out = torch.empty_like(src)
        rc = L.lib().ab_color_jitter(L.ptr(src), L.i(1), L.i(4096 * 4096), L.ptr(o), L.ptr(f), L.ptr(out), L.ptr(ws), L.stream())
        assert rc == 0
        got = out.cpu().numpy()
        bad = np.argwhere((got[..., :3] != ref[..., :3]).any(axis=-1))
        assert len(bad) == 0, (order, factor, bad[:5], got[tuple(bad[0])], ref[tuple(bad[0])])


def test_render_is_bit_stable_beside_mfma_kernels():
    """Round 6: with packed-fp32 VALU instructions in the shading code, the SAME render differed by a few levels in a few dozen pixels whenever its
    waves shared a SIMD with waves executing v_mfma_f32_32x32x16_bf16 (another stream: the DDP render overlap, ThreadedPrefetcher) -- the visibility
    keys stayed exact, the same thread shading the same pixel twice got two answers.  The library is built without packed fp32 (build.py); here the
    render runs beside this build's layer-3 convolution on a second stream and must equal the solo render byte for byte, every time."""
    from artiboost_amd import kernels as K
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.render import DeviceRenderer
    B, res = 24, 256
    assets = SceneAssets("HO3D", seed=1)
    Kc = np.array([[435.0, 0, 256.0], [0, 435.0, 256.0], [0, 0, 1.0]])
    sc = gen_scene.make_samples(assets, B, 7, out_res=(res, res))
    r = DeviceRenderer(assets, Kc)
    dev = r.dev
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    args = (t(sc["samples"].view(np.uint8).reshape(B, -1)), t(sc["hand_verts"]), t(sc["order"]), t(sc["factor"]), t(sc["inv_affine"]))
    blur = t(sc["blur"])

    def render():
        pad = torch.zeros((B, res + 6, res + 8, 4), dtype=torch.bfloat16, device=dev)
        o = r.render(*args, res, res, out_pad=pad, want_keys=True, want_rgbx=True, blur=blur, pad_code=2)
        return o["keys"].clone(), o["rgbx"].clone(), pad

    ref = render()
    torch.cuda.synchronize()
    x3, w3 = K.split(torch.randn(64, 16, 16, 256, device=dev)), K.split(torch.randn(256, 3, 3, 256, device=dev) * 0.05)
    K.conv2d_fwd_x3(x3, w3, 1, 1, want_stats=True)
    side = torch.cuda.Stream()
    for it in range(12):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(10):
                K.conv2d_fwd_x3(x3, w3, 1, 1, want_stats=True)      # 126 VGPRs, 122 KB of LDS: leaves room for the renderer's workgroups on its CU
        got = render()
        torch.cuda.synchronize()
        for name, a, b in zip(("keys", "rgbx", "output plane"), ref, got):
            assert torch.equal(a, b), (it, name, int((a != b).sum()))

"""Grasp refiner (anakin/artiboost/refiner.py).

CPU: oracle/refiner_oracle.py against tests/golden/refiner.npz -- the output of the REAL reference classes
(HORefiner.forward, _RefineNet, ResBlock, CRot2rotmat, point2point_signed) run in the build container by
oracle/gen_golden.py with stand-ins only for the three absent third-party libraries.
GPU: ab_nearest_dist / ab_linear_fused / the HIP HORefiner against that oracle and the same golden vectors."""
import os

import numpy as np
import pytest
import torch

import refiner_oracle as rfo

GOLD = os.path.join(os.path.dirname(__file__), "golden", "refiner.npz")


def _assets():
    from artiboost_amd.assets import SceneAssets, resample_objects
    assets = SceneAssets("HO3D", seed=1)
    return assets, resample_objects(assets, 10000, seed=7)


def test_resampled_objects_are_surface_points():
    assets, pts = _assets()
    assert pts.shape == (4, 10000, 3) and pts.dtype == np.float32
    for o, p in zip(assets.objects, pts):
        v = np.asarray(o["verts"])
        assert (p.min(0) >= v.min(0) - 1e-6).all() and (p.max(0) <= v.max(0) + 1e-6).all()
        assert len(np.unique(p, axis=0)) > 9900                  # drawn without replacement from distinct vertices
    _, again = _assets()
    np.testing.assert_array_equal(pts, again)


def test_oracle_matches_reference_golden():
    g = np.load(GOLD)
    assets, pts = _assets()
    p = rfo.fill_params(4)
    np.testing.assert_allclose(rfo.res_block(p, "rb1", torch.from_numpy(g["rb1.in"])).numpy(), g["rb1.out"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rfo.crot2rotmat(torch.from_numpy(g["crot.in"])).numpy(), g["crot.out"], rtol=1e-6, atol=1e-6)
    pose, tsl, rot, oi = g["in.hand_pose"], g["in.hand_tsl"], g["in.obj_rot"], g["in.obj_idx"]
    verts = rfo.mano_fn_of(assets.hand)(torch.from_numpy(pose))[0].numpy() + tsl[:, None]
    d, _ = rfo.nearest_dist(verts, rfo.rotate_points(rot, pts[oi]))
    np.testing.assert_allclose(d, g["h2o.first"], rtol=1e-5, atol=1e-7)
    res = rfo.ho_refiner(p, assets.hand, torch.from_numpy(pose), torch.from_numpy(tsl), torch.from_numpy(rot), pts[oi], n_iters=3)
    for k in ("hand_verts", "joints", "hand_pose", "hand_tsl"):
        np.testing.assert_allclose(res[k].numpy(), g["out." + k], rtol=1e-4, atol=2e-5, err_msg=k)
    # the refinement is not a no-op with these weights
    assert np.abs(g["out.hand_pose"] - pose).max() > 1e-3 and np.abs(g["out.hand_tsl"] - tsl).max() > 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("B,P1,P2,with_rot", [(3, 778, 10000, True), (2, 70, 2053, False), (1, 129, 64, True)])
def test_nearest_dist_vs_oracle(B, P1, P2, with_rot):
    from artiboost_amd.refiner import nearest_dist
    import pose_oracle as po
    rng = np.random.default_rng(B * 1000 + P1)
    x = rng.uniform(-0.1, 0.1, (B, P1, 3)).astype(np.float32)
    tab = rng.uniform(-0.12, 0.12, (5, P2, 3)).astype(np.float32)
    tab[:, 7] = tab[:, 3]                                        # exact ties: the first index must win
    oi = rng.integers(0, 5, B)
    rot = po.aa_to_rotmat(rng.standard_normal((B, 3))).astype(np.float32) if with_rot else None
    y = rfo.rotate_points(rot, tab[oi]) if with_rot else tab[oi]
    d_ref, i_ref = rfo.nearest_dist(x, y)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    d, i = nearest_dist(t(x), t(tab), t(oi.astype(np.int64)), t(rot) if with_rot else None, want_idx=True)
    np.testing.assert_array_equal(i.cpu().numpy(), i_ref)
    np.testing.assert_allclose(d.cpu().numpy(), d_ref, rtol=1e-6, atol=0)
    # per-vertex affine + pitched output (the BatchNorm1d(778) fold and the MLP feature matrix)
    sc, sh = rng.uniform(0.5, 2, P1).astype(np.float32), rng.standard_normal(P1).astype(np.float32)
    out = torch.full((B, P1 + 13), -7.0, device="cuda")
    nearest_dist(t(x), t(tab), t(oi.astype(np.int64)), t(rot) if with_rot else None, t(sc), t(sh), out=out[:, :P1])
    np.testing.assert_allclose(out[:, :P1].cpu().numpy(), d_ref * sc + sh, rtol=1e-6, atol=1e-7)
    assert (out[:, P1:] == -7.0).all()


@pytest.mark.gpu
def test_linear_fused_vs_torch():
    from artiboost_amd.refiner import linear_fused
    g = torch.Generator().manual_seed(0)
    M, N, K = 37, 99, 880
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / 30, torch.randn(N, generator=g)
    sc, sh, r = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g), torch.randn(M, N + 5, generator=g)
    ref = torch.nn.functional.leaky_relu(((x.double() @ w.double().t() + b.double()) * sc.double() + sh.double()) + r[:, :N].double(), 0.2)
    c = lambda a: a.cuda()      # noqa: E731
    out = torch.zeros((M, N + 3), device="cuda")
    linear_fused(c(x), c(w), c(b), c(sc), c(sh), residual=c(r)[:, :N], act=2, slope=0.2, out=out[:, :N])
    np.testing.assert_allclose(out[:, :N].cpu().numpy(), ref.float().numpy(), rtol=2e-5, atol=2e-5)
    assert (out[:, N:] == 0).all()
    plain = linear_fused(c(x), c(w))
    np.testing.assert_allclose(plain.cpu().numpy(), (x.double() @ w.double().t()).float().numpy(), rtol=2e-5, atol=2e-5)
    relu = linear_fused(c(x), c(w), c(b), act=1)
    np.testing.assert_allclose(relu.cpu().numpy(), torch.relu(x.double() @ w.double().t() + b.double()).float().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_ho_refiner_vs_reference_golden_and_oracle():
    from artiboost_amd.refiner import HORefiner, Refiner
    from artiboost_amd.synth import ManoLayerHIP
    g = np.load(GOLD)
    assets, pts = _assets()
    ref = Refiner.build("hand_obj", {"PRETRAINED": "", "ITERS": 3, "ALLOW_RANDOM_INIT": True}, ManoLayerHIP(assets.hand))
    assert isinstance(ref, HORefiner)
    ref.load_state_dict(rfo.fill_params(4))
    ref.setup(pts)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    res = ref({"hand_pose": t(g["in.hand_pose"]), "hand_tsl": t(g["in.hand_tsl"]), "obj_rot": t(g["in.obj_rot"])},
              t(g["in.obj_idx"].astype(np.int64)))
    for k in ("hand_verts", "joints", "hand_pose", "hand_tsl"):
        np.testing.assert_allclose(res[k].cpu().numpy(), g["out." + k], rtol=1e-3, atol=1e-4, err_msg=k)
    # a larger seeded batch against the oracle
    pose, tsl, rot, oi = rfo.make_inputs(assets, 16, 11)
    want = rfo.ho_refiner(rfo.fill_params(4), assets.hand, torch.from_numpy(pose), torch.from_numpy(tsl), torch.from_numpy(rot), pts[oi])
    got = ref({"hand_pose": t(pose), "hand_tsl": t(tsl), "obj_rot": t(rot)}, t(oi.astype(np.int64)))
    for k in ("hand_verts", "joints", "hand_pose", "hand_tsl"):
        np.testing.assert_allclose(got[k].cpu().numpy(), want[k].numpy(), rtol=1e-3, atol=1e-4, err_msg=k)
    with pytest.raises(FileNotFoundError):
        Refiner.build("hand_obj", {"PRETRAINED": "assets/GrabNet/refinenet.pt", "ITERS": 3}, ManoLayerHIP(assets.hand))


@pytest.mark.gpu
def test_loader_epoch_with_refiner():
    """prepare() with a REFINER block: the refined grasps differ from the unrefined ones and still render / assemble GT."""
    import copy
    from test_gpu_synth import _loader
    from artiboost_amd.synth import ArtiBoostLoader
    assets, plain = _loader()
    cfg = copy.deepcopy(plain.cfg)
    cfg["REFINER"] = {"TYPE": "hand_obj", "PRETRAINED": "", "ITERS": 3, "ALLOW_RANDOM_INIT": True}
    refined = ArtiBoostLoader.from_assets(assets, cfg, plain.preset, plain.batch_size, plain.synth_len, compute_dtype=plain.dtype, random_seed=3)
    refined.refiner.load_state_dict(rfo.fill_params(4))
    plain.prepare(); refined.prepare()
    a, b = plain.epoch["_hand_verts"], refined.epoch["_hand_verts"]
    assert a.shape == b.shape and torch.isfinite(b).all()
    d = (a - b).norm(dim=-1).amax(dim=1).median().item()
    assert 1e-4 < d < 0.2, d                                          # moved (stand-in weights), but still a hand near the object
    np.testing.assert_array_equal(plain.epoch["_samples"].cpu().numpy()[:, :32], refined.epoch["_samples"].cpu().numpy()[:, :32])
    batch = next(iter(refined))
    assert torch.isfinite(batch["image"]).all() and batch["joints_3d"].shape == (4, 21, 3)
    # zeroed output heads: the refiner only decodes the scrambled grasp (6-D rotation round trip + MANO), so the epoch must
    # equal the unrefined one -- pins the pose-generator wiring (preprocessor.py:73-88)
    sd = rfo.fill_params(4)
    for k in ("out_p.weight", "out_p.bias", "out_t.weight", "out_t.bias"):
        sd[k] = torch.zeros_like(sd[k])
    refined.refiner.load_state_dict(sd)
    refined.rng = np.random.default_rng(3); refined.torch_gen = torch.Generator().manual_seed(3)
    plain.rng = np.random.default_rng(3); plain.torch_gen = torch.Generator().manual_seed(3)
    plain.prepare(); refined.prepare()
    np.testing.assert_allclose(refined.epoch["_hand_verts"].cpu().numpy(), plain.epoch["_hand_verts"].cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(refined.epoch["joints_3d"].cpu().numpy(), plain.epoch["joints_3d"].cpu().numpy(), atol=2e-5)

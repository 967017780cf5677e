"""CPU: pin oracle/learner_oracle.py (our restatement) against golden vectors produced by the REAL reference
(oracle/gen_golden.py imported /root/reference).  These are the 'oracle is trustworthy' tests."""
import os
import random

import numpy as np
import pytest
import torch

import learner_oracle as lo
from gen_batch import make_batch


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("tag", ["tiny", "g224", "g256"])
def test_softargmax_head_matches_reference(golden_dir, tag):
    g = _load(golden_dir, "head.npz")
    seed, B, C, D, H, W = [int(x) for x in g[f"{tag}.seed"]]
    gen = torch.Generator().manual_seed(seed)
    logits = (4.0 * torch.randn(B, C * D, H, W, generator=gen)).requires_grad_(True)
    uvd, conf = lo.softargmax3d(logits, C, D, H, W)
    np.testing.assert_allclose(uvd.detach().numpy(), g[f"{tag}.uvd"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(conf.detach().numpy(), g[f"{tag}.conf"], rtol=1e-5, atol=1e-7)
    ((uvd * torch.from_numpy(g[f"{tag}.g_uvd"])).sum() + (conf * torch.from_numpy(g[f"{tag}.g_conf"])).sum()).backward()
    dl = logits.grad.reshape(B, C, -1)[:, :, ::97].numpy()
    np.testing.assert_allclose(dl, g[f"{tag}.dlogits.sample"], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("tag", ["g224", "g256"])
def test_learner_forward_backward_matches_reference(golden_dir, tag):
    g = _load(golden_dir, f"learner_{tag}.npz")
    size, heat, depth, B, seed = [int(x) for x in g["meta"]]
    params = lo.fill_params(lo.param_shapes(22, depth), seed=seed)
    batch = make_batch(B, size, seed + 100)
    # eval mode
    with torch.no_grad():
        keep = {}
        preds = lo.hybrid_forward(params, batch, [size, size], 22, depth, 0, training=False, keep=keep)
    for k in ("joints_3d_abs", "corners_3d_abs", "joints_3d", "corners_3d", "2d_uvd", "boxroot_3d_abs", "box_rot_rotmat"):
        np.testing.assert_allclose(preds[k].numpy(), g[f"eval.pred.{k}"], rtol=1e-4, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(keep["logits"][:, ::37, ::5, ::5].numpy(), g["eval.logits.sample"], rtol=1e-3, atol=1e-4)
    # train mode + losses + grads
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v)
            for k, v in params.items()}
    stats, keep = {}, {}
    preds = lo.hybrid_forward(leaf, batch, [size, size], 22, depth, 0, training=True, stats=stats, keep=keep)
    for k in ("joints_3d_abs", "corners_3d_abs", "2d_uvd", "box_rot_rotmat"):
        np.testing.assert_allclose(preds[k].detach().numpy(), g[f"train.pred.{k}"], rtol=1e-4, atol=2e-5, err_msg=k)
    for k in ("res_layer1", "res_layer2", "res_layer3", "res_layer4"):
        np.testing.assert_allclose(keep[k].detach().mean(dim=(0, 2, 3)).numpy(), g[f"train.feat.{k}.mean_c"],
                                   rtol=1e-3, atol=1e-4)
    random.seed(seed + 7)
    torch.manual_seed(seed + 7)
    total, losses, _ = lo.criterion(preds, batch)
    for k in ("joints_3d_loss", "corners_3d_loss", "joint_ord_loss", "part_ord_loss", "scene_ord_loss", "final_loss"):
        np.testing.assert_allclose(losses[k].detach().numpy().reshape(-1), g[f"loss.{k}"].reshape(-1),
                                   rtol=2e-4, atol=1e-7, err_msg=k)
    total.backward()
    names = [str(n) for n in g["grad.names"]]
    ref = dict(zip(names, g["grad.norms"]))
    for n in names:
        got = float(leaf[n].grad.norm())
        assert abs(got - ref[n]) <= 1e-2 * ref[n] + 1e-9, (n, got, ref[n])
    assert leaf["backbone.fc.weight"].grad is None
    np.testing.assert_allclose(leaf["hybrid_head.final_layer.bias"].grad.numpy(), g["grad.final_bias"], rtol=2e-3, atol=1e-9)
    ref_s = g["grad.conv1.sample"]  # deepest gradient: fp32 round-off through 36 layers differs by reduction order
    np.testing.assert_allclose(leaf["backbone.conv1.weight"].grad[::8, :, ::3, ::3].numpy(), ref_s,
                               rtol=2e-2, atol=3e-2 * np.abs(ref_s).max())  # restatement == reference to 1e-8 in fp64 (checked once)
    np.testing.assert_allclose(stats["backbone.bn1.running_var"].numpy(), g["stat.bn1.running_var"], rtol=1e-5)
    np.testing.assert_allclose(stats["backbone.layer4.2.bn2.running_var"].numpy(), g["stat.l4.2.bn2.running_var"], rtol=1e-4)
    # optimiser step (clip 0.001 + Adam lr 5e-5)
    names_g = [n for n in leaf if leaf[n].dtype.is_floating_point and getattr(leaf[n], "grad", None) is not None]
    ps = [leaf[n].detach().clone() for n in names_g]
    gs = [leaf[n].grad for n in names_g]
    m = [torch.zeros_like(p) for p in ps]
    v = [torch.zeros_like(p) for p in ps]
    tn = lo.clip_and_adam(ps, gs, m, v, step=1)
    np.testing.assert_allclose(float(tn), float(g["opt.total_norm"]), rtol=1e-3)
    i = names_g.index("hybrid_head.final_layer.bias")
    np.testing.assert_allclose((ps[i] - params["hybrid_head.final_layer.bias"]).numpy(), g["opt.final_bias.delta"],
                               rtol=2e-2, atol=2e-7)


def test_misc_helpers_match_reference(golden_dir):
    g = _load(golden_dir, "misc.npz")
    import pose_oracle as po
    for i in range(len(g["affine.scale"])):
        res = [224, 224] if i % 2 else [256, 256]
        tot, post = po.get_affine_transform(g["affine.center"][i], float(g["affine.scale"][i]), [256.0, 256.0], res,
                                            float(g["affine.rot"][i]))
        np.testing.assert_allclose(tot, g["affine.total"][i], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(post, g["affine.post"][i], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(po.transform_coords(g["coords.pts"], g["affine.total"][0]), g["coords.out"], rtol=1e-6, atol=1e-4)
    np.testing.assert_array_equal(po.annot_center(g["coords.pts"]), g["annot.center"])
    np.testing.assert_allclose(po.annot_scale(g["coords.pts"]), g["annot.scale"])
    ids = [tuple(int(x) for x in r) for r in g["ccv.ids"]]
    res = dict(zip(ids, g["ccv.vals"].tolist()))
    w = lo.ccv_update_method_1(torch.ones(4, 288, 50), res, 0.1, 10.0)
    np.testing.assert_allclose(np.array([float(w[i]) for i in res]), g["ccv.m1"], rtol=1e-6)
    b, r, c = lo.ccv_row_col(torch.from_numpy(g["ccv.tidx"]), 288, 50)
    np.testing.assert_array_equal(b.numpy(), g["ccv.bidx"])
    np.testing.assert_array_equal(r.numpy(), g["ccv.ridx"])
    np.testing.assert_array_equal(c.numpy(), g["ccv.cidx"])
    for v, m in zip(g["view.vecs"], g["view.align"]):
        np.testing.assert_allclose(po.align_mat(v), m, rtol=1e-9, atol=1e-12)


def test_mano_lbs_oracle_matches_reference_in_tree_layer(golden_dir):
    """R1 pinned: pose_oracle.mano_lbs vs the reference's own MANO forward (anakin/postprocess/iknet/manolayer.py:182-276,
    run under a jax.numpy -> numpy shim by oracle/gen_mano_golden.py) on the seeded stand-in hand model.  The reference layer
    returns wrist-relative vertices / joints (center_idx = 0) and computes Rodrigues through norm(v + 1e-8): 1e-8-level."""
    import pose_oracle as po
    from artiboost_amd.assets import make_hand_model
    g = np.load(os.path.join(golden_dir, "mano.npz"))
    hm = make_hand_model(int(g["hand_model_seed"]))
    v, j, T = po.mano_lbs(hm, g["pose"], g["betas"])
    np.testing.assert_allclose(v - j[:, :1], g["verts_rel_wrist"], rtol=0, atol=5e-8)
    np.testing.assert_allclose(j - j[:, :1], g["joints_rel_wrist"], rtol=0, atol=5e-8)
    # the translation the centred layer removes: the wrist joint is the regressed rest joint, untouched by the pose
    vs = hm["v_template"][None] + np.einsum("vkl,bl->bvk", hm["shapedirs"], g["betas"])
    np.testing.assert_allclose(j[:, 0], np.einsum("v,bvk->bk", hm["J_regressor"][0], vs), rtol=0, atol=1e-12)
    np.testing.assert_allclose(T[:, 0, :3, 3], j[:, 0], rtol=0, atol=1e-12)


def test_scrambler_oracle_and_product_match_reference_class(golden_dir):
    """RandomScrambler (scrambler.py:65-81): the oracle's restatement and the product's torch implementation against the
    reference class run with the same draws (tests/golden/scrambler.npz, oracle/gen_scrambler_golden.py)."""
    import torch
    import pose_oracle as po
    from artiboost_amd.synth import PoseGenerator
    g = np.load(os.path.join(golden_dir, "scrambler.npz"))
    pose, tsl = g["in.hand_pose"], g["in.hand_tsl"]
    for seed in (11, 12):
        ra, rt = g[f"s{seed}.rand_angle"], g[f"s{seed}.rand_tsl"]
        op, ot = po.scramble(pose.astype(np.float64), tsl.astype(np.float64), ra.astype(np.float64), rt.astype(np.float64))
        np.testing.assert_allclose(op, g[f"s{seed}.hand_pose"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ot, g[f"s{seed}.hand_tsl"], rtol=1e-6, atol=1e-7)
        pp, pt = PoseGenerator.scramble(torch.from_numpy(pose), torch.from_numpy(tsl), torch.from_numpy(ra), torch.from_numpy(rt))
        np.testing.assert_allclose(pp.numpy(), g[f"s{seed}.hand_pose"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(pt.numpy(), g[f"s{seed}.hand_tsl"], rtol=1e-6, atol=1e-7)
    assert np.allclose(g["s11.hand_pose"][0, 3:6], 0.0)        # a zero rotation stays zero (axis 0 / clip)


def test_pose_generator_oracle_matches_reference_class(golden_dir):
    """pose_oracle.pose_generator against the reference's own PreProcessorPoseGenerator.forward + RandomScrambler, run with
    stand-ins for manotorch / pytorch3d / the refiner only (tests/golden/posegen.npz, oracle/gen_posegen_golden.py): the frame
    algebra, offsets and ordering of preprocessor.py:20-99."""
    import pose_oracle as po
    from artiboost_amd.assets import make_hand_model
    g = np.load(os.path.join(golden_dir, "posegen.npz"))
    hm = make_hand_model(int(g["hand_model_seed"]))
    f64 = lambda k: g[k].astype(np.float64)      # noqa: E731
    obj_pose, verts, joints = po.pose_generator(hm, f64("hand_pose"), f64("hand_shape"), f64("hand_tsl"), f64("persp_rotmat"),
                                                f64("camera_free_transf"), f64("z_offset"), rand_pose_angle=f64("rand_angle"),
                                                rand_tsl=f64("rand_tsl"))
    np.testing.assert_allclose(obj_pose, g["final_obj_pose"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(verts, g["final_hand_verts"], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(joints, g["final_joints"], rtol=1e-5, atol=5e-6)


def test_simplebaseline_oracle_matches_reference_class(golden_dir):
    """lo.simple_forward (SimpleBaseline, simplebaseline.py:194-241, on a ResNet-18 backbone) against the reference's own class run on
    the same seeded weights (tests/golden/simplebaseline.npz, oracle/gen_simplebaseline_golden.py): eval and train predictions, the
    JointsLoss value, every parameter's gradient norm."""
    import torch
    from gen_batch import make_batch
    g = np.load(os.path.join(golden_dir, "simplebaseline.npz"))
    size, heat, depth, B, seed = [int(x) for x in g["meta"]]
    shapes = lo.param_shapes(29, depth, layers=(2, 2, 2, 2), head_prefix="pose_head", box_head=False)
    params = lo.fill_params(shapes, seed=seed)
    batch = make_batch(B, size, seed + 100)
    batch["corners_3d"] = torch.from_numpy(g["corners_3d"])
    with torch.no_grad():
        pe = lo.simple_forward(params, batch, [size, size], 29, depth, 0, training=False, layers=(2, 2, 2, 2))
    for k in ("joints_3d_abs", "corners_3d_abs", "2d_uvd"):
        np.testing.assert_allclose(pe[k].numpy(), g[f"eval.pred.{k}"], rtol=0, atol=2e-5, err_msg=k)
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in params.items()}
    pt = lo.simple_forward(leaf, batch, [size, size], 29, depth, 0, training=True, layers=(2, 2, 2, 2))
    for k in ("joints_3d_abs", "corners_3d_abs"):
        np.testing.assert_allclose(pt[k].detach().numpy(), g[f"train.pred.{k}"], rtol=0, atol=2e-5, err_msg=k)
    total, _ = lo.joints_loss(pt, batch)
    np.testing.assert_allclose(float(total), float(g["loss.total"]), rtol=1e-4)
    total.backward()
    ref = dict(zip([str(n) for n in g["grad.names"]], g["grad.norms"]))
    for n, r in ref.items():
        if n.startswith("backbone.fc"):
            continue
        np.testing.assert_allclose(float(leaf[n].grad.norm()), r, rtol=2e-3, atol=1e-9, err_msg=n)


def test_bottleneck_oracle_matches_reference_resnet50(golden_dir):
    """lo.hybrid_forward(bottleneck=True) -- HybridBaseline on ResNet50 (resnet.py:104-141,252-258) -- against the reference's own class on
    seeded weights (tests/golden/resnet50_hybrid.npz, oracle/gen_resnet50_golden.py)."""
    import torch
    from gen_batch import make_batch
    g = np.load(os.path.join(golden_dir, "resnet50_hybrid.npz"))
    size, heat, depth, B, seed = [int(x) for x in g["meta"]]
    params = lo.fill_params(lo.param_shapes(22, depth, bottleneck=True), seed=seed)
    batch = make_batch(B, size, seed + 100)
    with torch.no_grad():
        pe = lo.hybrid_forward(params, batch, [size, size], 22, depth, 0, training=False, bottleneck=True)
    for k in ("joints_3d_abs", "corners_3d_abs", "2d_uvd"):
        np.testing.assert_allclose(pe[k].numpy(), g[f"eval.pred.{k}"], rtol=0, atol=3e-5, err_msg=k)
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in params.items()}
    pt = lo.hybrid_forward(leaf, batch, [size, size], 22, depth, 0, training=True, bottleneck=True)
    for k in ("joints_3d_abs", "corners_3d_abs"):
        np.testing.assert_allclose(pt[k].detach().numpy(), g[f"train.pred.{k}"], rtol=0, atol=3e-5, err_msg=k)
    total, _ = lo.joints_loss(pt, batch)
    np.testing.assert_allclose(float(total), float(g["loss.total"]), rtol=1e-4)
    total.backward()
    ref = dict(zip([str(n) for n in g["grad.names"]], g["grad.norms"]))
    for n, r in ref.items():
        if not n.startswith("backbone.fc"):
            np.testing.assert_allclose(float(leaf[n].grad.norm()), r, rtol=1e-2, atol=1e-9, err_msg=n)   # 53 BatchNorms at batch size 2: fp32 summation order shows

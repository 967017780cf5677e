"""GPU parity of the whole learner (HybridBaseline forward, losses, backward, clip+Adam) against the REAL reference's
golden vectors (tests/golden/learner_*.npz) and against the CPU oracle on the same seeded inputs."""
import os
import random

import numpy as np
import pytest
import torch

import learner_oracle as lo
from gen_batch import make_batch

pytestmark = pytest.mark.gpu

REF_YAML_ARCH = {
    "TYPE": "HybridBaseline", "PRETRAINED": "",
    "BACKBONE": {"TYPE": "ResNet34", "PRETRAINED": False, "FREEZE_BATCHNORM": False},
    "HYBRID_HEAD": {"TYPE": "IntegralDeconvHead", "NCLASSES": 22, "DECONV_WITH_BIAS": False, "NORM_TYPE": "softmax",
                    "INPUT_CHANNEL": 512, "DEPTH_RESOLUTION": 28, "NUM_DECONV_LAYERS": 2,
                    "NUM_DECONV_FILTERS": [256, 256], "NUM_DECONV_KERNELS": [4, 4], "FINAL_CONV_KERNEL": 1},
    "BOX_HEAD": {"TYPE": "MLP_O", "LAYERS_N": [512, 256, 128], "OUT_CHANNEL": 6},
    "PREVIOUS": [],
}


def build(size, heat, dtype, seed, **arch_extra):
    from artiboost_amd import registry as R
    from artiboost_amd.models import Arch
    from artiboost_amd.criterions import Criterion
    import artiboost_amd.criterions  # noqa: F401 (registers losses)
    preset = {"IMAGE_SIZE": [size, size], "HEATMAP_SIZE": [heat, heat], "CENTER_IDX": 0}
    arch_cfg = dict(REF_YAML_ARCH, COMPUTE_DTYPE=dtype, **arch_extra)
    cfg = {"ARCH": arch_cfg, "LAMBDAS": [0.5, 0.2, 0.1],
           "CRITERION": [{"TYPE": "JointsLoss", "LAMBDA_JOINTS_3D": 1.0, "LAMBDA_CORNERS_3D": 0.2},
                         {"TYPE": "HandOrdLoss"}, {"TYPE": "SceneOrdLoss"}]}
    model = Arch(cfg, R.build_arch_model_list(arch_cfg, preset_cfg=preset))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=preset, LAMBDAS=cfg["LAMBDAS"]))
    params = lo.fill_params(lo.param_shapes(22, 28), seed=seed)
    model.model_list[0].load_state_dict(params)
    return model, crit, params


# Prediction tolerances (absolute; metres for joints / corners, heat-map units for 2d_uvd).  "bf16x3" -- split-bf16 MFMA, the
# benchmarked precision -- is held to the exact-f32 path's tolerance on the losses (3e-4), to 1.5 % on the gradient norms
# (f32: 1 %) and on the predictions to a bound 5x below the north star's 1e-3.  Measured against the reference goldens on MI355X
# (tests/parity_report.py, final build of round 2 -- block outputs held as (hi, lo) planes): train-mode joints 3.6e-7 m / corners 7.5e-6 m
# (f32: 1.2e-7 / 2.6e-7); eval mode on this deliberately ill-conditioned random-weight net (running statistics 0 / 1) joints 1.0e-4 m,
# 2d_uvd 2.5e-4 (f32: 2.2e-5 m, 5.6e-5; bf16: 9.5e-2 m, 2.4e-1).  With fp32 copies of the block outputs (AB_RES_PLANES=0): 1.2e-4 m, 3.1e-4.
# gsamp: element-wise check of sampled weight-gradient entries, as a fraction of the sample's largest entry (the gradient
# of the stem passes through 33 BatchNorms at batch size 2: its small entries are the most rounding-sensitive numbers here)
# gnorm: relative bound on every parameter tensor's gradient norm (110 tensors).  Measured worst case on MI355X: f32 0.47 %,
# bf16x3 0.75 % (1.18 % with AB_RES_PLANES=0: one BatchNorm bias of layer 1 in the batch-size-2 g224 golden), bf16 21 %.
PRED_TOL = {"f32": dict(eval=3e-5, train=3e-5, metres=3e-5, logits=2e-4, gsamp=3e-2, gnorm=1e-2),
            "bf16x3": dict(eval=5e-4, train=1e-4, metres=2e-4, logits=3e-3, gsamp=6e-2, gnorm=1.5e-2)}


@pytest.mark.parametrize("route", ["fused", "registry"])
@pytest.mark.parametrize("dtype", ["bf16x3", "f32"])
@pytest.mark.parametrize("tag", ["g224", "g256"])
def test_f32_path_matches_reference_golden(golden_dir, tag, dtype, route):
    """route: `Criterion.compute_losses` through the fused pose/loss kernel (the default when the predictions come from the
    HIP model) or through the registry losses' torch ops + autograd."""
    g = np.load(os.path.join(golden_dir, f"learner_{tag}.npz"))
    size, heat, depth, B, seed = [int(x) for x in g["meta"]]
    model, crit, params = build(size, heat, dtype, seed)
    crit.fused_route = route == "fused"
    hb = model.model_list[0]
    batch = make_batch(B, size, seed + 100)
    # ---- eval
    model.eval()
    with torch.no_grad():
        preds = model(batch)["HybridBaseline"]
    T = PRED_TOL[dtype]
    for k in ("joints_3d_abs", "corners_3d_abs", "joints_3d", "corners_3d", "2d_uvd", "boxroot_3d_abs", "box_rot_rotmat"):
        atol = T["metres"] if k.startswith(("joints", "corners", "boxroot")) else T["eval"]
        np.testing.assert_allclose(preds[k].cpu().numpy(), g[f"eval.pred.{k}"], rtol=1e-4, atol=atol, err_msg=k)
    lg = hb.net.last["logits"].float().cpu().reshape(B, heat, heat, 22, 32)[..., :28].permute(0, 3, 4, 1, 2).reshape(B, 616, heat, heat)
    np.testing.assert_allclose(lg[:, ::37, ::5, ::5].numpy(), g["eval.logits.sample"], rtol=2e-3, atol=T["logits"])   # |logits| ~ 60
    # ---- train: forward, losses with the reference's RNG seeding, backward
    model.train()
    preds = model(batch)["HybridBaseline"]
    for k in ("joints_3d_abs", "corners_3d_abs", "2d_uvd", "box_rot_rotmat"):
        atol = min(T["metres"], T["train"]) if k.startswith(("joints", "corners")) else T["train"]
        np.testing.assert_allclose(preds[k].detach().cpu().numpy(), g[f"train.pred.{k}"], rtol=1e-4, atol=atol, err_msg=k)
    random.seed(seed + 7)
    torch.manual_seed(seed + 7)
    total, losses = crit.compute_losses(preds, batch)
    assert (type(total.grad_fn).__name__ == "_FusedLossFnBackward") == (route == "fused")
    for k in ("joints_3d_loss", "corners_3d_loss", "joint_ord_loss", "part_ord_loss", "scene_ord_loss", "final_loss"):
        np.testing.assert_allclose(losses[k].detach().cpu().numpy().reshape(-1), g[f"loss.{k}"].reshape(-1),
                                   rtol=3e-4, atol=1e-7, err_msg=k)
    total.backward()
    grads = hb.store.reference_state_dict(grads=True)
    ref = dict(zip([str(n) for n in g["grad.names"]], g["grad.norms"]))
    for n, r in ref.items():
        got = float(grads[n].norm())
        assert abs(got - r) <= T["gnorm"] * r + 1e-9, (n, got, r)
    for got, ref_ in ((grads["hybrid_head.final_layer.bias"], g["grad.final_bias"]), (grads["box_head.layers.4.weight"], g["grad.box4.weight"])):
        # every element of two whole gradient tensors; bf16x3: + 1e-3 of the tensor's largest entry for the near-zero ones
        np.testing.assert_allclose(got.cpu().numpy(), ref_, rtol=3e-3, atol=1e-9 if dtype == "f32" else 1e-3 * np.abs(ref_).max())
    rs = g["grad.conv1.sample"]
    np.testing.assert_allclose(grads["backbone.conv1.weight"][::8, :, ::3, ::3].cpu().numpy(), rs, rtol=2e-2, atol=T["gsamp"] * np.abs(rs).max())
    rs = g["grad.deconv3.sample"]
    np.testing.assert_allclose(grads["hybrid_head.deconv_layers.3.weight"][::32, ::32].cpu().numpy(), rs, rtol=2e-2, atol=T["gsamp"] * np.abs(rs).max())
    rs = g["grad.l3.0.ds.sample"]
    np.testing.assert_allclose(grads["backbone.layer3.0.downsample.0.weight"][::16, ::16, 0, 0].cpu().numpy(), rs, rtol=2e-2, atol=T["gsamp"] * np.abs(rs).max())
    sd = hb.state_dict()
    np.testing.assert_allclose(sd["backbone.bn1.running_var"].numpy(), g["stat.bn1.running_var"], rtol=1e-4)
    np.testing.assert_allclose(sd["backbone.layer4.2.bn2.running_var"].numpy(), g["stat.l4.2.bn2.running_var"], rtol=1e-3)
    assert float(grads["backbone.fc.weight"].abs().max()) == 0.0


BF16_TOL = 3e-2   # metres.  Measured 1.7e-2 (GPU) / 2.1e-2 (CPU emulation rounding every conv/BN output to bf16) on this
# deliberately ill-conditioned random-weight net; fp16/TF32-class rounding gives 7e-3 on the same input, i.e. the
# north-star 1e-3 is only reachable with f32 arithmetic, which test_f32_path_matches_reference_golden demonstrates.


def test_bf16_path_tolerance(golden_dir):
    """bf16 conv operands / activations, fp32 accumulate (the bench configuration): bounded deviation from the
    reference, loss within 2 %, every gradient norm within 35 % (B=2 batch statistics amplify rounding)."""
    g = np.load(os.path.join(golden_dir, "learner_g256.npz"))
    size, heat, depth, B, seed = [int(x) for x in g["meta"]]
    model, crit, params = build(size, heat, "bf16", seed)
    batch = make_batch(B, size, seed + 100)
    model.eval()
    with torch.no_grad():
        preds = model(batch)["HybridBaseline"]
    for k in ("joints_3d_abs", "corners_3d_abs"):
        np.testing.assert_allclose(preds[k].cpu().numpy(), g[f"eval.pred.{k}"], rtol=0, atol=BF16_TOL, err_msg=k)
    model.train()
    preds = model(batch)["HybridBaseline"]
    for k in ("joints_3d_abs", "corners_3d_abs"):
        np.testing.assert_allclose(preds[k].detach().cpu().numpy(), g[f"train.pred.{k}"], rtol=0, atol=BF16_TOL, err_msg=k)
    random.seed(seed + 7)
    torch.manual_seed(seed + 7)
    total, losses = crit.compute_losses(preds, batch)
    np.testing.assert_allclose(float(total), float(g["loss.final_loss"].reshape(-1)[0]), rtol=2e-2)
    total.backward()
    grads = model.model_list[0].store.reference_state_dict(grads=True)
    ref = dict(zip([str(n) for n in g["grad.names"]], g["grad.norms"]))
    bad = [(n, float(grads[n].norm()), r) for n, r in ref.items() if abs(float(grads[n].norm()) - r) > 0.35 * r + 1e-9]
    assert not bad, bad[:5]


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16"])
def test_segment_graphs_equal_eager_loop(dtype):
    """The reference-shaped loop (model(batch) -> compute_losses -> backward -> clip_grad_norm_ -> optimizer.step) with the
    network halves replayed as hipGraphs (models._NetSegment, the default) is the same sequence of updates, bit for bit, as
    the kernel-by-kernel eager loop; eval-mode forwards in between do not disturb it.  (torch's deterministic algorithms
    are switched on for the comparison: the autograd backward of the registry losses otherwise accumulates with float atomics
    and two EAGER runs already differ in the last bits.)"""
    from artiboost_amd.netutils import build_optimizer
    size, heat, B, seed, steps = 64, 8, 4, 11, 5
    runs = {}
    det = torch.are_deterministic_algorithms_enabled()
    torch.use_deterministic_algorithms(True)
    try:
        _segment_runs(runs, size, heat, B, seed, steps, dtype, build_optimizer)
    finally:
        torch.use_deterministic_algorithms(det)
    assert runs[False][0] == runs[True][0]
    assert all(torch.equal(a, b) for a, b in zip(runs[False][1], runs[True][1]))
    assert torch.equal(runs[False][2], runs[True][2]) and torch.equal(runs[False][3], runs[True][3])


def _segment_runs(runs, size, heat, B, seed, steps, dtype, build_optimizer):
    for seg in (False, True):
        model, crit, _ = build(size, heat, dtype, seed, SEGMENT_GRAPHS=seg)
        hb = model.model_list[0]
        assert hb.segment_graphs is seg
        opt = build_optimizer(model.models_params, OPTIMIZER="adam", LR=1e-4, WEIGHT_DECAY=0)
        losses, evals = [], []
        for it in range(steps):
            batch = make_batch(B, size, seed + 100 + it)
            model.train()
            preds = model(batch)["HybridBaseline"]
            random.seed(seed + it)
            torch.manual_seed(seed + it)
            total, _ = crit.compute_losses(preds, batch)
            opt.zero_grad()
            total.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 0.001)
            opt.step()
            losses.append(float(total))
            model.eval()
            with torch.no_grad():
                evals.append(model(make_batch(B, size, seed + 500 + it))["HybridBaseline"]["joints_3d_abs"].cpu())
        if seg:
            segs = list(hb._segments.values())
            assert len(segs) == 2 and all(s.fwd is not None for s in segs) and any(s.bwd is not None for s in segs)
        runs[seg] = (losses, evals, hb.store.flat.detach().cpu().clone(), hb.store.stats.cpu().clone())


@pytest.mark.parametrize("g,B", [(224, 2), (256, 3)])
def test_eval_forward_with_folded_batchnorm_is_bit_identical(g, B, monkeypatch):
    """Eval-mode forward (BASELINE configs[1]: submit_reload.py / the TEST pass, resnet.py:85-101 under eval()) with the
    BatchNorm of every 3x3/s1 convolution folded into the conv epilogue (ab_conv2d_fwd_x3_evalbn: the fp32 conv outputs are never
    stored) == conv + ab_bn_apply_x3 as separate launches, bit for bit -- after a few training steps, so that running statistics,
    scales and shifts are non-trivial."""
    from artiboost_amd import kernels as K
    size, heat, seed = g, g // 8, 5
    model, crit, _ = build(size, heat, "bf16x3", seed, SEGMENT_GRAPHS=False)
    hb = model.model_list[0]
    model.train()
    for it in range(2):                     # move the running statistics away from (0, 1)
        model(make_batch(B, size, seed + it))
    model.eval()
    batch = make_batch(B, size, seed + 50)
    calls = []
    orig = K.conv2d_fwd_x3_evalbn
    monkeypatch.setattr(K, "conv2d_fwd_x3_evalbn", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    outs = {}
    for fold in (True, False):
        hb.net.eval_fold = fold
        with torch.no_grad():
            p = model(batch)["HybridBaseline"]
        outs[fold] = {k: v.detach().cpu().clone() for k, v in p.items()}
        outs[fold]["logits"] = hb.net.last["logits"].detach().cpu().clone()
        outs[fold]["feat"] = hb.net.last["feat"].detach().float().cpu().clone()
    assert len(calls) == 29                 # 32 3x3/s1 convolutions of ResNet-34 minus conv1 of the three down-sampling blocks
    for k in outs[True]:
        assert torch.equal(outs[True][k], outs[False][k]), k
    assert torch.isfinite(outs[True]["joints_3d_abs"]).all()


def test_eval_pose_assembly_kernel_matches_the_module_arithmetic():
    """M4 in eval mode: ab_pose_assemble (one launch) == the tensor-op restatement of hybridbaseline.py:49-96 in models.forward, for all
    nine output keys, on a row-pitched box-head output and non-trivial intrinsics."""
    size, heat, seed, B = 64, 8, 9, 5
    model, _, _ = build(size, heat, "bf16x3", seed, SEGMENT_GRAPHS=False)
    hb = model.model_list[0]
    model.eval()
    batch = make_batch(B, size, seed + 1)
    outs = {}
    for fused in (True, False):
        hb.fused_assembly = fused
        with torch.no_grad():
            outs[fused] = {k: v.detach().float().cpu().numpy() for k, v in model(batch)["HybridBaseline"].items()}
    assert set(outs[True]) == set(outs[False])
    for k in outs[False]:
        assert outs[True][k].shape == outs[False][k].shape, k
        np.testing.assert_allclose(outs[True][k], outs[False][k], rtol=2e-6, atol=2e-7, err_msg=k)


def test_frozen_batchnorm_backbone_matches_oracle():
    """BACKBONE.FREEZE_BATCHNORM: true (anakin/models/resnet.py:33-69,146-149: FrozenBatchNorm2d in every backbone slot) in TRAINING mode:
    the backbone BatchNorms are fixed affine maps of the running statistics, the head's stay ordinary BatchNorm2d.  Forward, losses and the
    gradient of every parameter vs the torch-CPU oracle with the same semantics; frozen weight / bias get zero gradient, running statistics
    of the backbone do not move, the state dict carries no num_batches_tracked for them."""
    size, heat, seed, B = 64, 8, 21, 4
    model, crit, params = build(size, heat, "bf16x3", seed, SEGMENT_GRAPHS=False,
                                BACKBONE={"TYPE": "ResNet34", "PRETRAINED": False, "FREEZE_BATCHNORM": True})
    hb = model.model_list[0]
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.clone() for k, v in params.items()}
    for k in sd:                                            # non-trivial frozen statistics and affine parameters
        if k.startswith("backbone.") and k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
        elif k.startswith("backbone.") and k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
    hb.load_state_dict(sd)
    stats0 = hb.store.stats.clone()
    batch = make_batch(B, size, seed + 100)
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
    random.seed(3); torch.manual_seed(3)
    preds_r = lo.hybrid_forward(leaf, batch, [size, size], 22, 28, 0, training=True, frozen_bn=True)
    total_r, losses_r, _ = lo.criterion(preds_r, batch)
    total_r.backward()
    model.train()
    preds = model(batch)["HybridBaseline"]
    random.seed(3); torch.manual_seed(3)
    total, losses = crit.compute_losses(preds, batch)
    total.backward()
    for k in ("joints_3d_abs", "corners_3d_abs"):
        np.testing.assert_allclose(preds[k].detach().cpu().numpy(), preds_r[k].detach().numpy(), rtol=0, atol=2e-4, err_msg=k)
    np.testing.assert_allclose(float(total), float(total_r), rtol=3e-4)
    grads = hb.store.reference_state_dict(grads=True)
    bad = []
    for k, v in leaf.items():
        if not (v.dtype.is_floating_point and v.requires_grad) or k.startswith("backbone.fc"):
            continue
        gr = v.grad if v.grad is not None else torch.zeros_like(v)
        if k.startswith("backbone.") and (".bn" in k or "downsample.1" in k):
            assert float(grads[k].abs().max()) == 0.0, k                       # frozen: buffers in the reference, zero gradient here
            continue
        a, r = float(grads[k].cpu().norm()), float(gr.norm())
        if abs(a - r) > 1.5e-2 * r + 1e-12:
            bad.append((k, a, r))
    assert not bad, bad[:5]
    # the backbone's running statistics did not move; the head's (ordinary BatchNorm2d) did
    off = {k: hb.store.buffers[k] for k in hb.store.buffers}
    for k, (o, c) in off.items():
        moved = not torch.equal(hb.store.stats[o:o + c], stats0[o:o + c])
        assert moved == (not k.startswith("backbone.")), k
    keys = set(hb.state_dict())
    assert "backbone.layer1.0.bn1.num_batches_tracked" not in keys and "hybrid_head.deconv_layers.1.num_batches_tracked" in keys


def test_simplebaseline_resnet18_vs_reference_golden(golden_dir):
    """SimpleBaseline (simplebaseline.py:194-241: 29 heat-map classes, no box head) on a ResNet-18 backbone (resnet.py:236-241) through the
    same HIP executor (`pose_head.*` parameters, a zero-weight padding class, registry JointsLoss + autograd) against the reference's own
    class on the same seeded weights (tests/golden/simplebaseline.npz): eval / train predictions, the loss, every gradient norm."""
    from artiboost_amd import registry as R
    from artiboost_amd.models import Arch
    g = np.load(os.path.join(golden_dir, "simplebaseline.npz"))
    size, heat, depth, B, seed = [int(x) for x in g["meta"]]
    arch = {"TYPE": "SimpleBaseline", "PRETRAINED": "", "PREVIOUS": [], "COMPUTE_DTYPE": "bf16x3", "SEGMENT_GRAPHS": False,
            "BACKBONE": {"TYPE": "ResNet18", "PRETRAINED": False, "FREEZE_BATCHNORM": False},
            "HEAD": {"TYPE": "IntegralDeconvHead", "NCLASSES": 29, "DECONV_WITH_BIAS": False, "NORM_TYPE": "softmax", "INPUT_CHANNEL": 512,
                     "DEPTH_RESOLUTION": depth, "NUM_DECONV_LAYERS": 2, "NUM_DECONV_FILTERS": [256, 256], "NUM_DECONV_KERNELS": [4, 4],
                     "FINAL_CONV_KERNEL": 1}}
    preset = {"IMAGE_SIZE": [size, size], "HEATMAP_SIZE": [heat, heat], "CENTER_IDX": 0}
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=preset))
    sb = model.model_list[0]
    shapes = lo.param_shapes(29, depth, layers=(2, 2, 2, 2), head_prefix="pose_head", box_head=False)
    sb.load_state_dict(lo.fill_params(shapes, seed=seed))
    assert list(sb.state_dict()) == [k for k, _ in shapes]
    batch = make_batch(B, size, seed + 100)
    batch["corners_3d"] = torch.from_numpy(g["corners_3d"])
    tol = PRED_TOL["bf16x3"]
    model.eval()
    with torch.no_grad():
        pe = model(batch)["SimpleBaseline"]
    assert set(pe) == {"joints_3d_abs", "corners_3d_abs", "joints_3d", "corners_3d", "2d_uvd"}
    for k in ("joints_3d_abs", "corners_3d_abs"):
        np.testing.assert_allclose(pe[k].cpu().numpy(), g[f"eval.pred.{k}"], rtol=0, atol=tol["eval"], err_msg=k)
    model.train()
    pt = model(batch)["SimpleBaseline"]
    for k in ("joints_3d_abs", "corners_3d_abs"):
        np.testing.assert_allclose(pt[k].detach().cpu().numpy(), g[f"train.pred.{k}"], rtol=0, atol=tol["train"], err_msg=k)
    dev = pt["joints_3d_abs"].device
    tb = {k: v.to(dev) for k, v in batch.items()}
    tj = (tb["joints_3d"] + tb["root_joint"][:, None]) * tb["joints_vis"][..., None]
    tc = (tb["corners_3d"] + tb["root_joint"][:, None]) * tb["corners_vis"][..., None]
    total = (torch.nn.functional.mse_loss(pt["joints_3d_abs"] * tb["joints_vis"][..., None], tj)
             + 0.2 * torch.nn.functional.mse_loss(pt["corners_3d_abs"] * tb["corners_vis"][..., None], tc))
    np.testing.assert_allclose(float(total), float(g["loss.total"]), rtol=3e-4)
    total.backward()
    grads = sb.store.reference_state_dict(grads=True)
    ref = dict(zip([str(n) for n in g["grad.names"]], g["grad.norms"]))
    bad = [(n, float(grads[n].norm()), r) for n, r in ref.items()
           if not n.startswith("backbone.fc") and abs(float(grads[n].norm()) - r) > tol["gnorm"] * r + 1e-12]
    assert not bad, bad[:5]


@pytest.mark.parametrize("dtype", ["f32", "bf16x3"])
def test_bottleneck_resnet50_hybridbaseline_vs_reference_golden(golden_dir, dtype):
    """HybridBaseline on the Bottleneck backbone ResNet50 (resnet.py:104-141,252-258; head on 2048 channels, MLP_O [2048, 256, 128]) through
    the same kernels (1x1 convolutions on the generic implicit GEMM, the strided 3x3 in conv2) against the reference's own class on seeded
    weights (tests/golden/resnet50_hybrid.npz): eval / train predictions, the loss, every gradient norm."""
    from artiboost_amd import registry as R
    from artiboost_amd.models import Arch
    g = np.load(os.path.join(golden_dir, "resnet50_hybrid.npz"))
    size, heat, depth, B, seed = [int(x) for x in g["meta"]]
    arch = dict(REF_YAML_ARCH, COMPUTE_DTYPE=dtype, SEGMENT_GRAPHS=False, BACKBONE={"TYPE": "ResNet50", "PRETRAINED": False, "FREEZE_BATCHNORM": False},
                HYBRID_HEAD=dict(REF_YAML_ARCH["HYBRID_HEAD"], INPUT_CHANNEL=2048, DEPTH_RESOLUTION=depth),
                BOX_HEAD={"TYPE": "MLP_O", "LAYERS_N": [2048, 256, 128], "OUT_CHANNEL": 6})
    preset = {"IMAGE_SIZE": [size, size], "HEATMAP_SIZE": [heat, heat], "CENTER_IDX": 0}
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=preset))
    hb = model.model_list[0]
    shapes = lo.param_shapes(22, depth, bottleneck=True)
    hb.load_state_dict(lo.fill_params(shapes, seed=seed))
    assert list(hb.state_dict()) == [k for k, _ in shapes]
    batch = make_batch(B, size, seed + 100)
    # 53 BatchNorms at batch size 2 on random weights: every rounding is amplified layer by layer (the imported reference and its
    # CPU restatement already differ by 0.4 % in single gradient norms).  The exact-f32 path shows the orchestration is right
    # (3e-5 m); bf16x3 is held to the 5e-4 m of its eval tolerance in both modes here -- 2x inside the north star's 1e-3.
    tol = dict(PRED_TOL[dtype], train=(3e-5 if dtype == "f32" else 5e-4), gnorm=3e-2)        # (f32 measured 1.7 % on two BatchNorm parameters of layer1.0, bf16x3 inside 3 %)
    model.eval()
    with torch.no_grad():
        pe = model(batch)["HybridBaseline"]
    for k in ("joints_3d_abs", "corners_3d_abs"):
        np.testing.assert_allclose(pe[k].cpu().numpy(), g[f"eval.pred.{k}"], rtol=0, atol=tol["eval"], err_msg=k)
    model.train()
    pt = model(batch)["HybridBaseline"]
    for k in ("joints_3d_abs", "corners_3d_abs"):
        np.testing.assert_allclose(pt[k].detach().cpu().numpy(), g[f"train.pred.{k}"], rtol=0, atol=tol["train"], err_msg=k)
    dev = pt["joints_3d_abs"].device
    tb = {k: v.to(dev) for k, v in batch.items()}
    total, _ = lo.joints_loss({k: pt[k] for k in ("joints_3d_abs", "corners_3d_abs")}, tb)
    np.testing.assert_allclose(float(total), float(g["loss.total"]), rtol=2e-3)
    total.backward()
    grads = hb.store.reference_state_dict(grads=True)
    ref = dict(zip([str(n) for n in g["grad.names"]], g["grad.norms"]))
    bad = [(n, float(grads[n].norm()), r) for n, r in ref.items()
           if not n.startswith("backbone.fc") and abs(float(grads[n].norm()) - r) > tol["gnorm"] * r + 1e-12]
    assert not bad, bad[:5]

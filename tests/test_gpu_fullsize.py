"""Full-size (BASELINE.json geometry: B = 64, 256 x 256) parity that is not a self-comparison: the whole bf16x3 training
step against the torch-CPU fp32 oracle on the SAME rendered batch, and the end-to-end determinism check at full size."""
import os
import random

import numpy as np
import pytest
import torch
import yaml

import learner_oracle as lo

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_image(xpad, plane, size):
    """The oracle's input image (x / 255 - 0.5, fp32 CHW) from the padded tensor the GPU step trained on: the fp32 image as is; the
    integer plane 2 v - 255 (what bench.py's headline runs on) divided by 510 -- the same pixel values to fp32 rounding."""
    x = xpad.float().cpu()
    if plane == "u8n":
        x = x / 510.0
    return x[:, 3:3 + size, 3:3 + size, :3].permute(0, 3, 1, 2).contiguous()


def test_the_benchmarked_image_plane_is_one_of_the_pinned_ones():
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.DEFAULT_IMAGE_PLANE in PLANES and bench.DEFAULT_IMAGE_PLANE == "u8n"


PLANES = ("f32", "u8n")


@pytest.mark.parametrize("image_plane", PLANES)
@pytest.mark.parametrize("dataset,cfg_name", [("HO3D", "ho3dv2_clasbased_artiboost_mi355x.yaml"),
                                              ("DexYCB", "dexycb_clasbased_sym_mi355x.yaml")])
def test_full_size_train_steps_match_cpu_oracle(dataset, cfg_name, image_plane):
    """Two graph-replayed bf16x3 steps of the benchmark workload (render -> forward -> fused criterion -> backward -> clip +
    Adam) vs learner_oracle (torch-CPU fp32 restatement pinned to the reference goldens) fed the images the GPU rendered:
    every loss term of both steps within 3e-4 relative (step 2 sees step 1's update), the pre-clip gradient norm within 1 %.
    Two workloads: BASELINE configs[2] (HO3D-like objects, 3 losses) and configs[4] on one GPU (DexYCB-like: 21 objects at 16 k
    faces, + SymCornerLoss from the DexYCB config, criterions/symcornerloss.py:18-102); each on both image planes the loader can hand the
    bf16x3 stem -- the fp32 image (three MFMA passes) and the integer plane "u8n" (two passes: bench.py's configuration)."""
    from artiboost_amd import registry as R
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.synth import ArtiBoostLoader
    from artiboost_amd.train import TrainStep
    B, size, lr, clip = 64, 256, 5e-5, 0.001
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", cfg_name)))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [size, size], [size // 8, size // 8]
    arch = dict(cfg["ARCH"], COMPUTE_DTYPE="bf16x3", INIT_SEED=3)
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=lr, max_norm=clip, model=hb)
    loader = ArtiBoostLoader.from_assets(SceneAssets(dataset, seed=1), cfg["MANAGER"], cfg["DATA_PRESET"], B, 2 * B,
                                         compute_dtype=torch.float32 if image_plane == "f32" else "u8n", random_seed=3)
    sym = next((l for l in crit.loss_list if type(l).__name__ == "SymCornerLoss"), None)
    assert (sym is not None) == (dataset == "DexYCB")
    lam = cfg["LAMBDAS"]
    loader.prepare()
    params0 = {k: v.clone() for k, v in hb.state_dict().items()}
    static = loader.new_static_batch()
    loader.load_batch(static, 0)
    model.train()
    ts = TrainStep(model, crit, opt, static, use_graph=True, renderer=loader)
    ts.static = static
    assert (ts.fused.sym is not None) == (sym is not None)
    assert hb.net.image_plane == image_plane == loader.image_plane
    assert static["image_nhwc4_padded"].dtype == (torch.bfloat16 if image_plane == "u8n" else torch.float32)
    # ---- oracle state
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in params0.items()}
    names = ms = vs = None
    for step in range(2):
        loader.load_batch(static, step)
        random.seed(100 + step); torch.manual_seed(100 + step)
        _, losses, _ = ts()
        got = {k: float(v) for k, v in zip(ts.fused.LOSS_KEYS, losses.float().cpu())}
        gnorm = float(opt.total_norm.cpu())
        # the batch the step just trained on, as the oracle's inputs
        batch = {"image": _oracle_image(static["image_nhwc4_padded"], image_plane, size)}
        for k in ("root_joint", "cam_intr", "corners_can", "joints_3d", "corners_3d", "joints_vis", "corners_vis", "obj_transf"):
            batch[k] = static[k].float().cpu()
        batch["obj_idx"] = static["obj_idx"].cpu()
        for v in leaf.values():
            if getattr(v, "grad", None) is not None:
                v.grad = None
        random.seed(100 + step); torch.manual_seed(100 + step)
        run_stats = {}
        preds = lo.hybrid_forward(leaf, batch, [size, size], 22, 28, 0, training=True, stats=run_stats)
        total, ref, _ = lo.criterion(preds, batch, lambdas=lam[:3])
        if sym is not None:      # Criterion.compute_losses adds LAMBDAS[3] * (LAMBDA_SYM_CORNERS_3D * loss) (criterion.py:57-67)
            sref = lo.sym_corner_loss(preds, batch, sym.R.cpu(), sym.t.cpu())
            total = total + lam[3] * float(sym.lambda_sym_corners_3d) * sref
            ref = dict(ref, final_loss=total)
            got["sym_corners_3d_loss"] = float(ts.fused.losses_dict()["sym_corners_3d_loss"])
            assert abs(got["sym_corners_3d_loss"] - float(sref)) <= 3e-4 * abs(float(sref)) + 1e-9, (step, got["sym_corners_3d_loss"], float(sref))
        total.backward()
        if names is None:
            names = [k for k, v in leaf.items() if v.dtype.is_floating_point and getattr(v, "grad", None) is not None]
            ms = [torch.zeros_like(leaf[k]) for k in names]
            vs = [torch.zeros_like(leaf[k]) for k in names]
        rnorm = float(lo.clip_and_adam([leaf[k].detach() for k in names], [leaf[k].grad for k in names], ms, vs, step + 1, lr=lr, max_norm=clip))
        for k, v in run_stats.items():             # the oracle's running statistics follow the same two training forwards
            leaf[k] = v.detach().clone()
        for k in ts.fused.LOSS_KEYS:
            r = float(ref[k])
            # step 0: same weights on both sides (measured 1e-6).  Later steps follow an Adam update, whose first step is
            # lr * sign(g) for EVERY parameter: rounding-level differences in near-zero gradient entries move whole weights,
            # and the smallest loss term (part_ord_loss ~ 2e-5, a hinge with a handful of active pairs) moves by 2.6e-4 .. 4.7e-4
            # between two equally valid tilings of the same kernel; the large terms stay within 1.6e-4.
            tol = 3e-4 if step == 0 or k != "part_ord_loss" else 1.5e-3
            assert abs(got[k] - r) <= tol * abs(r) + 1e-9, (step, k, got[k], r)
        # step 0: identical weights on both sides.  Step 1 runs on weights that differ by Adam's first update (lr * sign(g): entries whose
        # gradient is rounding-level flip sign between two equally valid summation orders), measured 0.05 - 1.3 % over tile geometries
        assert abs(gnorm - rnorm) <= (1e-2 if step == 0 else 2e-2) * rnorm, (step, gnorm, rnorm)
    # BASELINE configs[1] at ITS size: the eval-mode forward (running statistics of the two steps above, BatchNorm folded into the conv
    # epilogues, ab_pose_assemble) on the last batch vs the oracle in eval mode with ITS running statistics and updated weights
    sd_now = {k: v.clone() for k, v in hb.state_dict().items()}          # the GPU model's weights and running statistics after two steps
    for k in ("backbone.bn1.running_mean", "backbone.layer3.2.bn1.running_var", "hybrid_head.deconv_layers.4.running_var"):
        want = leaf[k].detach().numpy()          # the oracle's (step 2 ran on slightly different weights: a tolerance on the tensor's scale)
        np.testing.assert_allclose(sd_now[k].numpy(), want, rtol=2e-3, atol=2e-3 * float(np.abs(want).max()), err_msg=k)
    model.eval()
    with torch.no_grad():
        pe = model(static)["HybridBaseline"]
        pr = lo.hybrid_forward(sd_now, batch, [size, size], 22, 28, 0, training=False)
    for k in ("joints_3d_abs", "corners_3d_abs"):
        err = float((pe[k].float().cpu() - pr[k]).abs().max())
        assert err <= 2e-4, (k, err)               # metres (well-conditioned statistics here; the random-weight goldens get 5e-4)
    model.train()
    # after two updates the weights moved the same way: compare the update of a large early and a late tensor
    sd = hb.state_dict()
    for k in ("backbone.layer1.0.conv1.weight", "hybrid_head.final_layer.weight"):
        d_gpu, d_ref = (sd[k] - params0[k]).double(), (leaf[k].detach() - params0[k]).double()
        cos = float((d_gpu * d_ref).sum() / (d_gpu.norm() * d_ref.norm() + 1e-30))
        assert cos > 0.98, (k, cos)


def test_full_size_step_is_deterministic(monkeypatch):
    """tests/det_check.py's comparison at the benchmark geometry as a collected test: the graph-replayed bf16x3 step, run twice
    from the same seeds, gives bit-identical losses and weights (single-graph and split-backward capture alike)."""
    import test_gpu_synth as T
    l0, w0 = T._run_steps(monkeypatch, False, nsteps=3, bs=64, size=256, dtype="bf16x3")
    l1, w1 = T._run_steps(monkeypatch, False, nsteps=3, bs=64, size=256, dtype="bf16x3")
    l2, w2 = T._run_steps(monkeypatch, True, nsteps=3, bs=64, size=256, dtype="bf16x3")
    assert np.isfinite(l0).all()
    np.testing.assert_array_equal(l0, l1)
    np.testing.assert_array_equal(w0, w1)
    np.testing.assert_array_equal(l0, l2)
    np.testing.assert_array_equal(w0, w2)


@pytest.mark.parametrize("image_plane", PLANES)
def test_full_size_mixed_step_matches_cpu_oracle(image_plane):
    """The benchmarked `mixed_real_synth_step` (the reference's MixedDataset batch as ONE training step) against the oracle: B = 64 =
    40 real 640 x 480 frames served as .jpg files -- decoded on the device (ab_jpeg_decode_batch), flipped / blurred / jittered / cropped by
    ab_augment_batch -- + 24 samples rendered on the device, through MixedLoader's default schedule (frames of four batches per decode call,
    the next group on a side stream) and the graph-replayed bf16x3 step; learner_oracle is fed the batch the step trained on.  Same bounds
    as the synthetic-only test above: losses 3e-4 (step 2: part_ord_loss 1.5e-3), pre-clip gradient norm 1 % / 2 %."""
    pytest.importorskip("PIL")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_mixed
    from artiboost_amd import registry as R
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.realdata import MixedLoader, RealBatcher
    from artiboost_amd.synth import ArtiBoostLoader
    from artiboost_amd.train import TrainStep
    B, size, lr, clip = 64, 256, 5e-5, 0.001
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [size, size], [size // 8, size // 8]
    random.seed(5); torch.manual_seed(5); np.random.seed(5)
    src = bench_mixed.JpegFileSource(n=1024)
    synth_len = int(0.6 * len(src))
    n_synth = MixedLoader.n_synth_for(B, len(src), synth_len)
    cd = torch.float32 if image_plane == "f32" else "u8n"      # "u8n": what bench.py's mixed_real_synth_step leg (tools/bench_mixed.py) runs
    synth = ArtiBoostLoader.from_assets(SceneAssets("HO3D", seed=1), cfg["MANAGER"], cfg["DATA_PRESET"], n_synth, synth_len, compute_dtype=cd)
    synth.prepare()
    ml = MixedLoader(RealBatcher(src, cfg["DATA_PRESET"], compute_dtype=cd), synth, B)      # decode_group 4, decode_ahead: the defaults
    assert (ml.n_real, ml.n_synth) == (40, 24)
    arch = dict(cfg["ARCH"], COMPUTE_DTYPE="bf16x3", INIT_SEED=3)
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=lr, max_norm=clip, model=hb)
    model.train()
    params0 = {k: v.clone() for k, v in hb.state_dict().items()}
    it = iter(ml)
    first = next(it)
    ts = TrainStep(model, crit, opt, {k: v.clone() for k, v in first.items()}, use_graph=True, renderer=None)
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in params0.items()}
    names = ms = vs = None
    b = first
    for step in range(2):
        random.seed(100 + step); torch.manual_seed(100 + step)
        _, losses, _ = ts(b)
        got = {k: float(v) for k, v in zip(ts.fused.LOSS_KEYS, losses.float().cpu())}
        gnorm = float(opt.total_norm.cpu())
        assert b["is_synth"].tolist() == [False] * 40 + [True] * 24
        batch = {"image": _oracle_image(b["image_nhwc4_padded"], image_plane, size)}
        assert hb.net.image_plane == image_plane
        assert float(batch["image"][:40].std()) > 0.05 and float(batch["image"][40:].std()) > 0.05      # both halves carry pictures
        for k in ("root_joint", "cam_intr", "corners_can", "joints_3d", "corners_3d", "joints_vis", "corners_vis", "obj_transf"):
            batch[k] = b[k].float().cpu()
        batch["obj_idx"] = b["obj_idx"].cpu()
        for v in leaf.values():
            if getattr(v, "grad", None) is not None:
                v.grad = None
        random.seed(100 + step); torch.manual_seed(100 + step)
        run_stats = {}
        preds = lo.hybrid_forward(leaf, batch, [size, size], 22, 28, 0, training=True, stats=run_stats)
        total, ref, _ = lo.criterion(preds, batch, lambdas=cfg["LAMBDAS"][:3])
        total.backward()
        if names is None:
            names = [k for k, v in leaf.items() if v.dtype.is_floating_point and getattr(v, "grad", None) is not None]
            ms = [torch.zeros_like(leaf[k]) for k in names]
            vs = [torch.zeros_like(leaf[k]) for k in names]
        rnorm = float(lo.clip_and_adam([leaf[k].detach() for k in names], [leaf[k].grad for k in names], ms, vs, step + 1, lr=lr, max_norm=clip))
        for k, v in run_stats.items():
            leaf[k] = v.detach().clone()
        for k in ts.fused.LOSS_KEYS:
            r = float(ref[k])
            tol = 3e-4 if step == 0 or k != "part_ord_loss" else 1.5e-3
            assert abs(got[k] - r) <= tol * abs(r) + 1e-9, (step, k, got[k], r)
        assert abs(gnorm - rnorm) <= (1e-2 if step == 0 else 2e-2) * rnorm, (step, gnorm, rnorm)
        b = next(it)

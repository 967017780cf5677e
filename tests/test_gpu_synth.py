"""GPU parity of the synthesis control/data plane: MANO LBS kernel and pose generator vs the numpy oracle, and the
loader's rendered batches vs the CPU oracle renderer on the loader's own epoch records."""
import os

import numpy as np
import pytest
import torch

import pose_oracle as po
import render_oracle as ro

pytestmark = pytest.mark.gpu


def test_mano_lbs_vs_oracle():
    from artiboost_amd.assets import make_hand_model
    from artiboost_amd.synth import ManoLayerHIP
    hm = make_hand_model(1)
    rng = np.random.default_rng(0)
    B = 37
    pose = np.clip(0.4 * rng.standard_normal((B, 48)), -1.5, 1.5).astype(np.float32)
    pose[0] = 0
    betas = (0.5 * rng.standard_normal((B, 10))).astype(np.float32)
    v_ref, j_ref, T_ref = po.mano_lbs(hm, pose, betas)
    layer = ManoLayerHIP(hm)
    v, j, T = layer(torch.from_numpy(pose).cuda(), torch.from_numpy(betas).cuda())
    np.testing.assert_allclose(v.cpu().numpy(), v_ref, rtol=0, atol=2e-6)
    np.testing.assert_allclose(j.cpu().numpy(), j_ref, rtol=0, atol=2e-6)
    np.testing.assert_allclose(T.cpu().numpy(), T_ref, rtol=0, atol=2e-6)


def test_mano_lbs_vs_reference_golden(golden_dir):
    """R1 against the reference's own in-tree MANO forward (tests/golden/mano.npz, oracle/gen_mano_golden.py): fp32 kernel,
    wrist-relative as that layer returns them."""
    from artiboost_amd.assets import make_hand_model
    from artiboost_amd.synth import ManoLayerHIP
    g = np.load(os.path.join(golden_dir, "mano.npz"))
    hm = make_hand_model(int(g["hand_model_seed"]))
    layer = ManoLayerHIP(hm)
    v, j, T = layer(torch.from_numpy(g["pose"].astype(np.float32)).cuda(), torch.from_numpy(g["betas"].astype(np.float32)).cuda())
    v, j = v.cpu().numpy().astype(np.float64), j.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(v - j[:, :1], g["verts_rel_wrist"], rtol=0, atol=3e-6)
    np.testing.assert_allclose(j - j[:, :1], g["joints_rel_wrist"], rtol=0, atol=3e-6)


def test_pose_generator_vs_oracle():
    from artiboost_amd.assets import make_hand_model
    from artiboost_amd.synth import ManoLayerHIP, PoseGenerator
    hm = make_hand_model(1)
    rng = np.random.default_rng(1)
    B = 16
    pose = np.clip(0.3 * rng.standard_normal((B, 48)), -1.2, 1.2)
    shape = np.zeros((B, 10))
    tsl = rng.uniform(-0.05, 0.05, (B, 3))
    Rp = np.stack([po.perspective_from_id(int(p), rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5)) for p in rng.integers(0, 288, B)])
    fr = rng.uniform(0, 2 * np.pi, B)
    Tf = np.tile(np.eye(4), (B, 1, 1))
    Tf[:, 0, 0], Tf[:, 0, 1], Tf[:, 1, 0], Tf[:, 1, 1] = np.cos(fr), -np.sin(fr), np.sin(fr), np.cos(fr)
    z = np.zeros((B, 3)); z[:, 2] = rng.uniform(0.45, 0.55, B)
    dp, dt = 0.1 * rng.standard_normal((B, 16)), 0.01 * rng.standard_normal((B, 3))
    op_r, v_r, j_r = po.pose_generator(hm, pose, shape, tsl, Rp, Tf, z, dp, dt)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()   # noqa: E731
    gen = PoseGenerator(ManoLayerHIP(hm))
    op, v, j = gen(t(pose), t(shape), t(tsl), t(Rp), t(Tf), t(z), t(dp), t(dt))
    np.testing.assert_allclose(op.cpu().numpy(), op_r, rtol=0, atol=5e-6)
    np.testing.assert_allclose(v.cpu().numpy(), v_r, rtol=0, atol=2e-5)
    np.testing.assert_allclose(j.cpu().numpy(), j_r, rtol=0, atol=2e-5)


def _loader(dtype=torch.float32, bs=4, n=8, size=224):
    import yaml, os
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.synth import ArtiBoostLoader
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"] = [size, size]
    assets = SceneAssets("HO3D", seed=1)
    return assets, ArtiBoostLoader.from_assets(assets, cfg["MANAGER"], cfg["DATA_PRESET"], bs, n, compute_dtype=dtype, random_seed=3)


def test_loader_batches_match_oracle_render():
    assets, loader = _loader()
    loader.prepare()
    assert len(loader) == 2
    holder = ro.SceneHolder(assets)
    ep = loader.epoch
    for bi, batch in enumerate(loader):
        s0 = bi * 4
        smp = ep["_samples"][s0:s0 + 4].cpu().numpy().view(ro.SAMPLE_DTYPE).reshape(-1)
        ref, _, keys = holder.render_batch(smp, ep["_hand_verts"][s0:s0 + 4].cpu().numpy(), ep["_order"][s0:s0 + 4].cpu().numpy(),
                                           ep["_factor"][s0:s0 + 4].cpu().numpy(), ep["_inv_affine"][s0:s0 + 4].cpu().numpy(), 224, 224,
                                           blur=ep["_blur"][s0:s0 + 4].cpu().numpy())
        np.testing.assert_array_equal(batch["image"].cpu().numpy(), ref)
        pad = batch["image_nhwc4_padded"].cpu().numpy()
        np.testing.assert_array_equal(pad[:, 3:-3, 3:-5, :3].transpose(0, 3, 1, 2), ref)
        assert (keys != np.uint64(0xFFFFFFFFFFFFFFFF)).mean() > 0.01
        for k in ("cam_intr", "root_joint", "joints_3d", "corners_3d", "joints_vis", "corners_vis", "corners_can", "obj_transf",
                  "obj_idx", "is_synth", "obj_id", "persp_id", "grasp_id", "sample_idx", "joints_2d", "corners_2d"):
            assert k in batch, k
        assert batch["joints_3d"].shape == (4, 21, 3) and batch["cam_intr"].shape == (4, 3, 3)
        # GT consistency: projecting (joints_3d + root) with the crop intrinsics lands on joints_2d
        P = (batch["joints_3d"] + batch["root_joint"][:, None]).cpu().numpy()
        Kc = batch["cam_intr"].cpu().numpy()
        uv = np.einsum("bij,bkj->bki", Kc, P)
        uv = uv[..., :2] / uv[..., 2:3]
        np.testing.assert_allclose(uv, batch["joints_2d"].cpu().numpy(), atol=0.6)   # int-truncated crop centre -> sub-pixel slack


DTYPES = ["bf16x3", "bf16"]          # the benchmarked precision first; one reduced-precision case each


def _ldt(dtype):
    return torch.bfloat16 if dtype == "bf16" else torch.float32


@pytest.mark.parametrize("dtype", DTYPES)
def test_end_to_end_train_steps_decrease_loss(dtype):
    """render -> forward -> fused loss -> backward -> clip/Adam for a few graph-replayed steps: finite, loss moves."""
    import yaml, os
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.train import TrainStep
    assets, loader = _loader(_ldt(dtype), bs=8, n=32, size=224)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    arch = dict(cfg["ARCH"], COMPUTE_DTYPE=dtype)
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=1e-3, max_norm=1.0, model=hb)
    loader.prepare()
    static = loader.new_static_batch()
    loader.load_batch(static, 0)
    model.train()
    ts = TrainStep(model, crit, opt, static, use_graph=True, renderer=loader)
    ts.static = static
    vals = []
    for i in range(12):
        loader.load_batch(static, 0)          # same batch every step: the loss must go down
        _, losses, _ = ts()
        vals.append(float(losses[5]))
    assert np.isfinite(vals).all()
    assert vals[-1] < vals[0], vals


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", [True, "opt"])
def test_pipelined_render_hands_over_next_batch(mode, dtype):
    """Pipelined TrainStep: step i learns from batch i while batch i+1 is rendered on the side stream; the image handed
    to the next step must be bit-identical to an inline render of that batch."""
    import yaml, os
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.train import TrainStep
    assets, loader = _loader(_ldt(dtype), bs=8, n=32, size=224)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    arch = dict(cfg["ARCH"], COMPUTE_DTYPE=dtype)
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=1e-3, max_norm=1.0, model=hb)
    loader.prepare()
    static = loader.new_static_batch()
    loader.load_batch(static, 0)
    model.train()
    ts = TrainStep(model, crit, opt, static, use_graph=True, renderer=loader, pipeline_render=mode)
    ts.static = static
    ref = loader.new_static_batch()

    def inline(bi):
        loader.load_batch(ref, bi)
        loader.render_into(ref)
        torch.cuda.synchronize()
        return ref["image_nhwc4_padded"].clone()

    ts.prime(loader, 0)
    torch.cuda.synchronize()
    assert torch.equal(static["image_nhwc4_padded"], inline(0))
    for bi in range(3):
        ts.stage(loader, bi)
        _, losses, _ = ts()
        torch.cuda.synchronize()
        assert torch.isfinite(losses).all()
        assert torch.equal(static["image_nhwc4_padded"], inline((bi + 1) % len(loader)))
        loader.load_batch(ref, bi)
        for k, v in ref.items():                      # ground truth in the learn buffers is still batch bi's
            if not k.startswith("_") and k != "image_nhwc4_padded":
                assert torch.equal(static[k], v), k


def _run_steps(monkeypatch, split, nsteps=6, bs=8, size=224, dtype="bf16"):
    import yaml, os, random
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.train import TrainStep
    monkeypatch.setenv("AB_DDP_SPLIT", "1" if split else "0")
    random.seed(7); torch.manual_seed(7); np.random.seed(7)
    assets, loader = _loader(torch.bfloat16 if dtype == "bf16" else torch.float32, bs=bs, n=4 * bs, size=size)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"] = [size, size]
    cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [size // 8, size // 8]
    arch = dict(cfg["ARCH"], COMPUTE_DTYPE=dtype, INIT_SEED=3)
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=1e-3, max_norm=1.0, model=hb)
    loader.prepare()
    static = loader.new_static_batch()
    loader.load_batch(static, 0)
    model.train()
    ts = TrainStep(model, crit, opt, static, use_graph=True, renderer=loader)
    ts.static = static
    assert ts.split == split
    vals = []
    for i in range(nsteps):
        loader.load_batch(static, i % len(loader))
        _, losses, _ = ts()
        vals.append(losses.float().cpu().numpy().copy())
    return np.stack(vals), hb.store.flat.detach().cpu().numpy().copy()


def test_split_backward_graphs_match_single_graph(monkeypatch):
    """The DDP-overlap capture (backward in two graphs around the layer4 boundary) runs the same kernels in the same
    order as the single-graph step: losses and final weights are bit-identical."""
    l0, w0 = _run_steps(monkeypatch, False)
    l1, w1 = _run_steps(monkeypatch, True)
    assert np.isfinite(l0).all()
    np.testing.assert_array_equal(l0, l1)
    np.testing.assert_array_equal(w0, w1)


def test_two_rank_ddp_step_keeps_weights_identical():
    """world_size 2 on this box's GPU(s) (gloo when both ranks share one device, RCCL otherwise): split-graph backward,
    side-stream gradient averaging, clip+Adam -- after 5 steps (including the capture warm-up) both ranks hold bit-identical
    weights and a finite loss."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "tools", "ddp_smoke.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "weights_identical_across_ranks=True" in r.stdout


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("graph", [False, True])
def test_train_step_with_symcorner_loss(graph, dtype):
    """A criterion list containing SymCornerLoss (DexYCB-style configs, symcornerloss.py:18-102) runs through the fused
    pose/loss kernel (ab_pose_loss_sym), eagerly and as replayed hipGraphs, and trains."""
    import yaml, os
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.train import TrainStep
    assets, loader = _loader(_ldt(dtype), bs=8, n=16, size=224)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    nobj = 21
    info = {str(i + 1): ({"symmetries_discrete": [[-1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]]} if i % 2 else
                         {"symmetries_continuous": [{"axis": [0, 0, 1], "offset": [0, 0, 0]}]}) for i in range(nobj)}
    cfg["CRITERION"] = cfg["CRITERION"] + [{"TYPE": "SymCornerLoss", "LAMBDA_SYM_CORNERS_3D": 1.0, "MODEL_INFO": info,
                                            "MAX_SYM_DISC_STEP": 0.2}]
    cfg["LAMBDAS"] = cfg["LAMBDAS"] + [0.3]
    arch = dict(cfg["ARCH"], COMPUTE_DTYPE=dtype)
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=1e-3, max_norm=1.0, model=hb)
    loader.prepare()
    static = loader.new_static_batch()
    loader.load_batch(static, 0)
    model.train()
    ts = TrainStep(model, crit, opt, static, use_graph=graph, renderer=loader)
    ts.static = static
    assert ts.fused is not None and ts.fused.sym is not None and ts.use_graph is graph
    vals, syms = [], []
    for i in range(8):
        loader.load_batch(static, 0)
        _, losses, o = ts()
        vals.append(float(losses[5])); syms.append(float(o["sym_loss"][0]))
    assert np.isfinite(vals).all() and np.isfinite(syms).all() and syms[0] > 0
    assert vals[-1] < vals[0], vals


@pytest.mark.parametrize("dtype", DTYPES)
def test_deferred_epoch_metrics_equal_per_step_feeding(dtype):
    """DeferredEpochMetrics (one transfer per epoch) == evaluator.feed_all after every batch (train_artiboost.py:96-98)."""
    import copy, os, yaml
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.metrics import Evaluator
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.train import DeferredEpochMetrics, TrainStep
    assets, loader = _loader(_ldt(dtype), bs=8, n=32, size=64)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [64, 64], [8, 8]
    arch = dict(cfg["ARCH"], COMPUTE_DTYPE=dtype)
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    ev_a = Evaluator(cfg, R.build_evaluator_metric_list(cfg["EVALUATOR"], preset_cfg=cfg["DATA_PRESET"]))
    ev_b = Evaluator(cfg, R.build_evaluator_metric_list(copy.deepcopy(cfg["EVALUATOR"]), preset_cfg=cfg["DATA_PRESET"]))
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=1e-3, max_norm=1.0, model=hb)
    loader.prepare()
    static = loader.new_static_batch()
    loader.load_batch(static, 0)
    model.train()
    ts = TrainStep(model, crit, opt, static, use_graph=True, renderer=loader)
    rec = DeferredEpochMetrics(ts, 2 * len(loader), ev_b)
    for bi in list(range(len(loader))) * 2:          # every triplet twice: the later write must win in both evaluators
        ts.stage(loader, bi)
        preds, _, _ = ts()
        ev_a.feed_all(preds, ts.static, ts.fused.losses_dict())
        rec.collect()
    rec.flush(ev_b)
    ma, mb = ev_a.get_measures_all(), ev_b.get_measures_all()
    assert set(ma) == set(mb)
    for k in ma:
        if isinstance(ma[k], dict):
            assert set(ma[k]) == set(mb[k]) and len(ma[k]) > 0
            for t in ma[k]:
                np.testing.assert_allclose(mb[k][t], ma[k][t], rtol=2e-5)
        else:
            np.testing.assert_allclose(float(mb[k]), float(ma[k]), rtol=2e-5)
    wa = copy.deepcopy(loader.sample_weight_map)
    loader.step_eval(0, ev_a)
    w1 = loader.sample_weight_map.clone()
    loader.sample_weight_map = wa
    loader.step_eval(0, ev_b)
    np.testing.assert_allclose(loader.sample_weight_map.numpy(), w1.numpy(), rtol=1e-5)      # same mining update


def test_train_step_predictions_match_the_model_forward():
    """TrainStep.predictions(): the nine HybridBaseline keys rebuilt from the fused criterion kernel's outputs == the model's own
    forward on the same batch (learning rate 0, so the weights the step just used are still in place)."""
    import yaml, os
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.train import TrainStep
    assets, loader = _loader(torch.float32, bs=4, n=8, size=64)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [64, 64], [8, 8]
    arch = dict(cfg["ARCH"], COMPUTE_DTYPE="bf16x3", SEGMENT_GRAPHS=False)
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    opt = FusedClipAdam(model.models_params, lr=0.0, max_norm=1.0, model=model.model_list[0])
    loader.prepare()
    static = loader.new_static_batch()
    loader.load_batch(static, 0)
    model.train()
    ts = TrainStep(model, crit, opt, static, use_graph=False, renderer=loader)
    ts.static = static
    ts()
    got = {k: v.detach().clone() for k, v in ts.predictions().items()}
    with torch.no_grad():
        ref = model(static)["HybridBaseline"]
    assert set(got) == set(ref)
    for k in ref:
        np.testing.assert_allclose(got[k].float().cpu().numpy(), ref[k].float().cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


def test_pose_generator_vs_reference_class_golden(golden_dir):
    """synth.PoseGenerator (ab_mano_lbs + device glue) against the reference's PreProcessorPoseGenerator.forward + RandomScrambler
    run with stand-ins for manotorch / pytorch3d / the refiner only (tests/golden/posegen.npz)."""
    from artiboost_amd.assets import make_hand_model
    from artiboost_amd.synth import ManoLayerHIP, PoseGenerator
    g = np.load(os.path.join(golden_dir, "posegen.npz"))
    hm = make_hand_model(int(g["hand_model_seed"]))
    gen = PoseGenerator(ManoLayerHIP(hm, "cuda"))
    t = lambda k: torch.from_numpy(g[k]).float().cuda()       # noqa: E731
    op, hv, jt = gen(t("hand_pose"), t("hand_shape"), t("hand_tsl"), t("persp_rotmat"), t("camera_free_transf"), t("z_offset"),
                     rand_pose_angle=t("rand_angle"), rand_tsl=t("rand_tsl"))
    np.testing.assert_allclose(op.cpu().numpy(), g["final_obj_pose"], rtol=1e-5, atol=5e-6)
    np.testing.assert_allclose(hv.cpu().numpy(), g["final_hand_verts"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(jt.cpu().numpy(), g["final_joints"], rtol=1e-5, atol=2e-5)


# ---- render <-> ground-truth alignment (the counterpart of the reference's only render check, script/viz_artiboost_render.py:34-129, which
# overlays the projected GT joints / corners / mesh on the rendered crop): the rasteriser's visibility keys (R3: unpinnable pixel oracle) are
# tied to the reference-pinned GT of R4 (joints_2d / corners_2d of assemble_gt_batch, tests/golden/misc.npz) through the crop affine.
def _alignment_stats(keys, inv_affine, joints_2d, joints_3d, root, joints_vis, corners_2d, size, n_hand_faces):
    """keys: uint64 [H, W] full-frame visibility keys of one sample (depth24 << 32 | face id; all ones = background).
    -> (joints checked, joints that land within 2 px of a hand pixel, object-pixel bbox in crop space or None, corners bbox in crop space)."""
    H, W = keys.shape
    A = inv_affine.reshape(2, 3).astype(np.float64)
    ys, xs = np.mgrid[0:size, 0:size]
    src = np.einsum("ij,jhw->ihw", A, np.stack([xs, ys, np.ones_like(xs)]).astype(np.float64))      # crop pixel -> full-frame (x, y)
    sx, sy = np.rint(src[0]).astype(np.int64), np.rint(src[1]).astype(np.int64)
    inside = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    kc = np.full((size, size), np.uint64(0xFFFFFFFFFFFFFFFF))
    kc[inside] = keys[sy[inside], sx[inside]]                                                   # the visibility keys seen through the crop
    cov = kc != np.uint64(0xFFFFFFFFFFFFFFFF)
    fid = (kc & np.uint64(0xFFFFFFFF)).astype(np.int64)
    hand, obj = cov & (fid < n_hand_faces), cov & (fid >= n_hand_faces)
    z24 = (kc >> np.uint64(32)).astype(np.float64) / 16777215.0
    depth = 1.0 / (20.0 + z24 * (0.01 - 20.0))                                                   # metres (the renderer's depth quantisation)
    checked = hit = 0
    for j in range(joints_2d.shape[0]):
        if joints_vis[j] < 0.5:
            continue
        u, v = joints_2d[j]
        x0, x1, y0, y1 = int(np.floor(u)) - 2, int(np.ceil(u)) + 3, int(np.floor(v)) - 2, int(np.ceil(v)) + 3
        if x0 < 0 or y0 < 0 or x1 > size or y1 > size:
            continue                                                                            # too close to the border for a 2-px window
        win_hand, win_obj, win_d = hand[y0:y1, x0:x1], obj[y0:y1, x0:x1], depth[y0:y1, x0:x1]
        zj = joints_3d[j, 2] + root[2]
        if not win_hand.any() and win_obj.any() and (win_d[win_obj] < zj).all():
            continue                                                                            # the object is in front of this joint: not its surface
        checked += 1
        hit += bool(win_hand.any())
    ob = None
    if obj.any():
        yy, xx = np.nonzero(obj)
        ob = (xx.min(), yy.min(), xx.max(), yy.max())
    cb = (corners_2d[:, 0].min(), corners_2d[:, 1].min(), corners_2d[:, 0].max(), corners_2d[:, 1].max())
    return checked, hit, ob, cb


@pytest.mark.parametrize("dataset,cfgname,B", [("HO3D", "ho3dv2_clasbased_artiboost_mi355x.yaml", 8), ("DexYCB", "dexycb_clasbased_sym_mi355x.yaml", 2)])
def test_rendered_pixels_line_up_with_the_ground_truth(dataset, cfgname, B):
    """(i) every visible GT joint not hidden behind the object lands within 2 px of a HAND pixel of the render; (ii) the bbox of the OBJECT
    pixels lies inside the bbox of the GT corners_2d (+- 2 px) and spans most of its in-image part; (iii) with the key image mirrored in x or
    in y the same check FAILS (the test can fail).  Keys from the HIP path and from render_oracle.c (they are bit-identical: both are run)."""
    import yaml
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.synth import ArtiBoostLoader
    size = 256
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", cfgname)))
    cfg["DATA_PRESET"]["IMAGE_SIZE"] = [size, size]
    assets = SceneAssets(dataset, seed=1)
    loader = ArtiBoostLoader.from_assets(assets, cfg["MANAGER"], cfg["DATA_PRESET"], B, B, compute_dtype=torch.float32, random_seed=11)
    loader.prepare()
    ep = loader.epoch
    smp = ep["_samples"][:B]
    o = loader.renderer.render(smp, ep["_hand_verts"][:B], ep["_order"][:B], ep["_factor"][:B], ep["_inv_affine"][:B], size, size,
                               out_chw=torch.empty((B, 3, size, size), dtype=torch.float32, device=smp.device), want_keys=True)
    keys_hip = o["keys"].cpu().numpy().view(np.uint64)
    holder = ro.SceneHolder(assets)
    _, _, keys_ref = holder.render_batch(smp.cpu().numpy().view(ro.SAMPLE_DTYPE).reshape(-1), ep["_hand_verts"][:B].cpu().numpy(),
                                         ep["_order"][:B].cpu().numpy(), ep["_factor"][:B].cpu().numpy(), ep["_inv_affine"][:B].cpu().numpy(),
                                         size, size, blur=ep["_blur"][:B].cpu().numpy())
    np.testing.assert_array_equal(keys_hip, keys_ref)
    g = {k: ep[k][:B].cpu().numpy() for k in ("joints_2d", "joints_3d", "root_joint", "joints_vis", "corners_2d", "_inv_affine")}
    nh = 1538                                                    # MANO faces come first in the scene's face list (test_render_properties)
    for name, keys in (("hip", keys_hip), ("oracle", keys_ref)):
        tot = hits = 0
        mirrored = {"x": [0, 0], "y": [0, 0]}
        for b in range(B):
            args = (g["_inv_affine"][b], g["joints_2d"][b], g["joints_3d"][b], g["root_joint"][b], g["joints_vis"][b], g["corners_2d"][b], size, nh)
            c, h, ob, cb = _alignment_stats(keys[b], *args)
            tot += c; hits += h
            assert c >= 4, f"{name}: sample {b}: only {c} joints could be checked"
            # (the stand-in hand model's distal joints can sit a few millimetres outside its own skin: at most two such joints per sample)
            assert h >= c - 2, f"{name}: sample {b}: {c - h} of {c} visible joints are not on the rendered hand"
            assert ob is not None, f"{name}: sample {b}: no object pixel in the crop"
            assert ob[0] >= cb[0] - 2 and ob[1] >= cb[1] - 2 and ob[2] <= cb[2] + 2 and ob[3] <= cb[3] + 2, (name, b, ob, cb)
            vis_w = min(cb[2], size - 1) - max(cb[0], 0) + 1                # in-image part of the corners' box
            vis_h = min(cb[3], size - 1) - max(cb[1], 0) + 1
            cover = ((ob[2] - ob[0] + 1) * (ob[3] - ob[1] + 1)) / max(vis_w * vis_h, 1.0)
            assert cover >= 0.35, f"{name}: sample {b}: object pixels span {cover:.2f} of the corners' box"
            for ax, km in (("x", keys[b][:, ::-1]), ("y", keys[b][::-1, :])):
                cm, hm, _, _ = _alignment_stats(np.ascontiguousarray(km), *args)
                mirrored[ax][0] += cm; mirrored[ax][1] += hm
        assert tot >= 6 * B and hits >= 0.95 * tot, (name, hits, tot)
        print(f"{dataset} {name}: {hits} / {tot} joints on the hand; mirrored x {mirrored['x']}, y {mirrored['y']}")
        for ax in ("x", "y"):      # a flipped rasteriser misses a large share of the joints: the check above is not vacuous
            assert mirrored[ax][1] < 0.8 * mirrored[ax][0], f"{name}: the keys mirrored in {ax} still pass ({mirrored[ax]})"


def test_loader_integer_image_plane_is_the_same_image():
    """compute_dtype "u8n": the padded image leaves the renderer as the bf16 plane 2 v - 255 of the SAME jittered uint8 pixels the fp32 path
    normalises: plane / 510 == v / 255 - 0.5 to fp32 rounding, every value an odd integer, the border zero."""
    import yaml
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.synth import ArtiBoostLoader
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"] = [128, 128]
    assets = SceneAssets("HO3D", seed=1)
    mk = lambda dt: ArtiBoostLoader.from_assets(assets, cfg["MANAGER"], cfg["DATA_PRESET"], 4, 8, compute_dtype=dt, random_seed=3)   # noqa: E731
    a, b = mk(torch.float32), mk("u8n")
    a.prepare(); b.prepare()
    assert b.image_plane == "u8n" and b.dtype == torch.bfloat16
    for ba, bb in zip(a, b):
        pa, pb = ba["image_nhwc4_padded"], bb["image_nhwc4_padded"]
        assert pb.dtype == torch.bfloat16 and pb.shape == pa.shape
        n = pb.float().cpu().numpy()
        inner = n[:, 3:-3, 3:-5, :3]
        assert np.all(inner == np.rint(inner)) and np.all(np.abs(inner) <= 255) and np.all(inner.astype(np.int64) % 2 != 0)
        v = (inner + 255.0) / 2.0
        np.testing.assert_array_equal((v / 255.0).astype(np.float32) - np.float32(0.5), pa.cpu().numpy()[:, 3:-3, 3:-5, :3])
        assert np.abs(n[:, :3]).max() == 0 and np.abs(n[:, :, :3]).max() == 0 and np.abs(n[..., 3]).max() == 0
        np.testing.assert_array_equal(ba["image"].cpu().numpy(), bb["image"].cpu().numpy())       # the float CHW image is unchanged


@pytest.mark.gpu
def test_train_steps_on_the_integer_image_plane_match_the_fp32_image():
    """The same graph-replayed bf16x3 step from the same weights, once with the fp32 padded image (split into planes by its own pass, three
    MFMA passes in the stem) and once with the loaders' integer plane (AB_DT_U8N, two passes): the image is the same, so the first step's
    losses and gradient agree to the stem's operand rounding (later steps drift apart as any two fp32 runs do: batch-8 BatchNorm and Adam's
    normalised steps amplify rounding noise); the loss trajectories stay within a few per cent over three steps."""
    import random
    import yaml
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.train import TrainStep
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    res = {}
    for plane in ("f32", "u8n"):
        torch.manual_seed(0); np.random.seed(0); random.seed(0)
        assets, loader = _loader(torch.float32 if plane == "f32" else "u8n", bs=8, n=32, size=224)
        arch = dict(cfg["ARCH"], COMPUTE_DTYPE="bf16x3", INIT_SEED=7)
        model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
        crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
        hb = model.model_list[0]
        opt = FusedClipAdam(model.models_params, lr=1e-4, max_norm=1.0, model=hb)
        loader.prepare()
        static = loader.new_static_batch()
        loader.load_batch(static, 0)
        model.train()
        ts = TrainStep(model, crit, opt, static, use_graph=True, renderer=loader)
        ts.static = static
        assert hb.net.image_plane == plane
        vals, g1 = [], None
        for i in range(3):
            loader.load_batch(static, i % len(loader))
            _, losses, _ = ts()
            vals.append(losses.float().cpu().numpy().copy())
            if i == 0:
                g1 = {n: hb.store.gview(n).detach().float().cpu().numpy().ravel().copy() for n in hb.store.entries}
        res[plane] = (np.stack(vals), g1)
    np.testing.assert_allclose(res["u8n"][0][0], res["f32"][0][0], rtol=1e-4, atol=1e-7)          # first step: the same losses
    np.testing.assert_allclose(res["u8n"][0][:, :6], res["f32"][0][:, :6], rtol=5e-2, atol=1e-5)   # three steps: the same trajectory to a few %
    # first step's gradient.  The head sees the same activations to rounding; deeper into the backbone a rounding-level change of the stem output
    # flips ReLU / max-pool decisions of this randomly initialised net at batch 8: the SAME fp32 image through another stem kernel
    # (AB_STEM_HALO_X3=0) moves these gradients by up to 1.8 % (cosine 0.9999), the integer plane by up to 3.1 % (cosine 0.9996).
    gu, gf = res["u8n"][1], res["f32"][1]
    for n in ("hybrid_head.final_layer.weight", "hybrid_head.final_layer.bias"):
        assert np.linalg.norm(gu[n] - gf[n]) <= 1e-4 * np.linalg.norm(gf[n]), n
    for n, a in gf.items():
        na = np.linalg.norm(a)
        if na > 1e-3:
            cos = float(a @ gu[n]) / (na * np.linalg.norm(gu[n]))
            assert cos >= 0.998 and abs(np.linalg.norm(gu[n]) / na - 1.0) <= 0.02, (n, cos)


def test_lazy_chw_image_equals_the_renderers_own_chw_store():
    """`batch["image"]` of the reference-shaped iteration is made on first access from the padded plane (synth.LazyImageBatch): byte-identical
    to what the renderer writes when asked for the CHW image directly, for the fp32 image and the integer plane; the model never touches it."""
    for dt in (torch.float32, "u8n"):
        _, loader = _loader(dt, bs=4, n=8, size=128)
        loader.prepare()
        static = loader.new_static_batch()
        for bi, batch in enumerate(loader):
            assert "image" in batch and "image" not in batch.keys()                  # promised, not made yet
            assert batch["image_plane"] == loader.image_plane
            loader.load_batch(static, bi)
            loader.render_into(static, want_chw=True)                                # the renderer's own CHW store of the same samples
            np.testing.assert_array_equal(batch["image"].cpu().numpy(), static["image"].cpu().numpy())
            assert "image" in batch.keys() and batch.get("image") is batch["image"]  # made once


def test_untagged_bf16_image_is_refused_and_tags_select_the_stem_path():
    """the image plane is carried by the loaders' tag, never guessed from dtype or value range."""
    import yaml
    from artiboost_amd import registry as R
    from artiboost_amd.models import Arch
    from artiboost_amd.registry import RUNTIME
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [64, 64], [8, 8]
    arch = dict(cfg["ARCH"], COMPUTE_DTYPE="bf16x3")
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
    assert RUNTIME["loader_compute_dtype"] == "u8n"          # what a loader built with the reference's keywords will follow
    _, loader = _loader("u8n", bs=4, n=4, size=64)
    loader.prepare()
    batch = next(iter(loader))
    model.eval()
    with torch.no_grad():
        ref = model(batch)["HybridBaseline"]["joints_3d_abs"].clone()
        assert model.model_list[0].net.image_plane == "u8n"
        stripped = {k: v for k, v in batch.items() if k != "image_plane"}
        stripped["image_nhwc4_padded"] = batch["image_nhwc4_padded"].clone()          # a clone carries no tag
        with pytest.raises(TypeError, match="integer plane"):
            model(stripped)
        # the same pixels as the fp32 image (tagged "f32" by its loader): the three-pass stem, same prediction to operand rounding
        _, lf = _loader(torch.float32, bs=4, n=4, size=64)
        lf.prepare()
        got = model(next(iter(lf)))["HybridBaseline"]["joints_3d_abs"]
        assert model.model_list[0].net.image_plane == "f32"
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-4)


def test_reference_keyword_loader_follows_the_model_it_feeds():
    """train_artiboost.py:117-190 builds the model, then the loader -- with no compute_dtype keyword.  The loader lands on the model's
    native image plane (bf16x3 -> "u8n"; an exact-f32 model -> the fp32 image)."""
    import types, yaml
    from artiboost_amd import registry as R
    from artiboost_amd.models import Arch
    from artiboost_amd.synth import ArtiBoostLoader
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [64, 64], [8, 8]
    cfg["MANAGER"]["SYNTH_LEN"] = 8
    cfg["MANAGER"].pop("REFINER", None)
    for cd, plane, dt in (("bf16x3", "u8n", torch.bfloat16), ("f32", "f32", torch.float32)):
        arch = dict(cfg["ARCH"], COMPUTE_DTYPE=cd)
        model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
        loader = ArtiBoostLoader(None, arg=types.SimpleNamespace(device="cuda:0", batch_size=4), cfg=cfg["MANAGER"], cfg_dataset=cfg["DATASET"],
                                 cfg_preset=cfg["DATA_PRESET"], batch_size=4, random_seed=1)
        loader.prepare()
        batch = next(iter(loader))
        assert loader.image_plane == plane and batch["image_plane"] == plane and batch["image_nhwc4_padded"].dtype == dt
        model.train()
        out = model(batch)["HybridBaseline"]
        assert torch.isfinite(out["joints_3d_abs"]).all() and model.model_list[0].net.image_plane == plane


def test_evaluator_late_reads_equal_blocking_feeds():
    """Evaluator.feed_all reads the device back one step late (pinned buffer + event); at every read of the measures the state equals
    per-step blocking feeds (max_lag 0) exactly -- Mean3DEPE sums, LossesMetric means, ValMetricMean3DEPE2's last-write-wins table --
    and the progress string never runs more than one feed behind."""
    import copy, yaml
    from artiboost_amd import registry as R
    from artiboost_amd.metrics import Evaluator
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    mk = lambda lag: Evaluator(cfg, R.build_evaluator_metric_list(copy.deepcopy(cfg["EVALUATOR"]), preset_cfg=cfg["DATA_PRESET"]), max_lag=lag)  # noqa: E731
    late, block = mk(1), mk(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    B = 8
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)      # noqa: E731
    for step in range(7):
        targs = {"joints_3d": 0.1 * r(B, 21, 3), "corners_3d": 0.1 * r(B, 8, 3), "root_joint": r(B, 3),
                 "obj_id": torch.randint(0, 3, (B,), device="cuda", generator=g), "persp_id": torch.randint(0, 2, (B,), device="cuda", generator=g),
                 "grasp_id": torch.randint(0, 2, (B,), device="cuda", generator=g), "is_synth": torch.rand(B, device="cuda", generator=g) > 0.3}
        preds = {"joints_3d_abs": targs["joints_3d"] + targs["root_joint"][:, None] + 0.01 * r(B, 21, 3),
                 "corners_3d_abs": targs["corners_3d"] + targs["root_joint"][:, None] + 0.01 * r(B, 8, 3)}
        losses = {"final_loss": r(1).abs()[0], "joints_3d_loss": r(1).abs()[0], "sym_corners_3d_loss": None, "host_number": 0.25 * step}
        late.feed_all(preds, targs, losses)
        block.feed_all(preds, targs, losses)
        assert len(late._inflight) <= 1 and len(block._inflight) == 0
        str(late)                                               # the progress string: no flush
        assert len(late._inflight) <= 1
    ma, mb = late.get_measures_all(), block.get_measures_all()
    assert len(late._inflight) == 0 and set(ma) == set(mb) and len(ma) >= 3
    for k in ma:
        if isinstance(ma[k], dict):
            assert ma[k].keys() == mb[k].keys() and len(ma[k]) > 0
            assert all(ma[k][t] == mb[k][t] for t in ma[k])
        else:
            assert ma[k] == mb[k], k
    assert str(late) == str(block) and "final_loss" in str(late)
    for a, b in zip(late.metrics_list, block.metrics_list):
        if hasattr(a, "get_measures_averaged"):
            assert a.get_measures_averaged() == b.get_measures_averaged()

"""GPU parity: fused pose-assembly + criterion (+backward) kernel vs the CPU oracle with identical RNG draws, and the
fused training path vs the reference's golden losses / gradient norms."""
import os
import random

import numpy as np
import pytest
import torch

import learner_oracle as lo
from gen_batch import make_batch

pytestmark = pytest.mark.gpu


def _crit():
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    cfgc = [{"TYPE": "JointsLoss", "LAMBDA_JOINTS_3D": 1.0, "LAMBDA_CORNERS_3D": 0.2}, {"TYPE": "HandOrdLoss"},
            {"TYPE": "SceneOrdLoss"}]
    return Criterion({"LAMBDAS": [0.5, 0.2, 0.1]}, R.build_criterion_loss_list(cfgc, preset_cfg={}, LAMBDAS=[0.5, 0.2, 0.1]))


@pytest.mark.parametrize("B,size,seed", [(1, 224, 0), (5, 256, 1), (64, 256, 2)])
def test_fused_pose_loss_vs_oracle(B, size, seed):
    from artiboost_amd.criterions import FusedPoseCriterion
    batch = make_batch(B, size, seed + 10)
    g = torch.Generator().manual_seed(seed)
    kp3d = torch.rand(B, 22, 3, generator=g).requires_grad_(True)
    box6d = torch.randn(B, 6, generator=g).requires_grad_(True)
    # oracle: pose assembly + criterion with explicit draws
    random.seed(seed + 3); torch.manual_seed(seed + 3)
    pose = lo.uvd2xyz(kp3d, batch["root_joint"], batch["cam_intr"], [size, size])
    R = lo.ortho6d_to_rotmat(box6d)
    corners = torch.matmul(R, batch["corners_can"].permute(0, 2, 1)).permute(0, 2, 1) + pose[:, 21:22]
    preds = {"joints_3d_abs": pose[:, :21], "corners_3d_abs": corners}
    total, losses, draws = lo.criterion(preds, batch)
    total.backward()
    # fused kernel with the same draws (same seeds -> same RNG stream, reference order)
    crit = _crit()
    fused = FusedPoseCriterion(crit, [size, size], 0)
    random.seed(seed + 3); torch.manual_seed(seed + 3)
    dev = torch.device("cuda")
    fused.draw(dev)
    box_buf = torch.zeros(B, 64).cuda()
    box_buf[:, :6] = box6d.detach().cuda()
    tb = {k: v.cuda() for k, v in batch.items()}
    o = fused(kp3d.detach().cuda(), box_buf, 64, tb)
    np.testing.assert_allclose(o["joints_3d_abs"].cpu().numpy(), pose[:, :21].detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o["corners_3d_abs"].cpu().numpy(), corners.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(o["box_rot_rotmat"].cpu().numpy(), R.detach().numpy(), rtol=1e-5, atol=1e-6)
    ld = fused.losses_dict()
    for k in ("joints_3d_loss", "corners_3d_loss", "joint_ord_loss", "part_ord_loss", "scene_ord_loss", "final_loss"):
        np.testing.assert_allclose(float(ld[k]), float(losses[k]), rtol=2e-5, err_msg=k)
    gk = kp3d.grad.numpy()
    np.testing.assert_allclose(o["g_kp3d"].cpu().numpy(), gk, rtol=2e-4, atol=2e-5 * np.abs(gk).max())
    gb = box6d.grad.numpy()
    np.testing.assert_allclose(o["g_box6d"].cpu().numpy(), gb, rtol=2e-4, atol=2e-5 * np.abs(gb).max())
    epe = lo.mean_epe_mm(pose[:, :21].detach(), batch["joints_3d"], batch["root_joint"])
    np.testing.assert_allclose(o["sample_part"][:, 5].cpu().numpy(), epe.numpy(), rtol=1e-4)


@pytest.mark.parametrize("graph", [False, True])
def test_fused_train_step_matches_reference_golden(golden_dir, graph):
    from test_gpu_learner import build
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.train import TrainStep
    g = np.load(os.path.join(golden_dir, "learner_g224.npz"))
    size, heat, depth, B, seed = [int(x) for x in g["meta"]]
    model, crit, params = build(size, heat, "f32", seed)
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=5e-5, max_norm=0.001, model=hb)
    batch = make_batch(B, size, seed + 100)
    model.train()
    ts = TrainStep(model, crit, opt, batch, use_graph=graph)
    if graph:
        ts._capture()          # warm-up performs one real update: restore the golden weights and optimiser state
        hb.load_state_dict(params)
        hb.net.pack_weights()
        for st in opt.state.values():
            st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
        opt.graph_steps = 0
    random.seed(seed + 7)
    torch.manual_seed(seed + 7)
    preds, losses, _ = ts()
    lv = losses.cpu().numpy()
    for i, k in enumerate(("joints_3d_loss", "corners_3d_loss", "joint_ord_loss", "part_ord_loss", "scene_ord_loss", "final_loss")):
        np.testing.assert_allclose(lv[i], g[f"loss.{k}"].reshape(-1)[0], rtol=3e-4, atol=1e-7, err_msg=k)
    np.testing.assert_allclose(preds["joints_3d_abs"].cpu().numpy(), g["train.pred.joints_3d_abs"], rtol=1e-4, atol=3e-5)
    grads = hb.store.reference_state_dict(grads=True)
    ref = dict(zip([str(n) for n in g["grad.names"]], g["grad.norms"]))
    for n, r in ref.items():
        got = float(grads[n].norm())
        assert abs(got - r) <= 1e-2 * r + 1e-9, (n, got, r)
    np.testing.assert_allclose(float(opt.total_norm), float(g["opt.total_norm"]), rtol=2e-3)
    d = hb.state_dict()["hybrid_head.final_layer.bias"] - params["hybrid_head.final_layer.bias"]
    np.testing.assert_allclose(d.numpy(), g["opt.final_bias.delta"], rtol=3e-2, atol=3e-7)


@pytest.mark.parametrize("B,seed", [(1, 0), (7, 1), (64, 2)])
def test_fused_symcorner_loss_vs_oracle(B, seed):
    """SymCornerLoss inside the fused kernel (ab_pose_loss_sym) vs the oracle's restatement of symcornerloss.py:49-102
    (itself checked against the imported reference): loss value, the total including it, and the gradients."""
    from artiboost_amd import registry as R_
    from artiboost_amd.criterions import Criterion, FusedPoseCriterion
    size = 256
    info = {str(i + 1): ({"symmetries_discrete": [[-1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]]} if i % 3 == 1 else
                         {"symmetries_continuous": [{"axis": [0, 0, 1], "offset": [0, 0, 0]}]} if i % 3 == 2 else {})
            for i in range(21)}
    cfgc = [{"TYPE": "JointsLoss", "LAMBDA_JOINTS_3D": 1.0, "LAMBDA_CORNERS_3D": 0.2}, {"TYPE": "HandOrdLoss"},
            {"TYPE": "SceneOrdLoss"}, {"TYPE": "SymCornerLoss", "LAMBDA_SYM_CORNERS_3D": 0.7, "MODEL_INFO": info,
                                       "MAX_SYM_DISC_STEP": 0.05}]
    lambdas = [0.5, 0.2, 0.1, 0.3]
    crit = Criterion({"LAMBDAS": lambdas}, R_.build_criterion_loss_list(cfgc, preset_cfg={}, LAMBDAS=lambdas))
    sym = crit.loss_list[3]
    batch = make_batch(B, size, seed + 20)
    g = torch.Generator().manual_seed(seed)
    batch["obj_idx"] = torch.randint(1, 22, (B,), generator=g)
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, :3, :3] = lo.ortho6d_to_rotmat(torch.randn(B, 6, generator=g))
    T[:, :3, 3] = batch["root_joint"] + 0.05 * torch.randn(B, 3, generator=g)
    batch["obj_transf"] = T
    kp3d = torch.rand(B, 22, 3, generator=g).requires_grad_(True)
    box6d = torch.randn(B, 6, generator=g).requires_grad_(True)
    random.seed(seed + 3); torch.manual_seed(seed + 3)
    pose = lo.uvd2xyz(kp3d, batch["root_joint"], batch["cam_intr"], [size, size])
    Rm = lo.ortho6d_to_rotmat(box6d)
    corners = torch.matmul(Rm, batch["corners_can"].permute(0, 2, 1)).permute(0, 2, 1) + pose[:, 21:22]
    preds = {"joints_3d_abs": pose[:, :21], "corners_3d_abs": corners}
    total, losses, _ = lo.criterion(preds, batch)
    sym_ref = lo.sym_corner_loss(preds, batch, sym.R, sym.t)
    total = total + lambdas[3] * 0.7 * sym_ref
    total.backward()
    fused = FusedPoseCriterion(crit, [size, size], 0)
    random.seed(seed + 3); torch.manual_seed(seed + 3)
    fused.draw(torch.device("cuda"))
    box_buf = torch.zeros(B, 64).cuda()
    box_buf[:, :6] = box6d.detach().cuda()
    tb = {k: v.cuda() for k, v in batch.items()}
    o = fused(kp3d.detach().cuda(), box_buf, 64, tb)
    ld = fused.losses_dict()
    np.testing.assert_allclose(float(ld["sym_corners_3d_loss"]), float(sym_ref), rtol=2e-5)
    np.testing.assert_allclose(float(ld["final_loss"]), float(total), rtol=2e-5)
    gk = kp3d.grad.numpy()
    np.testing.assert_allclose(o["g_kp3d"].cpu().numpy(), gk, rtol=2e-4, atol=2e-5 * np.abs(gk).max())
    gb = box6d.grad.numpy()
    np.testing.assert_allclose(o["g_box6d"].cpu().numpy(), gb, rtol=2e-4, atol=2e-5 * np.abs(gb).max())


@pytest.mark.parametrize("with_sym", [False, True])
def test_compute_losses_fused_route_equals_registry_route(with_sym):
    """`Criterion.compute_losses` on predictions of the HIP model: the fused pose/loss kernel (default) and the registry
    losses' torch ops + autograd give the same loss entries and the same gradient at the network's raw outputs."""
    from artiboost_amd import registry as R_
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import batch_uvd2xyz, ortho6d_to_rotmat
    size, B, seed = 256, 16, 5
    cfgc = [{"TYPE": "JointsLoss", "LAMBDA_JOINTS_3D": 1.0, "LAMBDA_CORNERS_3D": 0.2}, {"TYPE": "HandOrdLoss"}, {"TYPE": "SceneOrdLoss"}]
    lambdas = [0.5, 0.2, 0.1]
    batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in make_batch(B, size, seed + 20).items()}
    g = torch.Generator().manual_seed(seed)
    if with_sym:
        info = {str(i + 1): ({"symmetries_discrete": [[-1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]]} if i % 3 == 1 else
                             {"symmetries_continuous": [{"axis": [0, 0, 1], "offset": [0, 0, 0]}]} if i % 3 == 2 else {}) for i in range(21)}
        cfgc = cfgc + [{"TYPE": "SymCornerLoss", "LAMBDA_SYM_CORNERS_3D": 0.7, "MODEL_INFO": info, "MAX_SYM_DISC_STEP": 0.05}]
        lambdas = lambdas + [0.3]
        batch["obj_idx"] = torch.randint(1, 22, (B,), generator=g).cuda()
        T = torch.eye(4).repeat(B, 1, 1)
        T[:, :3, :3] = lo.ortho6d_to_rotmat(torch.randn(B, 6, generator=g))
        T[:, :3, 3] = batch["root_joint"].cpu() + 0.05 * torch.randn(B, 3, generator=g)
        batch["obj_transf"] = T.cuda()
    crit = Criterion({"LAMBDAS": lambdas}, R_.build_criterion_loss_list(cfgc, preset_cfg={}, LAMBDAS=lambdas))
    kp0, box0 = torch.rand(B, 22, 3, generator=g), torch.randn(B, 6, generator=g)
    res = {}
    for route in ("fused", "registry"):
        kp3d, box6d = kp0.clone().cuda().requires_grad_(True), box0.clone().cuda().requires_grad_(True)
        pose = batch_uvd2xyz(kp3d, batch["root_joint"], batch["cam_intr"], [size, size])
        corners = torch.matmul(ortho6d_to_rotmat(box6d), batch["corners_can"].permute(0, 2, 1)).permute(0, 2, 1) + pose[:, 21:22]
        joints = pose[:, :21]
        joints._ab_fuse = dict(kp3d=kp3d, box6d=box6d, inp_res=[size, size], center_idx=0)      # what HybridBaseline.forward attaches
        crit.fused_route = route == "fused"
        random.seed(seed + 3); torch.manual_seed(seed + 3)
        total, losses = crit.compute_losses({"joints_3d_abs": joints, "corners_3d_abs": corners}, batch)
        assert (type(total.grad_fn).__name__ == "_FusedLossFnBackward") == (route == "fused")
        (2.0 * total).backward()
        res[route] = ({k: float(v) for k, v in losses.items() if v is not None}, kp3d.grad.cpu().numpy(), box6d.grad.cpu().numpy())
    a, b = res["fused"], res["registry"]
    assert set(a[0]) == set(b[0])
    for k in b[0]:
        np.testing.assert_allclose(a[0][k], b[0][k], rtol=3e-5, atol=1e-9, err_msg=k)
    np.testing.assert_allclose(a[1], b[1], rtol=3e-4, atol=3e-5 * np.abs(b[1]).max())
    np.testing.assert_allclose(a[2], b[2], rtol=3e-4, atol=3e-5 * np.abs(b[2]).max())

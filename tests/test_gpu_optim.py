"""GPU parity: fused clip+Adam (HIP) vs the CPU oracle (== torch clip_grad_norm_ + Adam, pinned by the golden test),
and one full optimiser step of the learner vs the reference's golden parameter deltas."""
import os
import random

import numpy as np
import pytest
import torch

import learner_oracle as lo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [0, 1, 3, 4, 7, 1000, 4096 * 257 + 5])
@pytest.mark.parametrize("off", [0, 1, 2, 3])
def test_scale_f32_is_the_elementwise_product(n, off):
    """ab_scale_f32 (the 1 / world of the gradient average, train.allreduce_flat_): bit-equal to x * s for any slice offset / length,
    and leaves the elements either side of the slice alone."""
    from artiboost_amd import _lib as L
    base = torch.randn(n + 16, generator=torch.Generator().manual_seed(n + off)).cuda()
    want = base.clone()
    want[off:off + n] *= 0.125
    want3 = base.clone()
    want3[off:off + n] *= (1.0 / 3.0)
    for s, ref in ((0.125, want), (1.0 / 3.0, want3)):
        x = base.clone()
        L.check(L.lib().ab_scale_f32(L.ptr(x[off:off + n]) if n else L.ptr(x[off:off + 1]), L.l(n), L.f(s), L.stream()), "ab_scale_f32")
        assert torch.equal(x, ref)


def test_allreduce_flat_over_one_rank_rccl_group_is_the_identity():
    """train.allreduce_flat_ on the GPU path: SUM all-reduce + ab_scale_f32 (no ReduceOp.AVG: its RCCL kernels use packed fp32).  One rank is all a
    one-GPU box can run: the gradient must come back bit-identical, for ranges that start off a 16-byte boundary too."""
    import socket
    import torch.distributed as dist
    from artiboost_amd.train import allreduce_flat_, rccl_env_defaults
    if dist.is_initialized():
        pytest.skip("a process group is already alive in this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    rccl_env_defaults()
    assert os.environ["NCCL_ALGO"]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        g = torch.randn((1 << 22) + 7, device="cuda")
        ref = g.clone()
        allreduce_flat_(g, 1, bucket_elems=1 << 20)
        allreduce_flat_(g[5:-1], 1, bucket_elems=1 << 20)
        torch.cuda.synchronize()
        assert torch.equal(g, ref)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,max_norm", [(4, 0.001), (1000, 0.001), (4096 * 257, 0.5), (64, None)])
def test_clip_adam_matches_oracle(n, max_norm):
    from artiboost_amd.optim import FusedClipAdam
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    p = torch.nn.Parameter(p0.clone().cuda())
    opt = FusedClipAdam([p], lr=5e-5, max_norm=max_norm)
    ps, m, v = [p0.clone()], [torch.zeros(n)], [torch.zeros(n)]
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (10.0 ** (step - 3))
        p.grad = grad.clone().cuda()
        opt.step()
        tn = lo.clip_and_adam(ps, [grad], m, v, step, lr=5e-5, max_norm=max_norm if max_norm else 1e30)
        if max_norm:
            np.testing.assert_allclose(float(opt.total_norm), float(tn), rtol=1e-5)
        np.testing.assert_allclose(p.detach().cpu().numpy(), ps[0].numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(opt.state[p]["exp_avg_sq"].cpu().numpy(), v[0].numpy(), rtol=1e-4, atol=1e-30)


def test_learner_step_matches_reference_golden(golden_dir):
    from test_gpu_learner import build
    from gen_batch import make_batch
    from artiboost_amd.optim import FusedClipAdam
    g = np.load(os.path.join(golden_dir, "learner_g224.npz"))
    size, heat, depth, B, seed = [int(x) for x in g["meta"]]
    model, crit, params = build(size, heat, "f32", seed)
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=5e-5, max_norm=0.001, model=hb)
    batch = make_batch(B, size, seed + 100)
    model.train()
    preds = model(batch)["HybridBaseline"]
    random.seed(seed + 7)
    torch.manual_seed(seed + 7)
    total, _ = crit.compute_losses(preds, batch)
    opt.zero_grad()
    total.backward()
    opt.step()
    np.testing.assert_allclose(float(opt.total_norm), float(g["opt.total_norm"]), rtol=2e-3)
    sd = hb.state_dict()
    d = sd["hybrid_head.final_layer.bias"] - params["hybrid_head.final_layer.bias"]
    np.testing.assert_allclose(d.numpy(), g["opt.final_bias.delta"], rtol=3e-2, atol=3e-7)
    d = (sd["backbone.conv1.weight"] - params["backbone.conv1.weight"])[::8, :, ::3, ::3]
    # first Adam step moves every weight by ~lr*sign(g): check sign agreement where the reference moved clearly
    ref = g["opt.conv1.delta.sample"]
    big = np.abs(ref) > 2e-5
    assert (np.sign(d.numpy()[big]) == np.sign(ref[big])).mean() > 0.97
    # padded parameters never move
    st = hb.store
    assert float(st.view("backbone.conv1.weight")[:, :, 7, :].abs().max()) == 0.0
    assert float(st.view("backbone.conv1.weight")[:, :, :, 3].abs().max()) == 0.0
    assert float(st.view("box_head.layers.4.weight")[6:].abs().max()) == 0.0
    assert float(st.view("hybrid_head.final_layer.weight").reshape(22, 32, 256)[:, 28:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16"])
@pytest.mark.parametrize("graph", [False, True])
def test_checkpoint_resume_is_bit_exact(graph, dtype, tmp_path):
    """Model state_dict (the reference's keys) + optimizer state_dict saved after 3 steps and loaded into fresh objects:
    the following steps equal those of the uninterrupted run bit for bit (Adam moments, step count / bias corrections)."""
    import random
    import yaml
    from gen_batch import make_batch
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.train import TrainStep
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [64, 64], [8, 8]
    batches = [{k: v.cuda() for k, v in make_batch(4, 64, 50 + i).items()} for i in range(6)]

    def fresh():
        arch = dict(cfg["ARCH"], COMPUTE_DTYPE=dtype, INIT_SEED=3)
        model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
        crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
        hb = model.model_list[0]
        opt = FusedClipAdam(model.models_params, lr=1e-3, max_norm=1.0, model=hb)
        model.train()
        return model, crit, hb, opt

    def run(ts, lo, hi):
        out = []
        for i in range(lo, hi):
            random.seed(100 + i); torch.manual_seed(100 + i)
            _, losses, _ = ts(batches[i])
            out.append(losses.float().cpu().numpy().copy())
        return np.stack(out)

    model, crit, hb, opt = fresh()
    ts = TrainStep(model, crit, opt, {k: v.clone() for k, v in batches[0].items()}, use_graph=graph)
    run(ts, 0, 3)
    # <bn>.num_batches_tracked counts training forwards, graph replays included (the capture warm-up is undone)
    assert int(hb.state_dict()["backbone.layer3.2.bn1.num_batches_tracked"]) == 3
    torch.save({"model": hb.state_dict(), "optimizer": opt.state_dict()}, tmp_path / "ckpt.pth.tar")
    tail_ref = run(ts, 3, 6)
    w_ref = hb.store.flat.detach().cpu().numpy().copy()

    ck = torch.load(tmp_path / "ckpt.pth.tar")
    model2, crit2, hb2, opt2 = fresh()
    hb2.load_state_dict(ck["model"])
    opt2.load_state_dict(ck["optimizer"])                  # before the first step: the graph capture leaves no trace
    ts2 = TrainStep(model2, crit2, opt2, {k: v.clone() for k, v in batches[0].items()}, use_graph=graph)
    np.testing.assert_array_equal(run(ts2, 3, 6), tail_ref)
    np.testing.assert_array_equal(hb2.store.flat.detach().cpu().numpy(), w_ref)


@pytest.mark.parametrize("dtype", ["bf16x3", "bf16"])
def test_graph_replay_equals_eager_from_the_first_step(dtype):
    """The capture warm-up is undone (weights, running stats, Adam state, RNG): graph and eager runs are the same updates."""
    import random
    import yaml
    from gen_batch import make_batch
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.train import TrainStep
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [64, 64], [8, 8]
    batches = [{k: v.cuda() for k, v in make_batch(4, 64, 70 + i).items()} for i in range(4)]
    res = []
    for graph in (False, True):
        random.seed(9); torch.manual_seed(9); np.random.seed(9)
        arch = dict(cfg["ARCH"], COMPUTE_DTYPE=dtype, INIT_SEED=3)
        model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
        crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
        hb = model.model_list[0]
        opt = FusedClipAdam(model.models_params, lr=1e-3, max_norm=1.0, model=hb)
        model.train()
        ts = TrainStep(model, crit, opt, {k: v.clone() for k, v in batches[0].items()}, use_graph=graph)
        ls = [ts(b)[1].float().cpu().numpy().copy() for b in batches]
        res.append((np.stack(ls), hb.store.flat.detach().cpu().numpy().copy(), hb.store.stats.cpu().numpy().copy()))
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, b)

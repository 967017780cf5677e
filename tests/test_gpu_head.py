"""GPU parity: HIP soft-argmax head (through the C ABI) vs the CPU oracle and vs the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

import learner_oracle as lo

pytestmark = pytest.mark.gpu


def _nhwc(logits_nchw):
    return logits_nchw.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("tag", ["tiny", "g224", "g256"])
def test_head_vs_reference_golden(golden_dir, tag):
    from artiboost_amd.head import softargmax3d
    g = np.load(os.path.join(golden_dir, "head.npz"))
    seed, B, C, D, H, W = [int(x) for x in g[f"{tag}.seed"]]
    gen = torch.Generator().manual_seed(seed)
    logits = 4.0 * torch.randn(B, C * D, H, W, generator=gen)
    x = _nhwc(logits).cuda().requires_grad_(True)
    uvd, conf = softargmax3d(x, C, D)
    np.testing.assert_allclose(uvd.detach().cpu().numpy(), g[f"{tag}.uvd"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(conf.detach().cpu().numpy(), g[f"{tag}.conf"], rtol=2e-5, atol=1e-7)
    gu = torch.from_numpy(g[f"{tag}.g_uvd"]).cuda()
    gc = torch.from_numpy(g[f"{tag}.g_conf"]).cuda()
    ((uvd * gu).sum() + (conf * gc).sum()).backward()
    dl = x.grad.permute(0, 3, 1, 2).reshape(B, C, -1)[:, :, ::97].cpu().numpy()
    ref = g[f"{tag}.dlogits.sample"]
    np.testing.assert_allclose(dl, ref, rtol=2e-4, atol=1e-6 * np.abs(ref).max() + 1e-9)
    np.testing.assert_allclose(float(x.grad.abs().sum()), float(g[f"{tag}.dlogits.abs_sum"]), rtol=1e-4)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-6), (torch.bfloat16, 3e-6)])
@pytest.mark.parametrize("shape", [(1, 1, 1, 1, 1), (3, 22, 28, 7, 5), (64, 22, 28, 32, 32), (2, 5, 9, 65, 3)])
def test_head_vs_oracle(dtype, tol, shape):
    """bf16 logits are upcast exactly, so the oracle is fed the same rounded values and the tolerance is the fp32 one."""
    from artiboost_amd.head import softargmax3d
    B, C, D, H, W = shape
    gen = torch.Generator().manual_seed(B * 131 + C)
    logits = (3.0 * torch.randn(B, C * D, H, W, generator=gen)).to(dtype)
    ref_in = logits.float().clone().requires_grad_(True)
    uvd_r, conf_r = lo.softargmax3d(ref_in, C, D, H, W)
    x = _nhwc(logits).cuda().requires_grad_(True)
    uvd, conf = softargmax3d(x, C, D)
    np.testing.assert_allclose(uvd.detach().cpu().numpy(), uvd_r.detach().numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(conf.detach().cpu().numpy(), conf_r.detach().numpy(), rtol=3e-5, atol=1e-8)
    gu = torch.randn(uvd_r.shape, generator=gen)
    (uvd_r * gu).sum().backward()
    (uvd * gu.cuda()).sum().backward()
    got = x.grad.float().permute(0, 3, 1, 2).cpu().numpy()
    ref = ref_in.grad.numpy()
    scale = np.abs(ref).max() + 1e-12
    atol = (1e-5 if dtype == torch.float32 else 1e-2) * scale   # bf16 output rounding: 2^-8 relative
    np.testing.assert_allclose(got, ref, rtol=1e-4 if dtype == torch.float32 else 1e-2, atol=atol)


def test_head_properties_full_size():
    """Size-independent properties at the benchmark geometry (B=64, 22x28x32x32): shift invariance, one-hot peak,
    uniform logits -> centre of mass of the grid."""
    from artiboost_amd.head import softargmax3d
    B, C, D, H, W = 64, 22, 28, 32, 32
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, C * D, generator=gen).cuda()
    u0, c0 = softargmax3d(x, C, D)
    u1, c1 = softargmax3d(x + 37.5, C, D)
    np.testing.assert_allclose(u1.cpu().numpy(), u0.cpu().numpy(), atol=2e-6)
    z = torch.zeros(B, H, W, C * D).cuda()
    uz, cz = softargmax3d(z, C, D)
    exp = np.array([(W - 1) / 2 / W, (H - 1) / 2 / H, (D - 1) / 2 / D], dtype=np.float32)
    np.testing.assert_allclose(uz.cpu().numpy(), np.broadcast_to(exp, (B, C, 3)), atol=2e-6)
    np.testing.assert_allclose(cz.cpu().numpy(), 1.0 / (D * H * W), rtol=1e-5)
    peak = torch.full((B, H, W, C * D), -50.0)
    peak[:, 5, 9, 3::D] = 50.0  # every class: d=3, h=5, w=9
    up, cp = softargmax3d(peak.cuda(), C, D)
    np.testing.assert_allclose(up.cpu().numpy(), np.broadcast_to(np.array([9 / W, 5 / H, 3 / D], np.float32), (B, C, 3)), atol=1e-6)
    np.testing.assert_allclose(cp.cpu().numpy(), 1.0, rtol=1e-6)


@pytest.mark.parametrize("shape", [(3, 22, 28, 32, 7, 5), (64, 22, 28, 32, 32, 32), (2, 5, 9, 10, 65, 3)])
@pytest.mark.parametrize("with_conf", [False, True])
def test_softargmax_bwd_x3_bias_sums(shape, with_conf):
    """ab_softargmax3d_bwd_x3_bias: the same dlogits planes as ab_softargmax3d_bwd_x3, bit for bit, and dbias = their column sums
    (the final layer's bias gradient, simplebaseline.py:148) against a float64 sum of the fp32 gradient."""
    from artiboost_amd.head import softargmax3d_fwd, softargmax3d_bwd, softargmax3d_bwd_x3
    B, C, D, DP, H, W = shape
    gen = torch.Generator().manual_seed(B + 7 * H)
    x = (3.0 * torch.randn(B, H, W, C * DP, generator=gen)).cuda()
    uvd, conf, stat = softargmax3d_fwd(x, C, D, DP)
    gu = torch.randn(uvd.shape, generator=gen).cuda()
    gc = torch.randn(conf.shape, generator=gen).cuda() if with_conf else None
    plain = softargmax3d_bwd_x3(x, C, D, DP, uvd, conf, stat, gu, gc)
    dbias = torch.full((C * DP,), float("nan"), device="cuda")
    fused = softargmax3d_bwd_x3(x, C, D, DP, uvd, conf, stat, gu, gc, dbias=dbias)
    assert getattr(fused, "_ab_bias_done", False) and torch.equal(fused, plain)
    full = softargmax3d_bwd(x, C, D, DP, uvd, conf, stat, gu, gc)          # fp32 gradient
    ref = full.double().sum((0, 1, 2))
    scale = float(full.abs().double().sum((0, 1, 2)).max()) + 1e-30
    assert float((dbias.double() - ref).abs().max()) <= 2e-6 * scale


def test_c_abi_from_a_plain_c_host(tmp_path):
    """No Python, no torch between the caller and the kernels: tests/c_host/abi_host_test.c (hipMalloc, its own stream) calls
    ab_softargmax3d_fwd and ab_split_f32 through the C ABI and checks them against arithmetic restated from the reference formulas."""
    import subprocess
    from test_abi import _build_c_host
    exe = _build_c_host(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C_ABI_HOST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("tag", ["tiny", "odd", "g256"])
def test_sigmoid_head_vs_reference_golden(golden_dir, tag):
    """IntegralDeconvHead NORM_TYPE: sigmoid (simplebaseline.py:16-40,183-189) on the same kernels in the log domain
    (ab_softargmax3d_fwd_norm / _bwd_norm), against vectors from the reference's own norm_heatmap + integral_heatmap3d
    (tests/golden/head_sigmoid.npz, oracle/gen_head_sigmoid_golden.py).  "odd": C*D odd -> the scalar kernels."""
    from artiboost_amd.head import softargmax3d
    g = np.load(os.path.join(golden_dir, "head_sigmoid.npz"))
    seed, B, C, D, H, W = [int(x) for x in g[f"{tag}.seed"]]
    gen = torch.Generator().manual_seed(seed)
    logits = 3.0 * torch.randn(B, C * D, H, W, generator=gen)
    x = _nhwc(logits).cuda().requires_grad_(True)
    uvd, conf = softargmax3d(x, C, D, norm_type="sigmoid")
    np.testing.assert_allclose(uvd.detach().cpu().numpy(), g[f"{tag}.uvd"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(conf.detach().cpu().numpy(), g[f"{tag}.conf"], rtol=3e-6, atol=0)
    gu = torch.from_numpy(g[f"{tag}.g_uvd"]).cuda()
    (uvd * gu).sum().backward()
    dl = x.grad.permute(0, 3, 1, 2).reshape(B, C, -1)[:, :, ::53].cpu().numpy()
    ref = g[f"{tag}.dlogits.sample"]
    np.testing.assert_allclose(dl, ref, rtol=3e-4, atol=2e-6 * np.abs(ref).max() + 1e-12)
    np.testing.assert_allclose(float(x.grad.abs().sum()), float(g[f"{tag}.dlogits.abs_sum"]), rtol=1e-4)


def test_sigmoid_head_split_plane_backward_and_padded_depth():
    """The bf16x3 form (fp32 logits, dlogits as (hi, lo) planes + the final layer's bias gradient, depth pitch 32 > depth 28) of the
    sigmoid head vs the oracle's autograd."""
    from artiboost_amd.head import softargmax3d_bwd_x3, softargmax3d_fwd
    B, C, D, DP, H, W = 3, 22, 28, 32, 8, 8
    gen = torch.Generator().manual_seed(7)
    core = (2.0 * torch.randn(B, C, D, H, W, generator=gen))
    ref_in = core.reshape(B, C * D, H, W).clone().requires_grad_(True)
    uvd_r, conf_r = lo.softargmax3d(ref_in, C, D, H, W, norm_type="sigmoid")
    gu = torch.randn(uvd_r.shape, generator=gen)
    (uvd_r * gu).sum().backward()
    pad = torch.full((B, C, DP, H, W), 7.0)                       # garbage in the padded depth slots must not matter
    pad[:, :, :D] = core
    x = pad.reshape(B, C * DP, H, W).permute(0, 2, 3, 1).contiguous().cuda()
    uvd, conf, stat = softargmax3d_fwd(x, C, D, DP, norm=1)
    np.testing.assert_allclose(uvd.cpu().numpy(), uvd_r.detach().numpy(), rtol=0, atol=3e-6)
    np.testing.assert_allclose(conf.cpu().numpy(), conf_r.detach().numpy(), rtol=3e-6)
    dbias = torch.zeros(C * DP, device="cuda")
    dl = softargmax3d_bwd_x3(x, C, D, DP, uvd, conf, stat, gu.cuda(), dbias=dbias, norm=1)
    got = (dl[0].float() + dl[1].float()).permute(0, 3, 1, 2).reshape(B, C, DP, H, W).cpu()
    ref = ref_in.grad.reshape(B, C, D, H, W)
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got[:, :, :D].numpy(), ref.numpy(), rtol=2e-4, atol=2e-5 * scale)
    assert float(got[:, :, D:].abs().max()) == 0.0
    np.testing.assert_allclose(dbias.cpu().reshape(C, DP)[:, :D].numpy(), ref.sum(dim=(0, 3, 4)).numpy(), rtol=1e-3, atol=1e-5 * scale * B * H * W)

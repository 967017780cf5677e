"""GPU parity of the conv-stack kernels (through the C ABI) against torch-CPU fp32 (what oracle/learner_oracle.py
is built from).  f32 path: MFMA f32 is an exact fmaf chain -> tight tolerance.  bf16 path: operands are rounded to
bf16 on the host first so both sides see identical inputs; the tolerance then covers fp32-accumulate ordering and
the bf16 rounding of the stored output (2^-8 relative)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def tol(dtype, ref):
    s = float(ref.abs().max()) + 1e-12
    return (dict(rtol=2e-5, atol=2e-5 * s) if dtype == torch.float32 else dict(rtol=1.6e-2, atol=8e-3 * s))


def rnd(shape, gen, dtype, scale=1.0):
    return (scale * torch.randn(shape, generator=gen)).to(dtype).float()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 64, 64, 3, 1, 1),
    (3, 14, 10, 64, 128, 3, 2, 1),
    (2, 8, 8, 128, 128, 3, 1, 1),
    (2, 14, 14, 64, 128, 1, 2, 0),
    (1, 7, 7, 512, 512, 3, 1, 1),
    (2, 9, 5, 256, 616, 1, 1, 0),     # final layer: ragged N (616), ragged M, with bias
    (5, 6, 6, 256, 256, 3, 1, 1),
    (2, 8, 8, 256, 512, 4, 2, 1),     # == ConvTranspose backward-data shape
    (2, 32, 32, 64, 64, 3, 1, 1),     # LDS-halo 3x3 kernel, 4x32 tiles, single chunk
    (1, 28, 28, 128, 128, 3, 1, 1),   # LDS-halo, ragged 8x32 tiles (224-geometry layer2), 2 chunks
    (3, 14, 14, 256, 256, 3, 1, 1),   # LDS-halo, ragged 8x16 tiles (224-geometry layer3), 4 chunks
    (2, 56, 40, 64, 64, 3, 1, 1),     # LDS-halo, ragged in both directions
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(case, dtype):
    from artiboost_amd import kernels as K
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = rnd((N, Cin, H, W), g, dtype)
    w = rnd((Cout, Cin, k, k), g, dtype, (2.0 / (Cin * k * k)) ** 0.5)
    b = torch.randn(Cout, generator=g) if Cout == 616 else None
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    y, stats = K.conv2d_fwd(nhwc(x).to(dtype).cuda(), w.permute(0, 2, 3, 1).contiguous().to(dtype).cuda(), s, p,
                            bias=b.cuda() if b is not None else None, want_stats=True)
    got = nchw(y.float().cpu())
    np.testing.assert_allclose(got.numpy(), ref.numpy(), **tol(dtype, ref))
    # fused BN partials: sum and sum of squares of the (stored-precision) output per channel
    st = stats.double().sum(0).cpu()
    yy = y.double().cpu().reshape(-1, Cout)
    np.testing.assert_allclose(st[:, 0].numpy(), yy.sum(0).numpy(), rtol=2e-3, atol=2e-3 * float(yy.abs().sum(0).max()))
    np.testing.assert_allclose(st[:, 1].numpy(), (yy * yy).sum(0).numpy(), rtol=2e-2)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("nhw", [(2, 32, 32), (1, 224, 224), (3, 20, 12)])
def test_stem_fwd_wgrad(nhw, dtype):
    from artiboost_amd import kernels as K
    N, H, W = nhw
    g = torch.Generator().manual_seed(N * H)
    x = rnd((N, 3, H, W), g, dtype)
    w = rnd((64, 3, 7, 7), g, dtype, 0.1)
    ref = F.conv2d(x, w, stride=2, padding=3)
    xpad = K.image_pad_nhwc4(x.cuda(), dtype)
    wst = torch.zeros(64, 7, 8, 4)
    wst[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    y = K.conv2d_stem_fwd(xpad, wst.to(dtype).cuda(), H, W)
    np.testing.assert_allclose(nchw(y.float().cpu()).numpy(), ref.numpy(), **tol(dtype, ref))
    dy = rnd(ref.shape, g, dtype)
    xr = x.clone().requires_grad_(False)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, stride=2, padding=3).backward(dy)
    dw = K.conv2d_stem_wgrad(xpad, nhwc(dy).to(dtype).cuda(), H, W).cpu()
    assert float(dw[:, :, 7, :].abs().max()) == 0.0 and float(dw[:, :, :, 3].abs().max()) == 0.0
    got = dw[:, :, :7, :3].permute(0, 3, 1, 2)
    t = tol(dtype, wr.grad)
    if dtype == torch.bfloat16:
        t = dict(rtol=2e-3, atol=2e-3 * float(wr.grad.abs().max()))   # fp32 output, only accumulation order differs
    np.testing.assert_allclose(got.numpy(), wr.grad.numpy(), **t)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CONV_CASES[:7] + CONV_CASES[8:])
def test_conv_dgrad_wgrad(case, dtype):
    from artiboost_amd import kernels as K
    N, H, W, Cin, Cout, k, s, p = case
    if s == 2 and (H % 2 or W % 2):
        H, W = H + H % 2, W + W % 2
    if Cout == 616:
        Cout = 640       # wgrad/dgrad need channel multiples of 64/32
    g = torch.Generator().manual_seed(hash(case) % 977)
    x = rnd((N, Cin, H, W), g, dtype).requires_grad_(True)
    w = rnd((Cout, Cin, k, k), g, dtype, (2.0 / (Cin * k * k)) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p)
    dy = rnd(y.shape, g, dtype)
    y.backward(dy)
    add = rnd(x.shape, g, dtype)
    dyd = nhwc(dy).to(dtype).cuda()
    wt = w.detach().permute(1, 2, 3, 0).contiguous().to(dtype).cuda()   # [Cin][kh][kw][Cout]
    dx = K.conv2d_dgrad(dyd, wt, (H, W), s, p, addend=nhwc(add).to(dtype).cuda())
    ref = x.grad + add
    np.testing.assert_allclose(nchw(dx.float().cpu()).numpy(), ref.numpy(), **tol(dtype, ref))
    dw = K.conv2d_wgrad(nhwc(x.detach()).to(dtype).cuda(), dyd, k, k, s, p).cpu().permute(0, 3, 1, 2)
    t = tol(dtype, w.grad)
    if dtype == torch.bfloat16:
        t = dict(rtol=2e-3, atol=2e-3 * float(w.grad.abs().max()))
    np.testing.assert_allclose(dw.numpy(), w.grad.numpy(), **t)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [(2, 7, 7, 512, 256), (2, 16, 16, 256, 256), (1, 5, 3, 64, 64)])
def test_conv_transpose_4x4s2(case, dtype):
    """ConvTranspose2d(k4,s2,p1) forward/backward (simplebaseline.py:161-170) expressed with the three conv kernels."""
    from artiboost_amd import kernels as K
    N, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(Ci + H)
    x = rnd((N, Ci, H, W), g, dtype).requires_grad_(True)
    wt = rnd((Ci, Co, 4, 4), g, dtype, (2.0 / (Ci * 4)) ** 0.5).requires_grad_(True)
    y = F.conv_transpose2d(x, wt, stride=2, padding=1)
    dy = rnd(y.shape, g, dtype)
    y.backward(dy)
    # mirrored conv C: (Co -> Ci), weight w_conv[co=Ci][ci=Co][kh][kw] = wt
    w_ihwo = wt.detach().permute(1, 2, 3, 0).contiguous().to(dtype).cuda()      # [Co][kh][kw][Ci] = dgrad layout of C
    w_ohwi = wt.detach().permute(0, 2, 3, 1).contiguous().to(dtype).cuda()      # [Ci][kh][kw][Co] = fwd layout of C
    xd = nhwc(x.detach()).to(dtype).cuda()
    yd = K.conv2d_dgrad(xd, w_ihwo, (2 * H, 2 * W), 2, 1)
    np.testing.assert_allclose(nchw(yd.float().cpu()).numpy(), y.detach().numpy(), **tol(dtype, y.detach()))
    dyd = nhwc(dy).to(dtype).cuda()
    dx = K.conv2d_fwd(dyd, w_ohwi, 2, 1)
    np.testing.assert_allclose(nchw(dx.float().cpu()).numpy(), x.grad.numpy(), **tol(dtype, x.grad))
    dwc = K.conv2d_wgrad(dyd, xd, 4, 4, 2, 1).cpu()          # [Ci][kh][kw][Co]
    got = dwc.permute(0, 3, 1, 2)                             # -> [Ci][Co][kh][kw]
    t = tol(dtype, wt.grad)
    if dtype == torch.bfloat16:
        t = dict(rtol=2e-3, atol=2e-3 * float(wt.grad.abs().max()))
    np.testing.assert_allclose(got.numpy(), wt.grad.numpy(), **t)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 8, 8, 64), (3, 5, 7, 256), (64, 4, 4, 512), (1, 60, 60, 128)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False)])
def test_batchnorm_train_fwd_bwd(shape, dtype, relu, res):
    from artiboost_amd import kernels as K
    N, H, W, C = shape
    g = torch.Generator().manual_seed(C + N)
    y = rnd((N, C, H, W), g, dtype, 1.5).add_(0.3).to(dtype).float().requires_grad_(True)
    gamma = (0.5 + torch.rand(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    r = rnd((N, C, H, W), g, dtype).requires_grad_(True) if res else None
    rm, rv = torch.zeros(C), torch.ones(C)
    ref = F.batch_norm(y, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    dout = rnd(ref.shape, g, dtype)
    ref.backward(dout)
    yd = nhwc(y.detach()).to(dtype).cuda()
    part = K.col_stats(yd)
    rm_d, rv_d = torch.zeros(C).cuda(), torch.ones(C).cuda()
    bnp = K.bn_finalize(part, N * H * W, gamma.detach().cuda(), beta.detach().cuda(), rm_d, rv_d)
    np.testing.assert_allclose(rm_d.cpu().numpy(), rm.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rv_d.cpu().numpy(), rv.numpy(), rtol=1e-4, atol=1e-6)
    rd = nhwc(r.detach()).to(dtype).cuda() if res else None
    out = K.bn_apply(yd, bnp, res=rd, relu=relu)
    np.testing.assert_allclose(nchw(out.float().cpu()).numpy(), ref.detach().numpy(), **tol(dtype, ref.detach()))
    dgamma, dbeta = torch.empty(C).cuda(), torch.empty(C).cuda()
    # feed the oracle's `out` (rounded to dtype) as the relu mask source so both sides agree on the mask
    dy, dz = K.bn_bwd(nhwc(dout).to(dtype).cuda(), out, yd, bnp, dgamma, dbeta, relu=relu, want_dz=True)
    t = tol(dtype, y.grad)
    if dtype == torch.bfloat16:
        t = dict(rtol=3e-2, atol=2e-2 * float(y.grad.abs().max()))
    np.testing.assert_allclose(nchw(dy.float().cpu()).numpy(), y.grad.numpy(), **t)
    np.testing.assert_allclose(dgamma.cpu().numpy(), gamma.grad.numpy(), rtol=2e-3, atol=2e-3 * float(gamma.grad.abs().max()))
    np.testing.assert_allclose(dbeta.cpu().numpy(), beta.grad.numpy(), rtol=2e-3, atol=2e-3 * float(beta.grad.abs().max()))
    if res:
        np.testing.assert_allclose(nchw(dz.float().cpu()).numpy(), r.grad.numpy(), **tol(dtype, r.grad))
    if relu and not res:
        # mask recomputed from y*scale+shift instead of read from `out`: same bits
        dg2, db2 = torch.empty(C).cuda(), torch.empty(C).cuda()
        dy2, dz2 = K.bn_bwd(nhwc(dout).to(dtype).cuda(), None, yd, bnp, dg2, db2, relu="recompute", want_dz=True)
        assert torch.equal(dy2, dy) and torch.equal(dz2, dz) and torch.equal(dg2, dgamma) and torch.equal(db2, dbeta)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 8, 8, 64), (1, 112, 112, 64), (3, 6, 10, 128)])
def test_maxpool_avgpool(shape, dtype):
    from artiboost_amd import kernels as K
    N, H, W, C = shape
    g = torch.Generator().manual_seed(H)
    x = rnd((N, C, H, W), g, dtype)
    x = F.relu(x).requires_grad_(True)            # post-ReLU input: many exact ties at 0 (first-max semantics matter)
    ref = F.max_pool2d(x, 3, 2, 1)
    dout = rnd(ref.shape, g, dtype)
    ref.backward(dout)
    xd = nhwc(x.detach()).to(dtype).cuda()
    out, idx = K.maxpool_fwd(xd)
    np.testing.assert_array_equal(nchw(out.float().cpu()).numpy(), ref.detach().numpy())
    dx = K.maxpool_bwd(idx, nhwc(dout).to(dtype).cuda(), (H, W))
    np.testing.assert_allclose(nchw(dx.float().cpu()).numpy(), x.grad.numpy(), **tol(dtype, x.grad))
    m = K.avgpool_fwd(xd)
    np.testing.assert_allclose(m.cpu().numpy(), x.detach().mean(3).mean(2).numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", DT)
def test_transpose_oki_batch(dtype):
    """One-launch OHWI -> IHWO re-layout of several weights == per-tensor permute (bit-exact cast)."""
    from artiboost_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 9, 64), (128, 1, 64), (704, 1, 256), (256, 16, 512), (72, 3, 36), (8, 2, 4)]
    pairs = []
    for O, Kk, I in shapes:
        src = torch.randn((O, Kk, I), generator=g).cuda()
        pairs.append((src, torch.empty((I, Kk, O), dtype=dtype, device="cuda")))
    plan = K.transpose_plan(pairs)
    assert plan is not None
    K.transpose_oki_batch(plan)
    torch.cuda.synchronize()
    for src, dst in pairs:
        assert torch.equal(dst, src.permute(2, 1, 0).contiguous().to(dtype))
    assert K.transpose_plan([(torch.zeros(8, 1, 6).cuda(), torch.empty((6, 1, 8), dtype=dtype, device="cuda"))]) is None


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (3, 6, 10, 128), (1, 128, 128, 64)])
def test_fused_stem_bn_relu_maxpool(shape, dtype):
    """BN+ReLU+max-pool in one pass (and its backward through the two BN passes) == the separate kernels, bit for bit."""
    from artiboost_amd import kernels as K
    N, H, W, C = shape
    g = torch.Generator().manual_seed(H * W)
    y = (torch.randn((N, H, W, C), generator=g) * 1.3 + 0.2).to(dtype).cuda()
    gamma = (0.5 + torch.rand(C, generator=g)).cuda(); beta = (0.1 * torch.randn(C, generator=g)).cuda()
    bnp = K.bn_finalize(K.col_stats(y), N * H * W, gamma, beta)
    a = K.bn_apply(y, bnp, relu=True)
    ref_out, ref_idx = K.maxpool_fwd(a)
    out, idx = K.bn_relu_maxpool_fwd(y, bnp)
    assert torch.equal(out, ref_out) and torch.equal(idx, ref_idx)
    dpool = torch.randn((N, H // 2, W // 2, C), generator=g).to(dtype).cuda()
    dg0, db0, dg1, db1 = (torch.empty(C).cuda() for _ in range(4))
    da = K.maxpool_bwd(ref_idx, dpool, (H, W))
    ref_dy = K.bn_bwd(da, None, y, bnp, dg0, db0, relu="recompute")
    dy = K.bn_relu_maxpool_bwd(dpool, idx, y, bnp, dg1, db1)
    assert torch.equal(dy, ref_dy) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    if dtype == torch.float32:      # the split-bf16 path's variant: same pooled tensor and winners, plus its (hi, lo) planes
        o3, i3 = K.bn_relu_maxpool_fwd_x3(y, bnp)
        assert torch.equal(o3, out) and torch.equal(i3, idx) and torch.equal(o3._ab_split, K.split(out))
        dg3, db3 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        pl = K.bn_relu_maxpool_bwd_x3(dpool, idx, y, bnp, dg3, db3)      # pool backward + mask + reduction in one pass, planes out
        assert pl is not None
        got = pl[0].float() + pl[1].float()
        assert float((got - ref_dy).abs().max()) <= 2e-5 * float(ref_dy.abs().max())
        np.testing.assert_allclose(dg3.cpu().numpy(), dg0.cpu().numpy(), rtol=2e-5, atol=2e-6 * float(dg0.abs().max()))
        np.testing.assert_allclose(db3.cpu().numpy(), db0.cpu().numpy(), rtol=2e-5, atol=2e-6 * float(db0.abs().max()))
        # ... and with the reduction over the pooled elements: the forward also hands out the raw conv output at the winners
        o4, i4, yw = K.bn_relu_maxpool_fwd_x3(y, bnp, want_win=True)
        assert torch.equal(o4, out) and torch.equal(i4, idx) and torch.equal(o4._ab_split, o3._ab_split)
        assert torch.equal(torch.relu(yw * bnp[0] + bnp[1]), out)           # bn + relu of the winner == the pooled activation
        dg4, db4 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        pw = K.bn_relu_maxpool_bwd_x3(dpool, idx, y, bnp, dg4, db4, ywin=yw)
        got = pw[0].float() + pw[1].float()
        assert float((got - ref_dy).abs().max()) <= 2e-5 * float(ref_dy.abs().max())
        np.testing.assert_allclose(dg4.cpu().numpy(), dg0.cpu().numpy(), rtol=2e-5, atol=2e-6 * float(dg0.abs().max()))
        np.testing.assert_allclose(db4.cpu().numpy(), db0.cpu().numpy(), rtol=2e-5, atol=2e-6 * float(db0.abs().max()))


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 128, False), (3, 32, 32, 64, 64, True), (2, 8, 8, 128, 192, True),
                                  (1, 20, 12, 64, 64, False)])
def test_dgrad_with_fused_bn_backward_reduction(case, monkeypatch):
    """ab_conv2d_dgrad_bnstats: the BN-backward sums accumulated in the data-gradient epilogue + ab_bn_bwd_apply give the
    same dx / dy / dgamma / dbeta as ab_conv2d_dgrad followed by the two-pass ab_bn_bwd."""
    from artiboost_amd import kernels as K
    if os.environ.get("AB_BNFUSE_MIN") != "1":
        pytest.skip("opt-in path: run with AB_BNFUSE_MIN=1 (the library reads the variable once per process)")
    N, H, W, Cin, Cout, residual = case
    g = torch.Generator().manual_seed(Cin + Cout + H)
    dt = torch.bfloat16
    dy = (0.5 * torch.randn((N, H, W, Cout), generator=g)).to(dt).cuda()
    wt = (0.05 * torch.randn((Cin, 3, 3, Cout), generator=g)).to(dt).cuda()
    y = (torch.randn((N, H, W, Cin), generator=g) * 1.2 + 0.1).to(dt).cuda()           # the lower BN's input
    res = torch.randn((N, H, W, Cin), generator=g).to(dt).cuda() if residual else None
    addend = (0.3 * torch.randn((N, H, W, Cin), generator=g)).to(dt).cuda() if residual else None
    gamma = (0.5 + torch.rand(Cin, generator=g)).cuda(); beta = (0.1 * torch.randn(Cin, generator=g)).cuda()
    bnp = K.bn_finalize(K.col_stats(y), N * H * W, gamma, beta)
    out = K.bn_apply(y, bnp, res=res, relu=True)
    relu = True if residual else "recompute"
    # reference: separate kernels
    dx0 = K.conv2d_dgrad(dy, wt, (H, W), 1, 1, addend=addend)
    dg0, db0 = torch.empty(Cin).cuda(), torch.empty(Cin).cuda()
    d0 = K.bn_bwd(dx0, out if residual else None, y, bnp, dg0, db0, relu=relu)
    # fused
    dx1, part = K.conv2d_dgrad(dy, wt, (H, W), 1, 1, addend=addend, bn=(y, out if residual else None, bnp))
    assert part is not None
    dg1, db1 = torch.empty(Cin).cuda(), torch.empty(Cin).cuda()
    d1 = K.bn_bwd(dx1, out if residual else None, y, bnp, dg1, db1, relu=relu, part=part)
    assert torch.equal(dx0, dx1)
    np.testing.assert_allclose(dg1.cpu().numpy(), dg0.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(dg0.abs().max()))
    np.testing.assert_allclose(db1.cpu().numpy(), db0.cpu().numpy(), rtol=2e-4, atol=2e-4 * float(db0.abs().max()))
    np.testing.assert_allclose(d1.float().cpu().numpy(), d0.float().cpu().numpy(), rtol=2e-2, atol=2e-2 * float(d0.float().abs().max()))


@pytest.mark.parametrize("M,N,K", [(64, 256, 512), (64, 64, 128), (3, 8, 8), (70, 12, 36)])
def test_small_batch_linear_kernels(M, N, K):
    """ab_linear_fwd / dgrad / wgrad (the fp32 box-rotation MLP) vs torch on CPU in float64."""
    from artiboost_amd import kernels as K_
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g); w = 0.1 * torch.randn((N, K), generator=g); b = torch.randn(N, generator=g)
    gy = torch.randn((M, N), generator=g)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    y = torch.relu(xd @ wd.t() + bd)
    y.backward(gy.double())
    yo = K_.linear_fwd(x.cuda(), w.cuda(), b.cuda(), relu=True)
    np.testing.assert_allclose(yo.cpu().numpy(), y.detach().numpy(), rtol=1e-5, atol=1e-5)
    gz = (gy * (y.detach() > 0)).float()                          # gradient after the ReLU mask
    wt = w.t().contiguous().cuda()
    gx = K_.linear_dgrad(gz.cuda(), wt)
    np.testing.assert_allclose(gx.cpu().numpy(), xd.grad.numpy(), rtol=1e-5, atol=1e-5)
    # masked variant: mask by the activation that fed the layer
    act = torch.relu(x)
    gxm = K_.linear_dgrad(gz.cuda(), wt, act_out=act.cuda())
    np.testing.assert_allclose(gxm.cpu().numpy(), (xd.grad * (act > 0)).numpy(), rtol=1e-5, atol=1e-5)
    dw = torch.empty((N, K), device="cuda"); db = torch.empty(N, device="cuda")
    K_.linear_wgrad(gz.cuda(), x.cuda(), dw, db)
    np.testing.assert_allclose(dw.cpu().numpy(), wd.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(db.cpu().numpy(), bd.grad.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("halo", [1, 0])
def test_stem_fast_path_repeatable_at_full_size(monkeypatch, halo):
    """The bf16 stem at B=64 on the LDS-resident halo kernel (halo=1) and on the LDS-DMA tap kernel (halo=0, fully unrolled
    4-step ring): 12 runs bit-identical to each other and equal to the register-staged kernel.  Before the
    lgkmcnt(0)-before-barrier fix the tap kernel produced ~15 wrong tiles per run."""
    from artiboost_amd import kernels as K
    torch.manual_seed(0)
    xpad = K.image_pad_nhwc4(torch.rand(64, 3, 256, 256, device="cuda") - 0.5, torch.bfloat16)
    w = (0.1 * torch.randn(64, 7, 8, 4, device="cuda")).to(torch.bfloat16)
    w[:, :, 7] = 0
    w[..., 3] = 0
    monkeypatch.setenv("AB_STEM_V1", "1")
    ref = K.conv2d_stem_fwd(xpad, w, 256, 256).clone()
    monkeypatch.delenv("AB_STEM_V1")
    monkeypatch.setenv("AB_STEM_HALO", str(halo))
    first = None
    for _ in range(12):
        junk = torch.randn(32 * 1024 * 1024, device="cuda")
        del junk
        o = K.conv2d_stem_fwd(xpad, w, 256, 256)
        if first is None:
            first = o.clone()
            assert (o.float() - ref.float()).abs().max() <= 2e-2 * float(ref.float().abs().max())
        assert torch.equal(o, first)


def test_fast_conv_paths_at_benchmark_shapes():
    """tools/fullsize_check.py: every fast conv path (halo 3x3, LDS-DMA generic, all-taps / generic wgrad, stem), bf16 and
    bf16x3, at the B=64 benchmark shapes: run-to-run bit-identical with a dirtied allocator in between, and in agreement
    with torch's own fp32 convolutions (MIOpen) computed at the same full size -- plus, for bf16, with the v1 kernels."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fullsize_check.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-6000:] + r.stderr[-2000:]


def test_deferred_wgrad_reductions_are_bit_identical():
    """ab_conv2d_wgrad_deferred / ab_conv2d_stem_wgrad_deferred + ONE ab_wgrad_reduce_batch launch for several layers (3x3
    halo path, strided 3x3, 1x1, 4x4 transpose-conv shape, stem) == the per-layer launches, bit for bit."""
    from artiboost_amd import kernels as K
    torch.manual_seed(3)
    dt = torch.bfloat16
    cases = [(8, 32, 32, 64, 64, 3, 1, 1), (8, 32, 32, 64, 128, 3, 2, 1), (8, 16, 16, 128, 256, 1, 2, 0), (4, 16, 16, 256, 256, 3, 1, 1),
             (8, 16, 16, 256, 704, 1, 1, 0)]
    pend = K.PendingReductions()
    got, ref = [], []
    for N, H, W, Cin, Cout, k, s, p in cases:
        x = (torch.randn(N, H, W, Cin, device="cuda") * 0.5).to(dt)
        Ho = (H + 2 * p - k) // s + 1
        dy = (torch.randn(N, Ho, Ho, Cout, device="cuda") * 0.5).to(dt)
        ref.append(K.conv2d_wgrad(x, dy, k, k, s, p).clone())
        got.append(K.conv2d_wgrad(x, dy, k, k, s, p, out=torch.full((Cout, k, k, Cin), float("nan"), device="cuda"), defer=pend))
    xpad = K.image_pad_nhwc4(torch.rand(8, 3, 64, 64, device="cuda") - 0.5, dt)
    dy0 = (torch.randn(8, 32, 32, 64, device="cuda") * 0.5).to(dt)
    ref.append(K.conv2d_stem_wgrad(xpad, dy0, 64, 64).clone())
    got.append(K.conv2d_stem_wgrad(xpad, dy0, 64, 64, out=torch.full((64, 7, 8, 4), float("nan"), device="cuda"), defer=pend))
    assert len(pend.descs) == len(cases) + 1 and any(d.nslices > 0 for d in pend.descs)
    pend.flush()
    assert not pend.descs and not pend.keep
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [(8, 8, 8, 512, 256), (4, 16, 16, 256, 256), (2, 6, 10, 128, 64)])
def test_transposed_conv_forward_with_fused_bn_partials(shape):
    """ConvTranspose2d(4, 2, 1) forward = ab_conv2d_dgrad of the mirrored conv; with `stats` the epilogue also writes the
    BatchNorm partial sums of the output (one row per parity class and M tile): same tensor, sums == a col_stats pass."""
    from artiboost_amd import kernels as K
    N, h, w, Ct, Co = shape
    torch.manual_seed(N + Ct)
    x = (torch.randn(N, h, w, Ct, device="cuda") * 0.5).to(torch.bfloat16)
    wt = (torch.randn(Co, 4, 4, Ct, device="cuda") * 0.05).to(torch.bfloat16)     # [Cin of the mirrored conv][kh][kw][Cout]
    ref = K.conv2d_dgrad(x, wt, (2 * h, 2 * w), 2, 1)
    got, part = K.conv2d_dgrad(x, wt, (2 * h, 2 * w), 2, 1, want_stats=True)
    assert torch.equal(got, ref)
    want = K.col_stats(ref).double().sum(0).cpu()
    have = part.double().sum(0).cpu()
    yy = ref.double().reshape(-1, Co)
    np.testing.assert_allclose(have[:, 0].numpy(), want[:, 0].numpy(), rtol=2e-3, atol=2e-3 * float(yy.abs().sum(0).max()))
    np.testing.assert_allclose(have[:, 1].numpy(), want[:, 1].numpy(), rtol=2e-2)

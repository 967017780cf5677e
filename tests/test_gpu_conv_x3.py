"""GPU parity of the split-bf16 ("bf16x3") convolution kernels against torch-CPU float64.

The kernels see fp32 inputs as hi + lo bf16 planes (2^-17 relative operand error) and accumulate in fp32, so against an
exact reference the error of an output is bounded by ~2^-16 * sum|a||b| over its reduction; the tolerance below is that
bound with the measured headroom (the exact-f32 MFMA path passes 2e-5 of the output scale, bf16 needs 8e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

X3_TOL = 3e-5        # of max|ref| (bf16: 8e-3, exact f32: 2e-5)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def close(got, ref, tol=X3_TOL):
    s = float(ref.abs().max()) + 1e-30
    err = float((got.double() - ref.double()).abs().max())
    assert err <= tol * s, f"max err {err:.3e} vs scale {s:.3e} (ratio {err / s:.2e})"


CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 64, 64, 3, 1, 1),
    (3, 14, 10, 64, 128, 3, 2, 1),
    (2, 8, 8, 128, 128, 3, 1, 1),
    (2, 14, 14, 64, 128, 1, 2, 0),
    (1, 7, 7, 512, 512, 3, 1, 1),
    (2, 9, 5, 256, 616, 1, 1, 0),     # final layer: ragged N, ragged M, bias
    (5, 6, 6, 256, 256, 3, 1, 1),
    (2, 8, 8, 256, 512, 4, 2, 1),     # ConvTranspose backward-data shape
    (2, 32, 32, 64, 64, 3, 1, 1),
    (1, 28, 28, 128, 128, 3, 1, 1),
    (3, 14, 14, 256, 256, 3, 1, 1),
    (2, 56, 40, 64, 64, 3, 1, 1),
    (2, 64, 64, 64, 64, 3, 1, 1),     # benchmark geometry layer1
    (2, 16, 16, 256, 256, 3, 1, 1),   # benchmark geometry layer3
]


def test_split_planes():
    from artiboost_amd import kernels as K
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 8, 8, 64, generator=g) * torch.exp(4 * torch.randn(4, 8, 8, 64, generator=g))
    sp = K.split(x.cuda()).cpu()
    hi = x.to(torch.bfloat16)
    assert torch.equal(sp[0], hi)
    assert torch.equal(sp[1], (x - hi.float()).to(torch.bfloat16))
    rel = ((sp[0].double() + sp[1].double() - x.double()).abs() / x.double().abs()).max()
    assert float(rel) <= 2.0 ** -17


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_x3(case):
    from artiboost_amd import kernels as K
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g) if Cout == 616 else None
    ref = F.conv2d(x.double(), w.double(), b.double() if b is not None else None, stride=s, padding=p)
    ws = K.split(w.permute(0, 2, 3, 1).contiguous().cuda())
    y, stats = K.conv2d_fwd_x3(nhwc(x).cuda(), ws, s, p, bias=b.cuda() if b is not None else None, want_stats=True)
    close(nchw(y.cpu()), ref)
    st = stats.double().sum(0).cpu()
    yy = y.double().cpu().reshape(-1, Cout)
    np.testing.assert_allclose(st[:, 0].numpy(), yy.sum(0).numpy(), rtol=1e-4, atol=1e-4 * float(yy.abs().sum(0).max()))
    np.testing.assert_allclose(st[:, 1].numpy(), (yy * yy).sum(0).numpy(), rtol=1e-4)


@pytest.mark.parametrize("shape", [(64, 32, 32, 128, 256, 3, 2, 1), (8, 16, 16, 256, 256, 4, 2, 1), (4, 16, 16, 256, 256, 3, 1, 1)])
def test_conv_fwd_x3_stats_with_bias_never_overruns_the_stat_rows(shape):
    """ab_conv2d_x3_stat_rows sizes the partial buffer for the kernel a launch WITHOUT bias / relu takes (one row per tile of
    convp / conv2x2 / conv3x3); a launch with statistics AND a bias falls through to the generic kernel, which writes one row per M tile of
    its own.  Where the counts differ the call is refused (AB_EINVAL) instead of writing past the buffer; where they agree it runs."""
    from artiboost_amd import _lib as L, kernels as K
    N, H, W, Cin, Cout, k, s, p = shape
    lib = L.lib()
    rows = lib.ab_conv2d_x3_stat_rows(L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout), L.i(k), L.i(k), L.i(s), L.i(p))
    g = torch.Generator().manual_seed(5)
    x = torch.randn((N, H, W, Cin), generator=g).cuda()
    ws = K.split((torch.randn((Cout, k, k, Cin), generator=g) * 0.05).cuda())
    b = torch.randn(Cout, generator=g).cuda()
    y0, st0 = K.conv2d_fwd_x3(x, ws, s, p, want_stats=True)                       # the model's own launch: specialised kernel, `rows` rows
    assert st0.shape[0] == rows
    guard = 4096
    buf = torch.full((rows * Cout * 2 + guard,), 123.0, device="cuda")            # the caller's buffer + a canary behind it
    xh, xl = K._planes(x)
    y = torch.empty_like(y0)
    try:
        rc = lib.ab_conv2d_fwd_x3(L.ptr(xh), L.ptr(xl), L.ptr(ws[0]), L.ptr(ws[1]), L.ptr(y), L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout),
                                  L.i(k), L.i(k), L.i(s), L.i(p), L.ptr(b), L.ptr(buf), L.i(0), L.stream())
    except RuntimeError as e:          # the torch-op binding raises on a non-zero return code; the ctypes binding returns it
        assert "code -1" in str(e), e
        rc = -1
    torch.cuda.synchronize()
    assert torch.all(buf[rows * Cout * 2:] == 123.0), "wrote past the stat rows"
    if rc == 0:      # the generic kernel's row count happens to match: results must be right
        np.testing.assert_allclose((y - b).cpu().numpy(), y0.cpu().numpy(), rtol=0, atol=2e-5 * float(y0.abs().max()))
        np.testing.assert_allclose(buf[:rows * Cout * 2].view(rows, Cout, 2).sum(0)[:, 0].cpu().numpy(), y.double().sum((0, 1, 2)).cpu().numpy(),
                                   rtol=1e-4, atol=1e-4 * float(y.abs().sum((0, 1, 2)).max()))
    else:
        assert rc == -1                   # AB_EINVAL


@pytest.mark.parametrize("case", CASES[:7] + CASES[8:])
def test_conv_dgrad_wgrad_x3(case):
    from artiboost_amd import kernels as K
    N, H, W, Cin, Cout, k, s, p = case
    if s == 2 and (H % 2 or W % 2):
        H, W = H + H % 2, W + W % 2
    if Cout == 616:
        Cout = 640
    g = torch.Generator().manual_seed(hash(case) % 977)
    x = torch.randn((N, Cin, H, W), generator=g).double().requires_grad_(True)
    w = (torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5).double().requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g).double()
    y.backward(dy)
    add = torch.randn(x.shape, generator=g)
    dyd = nhwc(dy.float()).cuda()
    wt = K.split(w.detach().float().permute(1, 2, 3, 0).contiguous().cuda())      # [Cin][kh][kw][Cout]
    dx = K.conv2d_dgrad_x3(dyd, wt, (H, W), s, p, addend=nhwc(add).cuda())
    close(nchw(dx.cpu()), x.grad + add.double())
    dw = K.conv2d_wgrad_x3(nhwc(x.detach().float()).cuda(), dyd, k, k, s, p).cpu().permute(0, 3, 1, 2)
    close(dw, w.grad)
    # accumulate=True adds onto the existing gradient
    base = torch.randn(Cout, k, k, Cin, generator=g).cuda()
    dw2 = K.conv2d_wgrad_x3(K.split(nhwc(x.detach().float()).cuda()), K.split(dyd), k, k, s, p, out=base.clone(), accumulate=True)
    close(dw2.cpu().permute(0, 3, 1, 2), w.grad + base.cpu().permute(0, 3, 1, 2).double())


def test_conv_transpose_4x4s2_x3():
    """ConvTranspose2d(4x4, s2, p1) forward == data gradient of the mirrored conv, with BN partials from the epilogue."""
    from artiboost_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 512, 8, 8, generator=g)
    w = torch.randn(512, 256, 4, 4, generator=g) * 0.02            # ConvT weight [Cin_t, Cout_t, kh, kw]
    ref = F.conv_transpose2d(x.double(), w.double(), stride=2, padding=1)
    wt = K.split(w.permute(1, 2, 3, 0).contiguous().cuda())        # [Co][kh][kw][Ci]: dgrad (IHWO) layout of the mirrored conv
    y, part = K.conv2d_dgrad_x3(nhwc(x).cuda(), wt, (16, 16), 2, 1, want_stats=True)
    close(nchw(y.cpu()), ref)
    yy = y.double().cpu().reshape(-1, 256)
    st = part.double().sum(0).cpu()
    np.testing.assert_allclose(st[:, 0].numpy(), yy.sum(0).numpy(), rtol=1e-4, atol=1e-4 * float(yy.abs().sum(0).max()))
    np.testing.assert_allclose(st[:, 1].numpy(), (yy * yy).sum(0).numpy(), rtol=1e-4)


@pytest.mark.parametrize("nhw", [(2, 32, 32), (1, 224, 224), (3, 20, 12)])
def test_stem_fwd_wgrad_x3(nhw):
    from artiboost_amd import kernels as K
    N, H, W = nhw
    g = torch.Generator().manual_seed(N * H)
    x = torch.rand((N, 3, H, W), generator=g) - 0.5
    w = torch.randn((64, 3, 7, 7), generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), stride=2, padding=3)
    xpad = K.image_pad_nhwc4(x.cuda(), torch.float32)
    wst = torch.zeros(64, 7, 8, 4)
    wst[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    y, stats = K.conv2d_stem_fwd_x3(xpad, K.split(wst.cuda()), H, W, want_stats=True)
    close(nchw(y.cpu()), ref)
    yy = y.double().cpu().reshape(-1, 64)
    st = stats.double().sum(0).cpu()
    np.testing.assert_allclose(st[:, 0].numpy(), yy.sum(0).numpy(), rtol=1e-4, atol=1e-4 * float(yy.abs().sum(0).max()))
    np.testing.assert_allclose(st[:, 1].numpy(), (yy * yy).sum(0).numpy(), rtol=1e-4)
    dy = torch.randn(ref.shape, generator=g)
    wr = w.double().clone().requires_grad_(True)
    F.conv2d(x.double(), wr, stride=2, padding=3).backward(dy.double())
    dw = K.conv2d_stem_wgrad_x3(xpad, nhwc(dy).cuda(), H, W).cpu()
    assert float(dw[:, :, 7, :].abs().max()) == 0.0 and float(dw[:, :, :, 3].abs().max()) == 0.0
    close(dw[:, :, :7, :3].permute(0, 3, 1, 2), wr.grad)


def test_bn_producers_write_split_planes():
    """ab_bn_apply_x3 / ab_bn_bwd_x3 == the fp32 kernels followed by ab_split_f32, bit for bit."""
    from artiboost_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    N, H, W, C = 3, 12, 10, 128
    y = torch.randn(N, H, W, C, generator=g).cuda()
    res = torch.randn(N, H, W, C, generator=g).cuda()
    part = K.col_stats(y)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    bnp = K.bn_finalize(part, N * H * W, gamma, beta)
    ref = K.bn_apply(y, bnp, res=res, relu=True)
    got = K.bn_apply_x3(y, bnp, res=res, relu=True, want_f32=True)
    assert torch.equal(got, ref) and torch.equal(got._ab_split, K.split(ref))
    assert torch.equal(K.bn_apply_x3(y, bnp, relu=True), K.split(K.bn_apply(y, bnp, relu=True)))
    dout = torch.randn(N, H, W, C, generator=g).cuda()
    for relu in (True, "recompute", False):
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        dg2, db2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        dy_ref, dz_ref = K.bn_bwd(dout, ref, y, bnp, dg, db, relu=relu, want_dz=True)
        dy, dz = K.bn_bwd_x3(dout, ref, y, bnp, dg2, db2, relu=relu, want_dz=True)
        assert torch.equal(dy, K.split(dy_ref)) and torch.equal(dz, dz_ref)
        assert torch.equal(dg, dg2) and torch.equal(db, db2)


def test_plane_producers_match_split_of_fp32_results():
    """The passes that write split planes themselves (soft-argmax backward, Adam weight refresh, IHWO transposes) == their
    fp32 counterparts followed by ab_split_f32, bit for bit; column sums of planes == column sums of hi + lo."""
    from artiboost_amd import _lib as L
    from artiboost_amd import kernels as K
    from artiboost_amd.head import softargmax3d_bwd, softargmax3d_bwd_x3, softargmax3d_fwd
    g = torch.Generator().manual_seed(9)
    B, H, W, C, D, DP = 3, 8, 8, 22, 28, 32
    logits = (3 * torch.randn(B, H, W, C * DP, generator=g)).cuda()
    uvd, conf, stat = softargmax3d_fwd(logits, C, D, DP)
    gu = torch.randn(B, C, 3, generator=g).cuda()
    ref = softargmax3d_bwd(logits, C, D, DP, uvd, conf, stat, gu)
    got = softargmax3d_bwd_x3(logits, C, D, DP, uvd, conf, stat, gu)
    assert torch.equal(got, K.split(ref))
    out = torch.empty(C * DP, device="cuda")
    K.col_sum_x3(got, out)
    np.testing.assert_allclose(out.cpu().numpy(), (got[0].float() + got[1].float()).sum((0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-6)
    # Adam: the planes written by ab_clip_adam_x3 are the split of the updated parameters
    n = 4096
    p = torch.randn(n, generator=g).cuda()
    gr, m, v = torch.randn(n, generator=g).cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    planes = torch.empty((2, n), dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().ab_clip_adam_x3(L.ptr(p), L.ptr(gr), L.ptr(m), L.ptr(v), L.l(n), L.ptr(None), L.f(0.0), L.f(1e-2), L.f(0.9), L.f(0.999),
                                    L.f(1e-8), L.i(1), L.ptr(None), L.ptr(planes[0]), L.ptr(planes[1]), L.stream()), "ab_clip_adam_x3")
    assert torch.equal(planes, K.split(p))
    # IHWO weight copies as planes
    w = torch.randn(64, 9, 32, generator=g).cuda()                      # [O][K][I]
    dst = torch.zeros((2, 32 * 9 * 64), dtype=torch.bfloat16, device="cuda")
    plan = K.transpose_plan([(w, dst[0].view(32, 9, 64))])
    K.transpose_oki_batch_x3(plan, dst.shape[1])
    assert torch.equal(dst.view(2, 32, 9, 64), K.split(w.permute(2, 1, 0).contiguous()))


BN_CASES = [(2, 16, 16, 64, 64), (1, 7, 7, 512, 512), (5, 6, 6, 256, 256), (2, 32, 32, 64, 64), (1, 28, 28, 128, 128),
            (3, 14, 14, 256, 256), (2, 56, 40, 64, 64), (2, 64, 64, 128, 128), (2, 16, 16, 256, 256), (2, 8, 8, 512, 512),
            # layer 1 at the benchmark's map size: the persistent resident-weight kernel's fused epilogue (conv3x3r.hip, BNR) -- fewer tiles
            # than workgroups (64), and more (9 x 32 = 288 > 256: the workgroups loop, the last tile's waits differ)
            (2, 64, 64, 64, 64), (9, 64, 64, 64, 64), (3, 24, 80, 64, 64)]


@pytest.mark.parametrize("mask_src", ["recompute", "hi_plane"])
@pytest.mark.parametrize("case", BN_CASES)
def test_conv_dgrad_x3_bn_fused(case, mask_src):
    """ab_conv2d_dgrad_x3_bn: the data gradient of a 3x3 conv, masked by the ReLU of the BatchNorm it arrives at, with that
    BatchNorm's backward partial sums from the epilogue -- against float64 torch, and the BatchNorm backward built on the
    partials against the one that runs its own reduction pass."""
    from artiboost_amd import kernels as K
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(hash(case) % 991)
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (Cin * 9)) ** 0.5).double()
    dy = torch.randn((N, Cout, H, W), generator=g).double()
    add = torch.randn((N, Cin, H, W), generator=g)
    ref_dx = F.conv_transpose2d(dy, w, stride=1, padding=1) + add.double()             # data gradient + skip gradient
    ybn = torch.randn((N, H, W, Cin), generator=g) * 2 + 0.3                              # the conv output the BatchNorm saw
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    stats = K.col_stats(ybn.cuda())
    bnp = K.bn_finalize(stats, N * H * W, gamma.cuda(), beta.cuda(), torch.zeros(Cin).cuda(), torch.ones(Cin).cuda())
    res = torch.randn((N, H, W, Cin), generator=g) if mask_src == "hi_plane" else None
    out = K.bn_apply_x3(ybn.cuda(), bnp, res=res.cuda() if res is not None else None, relu=True, want_f32=True)
    wt = K.split(w.float().permute(1, 2, 3, 0).contiguous().cuda())
    dyd = nhwc(dy.float()).cuda()
    dz, part = K.conv2d_dgrad_x3(dyd, wt, (H, W), 1, 1, addend=nhwc(add).cuda(), bn=(ybn.cuda(), out if res is not None else None, bnp))
    assert part is not None
    import os
    c3rb = os.environ.get("AB_C3_L1T16", "1") != "0" and os.environ.get("AB_C3RB", "1") != "0" and os.environ.get("AB_C3R_OFF", "0") == "0"
    if c3rb and (Cin, Cout) == (64, 64) and H % 8 == 0 and W % 16 == 0 and N * (H // 8) * (W // 16) >= 64:
        assert part.shape[0] == min(N * (H // 8) * (W // 16), 256)        # one partial row per persistent workgroup: conv3x3r.hip took it
        # ... and the same launch without the skip gradient (the form conv2's gradient takes in the block backward)
        dz0, part0 = K.conv2d_dgrad_x3(dyd, wt, (H, W), 1, 1, bn=(ybn.cuda(), out if res is not None else None, bnp))
        ref0 = (nhwc(ref_dx) - nhwc(add).double()) * (out.cpu() > 0).double()
        close(dz0.cpu(), ref0)
        np.testing.assert_allclose(part0.double().sum(0).cpu()[:, 0].numpy(), ref0.sum((0, 1, 2)).numpy(), rtol=1e-4,
                                   atol=3e-5 * float(ref0.abs().sum((0, 1, 2)).max()))
    mask = (out.cpu() > 0).double()
    ref_dz = nhwc(ref_dx) * mask
    close(dz.cpu(), ref_dz)
    mean, istd = bnp[2].double().cpu(), bnp[3].double().cpu()
    xhat = (ybn.double() - mean) * istd
    sums = part.double().sum(0).cpu()
    scale = float(ref_dz.abs().sum((0, 1, 2)).max())
    np.testing.assert_allclose(sums[:, 0].numpy(), ref_dz.sum((0, 1, 2)).numpy(), rtol=1e-4, atol=3e-5 * scale)
    np.testing.assert_allclose(sums[:, 1].numpy(), (ref_dz * xhat).sum((0, 1, 2)).numpy(), rtol=1e-4, atol=1e-4 * scale)
    # BatchNorm backward on (dz, part) == the standalone one on the raw gradient
    dg1, db1, dg2, db2 = (torch.zeros(Cin).cuda() for _ in range(4))
    dy1 = K.bn_bwd_x3(dz, None, ybn.cuda(), bnp, dg1, db1, part=part, premasked=True)
    raw = K.conv2d_dgrad_x3(dyd, wt, (H, W), 1, 1, addend=nhwc(add).cuda())
    dy2 = K.bn_bwd_x3(raw, out if res is not None else None, ybn.cuda(), bnp, dg2, db2, relu=True if res is not None else "recompute")
    a = dy1[0].float() + dy1[1].float()
    b = dy2[0].float() + dy2[1].float()
    close(a.cpu(), b.double().cpu(), tol=2e-5)
    np.testing.assert_allclose(dg1.cpu().numpy(), dg2.cpu().numpy(), rtol=2e-4, atol=1e-4 * float(dg2.abs().max()))
    np.testing.assert_allclose(db1.cpu().numpy(), db2.cpu().numpy(), rtol=2e-4, atol=1e-4 * float(db2.abs().max()))


@pytest.mark.parametrize("case", [(2, 32, 32, 256, 704), (3, 16, 16, 64, 96), (1, 32, 32, 128, 64), (2, 24, 20, 256, 160)])
def test_conv1x1_dgrad_x3_bn_fused(case):
    """ab_conv2d_dgrad_x3_bn on a 1x1 convolution (the head's final layer, whose data gradient arrives at relu(bn(deconv output))): the masked
    gradient and the BatchNorm-backward partial rows from the generic kernel's epilogue, against float64 and against the standalone passes."""
    from artiboost_amd import kernels as K
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    w = (torch.randn((Cout, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5).double()
    dy = torch.randn((N, Cout, H, W), generator=g).double()
    ref_dx = F.conv_transpose2d(dy, w)
    ybn = torch.randn((N, H, W, Cin), generator=g) * 2 + 0.3
    gamma, beta = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    bnp = K.bn_finalize(K.col_stats(ybn.cuda()), N * H * W, gamma.cuda(), beta.cuda(), torch.zeros(Cin).cuda(), torch.ones(Cin).cuda())
    out = K.bn_apply_x3(ybn.cuda(), bnp, relu=True, want_f32=True)
    wt = K.split(w.float().permute(1, 2, 3, 0).contiguous().cuda())
    dyd = nhwc(dy.float()).cuda()
    dz, part = K.conv2d_dgrad_x3(dyd, wt, (H, W), 1, 0, bn=(ybn.cuda(), None, bnp))
    assert part is not None and part.shape[1:] == (Cin, 2)
    ref_dz = nhwc(ref_dx) * (out.cpu() > 0).double()
    close(dz.cpu(), ref_dz)
    xhat = (ybn.double() - bnp[2].double().cpu()) * bnp[3].double().cpu()
    sums = part.double().sum(0).cpu()
    scale = float(ref_dz.abs().sum((0, 1, 2)).max())
    np.testing.assert_allclose(sums[:, 0].numpy(), ref_dz.sum((0, 1, 2)).numpy(), rtol=1e-4, atol=3e-5 * scale)
    np.testing.assert_allclose(sums[:, 1].numpy(), (ref_dz * xhat).sum((0, 1, 2)).numpy(), rtol=1e-4, atol=1e-4 * scale)
    dg1, db1, dg2, db2 = (torch.zeros(Cin).cuda() for _ in range(4))
    dy1 = K.bn_bwd_x3(dz, None, ybn.cuda(), bnp, dg1, db1, part=part, premasked=True)
    raw = K.conv2d_dgrad_x3(dyd, wt, (H, W), 1, 0)
    dy2 = K.bn_bwd_x3(raw, None, ybn.cuda(), bnp, dg2, db2, relu="recompute")
    close((dy1[0].float() + dy1[1].float()).cpu(), (dy2[0].float() + dy2[1].float()).double().cpu(), tol=2e-5)
    np.testing.assert_allclose(dg1.cpu().numpy(), dg2.cpu().numpy(), rtol=2e-4, atol=1e-4 * float(dg2.abs().max()))
    np.testing.assert_allclose(db1.cpu().numpy(), db2.cpu().numpy(), rtol=2e-4, atol=1e-4 * float(db2.abs().max()))


@pytest.mark.parametrize("case", [(2, 64, 64, 64, 64), (2, 56, 40, 64, 64), (3, 32, 32, 64, 64)])
def test_conv3x3_x3_256x64_tile(case, monkeypatch):
    """The two tiles of the 64-channel 3x3 layers: 8 x 16 pixels for the forward / plain data gradient (round 3: tools/ab_l1_tiles.py) and
    256 pixels (8 x 32) for the data gradient with the fused BatchNorm-backward epilogue (picked on its own only when the launch has >= 512
    tiles, i.e. at benchmark size; forced here): forward + statistics, data gradient, fused epilogue -- and the partial-row counts that say
    which geometry ran."""
    from artiboost_amd import kernels as K
    monkeypatch.setenv("AB_C3_L1ALT", "2")
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(hash(case) % 983)
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (Cin * 9)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    y, stats = K.conv2d_fwd_x3(nhwc(x).cuda(), K.split(w.permute(0, 2, 3, 1).contiguous().cuda()), 1, 1, want_stats=True)
    close(nchw(y.cpu()), ref)
    import os
    t16 = W % 16 == 0 and H % 8 == 0 and os.environ.get("AB_C3_L1T16", "1") != "0"
    # one partial row per 8 x 16 (else 8 x 32) tile; round 5: from 64 tiles of 8 x 16 on, conv3x3r.hip's persistent workgroups (one row each, <= 256)
    nt16 = N * (H // 8) * (W // 16) if t16 else 0
    want_rows = min(nt16, 256) if nt16 >= 64 else N * ((H + 7) // 8) * ((W + 15) // 16 if t16 else (W + 31) // 32)
    assert stats.shape[0] == want_rows
    yy = y.double().cpu().reshape(-1, Cout)
    np.testing.assert_allclose(stats.double().sum(0).cpu()[:, 1].numpy(), (yy * yy).sum(0).numpy(), rtol=1e-4)
    dy = torch.randn((N, Cout, H, W), generator=g)
    add = torch.randn((N, Cin, H, W), generator=g)
    wt = K.split(w.permute(1, 2, 3, 0).contiguous().cuda())
    ref_dx = F.conv_transpose2d(dy.double(), w.double(), padding=1) + add.double()
    dx = K.conv2d_dgrad_x3(nhwc(dy).cuda(), wt, (H, W), 1, 1, addend=nhwc(add).cuda())
    close(nchw(dx.cpu()), ref_dx)
    ybn = torch.randn((N, H, W, Cin), generator=g)
    bnp = K.bn_finalize(K.col_stats(ybn.cuda()), N * H * W, torch.ones(Cin).cuda(), torch.zeros(Cin).cuda(), torch.zeros(Cin).cuda(), torch.ones(Cin).cuda())
    dz, part = K.conv2d_dgrad_x3(nhwc(dy).cuda(), wt, (H, W), 1, 1, addend=nhwc(add).cuda(), bn=(ybn.cuda(), None, bnp))
    # the fused epilogue keeps the 8 x 32 tile of conv3x3.hip -- except where conv3x3r.hip's persistent kernel takes the launch (round 6: from
    # 64 tiles of 8 x 16 on, one partial row per workgroup), with the 8 x 32 tile again under AB_C3RB=0 / AB_C3R_OFF=1
    rb = nt16 >= 64 and os.environ.get("AB_C3RB", "1") != "0" and os.environ.get("AB_C3R_OFF", "0") == "0"
    assert part.shape[0] == (min(nt16, 256) if rb else N * ((H + 7) // 8) * ((W + 31) // 32))
    ref_dz = nhwc(ref_dx) * (((ybn - bnp[2].cpu()) * bnp[3].cpu()) > 0).double()
    close(dz.cpu(), ref_dz)
    np.testing.assert_allclose(part.double().sum(0).cpu()[:, 0].numpy(), ref_dz.sum((0, 1, 2)).numpy(), rtol=1e-4,
                               atol=3e-5 * float(ref_dz.abs().sum((0, 1, 2)).max()))


@pytest.mark.parametrize("G", [2, 3, 4, 7, 8])
@pytest.mark.parametrize("shape", [(4, 64, 64, 64, 64), (8, 32, 32, 128, 128), (16, 16, 16, 256, 256), (32, 8, 8, 512, 512), (2, 16, 16, 64, 128)])
def test_wgrad_x3_group_equals_the_single_launches(shape, G):
    """ab_conv2d_wgrad_x3_group (round 6): G same-shape 3x3 / stride-1 weight gradients from ONE slab launch + ONE reduction -- against float64
    and against G ab_conv2d_wgrad_x3 calls (the same products summed over another fixed partition of the pixels: fp32 reassociation only);
    destinations are slices of one flat buffer, as the model's gradient views are."""
    from artiboost_amd import kernels as K
    N, H, W, Cin, Cout = shape
    if (Cin // 64) * (Cout // 64) * G > 256:          # more problems than one round of workgroups holds: refused, the model groups fewer
        probe = K.split(torch.zeros((N, H, W, Cin), device="cuda")), K.split(torch.zeros((N, H, W, Cout), device="cuda"))
        assert not K.conv2d_wgrad_x3_group_ok(probe[0], probe[1], G)
        return
    g = torch.Generator().manual_seed(G * 1000 + sum(shape))
    flat = torch.full((G + 1, Cout, 3, 3, Cin), 7.0, device="cuda")                 # (the last slot must stay untouched)
    items, refs = [], []
    for i in range(G):
        x = torch.randn((N, H, W, Cin), generator=g) * (1.0 + i)
        dy = torch.randn((N, H, W, Cout), generator=g)
        xs, ds = K.split(x.cuda()), K.split(dy.cuda())
        items.append((xs, ds if i % 2 else (ds[0], ds[1]), flat[i]))               # split tensors and (hi, lo) tuples alike
        ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double(), padding=1)
        refs.append(ref.permute(0, 2, 3, 1))                                        # [Cout, 3, 3, Cin]
    assert K.conv2d_wgrad_x3_group_ok(items[0][0], items[0][1], G)
    K.conv2d_wgrad_x3_group(items)
    torch.cuda.synchronize()
    assert torch.all(flat[G] == 7.0)
    for i in range(G):
        single = K.conv2d_wgrad_x3(items[i][0], items[i][1], 3, 3, 1, 1)
        scale = float(refs[i].abs().max())
        np.testing.assert_allclose(flat[i].double().cpu().numpy(), refs[i].numpy(), rtol=0, atol=3e-5 * scale)
        np.testing.assert_allclose(flat[i].cpu().numpy(), single.cpu().numpy(), rtol=0, atol=2e-6 * scale)
    again = torch.empty_like(flat)
    K.conv2d_wgrad_x3_group([(a, b, again[i]) for i, (a, b, _) in enumerate(items)])
    assert torch.equal(again[:G], flat[:G])                                          # fixed summation order: bit-identical reruns


@pytest.mark.parametrize("case", [(64, 64, 64, 64, 128, 1, 2, 0), (8, 32, 32, 128, 256, 3, 2, 1), (4, 16, 16, 256, 256, 3, 1, 1),
                                  (16, 32, 32, 256, 704, 1, 1, 0)])
def test_wgrad_x3_deferred_reduction_with_exact_workspace(case):
    """Deferred slab reductions (several weight gradients reduced by ONE launch) give the same bits as the immediate ones, each
    on a workspace of exactly ab_conv2d_wgrad_x3_workspace bytes (the 1x1/s2 downsample shape at B = 64 slices its pixels 256
    ways: twice what the bf16 kernels' workspace formula provides)."""
    from artiboost_amd import kernels as K
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(Cin + Cout)
    x = K.split(torch.randn((N, H, W, Cin), generator=g).cuda())
    Ho, Wo = K.conv_out(H, k, s, p), K.conv_out(W, k, s, p)
    dy = K.split(torch.randn((N, Ho, Wo, Cout), generator=g).cuda())
    ref = K.conv2d_wgrad_x3(x, dy, k, k, s, p)
    pend = K.PendingReductions()
    a = K.conv2d_wgrad_x3(x, dy, k, k, s, p, defer=pend)
    b = K.conv2d_wgrad_x3(x, dy, k, k, s, p, defer=pend)
    assert len(pend.descs) == 2
    pend.flush()
    torch.cuda.synchronize()
    assert torch.equal(a, ref) and torch.equal(b, ref)


@pytest.mark.parametrize("case", [(64, 64, 64, 64, 128, 1, 2, 0), (8, 32, 32, 128, 256, 3, 2, 1), (4, 16, 16, 256, 256, 3, 1, 1),
                                  (16, 32, 32, 256, 704, 1, 1, 0), (64, 8, 8, 512, 512, 3, 1, 1)])
def test_wgrad_bf16_deferred_reduction_with_exact_workspace(case):
    """The same exact-size-workspace check for the bf16 weight gradients (ab_conv2d_wgrad_workspace)."""
    from artiboost_amd import kernels as K
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).cuda()
    Ho, Wo = K.conv_out(H, k, s, p), K.conv_out(W, k, s, p)
    dy = torch.randn((N, Ho, Wo, Cout), generator=g).to(torch.bfloat16).cuda()
    ref = K.conv2d_wgrad(x, dy, k, k, s, p)
    pend = K.PendingReductions()
    a = K.conv2d_wgrad(x, dy, k, k, s, p, defer=pend)
    b = K.conv2d_wgrad(x, dy, k, k, s, p, defer=pend)
    pend.flush()
    torch.cuda.synchronize()
    assert torch.equal(a, ref) and torch.equal(b, ref)


def test_stem_wgrad_x3_beyond_2p21_pixels():
    """A stem weight gradient over more than 2^21 output pixels (batch 128 at 256 x 256 -- the reference YAML's TRAIN.BATCH_SIZE on
    one GPU) runs as two half-batches; same result as the sum of the halves computed separately."""
    from artiboost_amd import kernels as K
    N, H, W = 130, 256, 256
    g = torch.Generator().manual_seed(1)
    xpad = K.split(K.image_pad_nhwc4((torch.rand((N, 3, H, W), generator=g) - 0.5).cuda(), torch.float32))
    dy = K.split(torch.randn((N, H // 2, W // 2, 64), generator=g).cuda())
    assert N * (H // 2) * (W // 2) >= 1 << 21
    dw = K.conv2d_stem_wgrad_x3(xpad, dy, H, W)
    h = N // 2
    a = K.conv2d_stem_wgrad_x3(xpad[:, :h].contiguous(), dy[:, :h].contiguous(), H, W)
    b = K.conv2d_stem_wgrad_x3(xpad[:, h:].contiguous(), dy[:, h:].contiguous(), H, W)
    ref = a.double() + b.double()
    assert float((dw.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("case", [(16, 32, 32, 256, 704, 1, 1, 0),     # 256 x 256 tiles, partial last channel tile (704 = 2.75 x 256)
                                  (16, 32, 32, 256, 320, 1, 1, 0),     # ... a last tile of 64 channels
                                  (64, 32, 32, 256, 256, 4, 2, 1),     # 256 x 256 tiles, 16 taps (the transposed 256->256 layer)
                                  (8, 32, 32, 128, 256, 3, 2, 1),      # 256 x 128 tiles
                                  (64, 16, 16, 256, 512, 3, 2, 1)])    # 256 x 128 tiles, 72 of them: the slice count follows rounds of 256
def test_wgrad_x3_large_tiles_vs_fp64(case):
    """The 256-channel weight-gradient tiles (two-stage ring for 256 x 256, guarded last channel tile) and the round-aware slice
    count against a float64 weight gradient of the same operands."""
    from artiboost_amd import kernels as K
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(Cin * 7 + Cout + k)
    x = torch.randn((N, H, W, Cin), generator=g).cuda()
    Ho, Wo = K.conv_out(H, k, s, p), K.conv_out(W, k, s, p)
    dy = torch.randn((N, Ho, Wo, Cout), generator=g).cuda()
    ref = torch.nn.grad.conv2d_weight(nchw(x).double(), (Cout, Cin, k, k), nchw(dy).double(), stride=s, padding=p)
    dw = K.conv2d_wgrad_x3(K.split(x), K.split(dy), k, k, s, p)
    close(dw.permute(0, 3, 1, 2).cpu(), ref.cpu())


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 128), (3, 12, 20, 128, 256), (2, 8, 8, 256, 512), (8, 64, 64, 64, 128)])
@pytest.mark.parametrize("with_addend", [False, True])
def test_conv_dgrad_x3_pair_vs_fp64(case, with_addend):
    """conv1 (3x3/s2) and downsample (1x1/s2) data gradients of a down-sampling block in one launch
    (ab_conv2d_dgrad_x3_pair) against the float64 sum of the two transposed convolutions, and against the two-launch route."""
    from artiboost_amd import kernels as K
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(Cin + Cout + H)
    w1 = (torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (Cin * 9)) ** 0.5).cuda()
    w2 = (torch.randn((Cout, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5).cuda()
    dy1 = torch.randn((N, Cout, H // 2, W // 2), generator=g).cuda()
    dy2 = torch.randn((N, Cout, H // 2, W // 2), generator=g).cuda()
    add = torch.randn((N, H, W, Cin), generator=g).cuda() if with_addend else None
    ref = (torch.nn.grad.conv2d_input((N, Cin, H, W), w1.double(), dy1.double(), stride=2, padding=1)
           + torch.nn.grad.conv2d_input((N, Cin, H, W), w2.double(), dy2.double(), stride=2, padding=0))
    if with_addend:
        ref = ref + nchw(add).double()
    wt1 = K.split(w1.permute(1, 2, 3, 0).contiguous())      # [Cin][kh][kw][Cout]
    wt2 = K.split(w2.permute(1, 2, 3, 0).contiguous())
    dx = K.conv2d_dgrad_x3_pair(K.split(nhwc(dy1)), wt1, K.split(nhwc(dy2)), wt2, (H, W), 1, addend=add)
    close(nchw(dx).cpu(), ref.cpu())
    two = K.conv2d_dgrad_x3(nhwc(dy2), wt2, (H, W), 2, 0, addend=K.conv2d_dgrad_x3(nhwc(dy1), wt1, (H, W), 2, 1, addend=add))
    close(dx.cpu(), two.cpu().double(), tol=2e-6)


def test_bn_apply_x3_with_downsample_bn_residual():
    """relu(bn2(y2) + bn_ds(yd)) in one pass (ab_bn_apply_x3_resbn) == the two-pass route that stores bn_ds(yd) first."""
    from artiboost_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    y2 = torch.randn((3, 10, 12, 128), generator=g).cuda()
    yd = torch.randn((3, 10, 12, 128), generator=g).cuda()
    bnp2 = torch.randn((4, 128), generator=g).cuda()
    bnpd = torch.randn((4, 128), generator=g).cuda()
    r = K.bn_apply(yd, bnpd, res=None, relu=False)
    ref = K.bn_apply_x3(y2, bnp2, res=r, relu=True, want_f32=True)
    got = K.bn_apply_x3(y2, bnp2, res=yd, relu=True, want_f32=True, res_bnp=bnpd)
    exact = torch.relu(y2.double() * bnp2[0].double() + bnp2[1].double() + yd.double() * bnpd[0].double() + bnpd[1].double())
    close(got.cpu(), exact.cpu(), tol=1e-6)
    close(got.cpu(), ref.cpu().double(), tol=1e-6)
    assert torch.equal(got._ab_split.cpu(), K.split(got).cpu())


def test_bn_apply_x3_with_residual_as_planes():
    """relu(bn(y) + (res_hi + res_lo)) (ab_bn_apply_x3_respl) == ab_bn_apply_x3 fed the fp32 tensor hi + lo, bit for bit: a block
    input that exists only as the planes its producer wrote adds exactly what those planes hold."""
    from artiboost_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    y = torch.randn((2, 9, 7, 256), generator=g).cuda()
    res = torch.randn((2, 9, 7, 256), generator=g).cuda()
    bnp = torch.randn((4, 256), generator=g).cuda()
    pl = K.split(res)
    held = pl[0].float() + pl[1].float()                  # what the planes hold (2^-17 of res)
    assert float((held - res).abs().max()) <= 2.0 ** -16 * float(res.abs().max())
    ref = K.bn_apply_x3(y, bnp, res=held, relu=True, want_f32=True)
    got = K.bn_apply_x3(y, bnp, res=pl, relu=True, want_f32=True)
    assert torch.equal(got, ref) and torch.equal(got._ab_split, ref._ab_split)
    only = K.bn_apply_x3(y, bnp, res=pl, relu=True)       # planes only: no fp32 copy is written
    assert only.dtype == torch.bfloat16 and torch.equal(only, ref._ab_split)


def test_previous_tile_choices_still_correct():
    """The tile switches are read once per process: the whole file again in a child with the round-3 choices off (AB_C3_L1T16=0: 256-pixel
    layer-1 tile, AB_C3_STACK=0: one 8 x 8 image x 128 channels, AB_C3_ALT16=0: half a 16 x 16 image x 128 channels)."""
    import os
    import subprocess
    import sys
    if os.environ.get("AB_C3_STACK") == "0":
        pytest.skip("already the child")
    env = dict(os.environ, AB_C3_L1T16="0", AB_C3_STACK="0", AB_C3_ALT16="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "--timeout", "900"], env=env, capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.mark.parametrize("C,rows", [(256, 64), (512, 32), (64, 7)])
@pytest.mark.parametrize("resmode", ["none", "f32", "planes", "resbn"])
def test_bn_finalize_inside_the_apply_passes(C, rows, resmode):
    """ab_bn_fin_apply_x3 (round 4: the statistics are finalized in the apply pass's prologue, one launch instead of two) == ab_bn_finalize +
    ab_bn_apply_x3* : same (scale, shift, mean, invstd) and running statistics to 1e-6 (the partial rows are summed in another fixed order),
    the same activation planes; and ab_bn_bwd_x3 given few partial rows (its in-kernel finalize) == the same call with the reduction redone
    by its own reduce + finalize launches."""
    from artiboost_amd import kernels as K
    g = torch.Generator().manual_seed(C + rows)
    N, H, W = rows, 6, 4                                         # one partial row per image
    y = (torch.randn(N, H, W, C, generator=g) * 2 + 0.3).cuda()
    part = torch.stack([y.reshape(N, -1, C).sum(1), (y * y).reshape(N, -1, C).sum(1)], -1).contiguous()
    assert K.bn_fin_apply_x3_ok(part, C) and not K.bn_fin_apply_x3_ok(torch.empty(65, C, 2), C)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    count = N * H * W
    res = res_bnp = None
    if resmode == "f32":
        res = torch.randn(N, H, W, C, generator=g).cuda()
    elif resmode == "planes":
        res = K.split(torch.randn(N, H, W, C, generator=g).cuda())
    elif resmode == "resbn":
        res = torch.randn(N, H, W, C, generator=g).cuda()
        res_bnp = torch.randn(4, C, generator=g).cuda()
    rm1, rv1, rm2, rv2 = torch.zeros(C).cuda(), torch.ones(C).cuda(), torch.zeros(C).cuda(), torch.ones(C).cuda()
    bnp_ref = K.bn_finalize(part, count, gamma, beta, rm1, rv1)
    ref = K.bn_apply_x3(y, bnp_ref, res=res, relu=True, want_f32=True, res_bnp=res_bnp)
    got, bnp = K.bn_fin_apply_x3(y, part, count, gamma, beta, rm2, rv2, res=res, relu=True, want_f32=True, res_bnp=res_bnp)
    torch.testing.assert_close(bnp, bnp_ref, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(rm2, rm1, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(rv2, rv1, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(got, ref, rtol=2e-6, atol=2e-6)
    sp, spr = got._ab_split.float(), ref._ab_split.float()
    torch.testing.assert_close(sp[0] + sp[1], spr[0] + spr[1], rtol=2e-6, atol=2e-6)
    # backward: few partial rows handed over (in-kernel finalize) vs the call that reduces for itself
    dz = torch.randn(N, H, W, C, generator=g).cuda()
    xhat = (y - bnp_ref[2]) * bnp_ref[3]
    bpart = torch.stack([dz.reshape(N, -1, C).sum(1), (dz * xhat).reshape(N, -1, C).sum(1)], -1).contiguous()
    dg1, db1, dg2, db2 = (torch.zeros(C).cuda() for _ in range(4))
    dy1 = K.bn_bwd_x3(dz, None, y, bnp_ref, dg1, db1, relu=False, part=bpart)
    dy2 = K.bn_bwd_x3(dz, None, y, bnp_ref, dg2, db2, relu=False)
    torch.testing.assert_close(dg1, dg2, rtol=2e-5, atol=2e-4)
    torch.testing.assert_close(db1, db2, rtol=2e-5, atol=2e-4)
    torch.testing.assert_close(dy1.float().sum(0), dy2.float().sum(0), rtol=1e-4, atol=2e-5)


# ---- round 5: the register-resident GEMM of the final layer (gemm_rw.hip) and its fused soft-argmax statistics
RW_CASES = [
    # B, H, W, Cin, C (classes; Cout = C * 32), D
    (2, 8, 8, 256, 22, 28),       # one 64-pixel tile per image, three channel groups (8 + 8 + 6 waves)
    (3, 16, 8, 256, 22, 28),      # W < 32: a 32-pixel block spans four image rows
    (2, 32, 32, 256, 22, 28),     # benchmark head geometry (per image)
    (1, 8, 16, 128, 3, 32),       # K = 128, one partial channel group, no padding bins
    (5, 8, 8, 64, 9, 17),         # K = 64, two groups, odd unit count
]


@pytest.mark.parametrize("case", RW_CASES)
def test_final_layer_gemm_rw_and_fused_softargmax(case):
    """ab_conv2d_fwd_x3 (1x1 route) and ab_conv1x1_sam_fwd_x3 against float64, the per-tile statistics against a float64 softmax of
    the kernel's own logits, and uvd / conf against the oracle formula (simplebaseline.py:43-71, 183-189)."""
    from artiboost_amd import kernels as K
    from artiboost_amd.head import softargmax3d_fwd, softargmax3d_stage2
    B, H, W, Cin, C, D = case
    Cout = C * 32
    g = torch.Generator().manual_seed(hash(case) % 991)
    x = torch.randn((B, H, W, Cin), generator=g)
    w = torch.randn((Cout, Cin), generator=g) * (4.0 / Cin) ** 0.5
    w.view(C, 32, Cin)[:, D:] = 0                       # padding depth bins carry zero weights
    b = torch.randn(Cout, generator=g)
    b.view(C, 32)[:, D:] = 0
    ref = x.double().reshape(-1, Cin) @ w.double().t() + b.double()
    ws = K.split(w.view(Cout, 1, 1, Cin).contiguous().cuda())
    xs = K.split(x.cuda())
    y = K.conv2d_fwd_x3(xs, ws, 1, 0, bias=b.cuda())                      # plain route (no statistics)
    close(y.cpu().reshape(-1, Cout), ref)
    assert K.conv1x1_sam_fwd_x3_ok(xs, ws, C, D, 32)
    y2, part = K.conv1x1_sam_fwd_x3(xs, ws, b.cuda(), C, D)
    assert torch.equal(y2, y)                                              # the same GEMM, bit for bit
    # statistics of every (image, tile, class) against float64 on the kernel's logits
    lg = y2.double().cpu().view(B, H * W // 64, 64, C, 32)[..., :D]        # [B, tile, pix, C, D]
    m = lg.amax(dim=(2, 4))
    e = torch.exp(lg - m[:, :, None, :, None])
    pix = torch.arange(H * W).view(H * W // 64, 64)
    cu = ((pix % W).double() / W)[None, :, :, None, None]
    cv = ((pix // W).double() / H)[None, :, :, None, None]
    cd = (torch.arange(D).double() / D)[None, None, None, None, :]
    pp = part.double().cpu()
    np.testing.assert_array_equal(pp[..., 0].numpy(), m.numpy())
    for k, ref_k in ((1, e.sum((2, 4))), (2, (e * cu).sum((2, 4))), (3, (e * cv).sum((2, 4))), (4, (e * cd).sum((2, 4)))):
        np.testing.assert_allclose(pp[..., k].numpy(), ref_k.numpy(), rtol=2e-6, atol=1e-7)
    # stage 2 of the fused rows == the two-stage kernel on the same logits (to rounding), and == the float64 integral
    uvd, conf, stat = softargmax3d_stage2(part, C)
    uvd0, conf0, stat0 = softargmax3d_fwd(y2, C, D, 32)
    np.testing.assert_allclose(uvd.cpu().numpy(), uvd0.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(conf.cpu().numpy(), conf0.cpu().numpy(), rtol=3e-6)
    np.testing.assert_array_equal(stat[..., 0].cpu().numpy(), stat0[..., 0].cpu().numpy())
    np.testing.assert_allclose(stat[..., 1].cpu().numpy(), stat0[..., 1].cpu().numpy(), rtol=3e-6)
    full = y2.double().cpu().view(B, H * W, C, 32)[..., :D]
    p = torch.softmax(full.permute(0, 2, 1, 3).reshape(B, C, -1), dim=-1).view(B, C, H * W, D)
    p = p / (p.sum((2, 3), keepdim=True) + 1e-7)
    allpix = torch.arange(H * W)
    u = (p.sum(3) * ((allpix % W).double() / W)).sum(2)
    v = (p.sum(3) * ((allpix // W).double() / H)).sum(2)
    dd = (p.sum(2) * (torch.arange(D).double() / D)).sum(2)
    np.testing.assert_allclose(uvd.cpu().numpy(), torch.stack([u, v, dd], -1).numpy(), atol=3e-6)


# ---- round 5: the transposed convolutions of the head on a resident patch (conv2x2.hip)
@pytest.mark.parametrize("Ci,Co,hin,nb", [(256, 256, 16, 3), (64, 128, 16, 3), (96, 64, 16, 3), (512, 256, 8, 4), (64, 64, 8, 8), (96, 128, 8, 4)])
def test_conv_transpose_4x4s2_patch_kernel_32x32(Ci, Co, hin, nb):
    """ConvTranspose2d(4x4, s2, p1) 16x16 -> 32x32 and 8x8 -> 16x16 (deconv_layers.3 / .0 of the head): forward with BatchNorm partials, and
    its data gradient (the 4x4 / s2 convolution over the four parity sub-grids), both against float64."""
    from artiboost_amd import kernels as K
    g = torch.Generator().manual_seed(Ci + Co + hin)
    x = torch.randn(nb, Ci, hin, hin, generator=g)
    w = torch.randn(Ci, Co, 4, 4, generator=g) * (2.0 / (Ci * 4)) ** 0.5          # ConvT weight [Cin_t, Cout_t, kh, kw]
    ref = F.conv_transpose2d(x.double(), w.double(), stride=2, padding=1)
    wt = K.split(w.permute(1, 2, 3, 0).contiguous().cuda())                     # [Co][kh][kw][Ci]
    y, part = K.conv2d_dgrad_x3(nhwc(x).cuda(), wt, (2 * hin, 2 * hin), 2, 1, want_stats=True)
    close(nchw(y.cpu()), ref)
    assert part.shape[0] == (nb * 4 if hin == 16 else nb)                       # the patch kernel's rows: the new path ran
    yy = y.double().cpu().reshape(-1, Co)
    st = part.double().sum(0).cpu()
    np.testing.assert_allclose(st[:, 0].numpy(), yy.sum(0).numpy(), rtol=1e-4, atol=1e-4 * float(yy.abs().sum(0).max()))
    np.testing.assert_allclose(st[:, 1].numpy(), (yy * yy).sum(0).numpy(), rtol=1e-4)
    # data gradient of the layer = conv2d(dy, W as [Cin_t (out), Cout_t (in)], 4x4, s2, p1)
    dy = torch.randn(nb, Co, 2 * hin, 2 * hin, generator=g)
    refg = F.conv2d(dy.double(), w.double(), stride=2, padding=1)               # weight [out = Ci, in = Co, 4, 4]
    ws = K.split(w.permute(0, 2, 3, 1).contiguous().cuda())                     # OHWI [Ci][kh][kw][Co]
    dx = K.conv2d_fwd_x3(nhwc(dy).cuda(), ws, 2, 1)
    close(nchw(dx.cpu()), refg)
    dx2, st2 = K.conv2d_fwd_x3(nhwc(dy).cuda(), ws, 2, 1, want_stats=True)
    assert torch.equal(dx2, dx)
    if Ci % 64 == 0:                                                            # (output channels of the data gradient: the patch kernel's tiling)
        assert st2.shape[0] == (nb if hin == 16 else nb // 2)
    dd = dx.double().cpu().reshape(-1, Ci)
    np.testing.assert_allclose(st2.double().sum(0).cpu()[:, 0].numpy(), dd.sum(0).numpy(), rtol=1e-4, atol=1e-4 * float(dd.abs().sum(0).max()))
    np.testing.assert_allclose(st2.double().sum(0).cpu()[:, 1].numpy(), (dd * dd).sum(0).numpy(), rtol=1e-4)


# ---- round 5: the stride-2 3x3 convolutions of the stage entries on resident patches (convp.hip)
@pytest.mark.parametrize("Ci,Co,hin,nb", [(64, 128, 64, 2), (128, 256, 32, 3), (256, 512, 16, 4), (160, 64, 32, 1), (32, 192, 16, 2), (128, 64, 96, 1), (192, 128, 16, 6)])
def test_stride2_3x3_patch_kernels(Ci, Co, hin, nb):
    """layerN.0.conv1 (3x3, s2, p1) forward with BatchNorm partials -- nine taps over the four parity sub-grid patches -- and its data gradient
    merged with the 1x1 / s2 downsample branch's (four output-parity classes of 1 / 2 / 2 / 4 taps, the downsample as a second unit of class
    (even, even)), both against float64 and against the tap-by-tap kernel they replace."""
    import os, subprocess, sys
    from artiboost_amd import kernels as K
    g = torch.Generator().manual_seed(Ci * 3 + Co + hin)
    x = torch.randn(nb, Ci, hin, hin, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (Ci * 9)) ** 0.5
    wd = torch.randn(Co, Ci, 1, 1, generator=g) * (2.0 / Ci) ** 0.5
    ref = F.conv2d(x.double(), w.double(), stride=2, padding=1)
    ws = K.split(w.permute(0, 2, 3, 1).contiguous().cuda())                      # OHWI
    y, part = K.conv2d_fwd_x3(nhwc(x).cuda(), ws, 2, 1, want_stats=True)
    close(nchw(y.cpu()), ref)
    ho = hin // 2
    if Ci >= 128:                                                                # (two 32-channel chunks at least; 64 input channels stay on the
        assert part.shape[0] == (nb * (ho // 16) ** 2 if ho % 16 == 0 else nb // 2)     #  tap-by-tap kernel, which is as fast there) the patch kernel's rows
    yy = y.double().cpu().reshape(-1, Co)
    st = part.double().sum(0).cpu()
    np.testing.assert_allclose(st[:, 0].numpy(), yy.sum(0).numpy(), rtol=1e-4, atol=1e-4 * float(yy.abs().sum(0).max()))
    np.testing.assert_allclose(st[:, 1].numpy(), (yy * yy).sum(0).numpy(), rtol=1e-4)
    y0 = K.conv2d_fwd_x3(nhwc(x).cuda(), ws, 2, 1)                                # no statistics asked: the same kernel, the same bits
    assert torch.equal(y0, y)
    code = ("import torch; from artiboost_amd import kernels as K; a = torch.load(r'%s'); "
            "y = K.conv2d_fwd_x3(a['x'].cuda(), K.split(a['w'].cuda()), 2, 1); torch.save(y.cpu(), r'%s')")
    import tempfile
    with tempfile.TemporaryDirectory() as td:                                     # the tap-by-tap kernel it replaces (AB_CP_OFF=1 in a child process)
        torch.save({"x": nhwc(x), "w": w.permute(0, 2, 3, 1).contiguous()}, td + "/in.pt")
        r = subprocess.run([sys.executable, "-c", code % (td + "/in.pt", td + "/out.pt")], env=dict(os.environ, AB_CP_OFF="1"), capture_output=True,
                           text=True, timeout=600, cwd=os.path.join(os.path.dirname(__file__), ".."))
        assert r.returncode == 0, r.stderr[-2000:]
        y1 = torch.load(td + "/out.pt")
    np.testing.assert_allclose(y.cpu().numpy(), y1.numpy(), rtol=0, atol=2e-5 * float(ref.abs().max()))
    # eval mode: the following BatchNorm (+ ReLU) as the epilogue's affine == this kernel + ab_bn_apply_x3, bit for bit
    gam, bet = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.3
    bnp = K.bn_eval_params(gam.cuda(), bet.cuda(), (torch.randn(Co, generator=g) * 0.1).cuda(), (torch.rand(Co, generator=g) + 0.5).cuda())
    fold = K.conv2d_fwd_x3_affine(nhwc(x).cuda(), ws, bnp, 2, 1, relu=True)
    two = K.bn_apply_x3(y0, bnp, relu=True)
    assert torch.equal(fold[0], two[0]) and torch.equal(fold[1], two[1])
    f32 = K.conv2d_fwd_x3_affine(nhwc(x).cuda(), ws, bnp, 2, 1, relu=False, planes=False)
    close(nchw(f32.cpu()), ref * bnp[0].double().cpu().view(1, -1, 1, 1) + bnp[1].double().cpu().view(1, -1, 1, 1))
    # data gradients: conv1 alone, and conv1 + downsample in one launch
    dy = torch.randn(nb, Co, ho, ho, generator=g)
    dyd = torch.randn(nb, Co, ho, ho, generator=g)
    xr = x.double().clone().requires_grad_(True)
    F.conv2d(xr, w.double(), stride=2, padding=1).backward(dy.double())
    ref1 = xr.grad.clone()
    xr.grad = None
    (F.conv2d(xr, w.double(), stride=2, padding=1) * dy.double()).sum().add((F.conv2d(xr, wd.double(), stride=2) * dyd.double()).sum()).backward()
    ref2 = xr.grad
    wt = K.split(w.permute(1, 2, 3, 0).contiguous().cuda())                      # IHWO
    wdt = K.split(wd.permute(1, 2, 3, 0).contiguous().cuda())
    dx1 = K.conv2d_dgrad_x3(nhwc(dy).cuda(), wt, (hin, hin), 2, 1)
    close(nchw(dx1.cpu()), ref1)
    dx2 = K.conv2d_dgrad_x3_pair(nhwc(dy).cuda(), wt, nhwc(dyd).cuda(), wdt, (hin, hin), 1)
    close(nchw(dx2.cpu()), ref2)
    # weight gradient of the strided convolution (the tap-by-tap kernel: an all-taps form over the four parity sub-grids measured slower)
    if Ci % 64 == 0 and Co % 64 == 0:                                             # (the split-bf16 weight gradients take channels in 64s)
        wr = w.double().clone().requires_grad_(True)
        F.conv2d(x.double(), wr, stride=2, padding=1).backward(dy.double())
        dw = K.conv2d_wgrad_x3(nhwc(x).cuda(), nhwc(dy).cuda(), 3, 3, 2, 1)
        close(dw.cpu().permute(0, 3, 1, 2), wr.grad)
        base = torch.randn(dw.shape, generator=g).cuda()
        dw2 = K.conv2d_wgrad_x3(K.split(nhwc(x).cuda()), K.split(nhwc(dy).cuda()), 3, 3, 2, 1, out=base.clone(), accumulate=True)
        close((dw2 - base).cpu().permute(0, 3, 1, 2), wr.grad)          # (accumulated onto a random base: its rounding rides along)
    # ... arriving at relu(bn(y) + residual) of the stage below: masked gradient + that BatchNorm's backward partial rows from the epilogue
    ybn = torch.randn((nb, hin, hin, Ci), generator=g) * 2 + 0.3
    gamma, beta = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.3
    bnp = K.bn_finalize(K.col_stats(ybn.cuda()), nb * hin * hin, gamma.cuda(), beta.cuda(), torch.zeros(Ci).cuda(), torch.ones(Ci).cuda())
    for with_res in (True, False):
        res = torch.randn((nb, hin, hin, Ci), generator=g).cuda() if with_res else None
        out = K.bn_apply_x3(ybn.cuda(), bnp, res=res, relu=True, want_f32=True)
        dz, part = K.conv2d_dgrad_x3_pair(nhwc(dy).cuda(), wt, nhwc(dyd).cuda(), wdt, (hin, hin), 1, bn=(ybn.cuda(), out if with_res else None, bnp))
        mask = (out.cpu() > 0).double()
        ref_dz = nhwc(ref2) * mask
        if Ci % 64 or (ho % 16 and (ho != 8 or nb % 2)):
            assert part is None                                                   # not the patch kernel's shape: raw gradient, no rows
            close(dz.cpu(), nhwc(ref2))
            continue
        half = 2 if os.environ.get("AB_CP_HALF", "0") == "1" else 1                     # (opt-in 8 x 16 tiles: twice the rows)
        assert part.shape[0] == 4 * (nb * (ho // 16) ** 2 * half if ho % 16 == 0 else nb // 2)
        close(dz.cpu(), ref_dz)
        xhat = (ybn.double() - bnp[2].double().cpu()) * bnp[3].double().cpu()
        sums = part.double().sum(0).cpu()
        scale = float(ref_dz.abs().sum((0, 1, 2)).max())
        np.testing.assert_allclose(sums[:, 0].numpy(), ref_dz.sum((0, 1, 2)).numpy(), rtol=1e-4, atol=3e-5 * scale)
        np.testing.assert_allclose(sums[:, 1].numpy(), (ref_dz * xhat).sum((0, 1, 2)).numpy(), rtol=1e-4, atol=1e-4 * scale)
        # the BatchNorm backward built on the rows == the one that runs its own reduction pass over the raw gradient
        dg1, db1, dg2, db2 = (torch.zeros(Ci).cuda() for _ in range(4))
        dy1 = K.bn_bwd_x3(dz, None, ybn.cuda(), bnp, dg1, db1, part=part, premasked=True)
        dy2 = K.bn_bwd_x3(dx2.clone(), out if with_res else None, ybn.cuda(), bnp, dg2, db2, relu=True if with_res else "recompute")
        close((dy1[0].float() + dy1[1].float()).cpu(), (dy2[0].float() + dy2[1].float()).double().cpu(), tol=2e-5)
        np.testing.assert_allclose(dg1.cpu().numpy(), dg2.cpu().numpy(), rtol=2e-4, atol=1e-4 * float(dg2.abs().max()))
        np.testing.assert_allclose(db1.cpu().numpy(), db2.cpu().numpy(), rtol=2e-4, atol=1e-4 * float(db2.abs().max()))


@pytest.mark.parametrize("nhw", [(2, 64, 64), (3, 128, 96), (2, 256, 256)])
def test_stem_on_the_integer_image_plane(nhw):
    """Round 5 (review item 4): the loaders can write the padded image as ONE bf16 plane of the odd integers n = 2 v - 255 (AB_DT_U8N); the
    stem's forward and weight gradient then run TWO MFMA passes (n . w_hi + n . w_lo; dy_hi . n + dy_lo . n) with 1 / 510 in the epilogue.
    Against float64 on the exact image v / 255 - 0.5 both are at least as close as the three-pass path on the fp32 image."""
    from artiboost_amd import kernels as K
    N, H, W = nhw
    g = torch.Generator().manual_seed(N * H + 1)
    v = torch.randint(0, 256, (N, 3, H, W), generator=g)
    x64 = v.double() / 255.0 - 0.5
    w = torch.randn((64, 3, 7, 7), generator=g) * 0.1
    ref = F.conv2d(x64, w.double(), stride=2, padding=3)
    n = torch.zeros((N, H + 6, W + 8, 4), dtype=torch.bfloat16)
    n[:, 3:-3, 3:-5, :3] = (2 * v - 255).permute(0, 2, 3, 1).to(torch.bfloat16)
    assert torch.equal(n[:, 3:-3, 3:-5, :3].float(), (2 * v - 255).permute(0, 2, 3, 1).float())        # exact in bf16
    xpad32 = K.image_pad_nhwc4((v.float() / 255.0 - 0.5).cuda(), torch.float32)
    wst = torch.zeros(64, 7, 8, 4)
    wst[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    ws = K.split(wst.cuda())
    y, stats = K.conv2d_stem_fwd_x3(n.cuda(), ws, H, W, want_stats=True)
    y3 = K.conv2d_stem_fwd_x3(xpad32, ws, H, W)
    close(nchw(y.cpu()), ref)
    e2 = float((nchw(y.cpu()).double() - ref).abs().max()), float((nchw(y3.cpu()).double() - ref).abs().max())
    assert e2[0] <= 1.25 * e2[1] + 1e-7, e2                      # not worse than the three-pass path (it is better: the image is exact)
    yy = y.double().cpu().reshape(-1, 64)
    st = stats.double().sum(0).cpu()
    np.testing.assert_allclose(st[:, 0].numpy(), yy.sum(0).numpy(), rtol=1e-4, atol=1e-4 * float(yy.abs().sum(0).max()))
    np.testing.assert_allclose(st[:, 1].numpy(), (yy * yy).sum(0).numpy(), rtol=1e-4)
    dy = torch.randn(ref.shape, generator=g)
    wr = w.double().clone().requires_grad_(True)
    F.conv2d(x64, wr, stride=2, padding=3).backward(dy.double())
    dw = K.conv2d_stem_wgrad_x3(n.cuda(), nhwc(dy).cuda(), H, W).cpu()
    assert float(dw[:, :, 7, :].abs().max()) == 0.0 and float(dw[:, :, :, 3].abs().max()) == 0.0
    close(dw[:, :, :7, :3].permute(0, 3, 1, 2), wr.grad)
    dw3 = K.conv2d_stem_wgrad_x3(xpad32, nhwc(dy).cuda(), H, W).cpu()
    e = [float((d[:, :, :7, :3].permute(0, 3, 1, 2).double() - wr.grad).abs().max()) for d in (dw, dw3)]
    assert e[0] <= 1.25 * e[1] + 1e-7, e

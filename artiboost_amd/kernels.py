"""Tensor-level wrappers over the C ABI (include/artiboost_hip.h).  No autograd here: every function launches HIP
kernels on the current stream and returns device tensors.  Layouts: activations NHWC, weights OHWI / IHWO."""
import ctypes

import torch

from . import _lib as L


def _empty(shape, like, dtype=None):
    return torch.empty(shape, dtype=dtype or like.dtype, device=like.device)


def conv_out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def conv2d_fwd(x, w_ohwi, stride, pad, bias=None, want_stats=False, relu=False):
    """x [N,H,W,Cin], w [Cout,kh,kw,Cin] -> y [N,Ho,Wo,Cout] (+ per-tile BN partials)."""
    N, H, W, Cin = x.shape
    Cout, kh, kw, _ = w_ohwi.shape
    Ho, Wo = conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad)
    y = _empty((N, Ho, Wo, Cout), x)
    lib = L.lib()
    stats = None
    if want_stats:
        nt = lib.ab_conv2d_stat_rows(L.i(L.dt(x)), L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.i(0))
        stats = torch.empty((nt, Cout, 2), dtype=torch.float32, device=x.device)
    L.check(lib.ab_conv2d_fwd(L.ptr(x), L.ptr(w_ohwi), L.ptr(y), L.i(L.dt(x)), L.i(N), L.i(H), L.i(W), L.i(Cin),
                              L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.ptr(bias), L.ptr(stats),
                              L.i(1 if relu else 0), L.stream()), "ab_conv2d_fwd")
    return (y, stats) if want_stats else y


def conv2d_stem_fwd(xpad, w_stem, H, W, want_stats=False):
    """xpad [N,H+6,W+8,4], w_stem [Cout,7,8,4] -> y [N,H/2,W/2,Cout]."""
    N = xpad.shape[0]
    Cout = w_stem.shape[0]
    y = _empty((N, H // 2, W // 2, Cout), xpad)
    lib = L.lib()
    stats = None
    if want_stats:
        nt = lib.ab_conv2d_stat_rows(L.i(L.dt(xpad)), L.i(N), L.i(H), L.i(W), L.i(4), L.i(Cout), L.i(7), L.i(7), L.i(2), L.i(3), L.i(1))
        stats = torch.empty((nt, Cout, 2), dtype=torch.float32, device=xpad.device)
    L.check(lib.ab_conv2d_stem_fwd(L.ptr(xpad), L.ptr(w_stem), L.ptr(y), L.i(L.dt(xpad)), L.i(N), L.i(H), L.i(W),
                                   L.i(Cout), L.ptr(stats), L.stream()), "ab_conv2d_stem_fwd")
    return (y, stats) if want_stats else y


def conv2d_dgrad(dy, w_ihwo, in_hw, stride, pad, addend=None, bn=None, want_stats=False):
    """dy [N,Ho,Wo,Cout], w [Cin,kh,kw,Cout] -> dx [N,H,W,Cin]  (== ConvTranspose2d forward when x:=dy).
    want_stats (transposed-conv forward): returns (dx, part) with the BatchNorm partial sums [rows, Cin, 2] of dx written by
    the kernel's epilogue, or by a col_stats pass where the shape has no fused path.
    bn=(bn_y, bn_out_or_None, bnp): also try to fuse the BatchNorm-backward reduction of the layer that produced this
    conv's input into the epilogue (ab_conv2d_dgrad_bnstats); returns (dx, part) with part=None when the shape has no
    fused path."""
    N, Ho, Wo, Cout = dy.shape
    Cin, kh, kw, _ = w_ihwo.shape
    H, W = in_hw
    dx = _empty((N, H, W, Cin), dy)
    if bn is not None:
        lib = L.lib()
        rows = lib.ab_conv2d_dgrad_bnstats_rows(L.i(L.dt(dy)), L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout), L.i(kh), L.i(kw),
                                                L.i(stride), L.i(pad))
        if rows > 0:
            bn_y, bn_out, bnp = bn
            part = torch.empty((rows, Cin, 2), dtype=torch.float32, device=dy.device)
            L.check(lib.ab_conv2d_dgrad_bnstats(L.ptr(dy), L.ptr(w_ihwo), L.ptr(dx), L.i(L.dt(dy)), L.i(N), L.i(H), L.i(W),
                                                L.i(Cin), L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.ptr(addend),
                                                L.ptr(bn_y), L.ptr(bn_out), L.ptr(bnp), L.ptr(part), L.stream()),
                    "ab_conv2d_dgrad_bnstats")
            return dx, part
    part = None
    if want_stats and addend is None:
        rows = L.lib().ab_conv2d_dgrad_stat_rows(L.i(L.dt(dy)), L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout), L.i(kh), L.i(kw),
                                                 L.i(stride), L.i(pad))
        if rows > 0:
            part = torch.empty((rows, Cin, 2), dtype=torch.float32, device=dy.device)
    L.check(L.lib().ab_conv2d_dgrad(L.ptr(dy), L.ptr(w_ihwo), L.ptr(dx), L.i(L.dt(dy)), L.i(N), L.i(H), L.i(W), L.i(Cin),
                                    L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.ptr(addend), L.ptr(part),
                                    L.stream()), "ab_conv2d_dgrad")
    if want_stats:
        return dx, (part if part is not None else col_stats(dx))
    return (dx, None) if bn is not None else dx


_ws = {}


def _workspace(nbytes, device):
    """Scratch for the slab-writing kernels.  Eager: one growing buffer per (device, stream) -- stream order makes the reuse safe.
    Under hipGraph capture: a fresh allocation from the capturing graph's own pool every time.  A cached buffer would be shared by
    every graph captured on torch's (global) capture stream, and replacing it when a later call needs more frees memory that
    an EARLIER graph still replays into."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    w = _ws.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = w
    return w


class PendingReductions:
    """Slab reductions recorded by deferred weight-gradient calls (`defer=`), run in one launch by flush().  Each deferred
    call gets its own slab workspace, kept alive here until the batched reduction has been enqueued."""

    def __init__(self):
        self.descs, self.keep = [], []

    def new(self, nbytes, device):
        ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
        d = L.WgradReduceDesc()
        self.keep.append(ws)
        self.descs.append(d)
        return ws, d

    def flush(self):
        live = [d for d in self.descs if d.nslices > 0]
        if live:
            arr = (L.WgradReduceDesc * len(live))(*live)
            L.check(L.lib().ab_wgrad_reduce_batch(arr, L.i(len(live)), L.stream()), "ab_wgrad_reduce_batch")
        self.descs, self.keep = [], []


def conv2d_wgrad(x, dy, kh, kw, stride, pad, out=None, accumulate=False, defer=None):
    """x [N,H,W,Cin], dy [N,Ho,Wo,Cout] -> dw float32 [Cout,kh,kw,Cin].  defer: a PendingReductions that receives the final
    slab reduction instead of it being launched here (dw is complete only after defer.flush())."""
    N, H, W, Cin = x.shape
    Cout = dy.shape[3]
    lib = L.lib()
    M = dy.shape[0] * dy.shape[1] * dy.shape[2]
    nbytes = lib.ab_conv2d_wgrad_workspace(L.i(M), L.i(Cout), L.i(kh * kw * Cin))
    dw = out if out is not None else torch.empty((Cout, kh, kw, Cin), dtype=torch.float32, device=x.device)
    if defer is not None:
        ws, d = defer.new(nbytes, x.device)
        L.check(lib.ab_conv2d_wgrad_deferred(L.ptr(x), L.ptr(dy), L.ptr(dw), L.i(L.dt(x)), L.i(N), L.i(H), L.i(W), L.i(Cin),
                                             L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.ptr(ws),
                                             L.i(1 if accumulate else 0), ctypes.byref(d), L.stream()), "ab_conv2d_wgrad_deferred")
        return dw
    ws = _workspace(nbytes, x.device)
    L.check(lib.ab_conv2d_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dw), L.i(L.dt(x)), L.i(N), L.i(H), L.i(W), L.i(Cin),
                                L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.ptr(ws),
                                L.i(1 if accumulate else 0), L.stream()), "ab_conv2d_wgrad")
    return dw


def conv2d_stem_wgrad(xpad, dy, H, W, out=None, defer=None):
    N = xpad.shape[0]
    Cout = dy.shape[3]
    lib = L.lib()
    nbytes = lib.ab_conv2d_stem_wgrad_workspace(L.i(N), L.i(H), L.i(W), L.i(Cout))
    dw = out if out is not None else torch.empty((Cout, 7, 8, 4), dtype=torch.float32, device=xpad.device)
    if defer is not None:
        ws, d = defer.new(nbytes, xpad.device)
        L.check(lib.ab_conv2d_stem_wgrad_deferred(L.ptr(xpad), L.ptr(dy), L.ptr(dw), L.i(L.dt(xpad)), L.i(N), L.i(H), L.i(W),
                                                  L.i(Cout), L.ptr(ws), ctypes.byref(d), L.stream()), "ab_conv2d_stem_wgrad_deferred")
        return dw
    ws = _workspace(nbytes, xpad.device)
    L.check(lib.ab_conv2d_stem_wgrad(L.ptr(xpad), L.ptr(dy), L.ptr(dw), L.i(L.dt(xpad)), L.i(N), L.i(H), L.i(W),
                                     L.i(Cout), L.ptr(ws), L.stream()), "ab_conv2d_stem_wgrad")
    return dw


def col_stats(x2d):
    M, C = x2d.shape[:-1].numel(), x2d.shape[-1]
    lib = L.lib()
    part = torch.empty((lib.ab_col_stats_nparts(L.l(M)), C, 2), dtype=torch.float32, device=x2d.device)
    L.check(lib.ab_col_stats(L.ptr(x2d), L.i(L.dt(x2d)), L.l(M), L.i(C), L.ptr(part), L.stream()), "ab_col_stats")
    return part


def bn_finalize(part, count, gamma, beta, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, out=None):
    C = gamma.numel()
    bnp = out if out is not None else torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    L.check(L.lib().ab_bn_finalize(L.ptr(part), L.i(part.shape[0]), L.i(C), L.l(count), L.ptr(gamma), L.ptr(beta),
                                   L.f(eps), L.f(momentum), L.ptr(running_mean), L.ptr(running_var), L.ptr(bnp),
                                   L.stream()), "ab_bn_finalize")
    return bnp


def bn_eval_params(gamma, beta, rm, rv, eps=1e-5):
    C = gamma.numel()
    bnp = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    L.check(L.lib().ab_bn_eval_params(L.i(C), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), L.f(eps), L.ptr(bnp),
                                      L.stream()), "ab_bn_eval_params")
    return bnp


def bn_eval_params_batch(flat, stats, desc_dev, n, max_c, out, eps=1e-5):
    """Eval-mode (scale, shift, mean, invstd) of n BatchNorms in one launch; desc_dev int32 [n,6] (see the header)."""
    L.check(L.lib().ab_bn_eval_params_batch(L.ptr(flat), L.ptr(stats), L.ptr(desc_dev), L.i(n), L.i(max_c), L.f(eps), L.ptr(out), L.stream()),
            "ab_bn_eval_params_batch")
    return out


def bn_apply(y, bnp, res=None, relu=True, out=None):
    C = y.shape[-1]
    M = y.numel() // C
    o = out if out is not None else torch.empty_like(y)
    L.check(L.lib().ab_bn_apply(L.ptr(y), L.ptr(res), L.ptr(bnp), L.i(L.dt(y)), L.l(M), L.i(C), L.i(1 if relu else 0),
                                L.ptr(o), L.stream()), "ab_bn_apply")
    return o


def bn_bwd(dout, out, y, bnp, dgamma, dbeta, relu=True, want_dz=False, dy_out=None, part=None):
    """-> dy (grad wrt the conv output y) [, dz = dout*relu_mask].
    part: per-tile sums from conv2d_dgrad(..., bn=...) -- skips the reduction pass.
    relu: False (no activation), True (mask from the stored activation `out`; required when a residual was added before
    the ReLU) or "recompute" (mask from y*scale+shift, `out` is not read)."""
    C = y.shape[-1]
    M = y.numel() // C
    lib = L.lib()
    bwdp = torch.empty((2, C), dtype=torch.float32, device=y.device)
    dy = dy_out if dy_out is not None else torch.empty_like(y)
    dz = torch.empty_like(y) if want_dz else None
    if part is not None:       # first pass already done in the epilogue of the conv that produced `dout`
        L.check(lib.ab_bn_bwd_apply(L.ptr(dout), L.ptr(out if relu is True else None), L.ptr(y), L.ptr(bnp), L.i(L.dt(y)),
                                    L.l(M), L.i(C), L.i(2 if relu == "recompute" else 1 if relu else 0), L.ptr(part),
                                    L.i(part.shape[0]), L.ptr(bwdp), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dy), L.ptr(dz),
                                    L.stream()), "ab_bn_bwd_apply")
        return (dy, dz) if want_dz else dy
    part = torch.empty((lib.ab_col_stats_nparts(L.l(M)), C, 2), dtype=torch.float32, device=y.device)
    L.check(lib.ab_bn_bwd(L.ptr(dout), L.ptr(out if relu is True else None), L.ptr(y), L.ptr(bnp), L.i(L.dt(y)), L.l(M), L.i(C),
                          L.i(2 if relu == "recompute" else 1 if relu else 0), L.ptr(part), L.ptr(bwdp), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dy),
                          L.ptr(dz), L.stream()), "ab_bn_bwd")
    return (dy, dz) if want_dz else dy


def add(a, b, out=None):
    o = out if out is not None else torch.empty_like(a)
    L.check(L.lib().ab_add(L.ptr(a), L.ptr(b), L.i(L.dt(a)), L.l(a.numel()), L.ptr(o), L.stream()), "ab_add")
    return o


def maxpool_fwd(x, want_idx=True):
    N, H, W, C = x.shape
    o = _empty((N, H // 2, W // 2, C), x)
    idx = torch.empty((N, H // 2, W // 2, C), dtype=torch.uint8, device=x.device) if want_idx else None
    L.check(L.lib().ab_maxpool3x3s2_fwd(L.ptr(x), L.i(L.dt(x)), L.i(N), L.i(H), L.i(W), L.i(C), L.ptr(o), L.ptr(idx),
                                        L.stream()), "ab_maxpool3x3s2_fwd")
    return (o, idx) if want_idx else o


def bn_relu_maxpool_fwd(y, bnp):
    """maxpool3x3/2(relu(bn(y))) in one pass -> (pooled, idx); the full-resolution activation is never written."""
    N, H, W, C = y.shape
    o = _empty((N, H // 2, W // 2, C), y)
    idx = torch.empty((N, H // 2, W // 2, C), dtype=torch.uint8, device=y.device)
    L.check(L.lib().ab_bn_relu_maxpool3x3s2_fwd(L.ptr(y), L.ptr(bnp), L.i(L.dt(y)), L.i(N), L.i(H), L.i(W), L.i(C),
                                                L.ptr(o), L.ptr(idx), L.stream()), "ab_bn_relu_maxpool3x3s2_fwd")
    return o, idx


def bn_relu_maxpool_fwd_x3(y, bnp, want_win=False, want_f32=True):
    """bn_relu_maxpool_fwd on fp32 -> (pooled fp32 with its (hi, lo) planes as `_ab_split`, idx): the planes come from the
    pooling pass itself instead of a separate split pass over the pooled tensor.  want_win: also the raw conv output at every
    window's winner (-> (pooled, idx, ywin)), which bn_relu_maxpool_bwd_x3(ywin=...) reduces instead of the full-resolution y."""
    N, H, W, C = y.shape
    o = torch.empty((N, H // 2, W // 2, C), dtype=torch.float32, device=y.device)
    pl = torch.empty((2, N, H // 2, W // 2, C), dtype=torch.bfloat16, device=y.device)
    idx = torch.empty((N, H // 2, W // 2, C), dtype=torch.uint8, device=y.device)
    if want_win and C % 8 == 0:
        ywin = torch.empty_like(o)
        if not want_f32:      # the pooled activation as planes only (its consumers read planes: conv1, the residual, the weight gradient)
            o = None
        L.check(L.lib().ab_bn_relu_maxpool3x3s2_fwd_x3w(L.ptr(y), L.ptr(bnp), L.i(N), L.i(H), L.i(W), L.i(C), L.ptr(o), L.ptr(pl[0]),
                                                        L.ptr(pl[1]), L.ptr(idx), L.ptr(ywin), L.stream()), "ab_bn_relu_maxpool3x3s2_fwd_x3w")
        if o is None:
            return pl, idx, ywin
        o._ab_split = pl
        return o, idx, ywin
    L.check(L.lib().ab_bn_relu_maxpool3x3s2_fwd_x3(L.ptr(y), L.ptr(bnp), L.i(N), L.i(H), L.i(W), L.i(C), L.ptr(o), L.ptr(pl[0]),
                                                   L.ptr(pl[1]), L.ptr(idx), L.stream()), "ab_bn_relu_maxpool3x3s2_fwd_x3")
    o._ab_split = pl
    return (o, idx, None) if want_win else (o, idx)


def bn_relu_maxpool_bwd(dpool, idx, y, bnp, dgamma, dbeta):
    """Backward of bn_relu_maxpool_fwd -> dy (gradient wrt the conv output y)."""
    N, H, W, C = y.shape
    lib = L.lib()
    part = torch.empty((lib.ab_col_stats_nparts(L.l(N * H * W)), C, 2), dtype=torch.float32, device=y.device)
    bwdp = torch.empty((2, C), dtype=torch.float32, device=y.device)
    dy = torch.empty_like(y)
    L.check(lib.ab_bn_relu_maxpool_bwd(L.ptr(dpool), L.ptr(idx), L.ptr(y), L.ptr(bnp), L.i(L.dt(y)), L.i(N), L.i(H), L.i(W),
                                       L.i(C), L.ptr(part), L.ptr(bwdp), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dy),
                                       L.stream()), "ab_bn_relu_maxpool_bwd")
    return dy


def bn_relu_maxpool_bwd_x3(dpool, idx, y, bnp, dgamma, dbeta, ywin=None):
    """Backward of bn_relu_maxpool_fwd on fp32 tensors -> dy as split planes [2, *y.shape], or None when the shape is not handled:
    the max-pool backward pass also masks and reduces (no separate BatchNorm-backward reduction over the full-size tensors).
    ywin (bn_relu_maxpool_fwd_x3(want_win=True)): the reduction runs over the pooled elements only."""
    N, H, W, C = y.shape
    lib = L.lib()
    if ywin is not None:
        part = torch.empty((lib.ab_col_stats_nparts(L.l(N * (H // 2) * (W // 2))), C, 2), dtype=torch.float32, device=y.device)
        bwdp = torch.empty((2, C), dtype=torch.float32, device=y.device)
        dy = torch.empty((2,) + tuple(y.shape), dtype=torch.bfloat16, device=y.device)
        L.check(lib.ab_bn_relu_maxpool_bwd_x3w(L.ptr(dpool), L.ptr(idx), L.ptr(ywin), L.ptr(y), L.ptr(bnp), L.i(N), L.i(H), L.i(W),
                                               L.i(C), L.ptr(part), L.ptr(bwdp), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dy[0]),
                                               L.ptr(dy[1]), L.stream()), "ab_bn_relu_maxpool_bwd_x3w")
        return dy
    np_ = lib.ab_bn_relu_maxpool_bwd_x3_nparts(L.i(N), L.i(H), L.i(W), L.i(C))
    if np_ <= 0:
        return None
    part = torch.empty((np_, C, 2), dtype=torch.float32, device=y.device)
    bwdp = torch.empty((2, C), dtype=torch.float32, device=y.device)
    dz = torch.empty_like(y)
    dy = torch.empty((2,) + tuple(y.shape), dtype=torch.bfloat16, device=y.device)
    L.check(lib.ab_bn_relu_maxpool_bwd_x3(L.ptr(dpool), L.ptr(idx), L.ptr(y), L.ptr(bnp), L.i(N), L.i(H), L.i(W), L.i(C),
                                          L.ptr(part), L.ptr(bwdp), L.ptr(dgamma), L.ptr(dbeta), L.ptr(dz), L.ptr(dy[0]),
                                          L.ptr(dy[1]), L.stream()), "ab_bn_relu_maxpool_bwd_x3")
    return dy


def maxpool_bwd(idx, dout, in_hw):
    N, Ho, Wo, C = dout.shape
    H, W = in_hw
    dx = _empty((N, H, W, C), dout)
    L.check(L.lib().ab_maxpool3x3s2_bwd(L.ptr(idx), L.ptr(dout), L.i(L.dt(dout)), L.i(N), L.i(H), L.i(W), L.i(C),
                                        L.ptr(dx), L.stream()), "ab_maxpool3x3s2_bwd")
    return dx


def avgpool_fwd(x):
    N, H, W, C = x.shape
    o = torch.empty((N, C), dtype=torch.float32, device=x.device)
    L.check(L.lib().ab_avgpool_fwd(L.ptr(x), L.i(L.dt(x)), L.i(N), L.i(H * W), L.i(C), L.ptr(o), L.stream()),
            "ab_avgpool_fwd")
    return o


def avgpool_bwd(g, dx, accumulate):
    N, H, W, C = dx.shape
    L.check(L.lib().ab_avgpool_bwd(L.ptr(g), L.i(L.dt(dx)), L.i(N), L.i(H * W), L.i(C), L.ptr(dx),
                                   L.i(1 if accumulate else 0), L.stream()), "ab_avgpool_bwd")
    return dx


def cast_bf16(src_f32, dst_bf16):
    L.check(L.lib().ab_cast_f32_bf16(L.ptr(src_f32), L.l(src_f32.numel()), L.ptr(dst_bf16), L.stream()),
            "ab_cast_f32_bf16")
    return dst_bf16


def transpose_oki(src_f32_oki, dst):
    O, K, I = src_f32_oki.shape
    L.check(L.lib().ab_transpose_oki(L.ptr(src_f32_oki), L.i(O), L.i(K), L.i(I), L.i(L.dt(dst)), L.ptr(dst),
                                     L.stream()), "ab_transpose_oki")
    return dst


def linear_fwd(x, w, bias=None, relu=False):
    """fp32 x [M,K] @ w[N,K]^T (+bias) (+ReLU) -> [M,N]   (ab_linear_fwd; the box-rotation MLP)."""
    M, Kd = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    L.check(L.lib().ab_linear_fwd(L.ptr(x), L.ptr(w), L.ptr(bias), L.i(M), L.i(N), L.i(Kd), L.i(1 if relu else 0), L.ptr(y),
                                  L.stream()), "ab_linear_fwd")
    return y


def linear_dgrad(g, wt, act_out=None):
    """fp32 g [M,N] @ W[N,K] -> [M,K] given wt = W^T [K,N]; zeroed where act_out <= 0 (the ReLU that fed this layer)."""
    M, N = g.shape
    Kd = wt.shape[0]
    gx = torch.empty((M, Kd), dtype=torch.float32, device=g.device)
    L.check(L.lib().ab_linear_dgrad(L.ptr(g), L.ptr(wt), L.ptr(act_out), L.i(M), L.i(N), L.i(Kd), L.ptr(gx), L.stream()),
            "ab_linear_dgrad")
    return gx


def linear_wgrad(g, x, dw, db=None):
    """dw[N,K] = g[M,N]^T @ x[M,K], db[N] = sum_m g (written in place into the flat gradient views)."""
    M, N = g.shape
    Kd = x.shape[1]
    L.check(L.lib().ab_linear_wgrad(L.ptr(g), L.ptr(x), L.i(M), L.i(N), L.i(Kd), L.ptr(dw), L.ptr(db), L.stream()),
            "ab_linear_wgrad")


def transpose_plan(pairs):
    """pairs: [(src f32 [O,K,I], dst [I,K,O])] -> (device descriptor table, n, total_tiles, dtype code) for
    transpose_oki_batch; None when a shape does not fit the 16-byte vector tiles."""
    import numpy as np
    rec = np.zeros(len(pairs), dtype=np.dtype([("src", "<u8"), ("dst", "<u8"), ("O", "<i4"), ("K", "<i4"), ("I", "<i4"),
                                               ("tile_begin", "<i4")]))
    tiles = 0
    for n, (src, dst) in enumerate(pairs):
        O, Kk, I = src.shape
        if I % 4 or O % 8 or not src.is_contiguous() or not dst.is_contiguous() or dst.dtype != pairs[0][1].dtype:
            return None
        rec[n] = (src.data_ptr(), dst.data_ptr(), O, Kk, I, tiles)
        tiles += ((O + 63) // 64) * Kk * ((I + 63) // 64)
    dev = torch.from_numpy(rec.view(np.uint8).copy()).to(pairs[0][0].device)
    return dev, len(pairs), tiles, L.dt(pairs[0][1])


def transpose_oki_batch(plan):
    dev, n, tiles, dt = plan
    L.check(L.lib().ab_transpose_oki_batch(L.ptr(dev), L.i(n), L.l(tiles), L.i(dt), L.stream()), "ab_transpose_oki_batch")


def transpose_oki_batch_x3(plan, lo_offset_elems):
    """transpose_oki_batch with bf16 destinations written as split planes (lo plane `lo_offset_elems` behind the hi plane)."""
    dev, n, tiles, dt = plan
    assert dt == L.DT_BF16
    L.check(L.lib().ab_transpose_oki_batch_x3(L.ptr(dev), L.i(n), L.l(tiles), L.l(lo_offset_elems), L.stream()), "ab_transpose_oki_batch_x3")


def image_pad_nhwc4(img_nchw_f32, dtype):
    N, C, H, W = img_nchw_f32.shape
    assert C == 3
    out = torch.empty((N, H + 6, W + 8, 4), dtype=dtype, device=img_nchw_f32.device)
    L.check(L.lib().ab_image_pad_nhwc4(L.ptr(img_nchw_f32), L.i(L.dt(out)), L.i(N), L.i(H), L.i(W), L.ptr(out),
                                       L.stream()), "ab_image_pad_nhwc4")
    return out


def relu_bwd(dout, out):
    dz = torch.empty_like(dout)
    L.check(L.lib().ab_relu_bwd(L.ptr(dout), L.ptr(out), L.i(L.dt(dout)), L.l(dout.numel()), L.ptr(dz), L.stream()),
            "ab_relu_bwd")
    return dz


def col_sum(x, out):
    C = x.shape[-1]
    M = x.numel() // C
    lib = L.lib()
    part = torch.empty((lib.ab_col_stats_nparts(L.l(M)), C, 2), dtype=torch.float32, device=x.device)
    L.check(lib.ab_col_sum(L.ptr(x), L.i(L.dt(x)), L.l(M), L.i(C), L.ptr(part), L.ptr(out), L.stream()), "ab_col_sum")
    return out


# ---------------------------------------------------------------- split-bf16 ("bf16x3") convolutions
# A split tensor is a bf16 tensor [2, ...]: plane 0 = bf16(v), plane 1 = bf16(v - plane0) (include/artiboost_hip.h).

def split(x_f32, out=None):
    """fp32 tensor -> split planes [2, *x.shape] bf16 (ab_split_f32)."""
    if x_f32.dtype != torch.float32:
        raise TypeError("split() takes a float32 tensor")
    sp = out if out is not None else torch.empty((2,) + tuple(x_f32.shape), dtype=torch.bfloat16, device=x_f32.device)
    L.check(L.lib().ab_split_f32(L.ptr(x_f32), L.l(x_f32.numel()), L.ptr(sp[0]), L.ptr(sp[1]), L.stream()), "ab_split_f32")
    return sp


def _planes(t):
    """(hi, lo) of a split tensor, or of the planes cached on an fp32 tensor by its producer; splits on the fly otherwise."""
    if isinstance(t, tuple):
        return t
    if t.dtype == torch.bfloat16:
        return t[0], t[1]
    sp = getattr(t, "_ab_split", None)
    if sp is None:
        # NOT cached on the tensor: a persistent buffer (the static image of a replayed step) changes its contents between
        # calls, and a cached split would silently keep serving the first image.  Producers attach `_ab_split` to the FRESH
        # tensors they allocate; whoever feeds the same fp32 tensor to two convolutions splits it once itself.
        sp = split(t)
    return sp[0], sp[1]


def conv2d_fwd_x3(x, w_split, stride, pad, bias=None, want_stats=False, relu=False):
    """x: fp32 [N,H,W,Cin] or split [2,N,H,W,Cin]; w_split [2,Cout,kh,kw,Cin] -> y fp32 [N,Ho,Wo,Cout] (+ BN partials)."""
    xh, xl = _planes(x)
    N, H, W, Cin = xh.shape
    _, Cout, kh, kw, _ = w_split.shape
    Ho, Wo = conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad)
    y = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=xh.device)
    lib = L.lib()
    stats = None
    if want_stats:
        nt = lib.ab_conv2d_x3_stat_rows(L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad))
        stats = torch.empty((nt, Cout, 2), dtype=torch.float32, device=xh.device)
    L.check(lib.ab_conv2d_fwd_x3(L.ptr(xh), L.ptr(xl), L.ptr(w_split[0]), L.ptr(w_split[1]), L.ptr(y), L.i(N), L.i(H), L.i(W),
                                 L.i(Cin), L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.ptr(bias), L.ptr(stats),
                                 L.i(1 if relu else 0), L.stream()), "ab_conv2d_fwd_x3")
    return (y, stats) if want_stats else y


def conv1x1_sam_fwd_x3_ok(x, w_split, C, D, DP):
    """True when ab_conv1x1_sam_fwd_x3 takes the final layer's shape (DEPTH_PITCH 32, Cin % 64 == 0, <= 256, H * W % 64 == 0)."""
    xh, _ = _planes(x) if x.dtype == torch.bfloat16 else (x, None)
    N, H, W, Cin = xh.shape
    return DP == 32 and tuple(w_split.shape[1:]) == (C * 32, 1, 1, Cin) and \
        bool(L.lib().ab_conv1x1_sam_fwd_x3_ok(L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(C), L.i(D)))


def conv1x1_sam_fwd_x3(x, w_split, bias, C, D):
    """Final layer + soft-argmax stage 1 (ab_conv1x1_sam_fwd_x3): x fp32 / split [.., N,H,W,Cin], w_split [2, C*32, 1, 1, Cin] ->
    (logits fp32 [N,H,W,C*32], part fp32 [N, H*W/64, C, 8])."""
    xh, xl = _planes(x)
    N, H, W, Cin = xh.shape
    y = torch.empty((N, H, W, C * 32), dtype=torch.float32, device=xh.device)
    part = torch.empty((N, H * W // 64, C, 8), dtype=torch.float32, device=xh.device)
    L.check(L.lib().ab_conv1x1_sam_fwd_x3(L.ptr(xh), L.ptr(xl), L.ptr(w_split[0]), L.ptr(w_split[1]), L.ptr(bias), L.ptr(y), L.i(N), L.i(H),
                                          L.i(W), L.i(Cin), L.i(C), L.i(D), L.ptr(part), L.stream()), "ab_conv1x1_sam_fwd_x3")
    return y, part


def conv2d_fwd_x3_evalbn_ok(x, w_split):
    xh = x[0] if x.dtype == torch.bfloat16 else x
    N, H, W, Cin = xh.shape
    _, Cout, kh, kw, _ = w_split.shape
    return (kh, kw) == (3, 3) and bool(L.lib().ab_conv2d_fwd_x3_evalbn_ok(L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout)))


def conv2d_fwd_x3_evalbn(x, w_split, bnp, res=None, relu=True, want_f32=False):
    """Eval-mode 3x3/s1 convolution + the BatchNorm after it (+ residual, ReLU) in one launch -> split planes [2,N,H,W,Cout]
    (want_f32: the fp32 tensor with the planes cached on it, as bn_apply_x3 returns).  res: split planes, fp32 tensor or None."""
    xh, xl = _planes(x)
    N, H, W, Cin = xh.shape
    Cout = w_split.shape[1]
    sp = torch.empty((2, N, H, W, Cout), dtype=torch.bfloat16, device=xh.device)
    o = torch.empty((N, H, W, Cout), dtype=torch.float32, device=xh.device) if want_f32 else None
    rh = rl = rf = None
    if res is not None:
        if res.dtype == torch.bfloat16:
            rh, rl = res[0], res[1]
        else:
            rf = res
    L.check(L.lib().ab_conv2d_fwd_x3_evalbn(L.ptr(xh), L.ptr(xl), L.ptr(w_split[0]), L.ptr(w_split[1]), L.i(N), L.i(H), L.i(W), L.i(Cin),
                                            L.i(Cout), L.ptr(bnp), L.ptr(rh), L.ptr(rl), L.ptr(rf), L.i(1 if relu else 0), L.ptr(sp[0]),
                                            L.ptr(sp[1]), L.ptr(o), L.stream()), "ab_conv2d_fwd_x3_evalbn")
    if o is None:
        return sp
    o._ab_split = sp
    return o


def conv2d_fwd_x3_affine(x, w_split, bnp, stride, pad, relu=True, planes=True):
    """Eval-mode generic convolution + following BatchNorm (+ ReLU) in the epilogue -> split planes [2,N,Ho,Wo,Cout] or fp32 [N,Ho,Wo,Cout]."""
    xh, xl = _planes(x)
    N, H, W, Cin = xh.shape
    _, Cout, kh, kw, _ = w_split.shape
    Ho, Wo = conv_out(H, kh, stride, pad), conv_out(W, kw, stride, pad)
    sp = torch.empty((2, N, Ho, Wo, Cout), dtype=torch.bfloat16, device=xh.device) if planes else None
    o = None if planes else torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=xh.device)
    L.check(L.lib().ab_conv2d_fwd_x3_affine(L.ptr(xh), L.ptr(xl), L.ptr(w_split[0]), L.ptr(w_split[1]), L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout),
                                            L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.ptr(bnp[0]), L.ptr(bnp[1]), L.i(1 if relu else 0), L.ptr(o),
                                            L.ptr(sp[0] if planes else None), L.ptr(sp[1] if planes else None), L.stream()), "ab_conv2d_fwd_x3_affine")
    return sp if planes else o


def conv2d_dgrad_x3_affine(dy, wt_split, in_hw, stride, pad, bnp, relu=True):
    """Eval-mode transposed convolution (data-gradient form) + following BatchNorm (+ ReLU) -> split planes [2,N,H,W,Cin]."""
    dh, dl = _planes(dy)
    N, Ho, Wo, Cout = dh.shape
    _, Cin, kh, kw, _ = wt_split.shape
    H, W = in_hw
    sp = torch.empty((2, N, H, W, Cin), dtype=torch.bfloat16, device=dh.device)
    L.check(L.lib().ab_conv2d_dgrad_x3_affine(L.ptr(dh), L.ptr(dl), L.ptr(wt_split[0]), L.ptr(wt_split[1]), L.i(N), L.i(H), L.i(W), L.i(Cin),
                                              L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.ptr(bnp[0]), L.ptr(bnp[1]), L.i(1 if relu else 0),
                                              L.ptr(None), L.ptr(sp[0]), L.ptr(sp[1]), L.stream()), "ab_conv2d_dgrad_x3_affine")
    return sp


def conv2d_dgrad_x3(dy, wt_split, in_hw, stride, pad, addend=None, want_stats=False, bn=None):
    """dy: fp32 / split [.., N,Ho,Wo,Cout]; wt_split [2,Cin,kh,kw,Cout] -> dx fp32 [N,H,W,Cin] (+ BN partials of dx).

    bn=(bn_y, bn_out_or_None, bnp): dx is the gradient arriving at relu(bn(bn_y) [+ residual]).  Returns (dx, part): where the
    kernel can, dx is already MASKED (dz) and `part` holds the BatchNorm-backward partial sums of its epilogue (pass it to
    bn_bwd_x3(part=..., premasked=True)); otherwise part is None and dx the raw gradient."""
    dh, dl = _planes(dy)
    N, Ho, Wo, Cout = dh.shape
    _, Cin, kh, kw, _ = wt_split.shape
    H, W = in_hw
    dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dh.device)
    lib = L.lib()
    part = None
    if bn is not None:
        bn_y, bn_out, bnp = bn
        rows = lib.ab_conv2d_dgrad_x3_bn_rows(L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad))
        mask = None
        if bn_out is not None:
            sp = bn_out if bn_out.dtype == torch.bfloat16 else getattr(bn_out, "_ab_split", None)
            if sp is None:
                rows = 0                   # the mask is read from the hi plane of the stored activation
            else:
                mask = sp[0]
        if rows > 0:
            part = torch.empty((rows, Cin, 2), dtype=torch.float32, device=dh.device)
            L.check(lib.ab_conv2d_dgrad_x3_bn(L.ptr(dh), L.ptr(dl), L.ptr(wt_split[0]), L.ptr(wt_split[1]), L.ptr(dx), L.i(N),
                                              L.i(H), L.i(W), L.i(Cin), L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad),
                                              L.ptr(addend), L.ptr(bn_y), L.ptr(mask), L.ptr(bnp), L.ptr(part), L.stream()),
                    "ab_conv2d_dgrad_x3_bn")
            return dx, part
    if want_stats and addend is None:
        rows = lib.ab_conv2d_dgrad_x3_stat_rows(L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad))
        if rows > 0:
            part = torch.empty((rows, Cin, 2), dtype=torch.float32, device=dh.device)
    L.check(lib.ab_conv2d_dgrad_x3(L.ptr(dh), L.ptr(dl), L.ptr(wt_split[0]), L.ptr(wt_split[1]), L.ptr(dx), L.i(N), L.i(H), L.i(W),
                                   L.i(Cin), L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.ptr(addend), L.ptr(part),
                                   L.stream()), "ab_conv2d_dgrad_x3")
    if want_stats:
        return dx, (part if part is not None else col_stats(dx))
    return (dx, None) if bn is not None else dx


def conv2d_dgrad_x3_pair(dy, wt_split, dy2, wt2_split, in_hw, pad, addend=None, bn=None):
    """dx = dgrad(dy, wt; k x k / stride 2) + dgrad(dy2, wt2; 1x1 / stride 2 / pad 0) [+ addend] in one launch: the conv1 and
    downsample branches of a down-sampling block (anakin/models/resnet.py:85-101 backwards).

    bn=(bn_y, bn_out_or_None, bnp) (no addend): dx is the gradient arriving at relu(bn(bn_y) [+ residual]), the output of the stage below.
    Returns (dx, part) as conv2d_dgrad_x3(bn=...) does: masked dx + the BatchNorm-backward partial rows where the kernel can, else (dx, None)."""
    dh, dl = _planes(dy)
    eh, el = _planes(dy2)
    N, Ho, Wo, Cout = dh.shape
    _, Cin, kh, kw, _ = wt_split.shape
    assert tuple(eh.shape) == tuple(dh.shape) and tuple(wt2_split.shape[1:]) == (Cin, 1, 1, Cout)
    H, W = in_hw
    dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dh.device)
    lib = L.lib()
    if bn is not None and addend is None:
        bn_y, bn_out, bnp = bn
        rows = lib.ab_conv2d_dgrad_x3_pair_bn_rows(L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout), L.i(kh), L.i(kw), L.i(pad))
        mask = None
        if bn_out is not None:
            sp = bn_out if bn_out.dtype == torch.bfloat16 else getattr(bn_out, "_ab_split", None)
            if sp is None:
                rows = 0                   # the mask is read from the hi plane of the stored activation
            else:
                mask = sp[0]
        if rows > 0:
            part = torch.empty((rows, Cin, 2), dtype=torch.float32, device=dh.device)
            L.check(lib.ab_conv2d_dgrad_x3_pair_bn(L.ptr(dh), L.ptr(dl), L.ptr(wt_split[0]), L.ptr(wt_split[1]), L.ptr(eh), L.ptr(el),
                                                   L.ptr(wt2_split[0]), L.ptr(wt2_split[1]), L.ptr(dx), L.i(N), L.i(H), L.i(W), L.i(Cin),
                                                   L.i(Cout), L.i(kh), L.i(kw), L.i(pad), L.ptr(bn_y), L.ptr(mask), L.ptr(bnp), L.ptr(part),
                                                   L.stream()), "ab_conv2d_dgrad_x3_pair_bn")
            return dx, part
    L.check(lib.ab_conv2d_dgrad_x3_pair(L.ptr(dh), L.ptr(dl), L.ptr(wt_split[0]), L.ptr(wt_split[1]), L.ptr(eh), L.ptr(el),
                                        L.ptr(wt2_split[0]), L.ptr(wt2_split[1]), L.ptr(dx), L.i(N), L.i(H), L.i(W), L.i(Cin),
                                        L.i(Cout), L.i(kh), L.i(kw), L.i(pad), L.ptr(addend), L.stream()),
            "ab_conv2d_dgrad_x3_pair")
    return (dx, None) if bn is not None else dx


def conv2d_wgrad_x3(x, dy, kh, kw, stride, pad, out=None, accumulate=False, defer=None):
    """x, dy: fp32 or split -> dw fp32 [Cout,kh,kw,Cin].  defer: see conv2d_wgrad."""
    xh, xl = _planes(x)
    dh, dl = _planes(dy)
    N, H, W, Cin = xh.shape
    Cout = dh.shape[3]
    lib = L.lib()
    nbytes = lib.ab_conv2d_wgrad_x3_workspace(L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad))
    if nbytes <= 0:
        raise RuntimeError(f"conv2d_wgrad_x3: shape not handled (Cin {Cin}, Cout {Cout}, {kh}x{kw})")
    dw = out if out is not None else torch.empty((Cout, kh, kw, Cin), dtype=torch.float32, device=xh.device)
    if defer is not None:
        ws, d = defer.new(nbytes, xh.device)
        L.check(lib.ab_conv2d_wgrad_x3_deferred(L.ptr(xh), L.ptr(xl), L.ptr(dh), L.ptr(dl), L.ptr(dw), L.i(N), L.i(H), L.i(W), L.i(Cin),
                                                L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.ptr(ws), L.i(1 if accumulate else 0),
                                                ctypes.byref(d), L.stream()), "ab_conv2d_wgrad_x3_deferred")
        return dw
    ws = _workspace(nbytes, xh.device)
    L.check(lib.ab_conv2d_wgrad_x3(L.ptr(xh), L.ptr(xl), L.ptr(dh), L.ptr(dl), L.ptr(dw), L.i(N), L.i(H), L.i(W), L.i(Cin),
                                   L.i(Cout), L.i(kh), L.i(kw), L.i(stride), L.i(pad), L.ptr(ws), L.i(1 if accumulate else 0),
                                   L.stream()), "ab_conv2d_wgrad_x3")
    return dw


WGRAD_GROUP_MAX = 8


def conv2d_wgrad_x3_group_ok(x, dy, G):
    """True when ab_conv2d_wgrad_x3_group takes G layers of this shape (3x3 / stride 1 / pad 1 on wgrad3x3.hip's map sizes)."""
    xh, dh = _planes(x)[0], _planes(dy)[0]
    N, H, W, Cin = xh.shape
    return dh.shape[:3] == xh.shape[:3] and 1 <= G <= WGRAD_GROUP_MAX and \
        L.lib().ab_conv2d_wgrad_x3_group_workspace(L.i(G), L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(dh.shape[3])) > 0


def conv2d_wgrad_x3_group(items, accumulate=False):
    """items: [(x, dy, out)] of ONE shape (3x3 / stride 1 / pad 1; x, dy fp32 or split; out fp32 [Cout,3,3,Cin]) -> the G weight gradients
    from one slab launch + one reduction launch (ab_conv2d_wgrad_x3_group)."""
    G = len(items)
    planes = [(_planes(x), _planes(dy), out) for x, dy, out in items]
    (xh, _), (dh, _), _ = planes[0]
    N, H, W, Cin = xh.shape
    Cout = dh.shape[3]
    lib = L.lib()
    nbytes = lib.ab_conv2d_wgrad_x3_group_workspace(L.i(G), L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout))
    if nbytes <= 0:
        raise RuntimeError(f"conv2d_wgrad_x3_group: shape not handled ({G} x [{N},{H},{W},{Cin}] -> {Cout})")
    arr = (L.WgradGroupItem * G)()
    for i, ((a, b), (c, d), out) in enumerate(planes):
        assert a.shape == xh.shape and c.shape == dh.shape and out.dtype == torch.float32 and out.numel() == Cout * 9 * Cin and out.is_contiguous()
        arr[i].x_hi, arr[i].x_lo, arr[i].dy_hi, arr[i].dy_lo, arr[i].dw = a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), out.data_ptr()
    ws = _workspace(nbytes, xh.device)
    L.check(lib.ab_conv2d_wgrad_x3_group(arr, L.i(G), L.i(N), L.i(H), L.i(W), L.i(Cin), L.i(Cout), L.ptr(ws), L.i(1 if accumulate else 0),
                                         L.stream()), "ab_conv2d_wgrad_x3_group")
    return [out for _, _, out in planes]


def bn_apply_x3(y, bnp, res=None, relu=True, want_f32=False, res_bnp=None):
    """fp32 y -> split planes [2, *y.shape] of relu(bn(y) + res); want_f32: also the fp32 tensor (returned, with the planes
    cached on it as `_ab_split`) -- needed where the activation is a residual input or a ReLU mask in the backward.
    res_bnp: `res` is the raw output of the downsample conv and the residual is its BatchNorm (scale, shift) applied on the fly."""
    C = y.shape[-1]
    M = y.numel() // C
    sp = torch.empty((2,) + tuple(y.shape), dtype=torch.bfloat16, device=y.device)
    o = torch.empty_like(y) if want_f32 else None
    if res is not None and res.dtype == torch.bfloat16:      # the residual as (hi, lo) planes
        assert res_bnp is None
        L.check(L.lib().ab_bn_apply_x3_respl(L.ptr(y), L.ptr(res[0]), L.ptr(res[1]), L.ptr(bnp), L.l(M), L.i(C), L.i(1 if relu else 0),
                                             L.ptr(o), L.ptr(sp[0]), L.ptr(sp[1]), L.stream()), "ab_bn_apply_x3_respl")
    elif res_bnp is not None:
        L.check(L.lib().ab_bn_apply_x3_resbn(L.ptr(y), L.ptr(res), L.ptr(bnp), L.ptr(res_bnp), L.l(M), L.i(C), L.i(1 if relu else 0),
                                             L.ptr(o), L.ptr(sp[0]), L.ptr(sp[1]), L.stream()), "ab_bn_apply_x3_resbn")
    else:
        L.check(L.lib().ab_bn_apply_x3(L.ptr(y), L.ptr(res), L.ptr(bnp), L.l(M), L.i(C), L.i(1 if relu else 0), L.ptr(o),
                                       L.ptr(sp[0]), L.ptr(sp[1]), L.stream()), "ab_bn_apply_x3")
    if o is None:
        return sp
    o._ab_split = sp
    return o


def bn_fin_apply_x3_ok(part, C):
    """True when ab_bn_fin_apply_x3 takes this BatchNorm (few partial rows, C % 64 == 0): finalize and apply in one launch."""
    return part is not None and bool(L.lib().ab_bn_fin_apply_x3_ok(L.i(part.shape[0]), L.i(C)))


def bn_fin_apply_x3(y, part, count, gamma, beta, running_mean=None, running_var=None, res=None, relu=True, want_f32=False, res_bnp=None,
                    eps=1e-5, momentum=0.1):
    """bn_finalize + bn_apply_x3 in ONE launch (see ab_bn_fin_apply_x3) -> (activation as bn_apply_x3 returns it, bnp [4, C])."""
    C = y.shape[-1]
    M = y.numel() // C
    bnp = torch.empty((4, C), dtype=torch.float32, device=y.device)
    sp = torch.empty((2,) + tuple(y.shape), dtype=torch.bfloat16, device=y.device)
    o = torch.empty_like(y) if want_f32 else None
    rh = rl = rf = None
    if res is not None and res.dtype == torch.bfloat16:
        assert res_bnp is None
        rh, rl = res[0], res[1]
    else:
        rf = res
    L.check(L.lib().ab_bn_fin_apply_x3(L.ptr(part), L.i(part.shape[0]), L.l(count), L.ptr(gamma), L.ptr(beta), L.f(eps), L.f(momentum),
                                       L.ptr(running_mean), L.ptr(running_var), L.ptr(bnp), L.ptr(y), L.ptr(rf), L.ptr(rh), L.ptr(rl),
                                       L.ptr(res_bnp), L.l(M), L.i(C), L.i(1 if relu else 0), L.ptr(o), L.ptr(sp[0]), L.ptr(sp[1]), L.stream()),
            "ab_bn_fin_apply_x3")
    if o is None:
        return sp, bnp
    o._ab_split = sp
    return o, bnp


def bn_bwd_x3(dout, out, y, bnp, dgamma, dbeta, relu=True, want_dz=False, part=None, premasked=False):
    """As bn_bwd on fp32 tensors, with dy returned as split planes [2, *y.shape] (-> dy [, dz fp32]).
    premasked (with part): `dout` is already the masked gradient dz and `part` its reduction (conv2d_dgrad_x3(bn=...))."""
    if premasked:
        assert part is not None
        dy = bn_bwd_x3(dout, None, y, bnp, dgamma, dbeta, relu=False, part=part)
        return (dy, dout) if want_dz else dy
    C = y.shape[-1]
    M = y.numel() // C
    lib = L.lib()
    bwdp = torch.empty((2, C), dtype=torch.float32, device=y.device)
    dy = torch.empty((2,) + tuple(y.shape), dtype=torch.bfloat16, device=y.device)
    dz = torch.empty_like(y) if want_dz else None
    given = 0
    if part is None:
        part = torch.empty((lib.ab_col_stats_nparts(L.l(M)), C, 2), dtype=torch.float32, device=y.device)
    else:
        given = part.shape[0]
    mask, is_hi = (out if relu is True else None), 0
    if mask is not None and mask.dtype == torch.bfloat16:
        mask, is_hi = mask[0], 1                     # the activation exists only as planes
    elif mask is not None and getattr(mask, "_ab_split", None) is not None:
        mask, is_hi = mask._ab_split[0], 1          # sign of the hi plane == sign of the activation; half the bytes
    L.check(lib.ab_bn_bwd_x3(L.ptr(dout), L.ptr(mask), L.i(is_hi), L.ptr(y), L.ptr(bnp), L.l(M), L.i(C),
                             L.i(2 if relu == "recompute" else 1 if relu else 0), L.ptr(part), L.i(given), L.ptr(bwdp),
                             L.ptr(dgamma), L.ptr(dbeta), L.ptr(dy[0]), L.ptr(dy[1]), L.ptr(dz), L.stream()), "ab_bn_bwd_x3")
    return (dy, dz) if want_dz else dy


def _image_planes(xpad):
    """(hi, lo) planes of the padded image, or (plane, None) for the integer plane 2 v - 255 the loaders write with AB_DT_U8N
    (bf16 [N, H+6, W+8, 4]: the network input is plane / 510; ab_conv2d_stem_*_x3 take it with xpad_lo = NULL)."""
    if xpad.dtype == torch.bfloat16 and xpad.dim() == 4:
        return xpad, None
    return _planes(xpad)


def conv2d_stem_fwd_x3(xpad, w_split, H, W, want_stats=False):
    """xpad fp32 / split [.., N,H+6,W+8,4] / integer plane (see _image_planes), w_split [2,64,7,8,4] -> y fp32 [N,H/2,W/2,64] (+ BN partials)."""
    xh, xl = _image_planes(xpad)
    N = xh.shape[0]
    Cout = w_split.shape[1]
    y = torch.empty((N, H // 2, W // 2, Cout), dtype=torch.float32, device=xh.device)
    lib = L.lib()
    stats = None
    if want_stats:
        stats = torch.empty((lib.ab_conv2d_stem_x3_stat_rows(L.i(N), L.i(H), L.i(W)), Cout, 2), dtype=torch.float32, device=xh.device)
    L.check(lib.ab_conv2d_stem_fwd_x3(L.ptr(xh), L.ptr(xl), L.ptr(w_split[0]), L.ptr(w_split[1]), L.ptr(y), L.i(N), L.i(H), L.i(W),
                                      L.i(Cout), L.ptr(stats), L.stream()), "ab_conv2d_stem_fwd_x3")
    return (y, stats) if want_stats else y


def conv2d_stem_wgrad_x3(xpad, dy, H, W, out=None, defer=None):
    xh, xl = _image_planes(xpad)
    dh, dl = _planes(dy)
    N = xh.shape[0]
    Cout = dh.shape[3]
    lib = L.lib()
    nbytes = lib.ab_conv2d_stem_wgrad_workspace(L.i(N), L.i(H), L.i(W), L.i(Cout))
    dw = out if out is not None else torch.empty((Cout, 7, 8, 4), dtype=torch.float32, device=xh.device)
    if defer is not None:
        ws, d = defer.new(nbytes, xh.device)
        L.check(lib.ab_conv2d_stem_wgrad_x3_deferred(L.ptr(xh), L.ptr(xl), L.ptr(dh), L.ptr(dl), L.ptr(dw), L.i(N), L.i(H), L.i(W),
                                                     L.i(Cout), L.ptr(ws), ctypes.byref(d), L.stream()), "ab_conv2d_stem_wgrad_x3_deferred")
        return dw
    ws = _workspace(nbytes, xh.device)
    L.check(lib.ab_conv2d_stem_wgrad_x3(L.ptr(xh), L.ptr(xl), L.ptr(dh), L.ptr(dl), L.ptr(dw), L.i(N), L.i(H), L.i(W), L.i(Cout),
                                        L.ptr(ws), L.stream()), "ab_conv2d_stem_wgrad_x3")
    return dw


def col_sum_x3(x_split, out):
    """Column sums of a split tensor [2, ..., C] (hi + lo) -> out fp32 [C]."""
    C = x_split.shape[-1]
    M = x_split[0].numel() // C
    lib = L.lib()
    part = torch.empty((lib.ab_col_stats_nparts(L.l(M)), C, 2), dtype=torch.float32, device=x_split.device)
    L.check(lib.ab_col_sum_x3(L.ptr(x_split[0]), L.ptr(x_split[1]), L.l(M), L.i(C), L.ptr(part), L.ptr(out), L.stream()), "ab_col_sum_x3")
    return out


def pose_assemble(kp3d, box6d, root_joint, cam_intr, corners_can, center_idx, inp_res):
    """hybridbaseline.py:49-96 in one launch (eval-mode forwards): kp3d [B,22,3] f32, box6d [B,>=6] (row-pitched view allowed),
    -> dict of the module's geometric outputs."""
    B, dev = kp3d.shape[0], kp3d.device
    z = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)   # noqa: E731
    o = dict(joints_3d_abs=z(B, 21, 3), corners_3d_abs=z(B, 8, 3), box_rot_rotmat=z(B, 3, 3), uvd2d=z(B, 30, 3), joints_3d=z(B, 21, 3),
             corners_3d=z(B, 8, 3), boxroot_3d_abs=z(B, 1, 3))
    L.check(L.lib().ab_pose_assemble(L.ptr(kp3d), L.view_ptr(box6d), L.i(box6d.stride(0)), L.ptr(root_joint), L.ptr(cam_intr), L.ptr(corners_can),
                                     L.i(B), L.i(center_idx), L.f(inp_res[0]), L.f(inp_res[1]), L.ptr(o["joints_3d_abs"]),
                                     L.ptr(o["corners_3d_abs"]), L.ptr(o["box_rot_rotmat"]), L.ptr(o["uvd2d"]), L.ptr(o["joints_3d"]),
                                     L.ptr(o["corners_3d"]), L.ptr(o["boxroot_3d_abs"]), L.stream()), "ab_pose_assemble")
    return o

"""Soft-argmax head op (reference: anakin/models/simplebaseline.py:16-71,183-189) on the HIP kernel."""
import torch

from . import _lib as L


def softargmax3d_fwd(logits, C, D, DP):
    """logits [B,H,W,C*DP] (f32|bf16, device) -> uvd [B,C,3], conf [B,C], stat [B,C,2] (all f32)."""
    B, H, W, CD = logits.shape
    if CD != C * DP or DP < D:
        raise ValueError(f"logits last dim {CD} != NCLASSES*DEPTH_PITCH {C * DP}")
    lib = L.lib()
    nt = lib.ab_softargmax3d_ntiles(L.i(H), L.i(W))
    dev = logits.device
    part = torch.empty((B, nt, C, 8), dtype=torch.float32, device=dev)
    uvd = torch.empty((B, C, 3), dtype=torch.float32, device=dev)
    conf = torch.empty((B, C), dtype=torch.float32, device=dev)
    stat = torch.empty((B, C, 2), dtype=torch.float32, device=dev)
    L.check(lib.ab_softargmax3d_fwd(L.ptr(logits), L.i(L.dt(logits)), L.i(B), L.i(C), L.i(D), L.i(DP), L.i(H), L.i(W),
                                    L.ptr(part), L.ptr(uvd), L.ptr(conf), L.ptr(stat), L.stream()),
            "ab_softargmax3d_fwd")
    return uvd, conf, stat


def softargmax3d_bwd(logits, C, D, DP, uvd, conf, stat, g_uvd, g_conf=None, inplace=False):
    """-> dlogits (same shape/dtype as logits; written over `logits` when inplace)."""
    B, H, W, _ = logits.shape
    g_uvd = g_uvd.contiguous().float()
    gc = g_conf.contiguous().float() if g_conf is not None else None
    dl = logits if inplace else torch.empty_like(logits)
    L.check(L.lib().ab_softargmax3d_bwd(L.ptr(logits), L.i(L.dt(logits)), L.i(B), L.i(C), L.i(D), L.i(DP), L.i(H),
                                        L.i(W), L.ptr(uvd), L.ptr(conf), L.ptr(stat), L.ptr(g_uvd), L.ptr(gc),
                                        L.ptr(dl), L.stream()), "ab_softargmax3d_bwd")
    return dl


def softargmax3d_bwd_x3(logits, C, D, DP, uvd, conf, stat, g_uvd, g_conf=None, dbias=None):
    """fp32 logits -> dlogits as split-bf16 planes [2, B, H, W, C*DP] (ab_softargmax3d_bwd_x3).
    dbias (fp32 [C*DP], written): the column sums of dlogits -- the bias gradient of the final layer -- from the same pass; the
    returned planes then carry `_ab_bias_done`."""
    B, H, W, _ = logits.shape
    g_uvd = g_uvd.contiguous().float()
    gc = g_conf.contiguous().float() if g_conf is not None else None
    dl = torch.empty((2,) + tuple(logits.shape), dtype=torch.bfloat16, device=logits.device)
    if dbias is not None:
        lib = L.lib()
        part = torch.empty((lib.ab_softargmax3d_bwd_x3_bias_rows(L.i(B), L.i(H), L.i(W)), C * DP), dtype=torch.float32, device=logits.device)
        L.check(lib.ab_softargmax3d_bwd_x3_bias(L.ptr(logits), L.i(B), L.i(C), L.i(D), L.i(DP), L.i(H), L.i(W), L.ptr(uvd), L.ptr(conf),
                                                L.ptr(stat), L.ptr(g_uvd), L.ptr(gc), L.ptr(dl[0]), L.ptr(dl[1]), L.ptr(part),
                                                L.ptr(dbias), L.stream()), "ab_softargmax3d_bwd_x3_bias")
        dl._ab_bias_done = True
        return dl
    L.check(L.lib().ab_softargmax3d_bwd_x3(L.ptr(logits), L.i(B), L.i(C), L.i(D), L.i(DP), L.i(H), L.i(W), L.ptr(uvd), L.ptr(conf),
                                           L.ptr(stat), L.ptr(g_uvd), L.ptr(gc), L.ptr(dl[0]), L.ptr(dl[1]), L.stream()),
            "ab_softargmax3d_bwd_x3")
    return dl


class _SoftArgmax3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, C, D, DP):
        uvd, conf, stat = softargmax3d_fwd(logits, C, D, DP)
        ctx.save_for_backward(logits, uvd, conf, stat)
        ctx.dims = (C, D, DP)
        return uvd, conf

    @staticmethod
    def backward(ctx, g_uvd, g_conf):
        logits, uvd, conf, stat = ctx.saved_tensors
        C, D, DP = ctx.dims
        if g_uvd is None:
            g_uvd = torch.zeros_like(uvd)
        return softargmax3d_bwd(logits, C, D, DP, uvd, conf, stat, g_uvd, g_conf), None, None, None


def softargmax3d(logits_nhwc: torch.Tensor, nclasses: int, depth: int, depth_pitch: int = None):
    """logits (B, H, W, nclasses*depth_pitch) NHWC, channel = c*depth_pitch + d (d < depth valid)
    ->  uvd (B, nclasses, 3), conf (B, nclasses).  Differentiable (autograd wrapper over the two HIP kernels)."""
    return _SoftArgmax3D.apply(logits_nhwc.contiguous(), nclasses, depth, depth_pitch or depth)

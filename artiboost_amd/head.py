"""Soft-argmax head op (reference: anakin/models/simplebaseline.py:16-71,183-189) on the HIP kernel."""
import torch

from . import _lib as L


class _SoftArgmax3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, C, D):
        B, H, W, CD = logits.shape
        if CD != C * D:
            raise ValueError(f"logits last dim {CD} != NCLASSES*DEPTH {C * D}")
        lib = L.lib()
        nt = lib.ab_softargmax3d_ntiles(L.i(H), L.i(W))
        dev = logits.device
        part = torch.empty((B, nt, C, 8), dtype=torch.float32, device=dev)
        uvd = torch.empty((B, C, 3), dtype=torch.float32, device=dev)
        conf = torch.empty((B, C), dtype=torch.float32, device=dev)
        stat = torch.empty((B, C, 2), dtype=torch.float32, device=dev)
        L.check(lib.ab_softargmax3d_fwd(L.ptr(logits), L.i(L.dt(logits)), L.i(B), L.i(C), L.i(D), L.i(H), L.i(W),
                                        L.ptr(part), L.ptr(uvd), L.ptr(conf), L.ptr(stat), L.stream()),
                "ab_softargmax3d_fwd")
        ctx.save_for_backward(logits, uvd, conf, stat)
        ctx.dims = (B, C, D, H, W)
        return uvd, conf

    @staticmethod
    def backward(ctx, g_uvd, g_conf):
        logits, uvd, conf, stat = ctx.saved_tensors
        B, C, D, H, W = ctx.dims
        lib = L.lib()
        if g_uvd is None:
            g_uvd = torch.zeros_like(uvd)
        g_uvd = g_uvd.contiguous().float()
        gc = g_conf.contiguous().float() if g_conf is not None else None
        dl = torch.empty_like(logits)
        L.check(lib.ab_softargmax3d_bwd(L.ptr(logits), L.i(L.dt(logits)), L.i(B), L.i(C), L.i(D), L.i(H), L.i(W),
                                        L.ptr(uvd), L.ptr(conf), L.ptr(stat), L.ptr(g_uvd), L.ptr(gc), L.ptr(dl),
                                        L.stream()), "ab_softargmax3d_bwd")
        return dl, None, None


def softargmax3d(logits_nhwc: torch.Tensor, nclasses: int, depth: int):
    """logits (B, H, W, nclasses*depth) NHWC, channel = c*depth + d  ->  uvd (B, nclasses, 3), conf (B, nclasses)."""
    return _SoftArgmax3D.apply(logits_nhwc.contiguous(), nclasses, depth)

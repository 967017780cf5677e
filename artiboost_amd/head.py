"""Soft-argmax head op (reference: anakin/models/simplebaseline.py:16-71,183-189) on the HIP kernel."""
import torch

from . import _lib as L


NORM_CODE = {"softmax": 0, "sigmoid": 1}       # IntegralDeconvHead NORM_TYPE (simplebaseline.py:16-40); "divide_sum": not implemented


def norm_code(norm_type):
    try:
        return NORM_CODE[norm_type]
    except KeyError:
        raise NotImplementedError(f"IntegralDeconvHead NORM_TYPE {norm_type!r}: softmax and sigmoid run on the HIP kernels; 'divide_sum' "
                                  f"(raw, possibly negative weights -- the reference advises against it) does not") from None


def softargmax3d_fwd(logits, C, D, DP, norm=0):
    """logits [B,H,W,C*DP] (f32|bf16, device) -> uvd [B,C,3], conf [B,C], stat [B,C,2] (all f32).  norm: NORM_CODE."""
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
    L.check(lib.ab_softargmax3d_fwd_norm(L.ptr(logits), L.i(L.dt(logits)), L.i(B), L.i(C), L.i(D), L.i(DP), L.i(H), L.i(W), L.i(norm),
                                         L.ptr(part), L.ptr(uvd), L.ptr(conf), L.ptr(stat), L.stream()),
            "ab_softargmax3d_fwd_norm")
    return uvd, conf, stat


def softargmax3d_stage2(part, C):
    """part [B, ntile, C, 8] (written by the final layer's GEMM epilogue: kernels.conv1x1_sam_fwd_x3) -> uvd, conf, stat as softargmax3d_fwd."""
    B, nt = int(part.shape[0]), int(part.shape[1])
    dev = part.device
    uvd = torch.empty((B, C, 3), dtype=torch.float32, device=dev)
    conf = torch.empty((B, C), dtype=torch.float32, device=dev)
    stat = torch.empty((B, C, 2), dtype=torch.float32, device=dev)
    L.check(L.lib().ab_softargmax3d_stage2(L.ptr(part), L.i(B), L.i(C), L.i(nt), L.ptr(uvd), L.ptr(conf), L.ptr(stat), L.stream()),
            "ab_softargmax3d_stage2")
    return uvd, conf, stat


def softargmax3d_bwd(logits, C, D, DP, uvd, conf, stat, g_uvd, g_conf=None, inplace=False, norm=0):
    """-> dlogits (same shape/dtype as logits; written over `logits` when inplace)."""
    B, H, W, _ = logits.shape
    g_uvd = g_uvd.contiguous().float()
    gc = g_conf.contiguous().float() if g_conf is not None else None
    dl = logits if inplace else torch.empty_like(logits)
    L.check(L.lib().ab_softargmax3d_bwd_norm(L.ptr(logits), L.i(L.dt(logits)), L.i(B), L.i(C), L.i(D), L.i(DP), L.i(H),
                                             L.i(W), L.i(norm), L.ptr(uvd), L.ptr(conf), L.ptr(stat), L.ptr(g_uvd), L.ptr(gc),
                                             L.ptr(dl), L.stream()), "ab_softargmax3d_bwd_norm")
    return dl


def softargmax3d_bwd_x3(logits, C, D, DP, uvd, conf, stat, g_uvd, g_conf=None, dbias=None, norm=0):
    """fp32 logits -> dlogits as split-bf16 planes [2, B, H, W, C*DP] (ab_softargmax3d_bwd_x3).
    dbias (fp32 [C*DP], written): the column sums of dlogits -- the bias gradient of the final layer -- from the same pass; the
    returned planes then carry `_ab_bias_done`."""
    B, H, W, _ = logits.shape
    g_uvd = g_uvd.contiguous().float()
    gc = g_conf.contiguous().float() if g_conf is not None else None
    dl = torch.empty((2,) + tuple(logits.shape), dtype=torch.bfloat16, device=logits.device)
    if norm:                      # sigmoid head (g_conf is not differentiated there)
        if gc is not None:
            raise NotImplementedError("gradient through the confidence of the sigmoid head")
        lib, part = L.lib(), None
        if dbias is not None:
            part = torch.empty((lib.ab_softargmax3d_bwd_x3_bias_rows(L.i(B), L.i(H), L.i(W)), C * DP), dtype=torch.float32, device=logits.device)
        L.check(lib.ab_softargmax3d_bwd_x3_norm(L.ptr(logits), L.i(B), L.i(C), L.i(D), L.i(DP), L.i(H), L.i(W), L.i(norm), L.ptr(uvd), L.ptr(conf),
                                                L.ptr(stat), L.ptr(g_uvd), L.ptr(dl[0]), L.ptr(dl[1]), L.ptr(part), L.ptr(dbias), L.stream()),
                "ab_softargmax3d_bwd_x3_norm")
        if dbias is not None:
            dl._ab_bias_done = True
        return dl
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


# ---- the differentiable op, registered with the dispatcher: torch.ops.artiboost_hip.softargmax3d
# (the raw kernels are torch.ops.artiboost_hip.softargmax3d_fwd / _bwd -- libartiboost_torch.so; this functional op allocates the
# outputs, and autograd / fake-tensor rules are registered on it, so it composes with torch.autograd and torch.compile's tracing)
_FRAG = torch.library.Library("artiboost_hip", "FRAGMENT")
_FRAG.define("softargmax3d(Tensor logits, int nclasses, int depth, int depth_pitch, int norm_type=0) -> (Tensor, Tensor, Tensor)")
_FRAG.define("softargmax3d_backward(Tensor logits, int nclasses, int depth, int depth_pitch, Tensor uvd, Tensor conf, Tensor stat, "
             "Tensor g_uvd, Tensor? g_conf, int norm_type=0) -> Tensor")


def _sam_fwd_impl(logits, nclasses, depth, depth_pitch, norm_type=0):
    return softargmax3d_fwd(logits.contiguous(), nclasses, depth, depth_pitch, norm_type)


def _sam_bwd_impl(logits, nclasses, depth, depth_pitch, uvd, conf, stat, g_uvd, g_conf, norm_type=0):
    return softargmax3d_bwd(logits, nclasses, depth, depth_pitch, uvd, conf, stat, g_uvd, g_conf, norm=norm_type)


_FRAG.impl("softargmax3d", _sam_fwd_impl, "CUDA")
_FRAG.impl("softargmax3d_backward", _sam_bwd_impl, "CUDA")


@torch.library.register_fake("artiboost_hip::softargmax3d")
def _sam_fake(logits, nclasses, depth, depth_pitch, norm_type=0):
    B = logits.shape[0]
    f = lambda *s: logits.new_empty(s, dtype=torch.float32)   # noqa: E731
    return f(B, nclasses, 3), f(B, nclasses), f(B, nclasses, 2)


@torch.library.register_fake("artiboost_hip::softargmax3d_backward")
def _sam_bwd_fake(logits, nclasses, depth, depth_pitch, uvd, conf, stat, g_uvd, g_conf, norm_type=0):
    return torch.empty_like(logits)


def _sam_setup(ctx, inputs, output):
    logits, C, D, DP = inputs[:4]
    uvd, conf, stat = output
    ctx.save_for_backward(logits, uvd, conf, stat)
    ctx.dims = (C, D, DP, inputs[4] if len(inputs) > 4 else 0)


def _sam_backward(ctx, g_uvd, g_conf, g_stat):
    logits, uvd, conf, stat = ctx.saved_tensors
    C, D, DP, norm = ctx.dims
    if g_uvd is None:
        g_uvd = torch.zeros_like(uvd)
    gc = g_conf.contiguous().float() if (g_conf is not None and not norm) else None
    dl = torch.ops.artiboost_hip.softargmax3d_backward(logits, C, D, DP, uvd, conf, stat, g_uvd.contiguous().float(), gc, norm)
    return dl, None, None, None, None


torch.library.register_autograd("artiboost_hip::softargmax3d", _sam_backward, setup_context=_sam_setup)


def softargmax3d(logits_nhwc: torch.Tensor, nclasses: int, depth: int, depth_pitch: int = None, norm_type: str = "softmax"):
    """logits (B, H, W, nclasses*depth_pitch) NHWC, channel = c*depth_pitch + d (d < depth valid)
    ->  uvd (B, nclasses, 3), conf (B, nclasses).  Differentiable: torch.ops.artiboost_hip.softargmax3d (dispatcher op with
    registered autograd) over the two HIP kernels."""
    if not logits_nhwc.is_cuda:
        raise RuntimeError("artiboost_hip ops need device tensors (HIP); got a CPU tensor")
    uvd, conf, _ = torch.ops.artiboost_hip.softargmax3d(logits_nhwc.contiguous(), nclasses, depth, depth_pitch or depth, norm_code(norm_type))
    return uvd, conf

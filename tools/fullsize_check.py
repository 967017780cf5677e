"""Every fast conv path at the benchmark shapes (B=64, 256x256 geometry), in both precisions of the fast kernels:
  * run-to-run bit-identity (3 runs with a dirtied allocator in between: no uninitialised reads, no races at 8192-workgroup
    grids);
  * agreement with an INDEPENDENT fp32 reference computed at the same full size: torch's own conv2d / conv2d-input-gradient /
    conv2d-weight-gradient on the device in fp32 (MIOpen behind torch -- none of this build's kernels);
  * bf16 only: agreement with the register-staged v1 kernels as well (AB_CONV_V1 / AB_WGRAD*_OFF).
Tolerances are relative to the largest reference entry: bf16 operands 2e-2 (outputs rounded to bf16) / 3e-3 (fp32 weight
gradients); bf16x3 (split-bf16, fp32 outputs) 5e-5.
usage: python tools/fullsize_check.py [bf16|bf16x3|both]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K   # noqa: E402

B = int(os.environ.get("B", 64))
SHAPES = [("l1 3x3", 64, 64, 64, 3, 1, 1), ("l2 3x3s2", 64, 64, 128, 3, 2, 1), ("l2 3x3", 32, 128, 128, 3, 1, 1),
          ("l2 ds", 64, 64, 128, 1, 2, 0), ("l3 3x3s2", 32, 128, 256, 3, 2, 1), ("l3 3x3", 16, 256, 256, 3, 1, 1),
          ("l3 ds", 32, 128, 256, 1, 2, 0), ("l4 3x3s2", 16, 256, 512, 3, 2, 1), ("l4 3x3", 8, 512, 512, 3, 1, 1),
          ("l4 ds", 16, 256, 512, 1, 2, 0), ("deconv1", 16, 256, 512, 4, 2, 1), ("deconv2", 32, 256, 256, 4, 2, 1),
          ("final", 32, 256, 704, 1, 1, 0)]
OFF = ("AB_CONV_V1", "AB_WGRAD3_OFF", "AB_WGRAD2_OFF")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False


def dirty():
    j = torch.randn(96 * 1024 * 1024, device="cuda")
    del j


def v1(fn):
    for k in OFF:
        os.environ[k] = "1"
    try:
        return fn()
    finally:
        for k in OFF:
            del os.environ[k]


def rel(o, r):
    return float((o.float() - r.float()).abs().max() / (r.float().abs().max() + 1e-30))


def check(name, fn, ref, tol, vs_v1=None):
    outs = []
    for _ in range(3):
        dirty()
        o = fn()
        torch.cuda.synchronize()
        outs.append(o.clone())
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    err = rel(outs[0], ref)
    msg = f"{name:30s} deterministic={same}  max err vs torch fp32 = {err:.2e} (tol {tol:.0e})"
    ok = same and err < tol
    if vs_v1 is not None:
        e1 = rel(outs[0], v1(fn))
        msg += f"  vs v1 = {e1:.2e}"
        ok = ok and e1 < vs_v1
    print(msg + ("  ok" if ok else "  FAIL"))
    return ok


def nchw(t):
    return t.permute(0, 3, 1, 2)


def run(mode):
    x3 = mode == "bf16x3"
    allok = True
    torch.manual_seed(0)
    for name, H, Ci, Co, k, s, p in SHAPES:
        x = torch.randn(B, H, H, Ci, device="cuda")
        w = torch.randn(Co, k, k, Ci, device="cuda") * 0.05
        Ho = K.conv_out(H, k, s, p)
        dy = torch.randn(B, Ho, Ho, Co, device="cuda")
        if not x3:                     # the bf16 kernels see bf16-rounded operands: give the reference the same values
            x, w, dy = x.bfloat16().float(), w.bfloat16().float(), dy.bfloat16().float()
        wt = w.permute(3, 1, 2, 0).contiguous()
        w_oihw = w.permute(0, 3, 1, 2).contiguous()
        # ---- independent fp32 references at full size (torch / MIOpen)
        y_ref = F.conv2d(nchw(x), w_oihw, stride=s, padding=p).permute(0, 2, 3, 1)
        dx_ref = torch.nn.grad.conv2d_input((B, Ci, H, H), w_oihw, nchw(dy), stride=s, padding=p).permute(0, 2, 3, 1)
        dw_ref = torch.nn.grad.conv2d_weight(nchw(x), (Co, Ci, k, k), nchw(dy), stride=s, padding=p).permute(0, 2, 3, 1)
        st_ref = torch.stack([y_ref.double().sum((0, 1, 2)), (y_ref.double() ** 2).sum((0, 1, 2))], 1).float()
        if x3:
            xs, ws, wts, dys = K.split(x), K.split(w), K.split(wt), K.split(dy)
            allok &= check(f"{mode} {name} fwd", lambda: K.conv2d_fwd_x3(xs, ws, s, p, want_stats=True)[0], y_ref, 5e-5)
            allok &= check(f"{mode} {name} fwd stats", lambda: K.conv2d_fwd_x3(xs, ws, s, p, want_stats=True)[1].sum(0), st_ref, 5e-5)
            allok &= check(f"{mode} {name} dgrad", lambda: K.conv2d_dgrad_x3(dys, wts, (H, H), s, p), dx_ref, 5e-5)
            if Co % 64 == 0:
                allok &= check(f"{mode} {name} wgrad", lambda: K.conv2d_wgrad_x3(xs, dys, k, k, s, p), dw_ref, 5e-5)
        else:
            xb, wb, wtb, dyb = x.bfloat16(), w.bfloat16(), wt.bfloat16(), dy.bfloat16()
            allok &= check(f"{mode} {name} fwd", lambda: K.conv2d_fwd(xb, wb, s, p, want_stats=True)[0], y_ref, 2e-2, vs_v1=2e-2)
            allok &= check(f"{mode} {name} dgrad", lambda: K.conv2d_dgrad(dyb, wtb, (H, H), s, p), dx_ref, 2e-2, vs_v1=2e-2)
            allok &= check(f"{mode} {name} wgrad", lambda: K.conv2d_wgrad(xb, dyb, k, k, s, p), dw_ref, 3e-3, vs_v1=2e-3)
    # ---- stem
    img = torch.rand(B, 3, 256, 256, device="cuda") - 0.5
    w7 = 0.1 * torch.randn(64, 3, 7, 7, device="cuda")
    dy = torch.randn(B, 128, 128, 64, device="cuda")
    if not x3:
        img, w7, dy = img.bfloat16().float(), w7.bfloat16().float(), dy.bfloat16().float()
    y_ref = F.conv2d(img, w7, stride=2, padding=3).permute(0, 2, 3, 1)
    dw_ref = torch.nn.grad.conv2d_weight(img, (64, 3, 7, 7), nchw(dy), stride=2, padding=3)
    wst = torch.zeros(64, 7, 8, 4, device="cuda")
    wst[:, :, :7, :3] = w7.permute(0, 2, 3, 1)
    unpack = lambda d: d[:, :, :7, :3].permute(0, 3, 1, 2)      # noqa: E731
    if x3:
        xpad = K.split(K.image_pad_nhwc4(img, torch.float32))
        wsp, dys = K.split(wst), K.split(dy)
        allok &= check(f"{mode} stem fwd", lambda: K.conv2d_stem_fwd_x3(xpad, wsp, 256, 256, want_stats=True)[0], y_ref, 5e-5)
        allok &= check(f"{mode} stem wgrad", lambda: unpack(K.conv2d_stem_wgrad_x3(xpad, dys, 256, 256)), dw_ref, 5e-5)
    else:
        xpad = K.image_pad_nhwc4(img, torch.bfloat16)
        allok &= check(f"{mode} stem fwd", lambda: K.conv2d_stem_fwd(xpad, wst.bfloat16(), 256, 256, want_stats=True)[0], y_ref, 2e-2, vs_v1=2e-2)
        allok &= check(f"{mode} stem wgrad", lambda: unpack(K.conv2d_stem_wgrad(xpad, dy.bfloat16(), 256, 256)), dw_ref, 3e-3, vs_v1=2e-3)
    return allok


ok = True
for m in (("bf16x3", "bf16") if which == "both" else (which,)):
    ok &= run(m)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)

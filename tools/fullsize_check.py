"""Every fast conv path at the benchmark shapes (B=64): run-to-run bit-identity (3 runs with a dirtied allocator in
between) and agreement with the register-staged v1 kernels (AB_CONV_V1 / AB_WGRAD*_OFF), which the unit tests pin
against torch at small sizes."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K

B = int(os.environ.get("B", 64))
SHAPES = [("l1 3x3", 64, 64, 64, 3, 1, 1), ("l2 3x3s2", 64, 64, 128, 3, 2, 1), ("l2 3x3", 32, 128, 128, 3, 1, 1),
          ("l2 ds", 64, 64, 128, 1, 2, 0), ("l3 3x3s2", 32, 128, 256, 3, 2, 1), ("l3 3x3", 16, 256, 256, 3, 1, 1),
          ("l3 ds", 32, 128, 256, 1, 2, 0), ("l4 3x3s2", 16, 256, 512, 3, 2, 1), ("l4 3x3", 8, 512, 512, 3, 1, 1),
          ("l4 ds", 16, 256, 512, 1, 2, 0), ("deconv1", 16, 256, 512, 4, 2, 1), ("deconv2", 32, 256, 256, 4, 2, 1),
          ("final", 32, 256, 704, 1, 1, 0)]
OFF = ("AB_CONV_V1", "AB_WGRAD3_OFF", "AB_WGRAD2_OFF")
dt = torch.bfloat16


def dirty():
    j = torch.randn(96 * 1024 * 1024, device="cuda"); del j


def ref(fn):
    for k in OFF:
        os.environ[k] = "1"
    try:
        return fn()
    finally:
        for k in OFF:
            del os.environ[k]


def check(name, fn, rtol):
    outs = []
    for _ in range(3):
        dirty()
        o = fn(); torch.cuda.synchronize(); outs.append(o.clone())
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    r = ref(fn).float(); o = outs[0].float()
    err = float((o - r).abs().max() / (r.abs().max() + 1e-30))
    ok = same and err < rtol
    print(f"{name:26s} deterministic={same}  max rel err vs v1 = {err:.2e}  {'ok' if ok else 'FAIL'}")
    return ok


allok = True
torch.manual_seed(0)
for name, H, Ci, Co, k, s, p in SHAPES:
    x = torch.randn(B, H, H, Ci, device="cuda").to(dt)
    w = (torch.randn(Co, k, k, Ci, device="cuda") * 0.05).to(dt)
    wt = w.permute(3, 1, 2, 0).contiguous()
    Ho = K.conv_out(H, k, s, p)
    dy = torch.randn(B, Ho, Ho, Co, device="cuda").to(dt)
    allok &= check(name + " fwd", lambda: K.conv2d_fwd(x, w, s, p, want_stats=True)[0], 2e-2)
    allok &= check(name + " fwd stats", lambda: K.conv2d_fwd(x, w, s, p, want_stats=True)[1].sum(0), 2e-2)
    allok &= check(name + " dgrad", lambda: K.conv2d_dgrad(dy, wt, (H, H), s, p), 2e-2)
    allok &= check(name + " wgrad", lambda: K.conv2d_wgrad(x, dy, k, k, s, p), 2e-3)
img = torch.rand(B, 3, 256, 256, device="cuda") - 0.5
xpad = K.image_pad_nhwc4(img, dt)
w = (0.1 * torch.randn(64, 7, 8, 4, device="cuda")).to(dt)
dy = torch.randn(B, 128, 128, 64, device="cuda").to(dt)
allok &= check("stem fwd", lambda: K.conv2d_stem_fwd(xpad, w, 256, 256, want_stats=True)[0], 2e-2)
allok &= check("stem wgrad", lambda: K.conv2d_stem_wgrad(xpad, dy, 256, 256), 2e-3)
print("ALL OK" if allok else "FAILURES")
sys.exit(0 if allok else 1)

"""Stem forward (7x7/2, 3 -> 64) at the benchmark geometry: halo kernel vs LDS-DMA tap kernel vs register-staged kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from artiboost_amd import kernels as K

B, S = 64, 256
torch.manual_seed(0)
xpad = K.image_pad_nhwc4(torch.rand(B, 3, S, S, device="cuda") - 0.5, torch.bfloat16)
w = (0.1 * torch.randn(64, 7, 8, 4, device="cuda")).to(torch.bfloat16)
w[:, :, 7] = 0
w[..., 3] = 0


def timed(env):
    for k in ("AB_STEM_V1", "AB_STEM_HALO", "AB_STEM_HALO_WGS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(5):
        y, st = K.conv2d_stem_fwd(xpad, w, S, S, want_stats=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        K.conv2d_stem_fwd(xpad, w, S, S, want_stats=True)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3, y.clone(), st.double().sum(0)


ref_t, ref, ref_st = timed({"AB_STEM_V1": "1"})
print(f"register-staged  {ref_t:7.1f} us")
for name, env in (("tap LDS-DMA", {"AB_STEM_HALO": "0"}), ("halo", {})):
    t, y, st = timed(env)
    err = float((y.float() - ref.float()).abs().max()) / float(ref.float().abs().max())
    serr = float((st - ref_st).abs().max() / ref_st.abs().max())
    print(f"{name:16s} {t:7.1f} us  max rel err {err:.2e}  stats rel err {serr:.2e}  out {y.numel() * 2 / t / 1e3:.0f} GB/s")

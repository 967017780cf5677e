"""Per-shape timing of the conv kernels (fwd / dgrad / wgrad) at the benchmark geometry (B=64, 256x256)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K

B = 64
SHAPES = [  # name, H, W, Cin, Cout, k, stride, pad
    ("l1 3x3 64->64 @64", 64, 64, 64, 64, 3, 1, 1),
    ("l2 3x3s2 64->128", 64, 64, 64, 128, 3, 2, 1),
    ("l2 3x3 128->128 @32", 32, 32, 128, 128, 3, 1, 1),
    ("l3 3x3 256->256 @16", 16, 16, 256, 256, 3, 1, 1),
    ("l4 3x3 512->512 @8", 8, 8, 512, 512, 3, 1, 1),
    ("l2 ds 1x1s2 64->128", 64, 64, 64, 128, 1, 2, 0),
    ("final 1x1 256->704 @32", 32, 32, 256, 704, 1, 1, 0),
    ("deconv-as-conv 4x4s2 256->256 @32", 32, 32, 256, 256, 4, 2, 1),
]
dt = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


for name, H, W, Ci, Co, k, s, p in SHAPES:
    x = torch.randn(B, H, W, Ci, device="cuda").to(dt)
    w = torch.randn(Co, k, k, Ci, device="cuda").to(dt) * 0.05
    wt = w.permute(3, 1, 2, 0).contiguous()
    Ho, Wo = K.conv_out(H, k, s, p), K.conv_out(W, k, s, p)
    dy = torch.randn(B, Ho, Wo, Co, device="cuda").to(dt)
    flop = 2.0 * B * Ho * Wo * Co * Ci * k * k
    t_f = timeit(lambda: K.conv2d_fwd(x, w, s, p, want_stats=True))
    t_d = timeit(lambda: K.conv2d_dgrad(dy, wt, (H, W), s, p))
    t_w = timeit(lambda: K.conv2d_wgrad(x, dy, k, k, s, p)) if Ci % 64 == 0 and Co % 64 == 0 else float("nan")
    print(f"{name:36s} GFLOP {flop/1e9:6.1f} | fwd {t_f:7.1f} us {flop/t_f/1e6:7.1f} TF | dgrad {t_d:7.1f} us {flop/t_d/1e6:7.1f} TF | wgrad {t_w:7.1f} us {flop/t_w/1e6:7.1f} TF")

"""Per-shape timing of the split-bf16 (bf16x3) conv kernels at the benchmark geometry (B=64, 256x256).
usage: python tools/bench_conv_x3.py [name-substring] [fwd|dgrad|wgrad|all] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K   # noqa: E402

B = 64
SHAPES = [  # name, H, W, Cin, Cout, k, stride, pad, calls per step
    ("l1 3x3 64->64 @64", 64, 64, 64, 64, 3, 1, 1, 6),
    ("l2 3x3s2 64->128", 64, 64, 64, 128, 3, 2, 1, 1),
    ("l2 3x3 128->128 @32", 32, 32, 128, 128, 3, 1, 1, 7),
    ("l2 ds 1x1s2 64->128", 64, 64, 64, 128, 1, 2, 0, 1),
    ("l3 3x3s2 128->256", 32, 32, 128, 256, 3, 2, 1, 1),
    ("l3 3x3 256->256 @16", 16, 16, 256, 256, 3, 1, 1, 11),
    ("l3 ds 1x1s2 128->256", 32, 32, 128, 256, 1, 2, 0, 1),
    ("l4 3x3s2 256->512", 16, 16, 256, 512, 3, 2, 1, 1),
    ("l4 3x3 512->512 @8", 8, 8, 512, 512, 3, 1, 1, 5),
    ("l4 ds 1x1s2 256->512", 16, 16, 256, 512, 1, 2, 0, 1),
    ("deconv1-as-conv 4x4s2 256->512 @16", 16, 16, 256, 512, 4, 2, 1, 1),
    ("deconv2-as-conv 4x4s2 256->256 @32", 32, 32, 256, 256, 4, 2, 1, 1),
    ("final 1x1 256->704 @32", 32, 32, 256, 704, 1, 1, 0, 1),
    # tile-shape probe: two 8x8 layer-4 images side by side as one 8x16 image (run with AB_C3_FORCE=4: 128 pixels x 64 channels
    # per workgroup = the weight stream of a workgroup halved at the same 256-workgroup grid; timing only, not layer 4's maths)
    ("probe l4 pair-tile 8x16 B32", 8, 16, 512, 512, 3, 1, 1, 0, 32),
    # weight-gradient grouping probe (round 6): the same layer at twice / four times the batch = what ONE launch over 2 / 4 same-shape layers
    # costs (same 256 workgroups, slabs written and reduced once): compare with 2 x / 4 x the B = 64 line
    ("probe grp l1 B128", 64, 64, 64, 64, 3, 1, 1, 0, 128), ("probe grp l1 B256", 64, 64, 64, 64, 3, 1, 1, 0, 256),
    ("probe grp l2 B128", 32, 32, 128, 128, 3, 1, 1, 0, 128), ("probe grp l2 B256", 32, 32, 128, 128, 3, 1, 1, 0, 256),
    ("probe grp l3 B128", 16, 16, 256, 256, 3, 1, 1, 0, 128), ("probe grp l3 B256", 16, 16, 256, 256, 3, 1, 1, 0, 256),
    ("probe grp l4 B128", 8, 8, 512, 512, 3, 1, 1, 0, 128), ("probe grp l4 B256", 8, 8, 512, 512, 3, 1, 1, 0, 256),
]


def timeit(fn, n):
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


only = sys.argv[1] if len(sys.argv) > 1 else ""
what = sys.argv[2] if len(sys.argv) > 2 else "all"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tot = [0.0, 0.0, 0.0]
for name, H, W, Ci, Co, k, s, p, cnt, *rest in SHAPES:
    if only not in name or (name.startswith("probe") and not only):
        continue
    B = rest[0] if rest else 64
    x = K.split(torch.randn(B, H, W, Ci, device="cuda"))
    w = K.split(torch.randn(Co, k, k, Ci, device="cuda") * 0.05)
    wt = K.split((torch.randn(Ci, k, k, Co, device="cuda") * 0.05))
    Ho, Wo = K.conv_out(H, k, s, p), K.conv_out(W, k, s, p)
    dy = K.split(torch.randn(B, Ho, Wo, Co, device="cuda"))
    flop = 2.0 * B * Ho * Wo * Co * Ci * k * k
    t_f = timeit(lambda: K.conv2d_fwd_x3(x, w, s, p, want_stats=True), iters) if what in ("all", "fwd") else float("nan")
    t_d = timeit(lambda: K.conv2d_dgrad_x3(dy, wt, (H, W), s, p), iters) if what in ("all", "dgrad") else float("nan")
    t_w = timeit(lambda: K.conv2d_wgrad_x3(x, dy, k, k, s, p), iters) if what in ("all", "wgrad") and Co % 64 == 0 else float("nan")
    tot[0] += cnt * t_f; tot[1] += cnt * t_d; tot[2] += cnt * t_w
    print(f"{name:36s} x{cnt:<2d} GFLOP {flop/1e9:6.1f} | fwd {t_f:7.1f} us {flop/t_f/1e6:6.1f} TF | dgrad {t_d:7.1f} us {flop/t_d/1e6:6.1f} TF | "
          f"wgrad {t_w:7.1f} us {flop/t_w/1e6:6.1f} TF   (x3 roof 833 TF)")
print(f"per step (us): fwd {tot[0]:.0f}  dgrad {tot[1]:.0f}  wgrad {tot[2]:.0f}  total {sum(tot):.0f}")

"""Final-layer timings at the benchmark geometry (B = 64, 32 x 32 x 256 -> 22 x 32 logits): the register-resident GEMM (gemm_rw.hip) with and
without the fused soft-argmax statistics against the implicit GEMM it replaces (AB_GRW_OFF=1 in a second process) + sam_stage1.
usage: python tools/bench_head.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K   # noqa: E402
from artiboost_amd.head import softargmax3d_fwd, softargmax3d_stage2   # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B, H, W, Cin, C, D = 64, 32, 32, 256, 22, 28


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = K.split(torch.randn(B, H, W, Cin, device="cuda"))
w = K.split(torch.randn(C * 32, 1, 1, Cin, device="cuda") * 0.1)
b = torch.randn(C * 32, device="cuda")
flop = 2.0 * B * H * W * C * 32 * Cin
t_plain = timeit(lambda: K.conv2d_fwd_x3(x, w, 1, 0, bias=b), iters)
print(f"final 1x1 fwd (ab_conv2d_fwd_x3, AB_GRW_OFF={os.environ.get('AB_GRW_OFF', '0')}): {t_plain:7.1f} us  {flop / t_plain / 1e6:6.1f} TF ({flop / t_plain / 1e6 / 833:.2f} of 833)")
y = K.conv2d_fwd_x3(x, w, 1, 0, bias=b)
t_sam = timeit(lambda: softargmax3d_fwd(y, C, D, 32), iters)
print(f"soft-argmax two-stage forward on the logits: {t_sam:7.1f} us")
if K.conv1x1_sam_fwd_x3_ok(x, w, C, D, 32) and not os.environ.get("AB_GRW_OFF"):
    t_f = timeit(lambda: K.conv1x1_sam_fwd_x3(x, w, b, C, D), iters)
    _, part = K.conv1x1_sam_fwd_x3(x, w, b, C, D)
    t_2 = timeit(lambda: softargmax3d_stage2(part, C), iters)
    print(f"fused final layer + stage 1: {t_f:7.1f} us  ({flop / t_f / 1e6 / 833:.2f} of 833);  stage 2 alone: {t_2:5.1f} us")

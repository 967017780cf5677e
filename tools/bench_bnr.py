"""Times the 3x3 data-gradient launches with the fused BatchNorm-backward epilogue (csrc/conv3x3.hip, X3 = 2) at the benchmark geometry,
in the two forms the basic-block backward issues (hybridnet._backward_blocks):
  A: conv2's gradient -> arrives at relu(bn1(y1)): mask recomputed from y, no addend          (reads dy planes + y, writes dz)
  B: conv1's gradient -> arrives at relu(bn2(y2) + residual) of the block below: + mask plane + addend (the skip gradient)
and the plain data gradient of the same shape beside them.      python tools/bench_bnr.py [layer 1..4] [iters]
Kernel variants are chosen by environment switches read once per process (AB_C3_L1EP=0|1, AB_C3_FORCE=n, ...): one process per variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K   # noqa: E402

GEOM = {1: (64, 64, 64), 2: (32, 32, 128), 3: (16, 16, 256), 4: (8, 8, 512)}


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
    return e0.elapsed_time(e1) / n * 1e3


def main():
    layers = [int(sys.argv[1])] if len(sys.argv) > 1 and sys.argv[1] != "all" else [1, 2, 3, 4]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    B = 64
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("AB_C3"))
    for li in layers:
        H, W, C = GEOM[li]
        g = torch.Generator(device="cuda").manual_seed(li)
        dy = K.split(torch.randn(B, H, W, C, device="cuda", generator=g))
        wt = K.split(torch.randn(C, 3, 3, C, device="cuda", generator=g) * 0.05)
        y = torch.randn(B, H, W, C, device="cuda", generator=g)
        act = K.split(torch.relu(torch.randn(B, H, W, C, device="cuda", generator=g)))
        add = torch.randn(B, H, W, C, device="cuda", generator=g)
        bnp = torch.cat([torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1,
                         torch.rand(C, device="cuda") + 0.5]).contiguous()
        flop = 2.0 * B * H * W * C * C * 9
        t_p = timeit(lambda: K.conv2d_dgrad_x3(dy, wt, (H, W), 1, 1), iters)
        t_pa = timeit(lambda: K.conv2d_dgrad_x3(dy, wt, (H, W), 1, 1, addend=add), iters)
        t_a = timeit(lambda: K.conv2d_dgrad_x3(dy, wt, (H, W), 1, 1, bn=(y, None, bnp)), iters)
        t_b = timeit(lambda: K.conv2d_dgrad_x3(dy, wt, (H, W), 1, 1, addend=add, bn=(y, act, bnp)), iters)
        mb = B * H * W * C * 4 / 1e6
        print(f"layer{li} {H}x{W}x{C} [{tag}] plain {t_p:6.1f} us  plain+addend {t_pa:6.1f}  A(fused, y) {t_a:6.1f} ({3 * mb / t_a:5.2f} TB/s of {3 * mb:.0f} MB)  "
              f"B(fused, y+mask+addend) {t_b:6.1f} ({4.5 * mb / t_b:5.2f} TB/s of {4.5 * mb:.0f} MB)   roof {flop / 833e6:5.1f} us", flush=True)


if __name__ == "__main__":
    main()

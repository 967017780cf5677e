"""Bandwidth of the BN elementwise kernels at the benchmark tensor sizes."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K


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
    return e0.elapsed_time(e1) / n * 1e3


B = 64
for name, H, C in [("stem 128x128x64", 128, 64), ("l1 64x64x64", 64, 64), ("l2 32x32x128", 32, 128), ("l3 16x16x256", 16, 256),
                   ("l4 8x8x512", 8, 512)]:
    y = torch.randn(B, H, H, C, device="cuda").bfloat16()
    res = torch.randn_like(y)
    g = torch.randn_like(y)
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    part = K.col_stats(y.view(-1, C))
    bnp = K.bn_finalize(part, B * H * H, gamma, beta)
    out = torch.empty_like(y)
    dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
    mb = y.numel() * 2 / 1e6
    t_s = timeit(lambda: K.col_stats(y.view(-1, C)))
    t_f = timeit(lambda: K.bn_finalize(part, B * H * H, gamma, beta, out=bnp))
    t_a = timeit(lambda: K.bn_apply(y, bnp, relu=True, out=out))
    t_ar = timeit(lambda: K.bn_apply(y, bnp, res=res, relu=True, out=out))
    dy = torch.empty_like(y)
    t_b = timeit(lambda: K.bn_bwd(g, out, y, bnp, dg, db, relu=True, want_dz=False, dy_out=dy))
    t_bz = timeit(lambda: K.bn_bwd(g, out, y, bnp, dg, db, relu=True, want_dz=True, dy_out=dy))
    print(f"{name:18s} {mb:6.1f} MB | stats {t_s:6.1f} us {mb/t_s:5.2f} TB/s | fin {t_f:5.1f} | apply {t_a:6.1f} us {2*mb/t_a:5.2f} TB/s | "
          f"apply+res {t_ar:6.1f} us {3*mb/t_ar:5.2f} TB/s | bwd {t_b:6.1f} us {7*mb/t_b:5.2f} TB/s | bwd+dz {t_bz:6.1f} us {8*mb/t_bz:5.2f} TB/s")

"""In-process A/B of conv3x3 tile configurations (AB_C3_FORCE is read per call): alternating rounds on one box and one clock state, the only
comparison that is stable to better than the 10 % run-to-run spread of separate processes.  usage: python tools/ab_l1_tiles.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from artiboost_amd import kernels as K   # noqa: E402

B = 64


def t(fn, n=40):
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


for name, HW, C, cfgs in (("l1 64ch @64", 64, 64, ("7", "4", "1")), ("l2 128ch @32", 32, 128, ("2", "3")), ("l3 256ch @16", 16, 256, ("3", "6"))):
    x = K.split(torch.randn(B, HW, HW, C, device="cuda"))
    w = K.split(torch.randn(C, 3, 3, C, device="cuda") * 0.05)
    wt = K.split(torch.randn(C, 3, 3, C, device="cuda") * 0.05)
    dy = K.split(torch.randn(B, HW, HW, C, device="cuda"))
    y = torch.randn(B, HW, HW, C, device="cuda")
    bnp = torch.rand(4, C, device="cuda") + 0.5
    res = {}
    for rnd in range(4):
        for cfg in cfgs:
            os.environ["AB_C3_FORCE"] = cfg
            f = t(lambda: K.conv2d_fwd_x3(x, w, 1, 1, want_stats=True))
            d = t(lambda: K.conv2d_dgrad_x3(dy, wt, (HW, HW), 1, 1))
            db = t(lambda: K.conv2d_dgrad_x3(dy, wt, (HW, HW), 1, 1, bn=(y, None, bnp)))
            res.setdefault(cfg, []).append((f, d, db))
    for cfg, v in res.items():
        v = v[1:]
        print(f"{name:14s} cfg {cfg}: fwd {sum(a for a, _, _ in v) / len(v):6.1f}  dgrad {sum(b for _, b, _ in v) / len(v):6.1f}  dgrad+bn {sum(c for _, _, c in v) / len(v):6.1f} us")
os.environ.pop("AB_C3_FORCE", None)

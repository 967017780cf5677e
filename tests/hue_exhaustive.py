"""The hue op of ab_color_jitter (PIL: RGB -> HSV, H += shift mod 256, -> RGB) over ALL 2^24 RGB triples x ALL 256 hue shifts against the C oracle
(oracle/render_oracle.c, pinned to Pillow): the branch-free rgb2hsv8 / hsv2rgb8 of render.hip are the same functions as PIL's.
usage: python tests/hue_exhaustive.py [first_shift [last_shift]]      (~2 s per shift on the host for the oracle)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")      # (a checker: lives under tests/, the only place besides smoke / cpu_baseline that may use oracle/)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import render_oracle as ro      # noqa: E402
from artiboost_amd import _lib as L      # noqa: E402

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 255
v = np.arange(1 << 24, dtype=np.uint32)
rgbx = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255, np.full_like(v, 255)], axis=1).astype(np.uint8).reshape(4096, 4096, 4)
src = torch.from_numpy(rgbx).cuda()
ws = torch.empty(8, dtype=torch.uint8, device="cuda")
out = torch.empty_like(src)
bad_total = 0
for k in range(lo, hi + 1):
    # a factor f with (uint8)(int)(f * 255.0f) == k: ColorJitter's hue factor is in [-0.5, 0.5]; (k + 0.5) / 255 wraps the byte the same way
    f = np.float32((k + 0.5) / 255.0)
    assert int(np.float32(f * np.float32(255.0))) & 255 == k
    order, factor = [2, 2, 2, 2], [float(f), 0.0, 0.0, 0.0]      # hue(shift k), then three identity hue ops (shift 0: RGB -> HSV -> RGB again)
    ref = ro.color_jitter(rgbx, order, factor)
    o = torch.tensor([order], dtype=torch.int32, device="cuda")
    ft = torch.tensor([factor], dtype=torch.float32, device="cuda")
    assert L.lib().ab_color_jitter(L.ptr(src), L.i(1), L.i(4096 * 4096), L.ptr(o), L.ptr(ft), L.ptr(out), L.ptr(ws), L.stream()) == 0
    got = out.cpu().numpy()
    nbad = int((got[..., :3] != ref[..., :3]).any(axis=-1).sum())
    bad_total += nbad
    if nbad or k % 16 == 0:
        print(f"shift {k:3d}: {nbad} mismatching triples", flush=True)
print(f"shifts {lo}..{hi}: {bad_total} mismatches over {(hi - lo + 1) * (1 << 24)} pixel-ops")
sys.exit(1 if bad_total else 0)

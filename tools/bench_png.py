"""PNG frames at HO3D's size (640 x 480, photograph-like, written by Pillow): Pillow on one thread, the pooled zlib inflate, and the device
share (upload of the inflated scanlines + ab_png_unfilter_batch), per batch of 40 / 160 frames.  usage: python tools/bench_png.py"""
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_jpeg import _photo   # noqa: E402
from PIL import Image   # noqa: E402
from artiboost_amd.png import PngDecoder, parse, pool   # noqa: E402

files = []
for i in range(16):
    b = io.BytesIO()
    Image.fromarray(_photo(640, 480, i)).save(b, "PNG")
    files.append(b.getvalue())
print(f"file size {sum(len(f) for f in files) / 16 / 1024:.0f} KB, {len(parse(files[0]).idat)} IDAT chunks")
t0 = time.perf_counter()
for f in files:
    np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
print(f"Pillow Image.open().convert('RGB'), one thread: {(time.perf_counter() - t0) / 16 * 1e3:.2f} ms per frame")
infos = [parse(f) for f in files]
dec = PngDecoder("cuda")
fs, inf = (files * 10)[:160], (infos * 10)[:160]
out = torch.empty((160, 480, 640, 4), dtype=torch.uint8, device="cuda")
for n in (40, 160):
    for _ in range(2):
        dec.decode(fs[:n], out=out, infos=inf[:n])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        job = dec.submit(fs[:n], inf[:n])
        for f in job["futs"]:
            f.result()
        dec._evs[job["k"]] = None
    t_inf = (time.perf_counter() - t0) / 5
    ts = []
    for _ in range(5):
        job = dec.submit(fs[:n], inf[:n])
        for f in job["futs"]:
            f.result()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); dec.complete(job, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    raw_mb = sum(it.raw_bytes for it in inf[:n]) / 1e6
    # the kernel alone: re-run it on the uploaded bytes
    print(f"{n} frames: inflate on {pool()._max_workers} threads {t_inf * 1e3:.2f} ms ({t_inf / n * 1e6:.0f} us per frame amortised); "
          f"upload of {raw_mb:.0f} MB + ab_png_unfilter_batch {sorted(ts)[2]:.2f} ms on the stream")

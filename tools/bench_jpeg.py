"""Decode throughput of ab_jpeg_decode_batch on a batch of 640 x 480 frames (run on the GPU box): device time per batch (HIP events),
host time of the header parse + plan + pack, and Pillow's decode of the same files on the host for scale.
    python tools/bench_jpeg.py [--n 64] [--quality 92] [--subsampling 2] [--sub-bytes 128]"""
import argparse
import io
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _photo(w, h, seed):
    """A frame with the statistics of a photograph (as tests/test_gpu_jpeg.py): smooth shading, edges, texture and sensor noise."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([120 + 80 * np.sin(x / 37.0 + seed) * np.cos(y / 53.0), 110 + 70 * np.cos(x / 91.0 - y / 45.0), 100 + 60 * np.sin((x + y) / 67.0)], -1)
    for _ in range(12):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(10, 90)
        m = ((x - cx) ** 2 + (y - cy) ** 2) < r * r
        img[m] = img[m] * 0.4 + rng.uniform(0, 255, 3) * 0.6
    img += rng.normal(0, 6, img.shape) + 10 * np.sin(x / 2.1)[..., None] * (y[..., None] > h / 2)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--quality", type=int, default=92)
    ap.add_argument("--subsampling", type=int, default=2)
    ap.add_argument("--sub-bytes", type=int, nargs="*", default=[128])
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from PIL import Image
    from artiboost_amd.jpeg import JpegDecoder, parse
    files = []
    for i in range(a.n):
        b = io.BytesIO()
        Image.fromarray(_photo(640, 480, i)).save(b, "JPEG", quality=a.quality, subsampling=a.subsampling)
        files.append(b.getvalue())
    nbytes = sum(len(f) for f in files)
    out = torch.empty((a.n, 480, 640, 4), dtype=torch.uint8, device="cuda")
    t0 = time.perf_counter()
    for f in files:
        np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
    t_pil = (time.perf_counter() - t0) / a.n
    for sb in a.sub_bytes:
        dec = JpegDecoder("cuda", sub_bytes=sb)
        for _ in range(3):
            dec.decode(files, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        infos = [parse(f) for f in files]
        t_parse = time.perf_counter() - t0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.iters):
            dec.decode(files, out=out, infos=infos)
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.iters
        dev = e0.elapsed_time(e1) / a.iters
        rr = dec.last_rounds()
        print(f"    synchronisation rounds per image: min {rr.min()} mean {rr.mean():.1f} max {rr.max()}")
        print(f"sub_bytes {sb}: {a.n} frames, {nbytes / a.n / 1024:.1f} KB/file: wall {wall * 1e3:.2f} ms/batch (incl. plan + pack + upload), "
              f"stream {dev:.2f} ms/batch = {a.n / dev * 1e3:.0f} frames/s; header parse {t_parse / a.n * 1e6:.0f} us/file; "
              f"Pillow on this host {t_pil * 1e3:.2f} ms/file = {1 / t_pil:.0f} frames/s/core")


if __name__ == "__main__":
    main()

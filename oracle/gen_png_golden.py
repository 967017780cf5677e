"""TEST INFRASTRUCTURE -- writes tests/golden/png_cases.npz: PNG files (bytes) and the pixels Pillow's own decoder
(`Image.open(...).convert("RGB")`, the reference's call on HO3D v2's rgb/NNNN.png frames: anakin/datasets/ho3d.py:181,228-231) returns for
them.  Data only: the files and the expected arrays.  Files come from two writers: Pillow's (adaptive filter choice per line) and the small
writer below, which forces a filter type per line so that every one of the five reconstruction rules, every supported sample layout
(8 / 16 bit, RGB / RGBA, 8-bit grey) and several IDAT chunkings are present.  Run in the build container (Pillow 12.2.0; the reference pins
Pillow==8.0.1, requirements.txt:94 -- the same libImaging/ZipDecode.c reconstruction)."""
import io
import os
import struct
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))


def write_png(samples, depth, ctype, filters, idat_split=1, level=6, extra_chunks=()):
    """samples: uint8 [h][w * bpp] (the file's sample bytes, big-endian for 16 bit); filters: the filter type of each line."""
    h, stride = samples.shape
    bpp = {(8, 2): 3, (8, 6): 4, (16, 2): 6, (16, 6): 8, (8, 0): 1}[(depth, ctype)]
    w = stride // bpp
    lines = []
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        cur = samples[y].astype(np.int32)
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if stride > bpp else np.zeros(stride, np.int32)
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if stride > bpp else np.zeros(stride, np.int32)
        ft = int(filters[y % len(filters)])
        pred = [0, a, prev, (a + prev) >> 1, _paeth(a, prev, c)][ft]
        lines.append(bytes([ft]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
        prev = cur
    z = zlib.compress(b"".join(lines), level)
    cuts = [len(z) * k // idat_split for k in range(idat_split + 1)]
    body = b"".join(_chunk(b"IDAT", z[cuts[k]:cuts[k + 1]]) for k in range(idat_split))
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
            b"".join(_chunk(t, d) for t, d in extra_chunks) + body + _chunk(b"IEND", b""))


def picture(w, h, ch, kind, rng):
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(x / 9.0 + y / 17.0 + k) + 20 * k for k in range(ch)], -1)
    if kind == "noise":
        base = base + rng.normal(0, 30, (h, w, ch))
    if kind == "hard":
        base = rng.integers(0, 256, (h, w, ch))
    return np.clip(base, 0, 255).astype(np.uint8)


def cases(rng):
    out = []
    # Pillow's writer (what produced the dataset's files): RGB / RGBA / grey, several sizes and compression levels
    for (w, h, mode, kind, kw) in [(64, 48, "RGB", "noise", {}), (70, 50, "RGB", "smooth", {"compress_level": 9}), (33, 17, "RGB", "hard", {"compress_level": 1}),
                                   (1, 1, "RGB", "noise", {}), (2, 3, "RGB", "noise", {}), (160, 120, "RGB", "noise", {"optimize": True}),
                                   (64, 48, "RGBA", "noise", {}), (31, 9, "RGBA", "hard", {}), (64, 48, "L", "noise", {}), (7, 9, "L", "hard", {})]:
        ch = {"RGB": 3, "RGBA": 4, "L": 1}[mode]
        pic = picture(w, h, ch, kind, rng)
        b = io.BytesIO()
        Image.fromarray(pic[..., 0] if ch == 1 else pic, mode).save(b, "PNG", **kw)
        out.append(b.getvalue())
    # forced filters, every layout
    for (w, h, depth, ctype, kind, filters, split) in [
            (64, 48, 8, 2, "noise", [0], 1), (64, 48, 8, 2, "noise", [1], 1), (64, 48, 8, 2, "noise", [2], 1), (64, 48, 8, 2, "noise", [3], 1),
            (64, 48, 8, 2, "noise", [4], 1), (67, 70, 8, 2, "hard", [4, 3, 1, 2, 0, 4, 4, 3], 3), (1, 5, 8, 2, "hard", [4, 3, 1, 2, 0], 1),
            (5, 1, 8, 2, "hard", [4], 1), (130, 66, 8, 6, "hard", [3, 4, 1], 2), (40, 30, 16, 2, "hard", [4, 1, 3, 2, 0], 1),
            (41, 29, 16, 6, "hard", [3, 4], 5), (50, 20, 8, 0, "hard", [4, 3, 2, 1, 0], 1), (160, 120, 8, 2, "smooth", [4, 4, 3, 1], 4)]:
        bpp = {(8, 2): 3, (8, 6): 4, (16, 2): 6, (16, 6): 8, (8, 0): 1}[(depth, ctype)]
        samples = picture(w, h, bpp, kind, rng).reshape(h, w * bpp)
        extra = ((b"gAMA", struct.pack(">I", 45455)), (b"tEXt", b"Comment\x00forced filters")) if split > 1 else ()
        out.append(write_png(samples, depth, ctype, filters, idat_split=split, extra_chunks=extra))
    return out


def main():
    rng = np.random.default_rng(11)
    files = cases(rng)
    out = {"n": np.int64(len(files))}
    for i, data in enumerate(files):
        out[f"file{i}"] = np.frombuffer(data, np.uint8)
        out[f"rgb{i}"] = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "png_cases.npz"), **out)
    print(len(files), "cases,", sum(len(f) for f in files), "file bytes")


if __name__ == "__main__":
    main()

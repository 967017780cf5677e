"""PNG path of the real-data half, host side and oracle -- no GPU.  oracle/png_oracle.c (scanline reconstruction + Pillow's convert("RGB")
sample selection) is pinned against the committed files + Pillow pixels of tests/golden/png_cases.npz (written by oracle/gen_png_golden.py
from the real Pillow) and, where Pillow is importable, against Pillow itself on a live sweep.  artiboost_amd/png.py: the chunk walk, the
refusals and the pooled inflate."""
import io
import os
import struct
import zlib

import numpy as np
import pytest

import png_oracle as po
from artiboost_amd import png as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png_cases.npz")


def golden_cases():
    g = np.load(GOLD, allow_pickle=False)
    return [(bytes(g[f"file{i}"]), g[f"rgb{i}"]) for i in range(int(g["n"]))]


def test_oracle_matches_pillow_goldens():
    cases = golden_cases()
    assert len(cases) >= 20
    for i, (data, rgb) in enumerate(cases):
        got = po.decode(data)
        assert got.shape == rgb.shape, i
        np.testing.assert_array_equal(got, rgb, err_msg=f"case {i}")


def test_goldens_cover_every_filter_and_layout():
    seen_filters, seen_layouts = set(), set()
    for data, rgb in golden_cases():
        it = P.parse(data)
        assert (it.height, it.width) == rgb.shape[:2]
        raw = np.empty(it.raw_bytes, np.uint8)
        P.inflate_into(data, it, raw)
        seen_filters |= set(raw.reshape(it.height, -1)[:, 0].tolist())
        seen_layouts.add((it.bpp, it.chan))
    assert seen_filters == {0, 1, 2, 3, 4}
    assert seen_layouts == set(P.LAYOUT.values())


def test_oracle_matches_pillow_live_sweep():
    pytest.importorskip("PIL")
    from PIL import Image
    import gen_png_golden as G
    rng = np.random.default_rng(5)
    for (w, h) in [(1, 1), (2, 2), (3, 7), (17, 5), (64, 48), (129, 67), (640, 480)]:
        for mode, ch in (("RGB", 3), ("RGBA", 4), ("L", 1)):
            for kind in ("smooth", "noise", "hard"):
                if (w, h) == (640, 480) and (mode != "RGB" or kind != "noise"):
                    continue
                pic = G.picture(w, h, ch, kind, rng)
                b = io.BytesIO()
                Image.fromarray(pic[..., 0] if ch == 1 else pic, mode).save(b, "PNG", compress_level=int(rng.integers(1, 10)))
                ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))
                np.testing.assert_array_equal(po.decode(b.getvalue()), ref, err_msg=str((w, h, mode, kind)))
        for depth, ctype in ((8, 2), (8, 6), (16, 2), (16, 6), (8, 0)):                 # forced filters, random per line
            bpp = P.LAYOUT[(depth, ctype)][0]
            data = G.write_png(G.picture(w, h, bpp, "hard", rng).reshape(h, w * bpp), depth, ctype, rng.integers(0, 5, 16).tolist(),
                               idat_split=int(rng.integers(1, 4)))
            ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
            np.testing.assert_array_equal(po.decode(data), ref, err_msg=str((w, h, depth, ctype)))


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)


def _file(w, h, depth, ctype, lace=0, payload=None):
    bpp = max(1, {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype] * depth // 8)
    raw = payload if payload is not None else zlib.compress(bytes(h * (1 + w * bpp)))
    return P.SIGNATURE + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, lace)) + _chunk(b"IDAT", raw) + _chunk(b"IEND", b"")


def test_parse_refuses_what_the_device_path_does_not_cover():
    for bad in (b"", b"\xff\xd8\xff\xe0" + bytes(64), P.SIGNATURE + bytes(10), _file(4, 4, 8, 3), _file(4, 4, 8, 4), _file(4, 4, 16, 0),
                _file(4, 4, 4, 0), _file(4, 4, 8, 2, lace=1), golden_cases()[0][0][:60]):
        with pytest.raises(P.PngUnsupported):
            P.parse(bad)
    it = P.parse(_file(4, 3, 8, 2))
    assert (it.width, it.height, it.bpp, it.chan, it.raw_bytes) == (4, 3, 3, (0, 1, 2), 3 * 13)
    dst = np.empty(it.raw_bytes, np.uint8)
    with pytest.raises(P.PngUnsupported):                 # corrupt stream
        P.inflate_into(_file(4, 3, 8, 2, payload=b"not zlib"), it, dst)
    with pytest.raises(P.PngUnsupported):                 # stream of another size than the header promises
        P.inflate_into(_file(4, 3, 8, 2, payload=zlib.compress(bytes(7))), it, dst)


def test_pooled_inflate_equals_serial():
    cases = golden_cases()
    infos = [P.parse(d) for d, _ in cases]
    outs = [np.empty(it.raw_bytes, np.uint8) for it in infos]
    list(P.pool().map(lambda a: P.inflate_into(a[0][0], a[1], a[2]), zip(cases, infos, outs)))
    for (d, _), it, o in zip(cases, infos, outs):
        ref = np.empty(it.raw_bytes, np.uint8)
        P.inflate_into(d, it, ref)
        np.testing.assert_array_equal(o, ref)


def test_filter_type_above_four_and_empty_idat_chunks():
    """Advisor finding (round 4): a scanline whose filter byte is above 4 is refused on the host (Pillow's ZipDecode raises there; the device
    kernel's output for it would be undefined), and a legal zero-length IDAT chunk does not break the ctypes inflate path."""
    w, h = 9, 12
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)
    rng = np.random.default_rng(7)
    lines = rng.integers(0, 256, (h, 1 + 3 * w), dtype=np.uint8)
    lines[:, 0] = rng.integers(0, 5, h)                                   # legal filter types
    good = P.SIGNATURE + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(lines.tobytes())) + _chunk(b"IEND", b"")
    info = P.parse(good)
    raw = np.empty(info.raw_bytes, np.uint8)
    P.inflate_into(good, info, raw)                                       # a good stream passes
    np.testing.assert_array_equal(raw.reshape(h, -1), lines)
    # (i) scanline 5's filter byte set to 7
    bad_lines = lines.copy()
    bad_lines[5, 0] = 7
    bad = P.SIGNATURE + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(bad_lines.tobytes())) + _chunk(b"IEND", b"")
    with pytest.raises(P.PngUnsupported):
        P.inflate_into(bad, P.parse(bad), np.empty(info.raw_bytes, np.uint8))
    # (ii) the good stream cut into IDAT chunks with empty ones in between
    z = zlib.compress(lines.tobytes())
    cut = len(z) // 2
    multi = (P.SIGNATURE + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", b"") + _chunk(b"IDAT", z[:cut]) + _chunk(b"IDAT", b"") +
             _chunk(b"IDAT", z[cut:]) + _chunk(b"IDAT", b"") + _chunk(b"IEND", b""))
    out = np.empty(info.raw_bytes, np.uint8)
    P.inflate_into(multi, P.parse(multi), out)
    np.testing.assert_array_equal(out, raw)

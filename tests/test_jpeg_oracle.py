"""oracle/jpeg_oracle.c (the sequential restatement of Pillow / libjpeg-turbo's default JPEG decode) pinned: against the committed files +
Pillow pixels of tests/golden/jpeg_cases.npz, and -- where Pillow is importable (the build container and the GPU image) -- against Pillow
itself on a sweep of sizes, samplings, qualities, custom Huffman tables and restart intervals.  Bit-exact."""
import io
import os

import numpy as np
import pytest

import jpeg_oracle as jo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_cases.npz")


def golden_cases():
    g = np.load(GOLD, allow_pickle=False)
    return [(bytes(g[f"file{i}"]), g[f"rgb{i}"]) for i in range(int(g["n"]))]


def test_oracle_matches_pillow_goldens():
    for i, (data, rgb) in enumerate(golden_cases()):
        got = jo.decode(data)
        assert got.shape == rgb.shape, i
        np.testing.assert_array_equal(got, rgb, err_msg=f"case {i}")


def test_oracle_matches_pillow_live_sweep():
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageFile
    import gen_jpeg_golden as G
    ImageFile.MAXBLOCK = 1 << 24
    rng = np.random.default_rng(11)
    n = 0
    for (w, h) in [(64, 48), (70, 50), (33, 17), (16, 16), (8, 8), (1, 1), (2, 3), (3, 5), (127, 129), (17, 33), (640, 480)]:
        for sub in (0, 1, 2):
            for q in (30, 75, 92, 100):
                for kind in ("smooth", "noise", "hard"):
                    for extra in ({}, {"optimize": True}, {"restart_marker_blocks": 3}, {"restart_marker_rows": 1}):
                        if (w, h) == (640, 480) and (kind != "noise" or q != 92 or extra):
                            continue
                        if (n % 3) and (w, h) != (640, 480):           # a third of the grid: the whole one (1 600 files) passes too, in a minute
                            n += 1
                            continue
                        n += 1
                        b = io.BytesIO()
                        Image.fromarray(G.picture(w, h, kind, rng)).save(b, "JPEG", quality=q, subsampling=sub, **extra)
                        ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("RGB"))
                        np.testing.assert_array_equal(jo.decode(b.getvalue()), ref, err_msg=str((w, h, sub, q, kind, extra)))


def test_oracle_refuses_what_it_does_not_restate():
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(b, "JPEG", progressive=True)
    with pytest.raises(ValueError):
        jo.decode(b.getvalue())
    with pytest.raises(ValueError):
        jo.decode(b"\x89PNG\r\n\x1a\n" + bytes(32))

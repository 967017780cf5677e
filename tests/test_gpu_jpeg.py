"""ab_jpeg_decode_batch (csrc/jpeg.hip: self-synchronising parallel Huffman decode, integer IDCT, fancy up-sampling, YCbCr -> RGB on the
device) against the files + Pillow pixels of tests/golden/jpeg_cases.npz and against oracle/jpeg_oracle.c (itself pinned to Pillow):
BIT-EXACT.  The reference call it stands in for: Image.open(path).convert("RGB"), anakin/datasets/ho3d.py:228-231."""
import io
import os

import numpy as np
import pytest
import torch

import jpeg_oracle as jo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_cases.npz")


def _golden():
    g = np.load(GOLD, allow_pickle=False)
    return [(bytes(g[f"file{i}"]), g[f"rgb{i}"]) for i in range(int(g["n"]))]


@pytest.mark.parametrize("sub_bytes", [16, 36, 64, 128])
@pytest.mark.parametrize("channels", [3, 4])
def test_jpeg_goldens_bit_exact(sub_bytes, channels):
    """Every golden file (4:4:4 / 4:2:2 / 4:2:0 / grey, custom Huffman tables, restart intervals down to one MCU, 1 x 1 to 160 x 120) in ONE
    ragged batch, for several subsequence lengths (16 bytes: chains cross many subsequences and more rounds than the launched ones, finished in
    jpeg_finish_kernel; 128: one thread per small file)."""
    from artiboost_amd.jpeg import JpegDecoder
    cases = _golden()
    dec = JpegDecoder("cuda", sub_bytes=sub_bytes)
    outs = dec.decode([c[0] for c in cases], channels=channels)
    torch.cuda.synchronize()
    for i, ((_, rgb), o) in enumerate(zip(cases, outs)):
        got = o.cpu().numpy()
        assert got.shape == rgb.shape[:2] + (channels,), i
        np.testing.assert_array_equal(got[..., :3], rgb, err_msg=f"case {i}")
        if channels == 4:
            assert not got[..., 3].any()


def test_jpeg_long_codes_by_canonical_search():
    """AB_JPEG_NO_LUT2=1 (read once per process: run in a child): codes longer than the first look-up level go through the canonical
    bit-by-bit search the kernels keep for tables whose long codes do not fit the second level."""
    import subprocess
    import sys
    code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from artiboost_amd.jpeg import JpegDecoder; g = np.load(%r); "
            "n = int(g['n']); outs = JpegDecoder('cuda').decode([bytes(g[f'file{i}']) for i in range(n)], channels=3); "
            "assert all(np.array_equal(o.cpu().numpy(), g[f'rgb{i}']) for i, o in enumerate(outs)); print('same')"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), GOLD))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AB_JPEG_NO_LUT2="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "same" in r.stdout, r.stderr[-2000:]


def _photo(w, h, seed):
    """A frame with the statistics of a photograph: smooth shading, edges, texture and sensor noise."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([120 + 80 * np.sin(x / 37.0 + seed) * np.cos(y / 53.0), 110 + 70 * np.cos(x / 91.0 - y / 45.0), 100 + 60 * np.sin((x + y) / 67.0)], -1)
    for _ in range(12):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(10, 90)
        m = ((x - cx) ** 2 + (y - cy) ** 2) < r * r
        img[m] = img[m] * 0.4 + rng.uniform(0, 255, 3) * 0.6
    img += rng.normal(0, 6, img.shape) + 10 * np.sin(x / 2.1)[..., None] * (y[..., None] > h / 2)
    return np.clip(img, 0, 255).astype(np.uint8)


def test_jpeg_full_size_batch_vs_oracle_and_pillow():
    """64 frames of 640 x 480 (HO3D / DexYCB frame size, ho3d.py:40) written by Pillow at the qualities / samplings cameras and OpenCV use,
    decoded in one batch into the RGBX frame tensor RealBatcher hands to ab_augment_batch: bit-exact vs the C oracle for every frame and vs
    Pillow (when importable) for the first eight."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    from artiboost_amd.jpeg import JpegDecoder
    files = []
    for i in range(64):
        b = io.BytesIO()
        Image.fromarray(_photo(640, 480, i)).save(b, "JPEG", quality=(75, 85, 92, 95)[i % 4], subsampling=(2, 2, 0, 1)[(i // 4) % 4],
                                                  **({"restart_marker_rows": 2} if i % 16 == 5 else {}))
        files.append(b.getvalue())
    out = torch.full((64, 480, 640, 4), 7, dtype=torch.uint8, device="cuda")
    dec = JpegDecoder("cuda")
    for _ in range(2):                              # second call: reused staging blobs / workspace
        dec.decode(files, out=out)
    got = out.cpu().numpy()
    for i, f in enumerate(files):
        np.testing.assert_array_equal(got[i, ..., :3], jo.decode(f), err_msg=f"frame {i}")
        if i < 8:
            np.testing.assert_array_equal(got[i, ..., :3], np.asarray(Image.open(io.BytesIO(f)).convert("RGB")))
    assert not got[..., 3].any()


def test_jpeg_large_and_odd_shapes_vs_oracle():
    """1920 x 1080 (several thousand subsequences per image, more than one workgroup of them), a 3000 x 17 strip, a 9 x 2000 column and a
    grey 1001 x 999 frame in one ragged batch: bit-exact vs the C oracle."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    from artiboost_amd.jpeg import JpegDecoder
    rng = np.random.default_rng(3)
    files = []
    for (w, h, sub, q, grey) in [(1920, 1080, 2, 90, False), (3000, 17, 1, 85, False), (9, 2000, 2, 95, False), (1001, 999, 0, 80, True),
                                 (1920, 1080, 0, 97, False)]:
        img = _photo(w, h, w + h)
        img = np.clip(img.astype(np.int32) + rng.integers(-20, 20, img.shape), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        if grey:
            Image.fromarray(img[..., 0]).save(b, "JPEG", quality=q)
        else:
            Image.fromarray(img).save(b, "JPEG", quality=q, subsampling=sub)
        files.append(b.getvalue())
    outs = JpegDecoder("cuda").decode(files, channels=3)
    for i, (f, o) in enumerate(zip(files, outs)):
        np.testing.assert_array_equal(o.cpu().numpy(), jo.decode(f), err_msg=f"file {i}")


def test_jpeg_random_sweep_vs_oracle():
    """120 files of random size (1 .. 230 pixels a side), content (smooth / photo-like / noise / saturated), quality 5 .. 100, sampling,
    Huffman tables (standard / optimised) and restart interval, decoded in two ragged batches with different subsequence lengths: every
    file bit-exact vs the C oracle (which is pinned to Pillow)."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageFile
    from artiboost_amd.jpeg import JpegDecoder
    ImageFile.MAXBLOCK = 1 << 24
    rng = np.random.default_rng(2024)
    files = []
    for i in range(120):
        w, h = int(rng.integers(1, 231)), int(rng.integers(1, 231))
        kind = i % 4
        if kind == 0:
            img = _photo(w, h, i)
        elif kind == 1:
            img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        elif kind == 2:
            img = np.clip(_photo(w, h, i).astype(np.int32) * 3 - 200, 0, 255).astype(np.uint8)          # saturated: range-limit paths
        else:
            y, x = np.mgrid[0:h, 0:w]
            img = np.stack([(x * 5) % 256, (y * 7) % 256, ((x + y) * 3) % 256], -1).astype(np.uint8)
        kw = {}
        r = int(rng.integers(0, 4))
        if r == 1:
            kw["optimize"] = True
        elif r == 2:
            kw["restart_marker_blocks"] = int(rng.integers(1, 9))
        elif r == 3:
            kw["restart_marker_rows"] = int(rng.integers(1, 4))
        b = io.BytesIO()
        if i % 11 == 0:
            Image.fromarray(img[..., 0]).save(b, "JPEG", quality=int(rng.integers(5, 101)), **kw)
        else:
            Image.fromarray(img).save(b, "JPEG", quality=int(rng.integers(5, 101)), subsampling=int(rng.integers(0, 3)), **kw)
        files.append(b.getvalue())
    for part, sb in ((files[:60], 128), (files[60:], 48)):
        outs = JpegDecoder("cuda", sub_bytes=sb).decode(part, channels=3)
        for i, (f, o) in enumerate(zip(part, outs)):
            np.testing.assert_array_equal(o.cpu().numpy(), jo.decode(f), err_msg=f"file {i} (sub_bytes {sb})")


def test_jpeg_refuses_unsupported_before_device_work():
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    from artiboost_amd.jpeg import JpegDecoder, JpegUnsupported
    b = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(b, "JPEG", progressive=True)
    with pytest.raises(JpegUnsupported):
        JpegDecoder("cuda").decode([b.getvalue()])

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


@pytest.mark.parametrize("sub_bytes", [16, 64, 128, 1024])
@pytest.mark.parametrize("channels", [3, 4])
def test_jpeg_goldens_bit_exact(sub_bytes, channels):
    """Every golden file (4:4:4 / 4:2:2 / 4:2:0 / grey, custom Huffman tables, restart intervals down to one MCU, 1 x 1 to 160 x 120) in ONE
    ragged batch, for several subsequence lengths (16 bytes: chains cross many subsequences; 1024: one thread per small file)."""
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


def test_jpeg_refuses_unsupported_before_device_work():
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    from artiboost_amd.jpeg import JpegDecoder, JpegUnsupported
    b = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(b, "JPEG", progressive=True)
    with pytest.raises(JpegUnsupported):
        JpegDecoder("cuda").decode([b.getvalue()])

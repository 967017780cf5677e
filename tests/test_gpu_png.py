"""ab_png_unfilter_batch (csrc/png.hip: skewed-wavefront scanline reconstruction + RGB(X) packing on the device, zlib inflate on the host
pool) against the files + Pillow pixels of tests/golden/png_cases.npz and against oracle/png_oracle.c (itself pinned to Pillow): BIT-EXACT.
The reference call it stands in for: Image.open(path).convert("RGB") on HO3D v2's rgb/NNNN.png, anakin/datasets/ho3d.py:181,228-231."""
import io
import os

import numpy as np
import pytest
import torch

import png_oracle as po

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png_cases.npz")



def _same_entry(a, b, msg):
    """Batch entries are device tensors -- bit-identical -- or plain tags (the IMAGE_PLANE_KEY string)."""
    if torch.is_tensor(a):
        torch.testing.assert_close(a, b, rtol=0, atol=0, msg=str(msg))
    else:
        assert a == b, msg


def _golden():
    g = np.load(GOLD, allow_pickle=False)
    return [(bytes(g[f"file{i}"]), g[f"rgb{i}"]) for i in range(int(g["n"]))]


@pytest.mark.parametrize("channels", [3, 4])
def test_png_goldens_bit_exact(channels):
    """Every golden file (all five filter types, 8 / 16-bit RGB and RGBA, 8-bit grey, 1 x 1 to 160 x 120, split IDAT streams, ancillary chunks)
    in ONE ragged batch."""
    from artiboost_amd.png import PngDecoder
    cases = _golden()
    outs = PngDecoder("cuda").decode([c[0] for c in cases], channels=channels)
    torch.cuda.synchronize()
    for i, ((_, rgb), o) in enumerate(zip(cases, outs)):
        got = o.cpu().numpy()
        assert got.shape == rgb.shape[:2] + (channels,), i
        np.testing.assert_array_equal(got[..., :3], rgb, err_msg=f"case {i}")
        if channels == 4:
            assert not got[..., 3].any()


def test_png_full_frames_vs_oracle_and_pillow():
    """640 x 480 frames (HO3D's size: 7.5 bands of 64 lines per wave), Pillow-written and with random forced filters, 8-bit RGB and 16-bit RGBA:
    device == C oracle == live Pillow; frames land in a preallocated [n, H, W, 4] tensor (the augmentation chain's input)."""
    pytest.importorskip("PIL")
    from PIL import Image
    import gen_png_golden as G
    from artiboost_amd.png import PngDecoder
    rng = np.random.default_rng(3)
    files = []
    for i in range(3):
        b = io.BytesIO()
        Image.fromarray(G.picture(640, 480, 3, ("smooth", "noise", "hard")[i], rng)).save(b, "PNG")
        files.append(b.getvalue())
    files.append(G.write_png(G.picture(640, 480, 3, "hard", rng).reshape(480, -1), 8, 2, rng.integers(0, 5, 64).tolist(), idat_split=7))
    files.append(G.write_png(G.picture(640, 480, 8, "hard", rng).reshape(480, -1), 16, 6, rng.integers(0, 5, 64).tolist()))
    out = torch.zeros((len(files) + 1, 480, 640, 4), dtype=torch.uint8, device="cuda")
    res = PngDecoder("cuda").decode(files, out=out)
    assert res.shape[0] == len(files) and not out[len(files)].any()
    for i, f in enumerate(files):
        got = res[i].cpu().numpy()
        np.testing.assert_array_equal(got[..., :3], po.decode(f), err_msg=f"file {i} vs oracle")
        np.testing.assert_array_equal(got[..., :3], np.asarray(Image.open(io.BytesIO(f)).convert("RGB")), err_msg=f"file {i} vs Pillow")


def test_png_refusals_and_corrupt_streams():
    from artiboost_amd.png import PngDecoder, PngUnsupported
    dec = PngDecoder("cuda")
    good = _golden()[0][0]
    with pytest.raises(PngUnsupported):
        dec.decode([good, b"\xff\xd8\xff\xe0 not a png"])
    bad = bytearray(good)
    i = bad.index(b"IDAT") + 4
    bad[i + 8:i + 24] = bytes(16)                                   # the deflate stream no longer decodes
    with pytest.raises(PngUnsupported):
        dec.decode([good, bytes(bad)])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dec.decode([good], channels=3)[0].cpu().numpy(), _golden()[0][1])      # the decoder is still usable
    with pytest.raises(ValueError):
        dec.decode([good], out=torch.empty((1, 5, 5, 4), dtype=torch.uint8, device="cuda"))


class _PngSource:
    """The golden real-data frames served as .png files (index modulo 3 for the annotations, twelve distinct images)."""
    sides = "right"

    def __init__(self, serve_bytes=True):
        from PIL import Image
        from test_realdata import GoldenSource
        self.base = GoldenSource()
        self.raw_size, self.n = self.base.raw_size, self.base.n
        self.files = []
        for i in range(self.n):
            b = io.BytesIO()
            Image.fromarray(np.roll(self.base.g["frames"][i % 3], 5 * i, axis=1)).save(b, "PNG")
            self.files.append(b.getvalue())
        if not serve_bytes:
            self.get_image_bytes = None

    def __len__(self):
        return self.n

    def get_image(self, idx):
        from PIL import Image
        return np.asarray(Image.open(io.BytesIO(self.files[idx])).convert("RGB"))

    def get_image_bytes(self, idx):
        return self.files[idx]

    def get_annots(self, idx):
        return self.base.get_annots(idx)


def test_real_batches_from_png_files_equal_the_pillow_path():
    """A source that serves .png FILES gives bit-identical batches (augmented images and ground truth) to the same source decoded with
    Pillow on the decode pool (get_image, the reference's call), batch by batch and through MixedLoader's grouped decode-ahead schedule with
    the inflates started a group early."""
    pytest.importorskip("PIL")
    from artiboost_amd.realdata import MixedLoader, RealBatcher
    cfg = {"IMAGE_SIZE": [64, 64], "CENTER_IDX": 0, "BBOX_EXPAND_RATIO": 1.2}
    a, b = _PngSource(), _PngSource(serve_bytes=False)
    ra, rb = RealBatcher(a, cfg, compute_dtype=torch.float32, seed=5), RealBatcher(b, cfg, compute_dtype=torch.float32, seed=5, num_workers=4)
    ha = ra.assemble([0, 1, 2, 7])
    hb = rb.assemble([0, 1, 2, 7])
    assert ha["file_kind"] == "png" and ha["frames"] is None and hb["files"] is None and hb["frames"] is not None
    ra.rng, rb.rng = np.random.default_rng(5), np.random.default_rng(5)
    for idxs in ([0, 1, 2, 7], [3, 11]):
        ba, bb = ra.batch(idxs), rb.batch(idxs)
        for k in ba:
            _same_entry(ba[k], bb[k], k)
    for seed in (1, 2):
        ref = [{k: v.clone() for k, v in x.items()} for x in
               MixedLoader(RealBatcher(b, cfg, compute_dtype=torch.float32, seed=9), None, 2, seed=seed, decode_group=1, decode_ahead=False)]
        ml = MixedLoader(RealBatcher(a, cfg, compute_dtype=torch.float32, seed=9), None, 2, seed=seed, decode_group=2, decode_ahead=True)
        got = [{k: v.clone() for k, v in x.items()} for x in ml]
        torch.cuda.synchronize()
        assert len(got) == len(ref) == 6 and ml.real._png_side is not None and not ml.real._jobs
        for x, r in zip(got, ref):
            for k in r:
                _same_entry(x[k], r[k], k)

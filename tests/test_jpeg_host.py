"""Host side of the device JPEG decoder (artiboost_amd/jpeg.py): the marker walk and the batch plan -- no GPU."""
import io
import os

import numpy as np
import pytest

from artiboost_amd import jpeg as J

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_cases.npz")


def _files():
    g = np.load(GOLD, allow_pickle=False)
    return [(bytes(g[f"file{i}"]), g[f"rgb{i}"]) for i in range(int(g["n"]))]


def test_parse_reads_frame_tables_and_scan_extent():
    for data, rgb in _files():
        it = J.parse(data)
        assert (it.height, it.width) == rgb.shape[:2]
        assert it.ncomp in (1, 3) and it.bpm == sum(c[1] * c[2] for c in it.comps)
        assert it.mcux == -(-it.width // (8 * it.hmax)) and it.mcuy == -(-it.height // (8 * it.vmax))
        assert it.qt.shape == (4, 64) and it.ht.shape == (8, 272)
        assert data[it.scan_off - 1] == 0 and data[it.scan_off + it.scan_len:it.scan_off + it.scan_len + 2] == b"\xff\xd9"
        nseg = 1 if not it.ri else -(-it.mcux * it.mcuy // it.ri)
        assert len(it.segs) == nseg and it.segs[0, 0] == 0
        if it.ri:                                    # every restart interval is followed by its RSTn marker
            for k, (o, ln) in enumerate(it.segs[:-1]):
                m = data[it.scan_off + o + ln:it.scan_off + o + ln + 2]
                assert m[0] == 0xFF and m[1] == 0xD0 + (k & 7)


def test_parse_refuses_truncated_and_foreign_files():
    data = _files()[0][0]
    for cut in (1, 3, 20, 100, 300):
        with pytest.raises(J.JpegUnsupported):
            J.parse(data[:cut])
    with pytest.raises(J.JpegUnsupported):
        J.parse(b"\x89PNG\r\n\x1a\n" + bytes(64))
    with pytest.raises(J.JpegUnsupported):
        J.parse(b"")
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    for kw in ({"progressive": True}, ):
        b = io.BytesIO()
        Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(b, "JPEG", **kw)
        with pytest.raises(J.JpegUnsupported):
            J.parse(b.getvalue())
    b = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 4), np.uint8), "CMYK").save(b, "JPEG")
    with pytest.raises(J.JpegUnsupported):
        J.parse(b.getvalue())


def test_plan_lays_out_one_blob():
    files = [f for f, _ in _files()]
    infos = [J.parse(f) for f in files]
    sizes = [it.width * it.height for it in infos]
    plan = J._Plan(files, infos, 128, list(np.cumsum([0] + sizes[:-1])), [it.width for it in infos])
    d = plan.desc
    assert d.shape == (len(files), J.DESC_INTS) and (d[:, 26] >= d[:, 24]).all()              # >= one subsequence per restart interval
    assert plan.total_sub == d[:, 26].sum() and plan.total_blocks == d[:, 28].sum()
    assert (np.diff(d[:, 25]) == d[:-1, 26]).all() and (np.diff(d[:, 27]) == d[:-1, 28]).all()
    assert len(plan.ht) <= len(files) and d[:, 36].max() == len(plan.ht) - 1                  # Huffman table sets shared between files
    blob = np.zeros(plan.blob_bytes(), np.uint8)
    offs, used = plan.pack(files, blob)
    assert used == plan.blob_bytes() and all(o % 256 == 0 for o in offs)
    for i, f in enumerate(files):                    # each descriptor's scan offset points at its file's entropy-coded bytes
        so = infos[i].scan_off
        assert bytes(blob[offs[0] + d[i, 0]:offs[0] + d[i, 0] + 8]) == f[so:so + 8]
    with pytest.raises(ValueError):
        J.JpegDecoder.__init__(J.JpegDecoder.__new__(J.JpegDecoder), "cpu", sub_bytes=512)


def test_scan_ends_at_the_first_eoi_not_the_last():
    """A file with data appended behind its EOI -- an MPF second image, a preview -- that carries an EOI of its own: the entropy segment
    handed to the device must end at the FIRST image's EOI (the extent used to be rfind(FFD9) over the whole file)."""
    for data, _ in _files():
        it = J.parse(data)
        tail = b"\xff\xd8\xff\xe0\x00\x04\x00\x00" + bytes(range(1, 200)) + b"\xff\xd9" + b"trailing junk"
        it2 = J.parse(data + tail)
        assert (it2.scan_off, it2.scan_len) == (it.scan_off, it.scan_len)
        np.testing.assert_array_equal(it2.segs, it.segs)

"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of oracle/png_oracle.c (sequential CPU restatement of the PNG scanline reconstruction +
Pillow's `convert("RGB")` sample selection: what the reference's `Image.open(path).convert("RGB")` returns for a .png frame,
ho3d.py:181,228-231).  The container walk and the inflate (zlib, as in Pillow) are done here in Python."""
import ctypes
import os
import struct
import subprocess
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libpng_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "png_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = ctypes.CDLL(SO)
        _lib.png_oracle_unfilter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p]
    return _lib


# (bit depth, colour type) -> (bytes per pixel, byte offsets of R, G, B)
LAYOUT = {(8, 2): (3, (0, 1, 2)), (8, 6): (4, (0, 1, 2)), (16, 2): (6, (0, 2, 4)), (16, 6): (8, (0, 2, 4)), (8, 0): (1, (0, 0, 0))}


def decode(data: bytes):
    """bytes of a .png -> uint8 [H, W, 3] RGB.  ValueError for files outside LAYOUT / interlaced / malformed."""
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG")
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + n
    if hdr is None or not idat:
        raise ValueError("no IHDR / IDAT")
    w, h, depth, ctype, comp, flt, lace = hdr
    if (depth, ctype) not in LAYOUT or comp or flt or lace:
        raise ValueError("unsupported PNG")
    bpp, (c0, c1, c2) = LAYOUT[(depth, ctype)]
    raw = zlib.decompress(b"".join(idat))
    if len(raw) != h * (1 + w * bpp):
        raise ValueError("scanline bytes")
    out = np.empty((h, w, 3), np.uint8)
    rawa = np.frombuffer(raw, np.uint8)
    rc = lib().png_oracle_unfilter(rawa.ctypes.data, w, h, bpp, c0, c1, c2, out.ctypes.data)
    if rc:
        raise ValueError(rc)
    return out

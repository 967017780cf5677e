"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of oracle/jpeg_oracle.c (sequential CPU restatement of Pillow / libjpeg-turbo's default
JPEG decode: what the reference's `Image.open(path).convert("RGB")` returns, ho3d.py:228-231 / dexycb.py:226-229)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libjpeg_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "jpeg_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = ctypes.CDLL(SO)
        _lib.jpeg_oracle_decode.argtypes = [ctypes.c_char_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.POINTER(ctypes.c_int),
                                            ctypes.POINTER(ctypes.c_int)]
    return _lib


def decode(data: bytes, max_pixels=1 << 24):
    """bytes of a .jpg -> uint8 [H, W, 3] RGB.  Raises ValueError(code) for malformed (-1) / unsupported (-2) files."""
    out = np.empty(max_pixels * 3, np.uint8)
    w, h = ctypes.c_int(0), ctypes.c_int(0)
    rc = lib().jpeg_oracle_decode(data, len(data), out.ctypes.data, out.size, ctypes.byref(w), ctypes.byref(h))
    if rc:
        raise ValueError(rc)
    return out[:h.value * w.value * 3].reshape(h.value, w.value, 3).copy()

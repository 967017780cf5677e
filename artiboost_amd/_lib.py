"""ctypes binding of libartiboost_hip.so (the C ABI declared in include/artiboost_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is raised."""
import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# ARTIBOOST_HIP_LIB selects another build of the same library (A/B timing of kernel changes); still no CPU fallback
LIB_PATH = os.environ.get("ARTIBOOST_HIP_LIB") or os.path.join(HERE, "libartiboost_hip.so")
HEADER = os.path.join(HERE, "..", "include", "artiboost_hip.h")

_lib = None

DT_F32, DT_BF16 = 0, 1
_DT = {torch.float32: DT_F32, torch.bfloat16: DT_BF16}


def declared_symbols():
    """Names of every function declared in include/artiboost_hip.h."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(n for _, n in re.findall(r"\b(int|long|void|int64_t|size_t)\s+(ab_\w+)\s*\(", txt)))


def _long_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\blong\s+(ab_\w+)\s*\(", txt))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -m artiboost_amd.build` (there is no CPU fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        longs = _long_symbols()
        for name in declared_symbols():
            fn = getattr(_lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = ctypes.c_long if name in longs else ctypes.c_int
    return _lib


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype} (float32 or bfloat16 expected)")


def ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("artiboost_hip ops need device tensors (HIP); got a CPU tensor")
    if not t.is_contiguous():
        raise RuntimeError("artiboost_hip ops need contiguous tensors")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}" + (" (hipError)" if rc > 0 else " (argument error)"))


def f(x):
    return ctypes.c_float(float(x))


def i(x):
    return ctypes.c_int(int(x))


def l(x):
    return ctypes.c_long(int(x))


class WgradReduceDesc(ctypes.Structure):
    """struct ab_wgrad_reduce_desc (include/artiboost_hip.h)."""
    _fields_ = [("slabs", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("slab_elems", ctypes.c_long), ("nslices", ctypes.c_int),
                ("src_j", ctypes.c_int), ("dst_j", ctypes.c_int), ("accumulate", ctypes.c_int), ("stem_mask", ctypes.c_int)]


class SymCorner(ctypes.Structure):
    """struct ab_symcorner (include/artiboost_hip.h)."""
    _fields_ = [("R", ctypes.c_void_p), ("t", ctypes.c_void_p), ("K", ctypes.c_int32), ("obj_idx", ctypes.c_void_p),
                ("obj_transf", ctypes.c_void_p), ("lam", ctypes.c_float), ("weight", ctypes.c_float),
                ("loss_out", ctypes.c_void_p)]

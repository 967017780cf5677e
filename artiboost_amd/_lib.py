"""Binding of libartiboost_hip.so (the C ABI declared in include/artiboost_hip.h).

Default: through the PyTorch dispatcher -- `lib().ab_xxx(...)` calls `torch.ops.artiboost_hip.xxx` (libartiboost_torch.so: one
TORCH_LIBRARY registration per entry point, generated from the header by gen_torch_ops.py; tensors in, current HIP stream inside).
AB_BINDING=ctypes calls the same entry points through ctypes instead (A/B of the binding cost; same library, same kernels).

The product path has NO fallback: if a library is missing or a call fails, a RuntimeError is raised."""
import ctypes
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# ARTIBOOST_HIP_LIB selects another build of the same library (A/B timing of kernel changes); still no CPU fallback
LIB_PATH = os.environ.get("ARTIBOOST_HIP_LIB") or os.path.join(HERE, "libartiboost_hip.so")
TORCH_LIB_PATH = os.path.join(HERE, "libartiboost_torch.so")
HEADER = os.path.join(HERE, "..", "include", "artiboost_hip.h")
BINDING = os.environ.get("AB_BINDING", "ctypes" if os.environ.get("ARTIBOOST_HIP_LIB") else "torch")

_lib = None
_cdll = None

DT_F32, DT_BF16 = 0, 1
_DT = {torch.float32: DT_F32, torch.bfloat16: DT_BF16}
_STREAM = object()          # placeholder for the trailing `void* stream` argument (see stream())


def declared_symbols():
    """Names of every function declared in include/artiboost_hip.h."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(n for _, n in re.findall(r"\b(int|long|void|int64_t|size_t)\s+(ab_\w+)\s*\(", txt)))


def _long_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\blong\s+(ab_\w+)\s*\(", txt))


def cdll():
    """The raw shared library (symbol checks, the ctypes binding)."""
    global _cdll
    if _cdll is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -m artiboost_amd.build` (there is no CPU fallback)")
        _cdll = ctypes.CDLL(LIB_PATH)
        longs = _long_symbols()
        for name in declared_symbols():
            fn = getattr(_cdll, name)  # AttributeError if a declared symbol is not exported
            fn.restype = ctypes.c_long if name in longs else ctypes.c_int
    return _cdll


def _host_bytes(a):
    """ctypes Structure / array / byref(...) -> CPU uint8 tensor over the SAME memory (host structs of the C ABI)."""
    if hasattr(a, "_obj"):
        a = a._obj
    return torch.frombuffer(a, dtype=torch.uint8)


class _TorchOps:
    """lib().ab_xxx(args..., stream()) -> torch.ops.artiboost_hip.xxx(args...)."""

    def __init__(self):
        if not os.path.exists(TORCH_LIB_PATH):
            raise RuntimeError(f"{TORCH_LIB_PATH} is missing: run `python -m artiboost_amd.build` (there is no CPU fallback)")
        cdll()                                   # fail loudly on a missing / incomplete libartiboost_hip.so first
        torch.ops.load_library(TORCH_LIB_PATH)
        self._ns = torch.ops.artiboost_hip
        from . import gen_torch_ops
        self._sig = {name: params for _, name, params in gen_torch_ops.declarations()}

    def __getattr__(self, name):
        op = getattr(self._ns, name[3:]).default            # the OpOverload: no overload resolution per call
        params = self._sig[name]
        has_stream = bool(params) and params[-1] == ("void*", "stream")
        # positions that may carry a HOST object of the C ABI (ctypes struct / array / byref): passed on as CPU byte tensors
        # (the header names host pointers `*_host`; struct pointers are host structs, or device tables that arrive as tensors anyway)
        host = tuple(i for i, (ty, pn) in enumerate(params) if "ab_" in ty or pn.endswith("_host"))
        Tensor = torch.Tensor
        if not host:
            if has_stream:
                def call(*args):
                    return op(*args[:-1]) or 0
            else:
                def call(*args):
                    return op(*args) or 0
        else:
            def call(*args):
                a = list(args[:-1] if has_stream else args)
                for i in host:
                    x = a[i]
                    if x is not None and not isinstance(x, Tensor):
                        a[i] = _host_bytes(x)
                return op(*a) or 0
        self.__dict__[name] = call
        return call


class _Ctypes:
    """The same call convention on ctypes (AB_BINDING=ctypes)."""

    def __init__(self):
        from . import gen_torch_ops
        self._c = cdll()
        self._sig = {name: params for _, name, params in gen_torch_ops.declarations()}

    def __getattr__(self, name):
        fn, params = getattr(self._c, name), self._sig[name]

        def call(*args):
            conv = []
            for a, (ty, _) in zip(args, params):
                if a is _STREAM:
                    conv.append(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                elif ty.endswith("*"):
                    conv.append(ctypes.c_void_p(a.data_ptr()) if isinstance(a, torch.Tensor) else (ctypes.c_void_p(0) if a is None else a))
                elif ty == "float":
                    conv.append(ctypes.c_float(a))
                elif ty == "long":
                    conv.append(ctypes.c_long(a))
                else:
                    conv.append(ctypes.c_int(a))
            return fn(*conv)
        self.__dict__[name] = call
        return call


def lib():
    global _lib
    if _lib is None:
        _lib = _TorchOps() if BINDING == "torch" else _Ctypes()
    return _lib


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype} (float32 or bfloat16 expected)")


def ptr(t):
    """A device buffer argument: the tensor itself (None = NULL), checked."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("artiboost_hip ops need device tensors (HIP); got a CPU tensor")
    if not t.is_contiguous():
        raise RuntimeError("artiboost_hip ops need contiguous tensors")
    return t


def view_ptr(t):
    """A (possibly row-pitched) device view whose pitch is passed separately: no contiguity check."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("artiboost_hip ops need device tensors (HIP); got a CPU tensor")
    return t


def addr(t):
    """Device address for a field of a host-side struct of the C ABI (ab_symcorner, ab_scene)."""
    return ctypes.c_void_p(ptr(t).data_ptr()) if t is not None else ctypes.c_void_p(0)


def stream():
    return _STREAM


def check(rc, what):
    if rc:
        raise RuntimeError(f"{what} failed with code {rc}" + (" (hipError)" if rc > 0 else " (argument error)"))


def f(x):
    return float(x)


def i(x):
    return int(x)


def l(x):
    return int(x)


class WgradReduceDesc(ctypes.Structure):
    """struct ab_wgrad_reduce_desc (include/artiboost_hip.h)."""
    _fields_ = [("slabs", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("slab_elems", ctypes.c_long), ("nslices", ctypes.c_int),
                ("src_j", ctypes.c_int), ("dst_j", ctypes.c_int), ("accumulate", ctypes.c_int), ("stem_mask", ctypes.c_int)]


class WgradGroupItem(ctypes.Structure):
    """struct ab_wgrad_group_item (include/artiboost_hip.h)."""
    _fields_ = [("x_hi", ctypes.c_void_p), ("x_lo", ctypes.c_void_p), ("dy_hi", ctypes.c_void_p), ("dy_lo", ctypes.c_void_p),
                ("dw", ctypes.c_void_p)]


class SymCorner(ctypes.Structure):
    """struct ab_symcorner (include/artiboost_hip.h)."""
    _fields_ = [("R", ctypes.c_void_p), ("t", ctypes.c_void_p), ("K", ctypes.c_int32), ("obj_idx", ctypes.c_void_p),
                ("obj_transf", ctypes.c_void_p), ("lam", ctypes.c_float), ("weight", ctypes.c_float),
                ("loss_out", ctypes.c_void_p)]

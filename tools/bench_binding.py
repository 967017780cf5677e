"""Host cost per call of the two bindings of the C ABI on the same tiny launches: torch.ops.artiboost_hip.* (argument checks included) vs ctypes.
usage: python tools/bench_binding.py        (spawns itself once per binding: the binding is chosen at import)"""
import os
import subprocess
import sys
import time

if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from artiboost_amd import kernels as K
    y = torch.randn(2, 4, 4, 64, device="cuda")
    bnp = torch.randn(4, 64, device="cuda")
    src = torch.randn(64, device="cuda")
    hi = torch.empty(2, 64, dtype=torch.bfloat16, device="cuda")

    def t(fn, n=20000):
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        return dt / n * 1e6

    from artiboost_amd import _lib as L
    lib = L.lib()
    a = t(lambda: lib.ab_split_f32(L.ptr(src), L.l(64), L.ptr(hi[0]), L.ptr(hi[1]), L.stream()))
    sp, o = torch.empty((2, 2, 4, 4, 64), dtype=torch.bfloat16, device="cuda"), torch.empty_like(y)
    b = t(lambda: lib.ab_bn_apply_x3(L.ptr(y), L.ptr(None), L.ptr(bnp), L.l(32), L.i(64), L.i(1), L.ptr(o), L.ptr(sp[0]), L.ptr(sp[1]), L.stream()))
    print(f"{sys.argv[1]:7s}: ab_split_f32 (3 tensors) {a:.2f} us per call, ab_bn_apply_x3 (6 tensors, 3 clauses) {b:.2f} us per call (host, launch included)")
else:
    for b in ("torch", "ctypes"):
        subprocess.run([sys.executable, os.path.abspath(__file__), b], env=dict(os.environ, AB_BINDING=b), check=False)

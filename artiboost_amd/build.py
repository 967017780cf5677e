"""Builds libartiboost_hip.so (all hand-written HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels to the GPU box
with the repo snapshot (it is git-ignored, not gpurun-ignored)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libartiboost_hip.so")
TORCH_LIB = os.path.join(HERE, "libartiboost_torch.so")      # the same entry points as torch.ops.artiboost_hip.* (gen_torch_ops.py)
# -packed-fp32-ops (round 6): no v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32.  Measured on MI355X (tools/render_race_debug.py + tools/probe_neighbour.hip):
# while ANOTHER wave on the same SIMD executes v_mfma_f32_32x32x16_bf16, packed-fp32 VALU instructions of a co-resident wave return wrong results
# (the shading of the renderer evaluated twice in one thread on identical inputs differed; a register-only MFMA loop as the neighbour is enough;
# scalar-fp32 code beside the same neighbour is exact).  Alone on its stream no kernel here shares a SIMD with another kernel's MFMA waves, but the
# overlap modes do (DDP render overlap, realdata.ThreadedPrefetcher, pipeline_render), and so do the tail waves of the MFMA kernels themselves.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "-Wno-array-bounds", "-Wno-int-to-pointer-cast",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    srcs = sources()
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = ["hipcc"] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {s}")
    if force or procs or _stale(LIB, objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        subprocess.check_call(cmd)
    build_torch_ops(force=force, verbose=verbose)
    return LIB


def build_torch_ops(force=False, verbose=False):
    """libartiboost_torch.so: TORCH_LIBRARY(artiboost_hip) registrations over the C ABI (one generated translation unit, host
    compiler + torch headers; links libartiboost_hip.so by $ORIGIN so both travel together in-tree)."""
    import torch
    from torch.utils import cpp_extension as ce
    from . import gen_torch_ops
    src, _ = gen_torch_ops.generate()
    hdr = os.path.join(HERE, "..", "include", "artiboost_hip.h")
    if not (force or _stale(TORCH_LIB, [src, hdr, LIB])):
        return TORCH_LIB
    try:
        inc = ce.include_paths(device_type="cuda")
    except TypeError:
        inc = ce.include_paths(True)
    libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)] + ["-I" + i for i in inc] + ["-I/opt/rocm/include"] +
           [src, "-o", TORCH_LIB, "-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip", "-L" + HERE, "-lartiboost_hip",
            "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + libdir])
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        sys.stderr.write(r.stdout.decode())
        raise RuntimeError("g++ failed on torch_ops_gen.cpp")
    return TORCH_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

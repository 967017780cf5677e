"""ISA lint for the LDS-ring kernels: every raw s_barrier of a kernel that restages an LDS buffer right after it must be
preceded by an s_waitcnt lgkmcnt(0) with no fragment ds_read in between, i.e. every operand read issued before the barrier
has RETIRED (a linear scan of the ISA text: control flow is ignored, which is conservative for these straight-line loops).
Otherwise the compiler is free to sink the last MFMA of a K step (and the lgkmcnt wait it needs) below the barrier, and
the LDS-DMA of the next stage can overwrite data a slower wave has not read yet (write-after-read): sporadic wrong tiles
that differ from run to run -- found in the fully unrolled 4-step stem variant of conv_gemm2_kernel; the halo 3x3 kernel
had 130 of 364 barriers exposed the same way.  Usage: python tools/check_barrier_war.py   (compiles to ISA, ~1 min)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
FILES = ["conv_gemm2.hip", "conv3x3.hip", "wgrad3x3.hip", "wgrad_gemm2.hip", "stem_halo.hip"]


def scan(asm):
    kernels, cur, pending, bad, nbar = {}, None, False, 0, 0
    for line in asm.split("\n"):
        t = line.strip()
        if t.startswith("_Z") and t.split(":")[0].endswith(("Args", "E")) and ":" in t:
            if cur:
                kernels[cur] = (nbar, bad)
            cur, pending, bad, nbar = t.split(":")[0], False, 0, 0
        elif t.startswith(("ds_read_b128", "ds_read_b64", "ds_read2_b64")):      # operand-fragment reads (tap-table reads are b32)
            pending = True
        elif t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
            pending = False
        elif t.startswith("s_barrier"):
            nbar += 1
            bad += pending
    if cur:
        kernels[cur] = (nbar, bad)
    return kernels


def main():
    worst = 0
    for f in FILES:
        with tempfile.NamedTemporaryFile(suffix=".s") as out:
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                                   "-o", out.name, os.path.join(ROOT, "artiboost_amd", "csrc", f)], stderr=subprocess.DEVNULL)
            res = scan(open(out.name).read())
        nb = sum(v[0] for v in res.values()); bad = sum(v[1] for v in res.values())
        print(f"{f:18s} kernels {len(res):3d}  barriers {nb:4d}  barriers with un-retired LDS reads {bad}")
        worst += bad
    print("OK" if worst == 0 else "EXPOSED BARRIERS FOUND")
    return 1 if worst else 0


if __name__ == "__main__":
    sys.exit(main())

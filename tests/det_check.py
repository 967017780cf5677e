"""TEST INFRASTRUCTURE (run by hand: python tests/det_check.py; BS=64 SIZE=256 for the benchmark geometry): the small
end-to-end training of test_gpu_synth twice per mode, reporting where losses start to differ (determinism check)."""
import os, sys, types
import numpy as np, torch, pytest
HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.join(HERE, ".."), os.path.join(HERE, "..", "oracle")):
    sys.path.insert(0, p)
import test_gpu_synth as T

class MP:
    def setenv(self, k, v): os.environ[k] = v

runs = {}
BS, SIZE = int(os.environ.get("BS", 8)), int(os.environ.get("SIZE", 224))     # BS=64 SIZE=256: the benchmark geometry
for tag, split in (("a0", False), ("a1", False), ("s0", True), ("s1", True)):
    runs[tag] = T._run_steps(MP(), split, bs=BS, size=SIZE)
for x, y in (("a0", "a1"), ("s0", "s1"), ("a0", "s0")):
    l0, w0 = runs[x]; l1, w1 = runs[y]
    d = np.abs(l0 - l1).max(axis=1)
    print(x, y, "max |dloss| per step:", np.array2string(d, precision=6), " weights equal:", np.array_equal(w0, w1))

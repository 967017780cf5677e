"""Run the small end-to-end training twice per mode and report where losses start to differ (determinism check)."""
import os, sys, types
import numpy as np, torch, pytest
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
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

"""A few launches of one bf16x3 3x3 convolution (forward with BatchNorm partials, plain data gradient) for rocprofv3 runs.
usage: python tools/one_c3v.py H C [reps] [Cout]   (B = 64, Cout = C by default)"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K
B = 64
H, C = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
x = K.split(torch.randn(B, H, H, C, device="cuda"))
Co = int(sys.argv[4]) if len(sys.argv) > 4 else C
w = K.split(torch.randn(Co, 3, 3, C, device="cuda") * 0.05)
wt = K.split(torch.randn(C, 3, 3, Co, device="cuda") * 0.05)
dy = K.split(torch.randn(B, H, H, Co, device="cuda"))
for _ in range(reps):
    K.conv2d_fwd_x3(x, w, 1, 1, want_stats=True)
    K.conv2d_dgrad_x3(dy, wt, (H, H), 1, 1)
torch.cuda.synchronize()

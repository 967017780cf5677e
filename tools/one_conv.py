import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K
B, H, W, Ci, Co = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 16, 0, int(sys.argv[2]) if len(sys.argv) > 2 else 256, 0
W = H; Co = Ci
x = torch.randn(B, H, W, Ci, device="cuda").bfloat16(); w = (torch.randn(Co, 3, 3, Ci, device="cuda") * 0.05).bfloat16()
for _ in range(10):
    y = K.conv2d_fwd(x, w, 1, 1)
torch.cuda.synchronize()

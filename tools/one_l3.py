"""A few launches of the l3 3x3 conv kernels (fwd, dgrad, wgrad) for PMC runs."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K
B, H, C = 64, int(os.environ.get("HW", 16)), int(os.environ.get("CH", 256))
x = torch.randn(B, H, H, C, device="cuda").bfloat16()
w = (torch.randn(C, 3, 3, C, device="cuda") * 0.05).bfloat16()
wt = w.permute(3, 1, 2, 0).contiguous()
dy = torch.randn(B, H, H, C, device="cuda").bfloat16()
for _ in range(5):
    K.conv2d_fwd(x, w, 1, 1, want_stats=True)
    K.conv2d_dgrad(dy, wt, (H, H), 1, 1)
    K.conv2d_wgrad(x, dy, 3, 3, 1, 1)
torch.cuda.synchronize()

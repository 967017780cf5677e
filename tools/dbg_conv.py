import sys, torch
sys.path.insert(0, '/root/repo')
from artiboost_amd import kernels as K
cases = [(2,16,16,64,64,3,1,1),(3,14,10,64,128,3,2,1),(2,8,8,128,128,3,1,1),(2,14,14,64,128,1,2,0),(1,7,7,512,512,3,1,1),(2,9,5,256,616,1,1,0),(5,6,6,256,256,3,1,1)]
for c in cases:
    N,H,W,Ci,Co,k,s,p = c
    x = torch.randn(N,H,W,Ci,device='cuda').bfloat16(); w = torch.randn(Co,k,k,Ci,device='cuda').bfloat16()
    print('fwd', c, flush=True)
    y, st = K.conv2d_fwd(x, w, s, p, want_stats=True); torch.cuda.synchronize()
    print(' ok', float(y.float().abs().mean()), flush=True)

"""Sustained time of one bf16x3 3x3 convolution launch: 20 launches captured in a graph, replayed back to back.
usage: [AB_C3V=0|1] [RELU=1] python tools/ab_c3v.py H C [fwd|dgrad|bn]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K
B = 64
H, C = int(sys.argv[1]), int(sys.argv[2])
what = sys.argv[3] if len(sys.argv) > 3 else "fwd"
x = torch.randn(B, H, H, C, device="cuda")
if os.environ.get("RELU", "0") == "1":
    x = torch.relu(x)
x = K.split(x)
w = K.split(torch.randn(C, 3, 3, C, device="cuda") * 0.05)
N = 20


def body():
    for _ in range(N):
        if what == "fwd":
            K.conv2d_fwd_x3(x, w, 1, 1, want_stats=True)
        else:
            K.conv2d_dgrad_x3(x, w, (H, H), 1, 1)


body()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g):
        body()
torch.cuda.synchronize()
res = []
for rep in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / (10 * N) * 1e3)
print(f"H={H} C={C} {what} AB_C3V={os.environ.get('AB_C3V', '0')} RELU={os.environ.get('RELU', '0')}: us per launch (6 x 200 launches):", " ".join(f"{r:.1f}" for r in res))

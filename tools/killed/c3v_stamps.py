"""Per-wave s_memtime stamps of conv3x3v_kernel (0 entry, 1 loop start, 2 chunk 1 start, 3 loop end, 4 staged, 5 rows stored, 6 end).
usage: AB_C3V=1 python tools/c3v_stamps.py H C [Cout]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from artiboost_amd import kernels as K, _lib as L
B = 64
H, C = int(sys.argv[1]), int(sys.argv[2])
Co = int(sys.argv[3]) if len(sys.argv) > 3 else C
x = torch.randn(B, H, H, C, device="cuda")
if os.environ.get("RELU", "0") == "1":
    x = torch.relu(x)
x = K.split(x)
w = K.split(torch.randn(Co, 3, 3, C, device="cuda") * 0.05)
for _ in range(3):
    K.conv2d_fwd_x3(x, w, 1, 1, want_stats=True)
def wall(n=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        K.conv2d_fwd_x3(x, w, 1, 1, want_stats=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
wall(20)
print("wall per call incl. pack (us):", [round(wall(), 1) for _ in range(3)])
dbg = torch.zeros(4096 * 4 * 8, dtype=torch.int64, device="cuda")
lib = L.cdll()
lib.ab_c3v_debug_buffer.argtypes = [ctypes.c_void_p]
lib.ab_c3v_debug_buffer(ctypes.c_void_p(dbg.data_ptr()))
K.conv2d_fwd_x3(x, w, 1, 1, want_stats=True)
torch.cuda.synchronize()
lib.ab_c3v_debug_buffer(ctypes.c_void_p(0))
d = dbg.cpu().view(-1, 4, 8)
d = d[d[:, 0, 0] != 0].double()
print("workgroups", d.shape[0])
t0 = d[:, :, 0].min()
names = ["entry", "loop", "chunk1", "loopend", "staged", "stored", "end"]
for k in range(7):
    v = d[:, :, k] - t0
    print(f"{names[k]:8s} min {v.min():9.0f} median {v.median():9.0f} max {v.max():9.0f}   (ticks since first entry)")
rt = d[:, :, 7]
print(f"realtime ticks (100 MHz) per wave: median {float(rt.median()):.0f} -> {float(rt.median()) / 100:.2f} us; cycles / realtime = {float((d[:, :, 6] - d[:, :, 0]).median()) / (float(rt.median()) * 10):.3f} GHz")
dur = d[:, :, 6] - d[:, :, 0]
print("per wave: total median", float(dur.median()), "loop", float((d[:, :, 3] - d[:, :, 1]).median()), "prologue", float((d[:, :, 1] - d[:, :, 0]).median()),
      "chunk0", float((d[:, :, 2] - d[:, :, 1]).median()), "loopend->staged", float((d[:, :, 4] - d[:, :, 3]).median()), "staged->stored", float((d[:, :, 5] - d[:, :, 4]).median()),
      "stored->end", float((d[:, :, 6] - d[:, :, 5]).median()))

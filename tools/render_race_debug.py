"""Is the batched render bit-stable while OTHER kernels share the GPU?  (Round 6: it was not -- see artiboost_amd/build.py on -packed-fp32-ops.)
Renders 24 samples repeatedly on one stream while a neighbour runs on another, and compares visibility keys, shaded RGBX and the jittered
output with the first solo render.  Neighbours: this build's layer-3 3x3 forward (v_mfma_f32_32x32x16_bf16 under an LDS ring), the replayed
training step, and -- if tools/probe_neighbour.hip has been built -- register-only MFMA loops, idle / written / LDS-DMA-filled LDS allocations.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/libprobe_neighbour.so tools/probe_neighbour.hip      # optional
    python tools/render_race_debug.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import yaml  # noqa: E402
from artiboost_amd import kernels as K  # noqa: E402
from artiboost_amd.assets import SceneAssets  # noqa: E402
from artiboost_amd.synth import ArtiBoostLoader  # noqa: E402

B, res = 24, 256
cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [res, res], [res // 8, res // 8]
loader = ArtiBoostLoader.from_assets(SceneAssets("HO3D", seed=1), cfg["MANAGER"], cfg["DATA_PRESET"], B, B, compute_dtype="u8n", random_seed=7)
loader.prepare()
st = loader.new_static_batch()
loader.load_batch(st, 0)                      # one batch of the loader's own epoch: CCV samples, hand vertices, jitter draws
r, dev = loader.renderer, loader.dev
smp, hv, od, fc, ia, bl = st["_samples"], st["_hand_verts"], st["_order"], st["_factor"], st["_inv_affine"], st["_blur"]


def render():
    pad = torch.zeros((B, res + 6, res + 8, 4), dtype=torch.bfloat16, device=dev)
    o = r.render(smp, hv, od, fc, ia, res, res, out_pad=pad, want_keys=True, want_rgbx=True, blur=bl, pad_code=2)
    return o["keys"].clone(), o["rgbx"].clone(), pad


ref = render(); torch.cuda.synchronize()
side = torch.cuda.Stream()
x3 = K.split(torch.randn(64, 16, 16, 256, device=dev)); w3 = K.split(torch.randn(256, 3, 3, 256, device=dev) * 0.05)
cases = {"this build's layer-3 3x3 forward": lambda: K.conv2d_fwd_x3(x3, w3, 1, 1, want_stats=True)}
so = os.path.join(ROOT, "tools", "libprobe_neighbour.so")
if os.path.exists(so):
    probe = ctypes.CDLL(so)
    src = torch.randint(0, 2 ** 31 - 1, (1 << 20, 4), dtype=torch.int32, device=dev); sink = torch.zeros(4, dtype=torch.int32, device=dev)
    for name, (blocks, lds, mode, spin) in {"register-only MFMA 32x32x16 bf16": (2048, 1024, 6, 100), "register-only MFMA 16x16x32 bf16": (2048, 1024, 7, 100),
                                            "120 KB of LDS, idle": (512, 120 << 10, 0, 60), "120 KB of LDS, ds_write fill": (512, 120 << 10, 1, 60),
                                            "120 KB of LDS, streaming LDS-DMA fill": (1024, 120 << 10, 2, 200), "streaming 16-byte loads": (1024, 64 << 10, 5, 200)}.items():
        cases[name] = (lambda a=(blocks, lds, mode, spin): probe.lds_neighbour_launch(ctypes.c_void_p(src.data_ptr()), a[0], a[1], a[2], ctypes.c_long(a[3]),
                                                                                     ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
for name, fn in cases.items():
    fn(); torch.cuda.synchronize()
    bad = {"keys": 0, "rgbx": 0, "out": 0}
    for it in range(16):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(10):
                fn()
        got = render()
        torch.cuda.synchronize()
        for k, a, b in zip(bad, ref, got):
            bad[k] += int(not torch.equal(a, b))
    print(f"render beside {name:40s}: renders differing from the solo one, of 16: {bad}", flush=True)

#!/usr/bin/env python
"""bench.py -- synthesized-samples/sec of the coupled hot path on N MI355X (one process per GPU).

A "step" = one pass of the hot path over one per-GPU batch of 64 synthetic CCV samples at 256x256:
   gather sample -> rasterise/z-buffer/shade 512x512 -> colour jitter -> affine crop (HIP render kernels)
   -> HybridBaseline ResNet-34 forward -> fused soft-argmax -> fused pose+criterion (+backward) -> backward
   -> [RCCL all-reduce of the flat gradient] -> global-norm clip + Adam (HIP)
Inputs (epoch pose cache, assets, weights) are resident in HBM when the timed region starts; the per-step host work
(loss RNG draws, Adam bias corrections, batch gather) is inside it.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel family (implicit-GEMM conv on MFMA): algorithmic
FLOPs of the conv stack per step / summed conv-kernel time measured with HIP events on the compute stream.
`cpu_baseline` times the CPU oracle of the same step (oracle/: C rasteriser + torch-CPU fp32 learner) on a bounded
sample on this box's host cores."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_FWD_BWD_PER_SAMPLE = {224: 24.336, 256: 31.785}      # SURVEY.md section 8d (torch flop counter on the reference module)
# MI355X_MICROARCH.md (dense): bf16 MFMA 2.5 PFLOP/s, f32-input MFMA 157.3 TFLOP/s.  "bf16x3" (split-bf16, fp32-grade: the
# reference's precision) spends three bf16 MFMA passes per product, so its roof for ALGORITHMIC flops is 2500 / 3.
MFMA_PEAK_TFLOPS = {"bf16x3": 2500.0 / 3.0, "bf16": 2500.0, "f32": 157.3}
DTYPE_NOTE = {"bf16x3": "fp32 conv outputs/gradients/BatchNorm/optimizer; post-activation tensors held as the (hi+lo) bf16 pairs the convolutions read; convolutions as split-bf16 (hi+lo) x3 MFMA passes with fp32 accumulation "
                        "(2^-17 operand precision; meets the f32 tolerances of tests/test_gpu_learner.py against the reference goldens)",
              "f32": "exact f32 MFMA everywhere", "bf16": "bf16 operands and activations (misses the 1e-3 parity bound)"}


CFG_OF = {"HO3D": "ho3dv2_clasbased_artiboost_mi355x.yaml",             # BASELINE configs[2] / [3]
          "DexYCB": "dexycb_clasbased_sym_mi355x.yaml"}                  # BASELINE configs[4]: 21 objects, + SymCornerLoss


# the padded image the headline's loader hands the bf16x3 stem: the integer plane 2 v - 255 (tests/test_gpu_fullsize.py pins THIS configuration
# to the CPU oracle at full size)
DEFAULT_IMAGE_PLANE = "u8n"


def ref_cfg(size, dataset="HO3D"):
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", CFG_OF[dataset])))
    cfg["DATA_PRESET"]["IMAGE_SIZE"] = [size, size]
    cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [size // 8, size // 8]
    cfg["ARCH"]["BACKBONE"]["PRETRAINED"] = False       # ImageNet weights are a download; random init (stated in `data`)
    return cfg


def build_everything(args, rank, world, device, wgrad_1pass=False):
    import torch
    from artiboost_amd import registry as R
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.synth import ArtiBoostLoader
    from artiboost_amd.train import TrainStep
    cfg = ref_cfg(args.size, args.dataset)
    arch_cfg = dict(cfg["ARCH"], COMPUTE_DTYPE=args.dtype, DEVICE=device, INIT_SEED=cfg["TRAIN"]["MANUAL_SEED"])
    model = Arch({"ARCH": arch_cfg}, R.build_arch_model_list(arch_cfg, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    hb = model.model_list[0]
    if wgrad_1pass:
        hb.net.wgrad_1pass = True           # (instance override of the AB_WGRAD_1PASS class default; read when the step is captured)
    opt = FusedClipAdam(model.models_params, lr=cfg["TRAIN"]["LR"], max_norm=cfg["TRAIN"]["GRAD_CLIP"], model=hb)
    assets = SceneAssets(args.dataset, seed=1)
    mgr = dict(cfg["MANAGER"], EPOCH=cfg["TRAIN"]["EPOCH"])
    synth_len = args.bs * world * max(args.steps + args.warmup + 2, 4)
    # bf16x3: the renderer hands the stem the integer image plane 2 v - 255 (AB_DT_U8N: two MFMA passes, no split pass; --image-plane f32: the
    # fp32 image and its (hi, lo) split, the round-4 path)
    u8n = args.dtype == "bf16x3" and getattr(args, "image_plane", DEFAULT_IMAGE_PLANE) == "u8n" and not wgrad_1pass
    loader = ArtiBoostLoader.from_assets(assets, mgr, cfg["DATA_PRESET"], args.bs, synth_len, device=device,
                             compute_dtype="u8n" if u8n else hb.net.dtype, random_seed=cfg["TRAIN"]["MANUAL_SEED"], rank=rank, world_size=world)
    loader.prepare()
    static = loader.new_static_batch()
    loader.load_batch(static, 0)
    group = torch.distributed.group.WORLD if (world > 1 or getattr(args, "rccl_single_rank", False)) else None
    model.train()
    # N > 1: the north-star schedule -- batch t+1 is rendered on a second stream while step t's last gradient range is
    # all-reduced and its clip + Adam runs (on one GPU that concurrency measured slower than the single queue, so it stays off)
    overlap = args.pipeline_opt or ((world > 1 or getattr(args, "rccl_single_rank", False)) and not args.no_render_overlap)
    ts = TrainStep(model, crit, opt, static, use_graph=not args.eager, dist_group=group, renderer=loader,
                   pipeline_render=("opt" if overlap else args.pipeline))
    args.render_overlap = bool(overlap)
    ts.static = static
    return cfg, model, crit, opt, loader, ts, static


def hbm_rooflines(args, loader, static, model):
    """The two HBM-bound halves of the step against the HBM roof (north_star: "rocprof HBM GB/s (rasterizer)"), measured live with HIP
    events on the compute stream; ALGORITHMIC bytes per sample from SURVEY.md section 8d: render chain (LBS output -> raster -> shade ->
    composite -> augment) 2.0 MB, soft-argmax head 2.52 MB forward + 2 x 2.52 MB backward at 256^2 (1.93 MB at 224^2).
    The committed PMC passes (profiles/) give the bytes these kernels actually move."""
    import torch
    from artiboost_amd.head import softargmax3d_fwd, softargmax3d_bwd_x3, softargmax3d_bwd
    out = {}

    def timed(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3

    t = timed(lambda: loader.render_into(static))
    byts = 2.0e6 * args.bs
    out["render"] = {"bound": "hbm", "achieved": round(byts / t / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(byts / t / 8e12, 4),
                     "traffic": None, "ms": round(t * 1e3, 3),
                     "kernel": "render chain: raster_setup, raster_shade (512^2, LDS z-buffer), gauss_blur, jitter_stats, warp_jitter -- latency / VALU bound "
                               "(LDS atomics, dependent texel fetches, PIL's HSV arithmetic), the reference's full-frame-then-crop flow moves 4.5x these bytes"}
    hb = model.model_list[0]
    C, D, DP = hb.nclasses, hb.depth_res, 32
    hw = args.size // 8
    logits = torch.randn((args.bs, hw, hw, C * DP), dtype=torch.float32, device=static["image_nhwc4_padded"].device)
    uvd, conf, stat = softargmax3d_fwd(logits, C, D, DP)
    g = torch.randn_like(uvd)
    tf = timed(lambda: softargmax3d_fwd(logits, C, D, DP))
    if hb.net.x3:
        tb = timed(lambda: softargmax3d_bwd_x3(logits, C, D, DP, uvd, conf, stat, g))
    else:
        tb = timed(lambda: softargmax3d_bwd(logits, C, D, DP, uvd, conf, stat, g))
    per = 22 * 28 * hw * hw * 4.0 * args.bs
    out["softargmax"] = {"bound": "hbm", "achieved": round(3 * per / (tf + tb) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(3 * per / (tf + tb) / 8e12, 4), "traffic": None, "ms_fwd": round(tf * 1e3, 3), "ms_bwd": round(tb * 1e3, 3),
                         "kernel": "sam_stage1/2 (logits read once) + sam_bwd (logits read once, dlogits written once; depth pitch 32: the kernels "
                                   "move 32/28 of the algorithmic bytes).  The standalone two-stage forward is timed here; in the bf16x3 step its first "
                                   "stage rides in the final layer's GEMM epilogue (gemm_rw_kernel<4, true>) and only sam_stage2 is launched"}
    return out


def wgrad_1pass_leg(args, device, steps=10, warmup=3):
    """PRECISION STUDY beside the headline (never the headline): the same step with the WEIGHT gradients computed from the hi planes only
    (bf16 operands, one MFMA pass instead of three; forward, data gradients, BatchNorm, losses, optimizer unchanged).  Weight gradients are
    leaves -- nothing but Adam consumes them -- so their rounding does not propagate; every parity test of tests/test_gpu_learner.py and
    tests/test_gpu_fullsize.py passes unchanged with it.  Reported so that the cost of the third pass on this path is on record."""
    import copy
    import torch
    a = copy.copy(args)
    a.steps, a.warmup, a.pipeline, a.pipeline_opt = steps, warmup, False, False
    cfg, model, crit, opt, loader, ts, static = build_everything(a, 0, 1, device, wgrad_1pass=True)
    nb = len(loader)
    ts.prime(loader, 0)
    for i in range(warmup):
        ts.stage(loader, i % nb)
        ts()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        ts.stage(loader, (warmup + i) % nb)
        ts()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(args.bs * steps / dt, 1), "unit": "samples/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "dtype": args.dtype + "+wgrad_bf16_1pass",
            "note": "NOT a parity configuration and not the headline: weight-gradient operands at bf16 (2^-9), below the reference's fp32; see DESIGN 12.1"}


def jpeg_leg(device, n=64, iters=20):  # noqa: C901
    """SURVEY 8f-3, the decode in front of the real-data half: n 640 x 480 .jpg frames (the HO3D / DexYCB frame size; synthetic photographs
    written by Pillow at quality 92, 4:2:0) -> RGBX on the device through ab_jpeg_decode_batch (Huffman decode included), HIP-event timed on the
    stream; the reference's decoder (Pillow = libjpeg-turbo, one DataLoader worker) timed on this host beside it.  Bit-identical outputs
    (tests/test_gpu_jpeg.py); sized to ~2 s."""
    import io
    import numpy as np
    import torch
    try:
        from PIL import Image
    except ImportError:
        return {"error": "Pillow not importable: no files to decode"}
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_jpeg import _photo
    from artiboost_amd.jpeg import JpegDecoder, parse
    files = []
    for i in range(n):
        b = io.BytesIO()
        Image.fromarray(_photo(640, 480, i)).save(b, "JPEG", quality=92, subsampling=2)
        files.append(b.getvalue())
    t0 = time.perf_counter()
    for f in files[:16]:
        np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
    t_pil = (time.perf_counter() - t0) / 16
    out = torch.empty((n, 480, 640, 4), dtype=torch.uint8, device=device)
    dec = JpegDecoder(device)
    for _ in range(3):
        dec.decode(files, out=out)
    torch.cuda.synchronize()
    infos = [parse(f) for f in files]             # the host's share: a marker walk, 15 us per file (tools/bench_jpeg.py)
    # Two numbers per batch (round-5 review: one event pair around whole decode() calls measured whichever of host and device was slower, and
    # on a box with a slow host that was the host): the DEVICE time of a call -- a HIP event pair around its upload + ab_jpeg_decode_batch,
    # calls separated by a synchronisation so no host work of the next call hides in it -- and the HOST time in front of it (plan + pack of
    # the entropy-coded segments into the pinned blob, one thread).  `value` is the device rate; MixedLoader hides the host share behind the
    # training step of the previous group (decode_ahead), so the device time is what a mixed step pays.
    dev_ms, host_ms = [], []
    for _ in range(iters):
        tm = {}
        dec.decode(files, out=out, infos=infos, timing=tm)
        torch.cuda.synchronize()
        dev_ms.append(tm["events"][0].elapsed_time(tm["events"][1]))
        host_ms.append(tm["host_s"] * 1e3)
    ms, hms = sorted(dev_ms)[len(dev_ms) // 2], sorted(host_ms)[len(host_ms) // 2]
    t0 = time.perf_counter()
    for _ in range(iters):
        dec.decode(files, out=out, infos=infos)
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / iters * 1e3
    rr = dec.last_rounds()
    return {"metric": "real frames/sec decoded (.jpg 640x480 -> RGBX on the device; device time of ab_jpeg_decode_batch + its upload)",
            "value": round(n / ms * 1e3, 1), "unit": "frames/s",
            "ms_per_batch": round(ms, 3), "device_ms_per_batch": round(ms, 3), "host_plan_pack_ms_per_batch": round(hms, 3),
            "back_to_back_calls_ms_per_batch": round(wall_ms, 3), "bound_of_back_to_back_calls": "host (plan + pack)" if hms > ms else "device",
            "batch": n, "file_kb": round(sum(len(f) for f in files) / n / 1024, 1), "dtype": "u8/int32",
            "sync_rounds_mean_max": [round(float(rr.mean()), 1), int(rr.max())],
            "bound": "latency of one thread's Huffman chain (VALU + LDS look-ups), not HBM: algorithmic bytes per frame "
                     f"{round(sum(len(f) for f in files) / n / 1e6 + 640 * 480 * 4 / 1e6, 2)} MB",
            "cpu_reference": {"kind": "reference", "decoder": "Pillow (libjpeg-turbo) Image.open().convert('RGB'), 1 thread = one DataLoader worker",
                              "value": round(1.0 / t_pil, 1), "unit": "frames/s", "sample": "16 of the same files"}}


def rccl_leg(args, ms_main):
    """The multi-rank schedule over a ONE-rank RCCL group (`--rccl-single-rank`, a child process: a process group cannot be added to this
    one after the fact): the RCCL path as far as a 1-GPU box can execute it, and what the schedule costs before any link time."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--rccl-single-rank", "--steps", str(args.steps), "--warmup", str(args.warmup), "--sustain", "0",
           "--bs", str(args.bs), "--size", str(args.size), "--dtype", args.dtype, "--dataset", args.dataset]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    if r.returncode != 0 or not lines:
        return {"error": (r.stderr or r.stdout)[-600:]}
    d = json.loads(lines[-1])
    return {"value": d["value"], "unit": "samples/s", "ms_per_step": d["ms_per_step"], "schedule_overhead_ms": round(d["ms_per_step"] - ms_main, 3),
            "parallelism": d["config"]["parallelism"], "render_overlap": d["config"]["render_overlap"], "final_loss": d["final_loss"],
            "what": "init_process_group('nccl', device_id=...), three backward graphs, bucketed SUM all-reduces + 1 / world on the comm stream, next "
                    "batch rendered under the last range -- over ONE rank: no link time, not a scaling measurement"}


def dropin_leg(args, ms_main, steps=30):
    """The drop-in boundary itself: the REFERENCE-SHAPED loop (train/train_artiboost.py:46-105 epoch_pass -- `for batch in artiboost_loader`
    -> arch_model(batch) -> compute_losses -> evaluator.feed_all -> zero_grad -> backward -> clip_grad_norm_ -> optimizer.step), every object
    built through the `anakin.*` import paths with the reference's keyword signatures (tools/bench_dropin.py, a child process: anakin.opt
    parses sys.argv at import, as the reference's does), same geometry as the headline.  Wall clock over `steps` iterations after 5 of
    warm-up, with and without feed_all; the ratio to the headline's graph-replayed step says what a user of the reference's own loop gets."""
    import subprocess
    out = {"loop": "train_artiboost.py epoch_pass through anakin.* (eager Python, network halves as replayed segment graphs)", "steps": steps}
    for key, extra in (("ms_per_step", []), ("ms_per_step_no_feed", ["--no-feed"])):
        cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_dropin.py"), "--steps", str(steps), "--batch", str(args.bs), "--size", str(args.size)] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith('{"loop"')]
        if r.returncode != 0 or not lines:
            return {"error": (r.stderr or r.stdout)[-600:]}
        d = json.loads(lines[-1])
        out[key] = d["ms_per_step"]
        if not extra:
            out.update(value=d["samples_per_s"], unit="samples/s", final_loss=d["final_loss"], image_plane=d.get("image_plane"))
    out["ratio_to_headline"] = round(out["ms_per_step"] / ms_main, 3)
    out["ratio_to_headline_no_feed"] = round(out["ms_per_step_no_feed"] / ms_main, 3)
    return out


def mixed_leg(args, steps=20):
    """The reference's actual training mix (MixedDataset: real frames + the epoch's synthetic samples, mixed_dataset.py:5-37) as one step:
    40 real 640 x 480 frames served as .jpg files (decoded on the device, augmented by ab_augment_batch) + 24 rendered samples per batch of 64,
    hipGraph-replayed bf16x3 step; MixedLoader's default schedule (frames of four batches per decode call, the next group on a side stream)."""
    import yaml
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_mixed
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", CFG_OF["HO3D"])))
    cfg["DATA_PRESET"]["IMAGE_SIZE"] = [args.size, args.size]
    cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [args.size // 8, args.size // 8]
    # round 6: the batch assembly (host + device side) on a worker thread two batches ahead (realdata.ThreadedPrefetcher), as train/train_artiboost.py
    # runs it; `same_thread_ms_per_step` is the loop with everything on the training thread (what the line reported through round 5)
    same = "same stream, frames of 4 batches decoded per call one group ahead on a side stream"
    mode = "worker thread two batches ahead, frames of 4 batches decoded per call one group ahead on a side stream"
    r = bench_mixed.train_loop(cfg, steps=steps, modes=(same, mode), quiet=True)
    out = {"metric": "mixed samples/sec (40 real .jpg frames decoded + augmented on the device, 24 rendered; fwd+bwd+optimizer)", "value": round(64 / r[mode] * 1e3, 1),
           "unit": "samples/s", "ms_per_step": round(r[mode], 3), "steps": steps, "batch": 64, "final_loss": r["final_loss"],
           "same_thread_ms_per_step": round(r[same], 3),
           "workload": "ThreadedPrefetcher(MixedLoader (decode_group 4, decode_ahead), depth 2) + TrainStep graph replay, bf16x3; SURVEY 8f-3"}
    try:      # the same step over .png files -- HO3D v2's own frame format (ho3d.py:181): zlib inflate on the host pool, reconstruction on the device
        from artiboost_amd import png as P
        rp = bench_mixed.train_loop(cfg, steps=steps, modes=(mode,), quiet=True, source="png")
        out["png"] = {"metric": "mixed samples/sec (40 real .png frames: pooled zlib inflate on the host + ab_png_unfilter_batch, 24 rendered)",
                      "value": round(64 / rp[mode] * 1e3, 1), "unit": "samples/s", "ms_per_step": round(rp[mode], 3), "final_loss": rp["final_loss"],
                      "decode_threads": P.pool()._max_workers, "host_cores": os.cpu_count()}
    except Exception as e:      # noqa: BLE001 -- a side leg must not take the headline line down
        out["png"] = {"error": repr(e)[:300]}
    return out


def dexycb_leg(args, device, steps=10, warmup=3):
    """BASELINE configs[4] on ONE GPU (its 8-GPU form is this step under the data-parallel schedule of configs[3]): DexYCB-like scenes
    (21 objects at 16 k faces) and the DexYCB criterion list (+ SymCornerLoss in the fused pose/loss kernel), same geometry and precision."""
    import copy
    import torch
    a = copy.copy(args)
    a.dataset, a.steps, a.warmup, a.pipeline, a.pipeline_opt = "DexYCB", steps, warmup, False, False
    cfg, model, crit, opt, loader, ts, static = build_everything(a, 0, 1, device)
    nb = len(loader)
    ts.prime(loader, 0)
    for i in range(warmup):
        ts.stage(loader, i % nb)
        ts()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        ts.stage(loader, (warmup + i) % nb)
        ts()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ld = ts.fused.losses_dict() if ts.fused is not None else {}
    return {"metric": f"synth samples/sec (render+fwd+bwd) {args.size}x{args.size} bs={args.bs}", "value": round(args.bs * steps / dt, 1),
            "unit": "samples/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "n_gpus": 1,
            "workload": f"train_artiboost DexYCB clasbased_sym ({CFG_OF['DexYCB']}: 21 objects at 16 k faces, JointsLoss + HandOrdLoss + SceneOrdLoss + "
                        f"SymCornerLoss) + online CCV render 512->{args.size}, batch {args.bs}, {args.dtype}",
            "final_loss": float(ld["final_loss"]) if "final_loss" in ld else None,
            "sym_corners_3d_loss": float(ld["sym_corners_3d_loss"]) if "sym_corners_3d_loss" in ld else None}


# every kernels.py entry that launches conv-stack MFMA kernels of the training step (forward, data gradient, weight gradient + its slab
# reduction; the final layer's GEMM carries the soft-argmax statistics in its epilogue)
CONV_FNS = ["conv2d_fwd", "conv2d_stem_fwd", "conv2d_dgrad", "conv2d_wgrad", "conv2d_stem_wgrad",
            "conv2d_fwd_x3", "conv2d_stem_fwd_x3", "conv2d_dgrad_x3", "conv2d_dgrad_x3_pair", "conv2d_wgrad_x3", "conv2d_wgrad_x3_group",
            "conv2d_stem_wgrad_x3", "conv1x1_sam_fwd_x3"]


def conv_kernel_time_graph_ms(ts, loader, reps=3):
    """Per-step time of the conv-stack launches INSIDE the replayed step: the step is captured once more with a one-thread launch that
    writes the device's constant-rate wall clock (ab_wall_stamp) in front of and behind every conv-stack call; after a replay the slot
    differences are the durations of those launches in graph-replay mode (rocprofv3 is not available in the driver's bench run, torch's
    external timing events are disallowed on ROCm).  A bracket also contains the leading stamp launch itself (its clock read comes first)
    and two sub-microsecond gaps: that price is the difference between back-to-back stamps of a captured chain of 33, removed once per
    bracket (checked under rocprofv3: bracket sum - 113 x chain difference = 7.087 ms against 7.073 ms of kernel durations in the same
    replay, profiles/round5_a).  -> (ms per step, launches, chain difference us)."""
    import torch
    from artiboost_amd import _lib as L
    from artiboost_amd import kernels as K
    lib = L.lib()
    khz = int(lib.ab_wall_clock_khz())
    if khz <= 0:
        raise RuntimeError("no wall clock rate")
    dev = ts.dev
    slots = torch.zeros(2048, dtype=torch.int64, device=dev)
    state = {"n": 0}
    orig = {n: getattr(K, n) for n in CONV_FNS}

    def stamp():
        i = state["n"]
        state["n"] = i + 1
        L.check(lib.ab_wall_stamp(L.ptr(slots[i:i + 1]), L.stream()), "ab_wall_stamp")

    def wrap(fn):
        def f(*a, **k):
            if not torch.cuda.is_current_stream_capturing():
                return fn(*a, **k)
            stamp()
            r = fn(*a, **k)
            stamp()
            return r
        return f

    try:
        for n in CONV_FNS:
            setattr(K, n, wrap(orig[n]))
        ts.g_fwd_bwd = None                   # the next call captures the step again, stamps included
        ts.stage(loader, 0)
        ts()
        nst = state["n"]
        # price of a boundary between two trivial launches in a graph: 33 stamps in a row
        chain = torch.zeros(33, dtype=torch.int64, device=dev)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for i in range(33):
                    L.check(lib.ab_wall_stamp(L.ptr(chain[i:i + 1]), L.stream()), "ab_wall_stamp")
        torch.cuda.current_stream(dev).wait_stream(s)
        tot, bnd = [], []
        for r in range(reps):
            ts.stage(loader, (r + 1) % max(len(loader), 1))
            ts()
            g.replay()
            torch.cuda.synchronize(dev)
            t = slots[:nst].cpu().numpy().astype("float64")
            c = chain.cpu().numpy().astype("float64")
            b = float(sorted(c[1:] - c[:-1])[16]) / khz * 1e3          # us
            d = (t[1::2] - t[0::2]) / khz * 1e3                            # us per bracket
            tot.append(float((d - b).clip(min=0.0).sum()) / 1e3)
            bnd.append(b)
    finally:
        for n in CONV_FNS:
            setattr(K, n, orig[n])
        ts.g_fwd_bwd = None                   # (recaptured without the stamps by whoever steps next)
    k = len(tot) // 2
    return sorted(tot)[k], nst // 2, sorted(bnd)[k]


def conv_kernel_time_ms(ts, loader, static, iters=3):
    """Average per-step time of the conv-stack MFMA kernels, measured with HIP events on the compute stream by
    running the step eagerly with events around every conv launch (kernels.py hooks).  Eager launches run 3 - 5 % longer than the same
    kernels inside the replayed graph (round-4 review: 7.33 vs 6.95 ms): kept as `conv_ms_per_step_eager`, the roofline uses
    conv_kernel_time_graph_ms."""
    import torch
    from artiboost_amd import kernels as K
    names = CONV_FNS
    orig = {n: getattr(K, n) for n in names}
    spans = []

    def wrap(fn):
        def f(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            spans.append((e0, e1))
            return r
        return f

    totals = []
    try:
        for n in names:
            setattr(K, n, wrap(orig[n]))
        for it in range(iters + 1):          # iteration 0 is not counted: the first eager step allocates its activations (slow hipMallocs
            spans.clear()                    # can outlast the spin below, and the spans would then hold launch latency)
            loader.load_batch(static, it % max(len(loader), 1))
            ts.crit.draw(ts.dev)
            # park the stream behind a long spin so that the whole eager step is ENQUEUED before any of it runs: the
            # event pairs then bracket back-to-back device execution, not the Python launch latency between them
            torch.cuda._sleep(120_000_000)
            ts._learn()
            empties = []
            for _ in range(16):              # cost of an empty event pair on this stream (marker overhead), removed below
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); e1.record()
                empties.append((e0, e1))
            torch.cuda.synchronize()
            empty_ms = sorted(a.elapsed_time(b) for a, b in empties)[len(empties) // 2]
            if it:
                totals.append(sum(max(a.elapsed_time(b) - empty_ms, 0.0) for a, b in spans))
            nl = len(spans)
    finally:
        for n in names:
            setattr(K, n, orig[n])
    return sorted(totals)[len(totals) // 2], nl          # median of the counted iterations


def pmc_traffic(args):
    """HBM bytes per step moved by the conv-stack kernels, from the committed PMC passes of this same workload
    (tools/pmc_traffic.py; FETCH_SIZE x2 + WRITE_SIZE).  None for any other configuration."""
    if args.bs != 64 or args.size != 256 or args.dataset != "HO3D":
        return None
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE[args.dtype])) as f:
            return round(float(json.load(f)["conv_stack_bytes_per_step"]))
    except (OSError, KeyError, ValueError):
        return None


PMC_FILE = {"bf16": "round1_pmc_hbm_traffic.json", "bf16x3": "round6_c_pmc_hbm_traffic.json", "f32": "none"}
GFLOP_FWD_PER_SAMPLE = {224: 8.191, 256: 10.698}            # forward only (BASELINE configs[1])


def host_info():
    """What BASELINE.md section 3.3 asks to be printed next to the CPU number."""
    import torch
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    return {"os_cpu_count": os.cpu_count(), "sched_affinity": aff, "cpu_model": model, "torch_num_threads": torch.get_num_threads()}


class ClockWatch:
    """tools/clock_watch.sh inside the process: samples the engine clock and the socket power with rocm-smi (~3 Hz, a
    background thread) while a block of steps runs; medians over the samples taken."""

    def __init__(self, device_index=0):
        import shutil
        self.smi = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
        self.dev, self.sclk, self.power, self._stop, self._t = device_index, [], [], False, None

    def _run(self):
        import re
        import subprocess
        while not self._stop:
            try:
                out = subprocess.run([self.smi, "-d", str(self.dev), "--showclocks", "--showpower"], capture_output=True, text=True,
                                     timeout=5).stdout
            except Exception:   # noqa: BLE001
                return
            m = re.search(r"sclk clock level:[^(]*\((\d+)Mhz\)", out)
            if m:
                self.sclk.append(int(m.group(1)))
            m = re.search(r"(?:Current Socket Graphics Package Power|Average Graphics Package Power) \(W\):\s*([0-9.]+)", out)
            if m:
                self.power.append(float(m.group(1)))

    def __enter__(self):
        if self.smi:
            import threading
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._t is not None:
            self._t.join(timeout=6)

    def summary(self):
        med = lambda v: sorted(v)[len(v) // 2] if v else None   # noqa: E731
        return {"sclk_mhz_median": med(self.sclk), "sclk_mhz_min": min(self.sclk) if self.sclk else None,
                "socket_power_w_median": med(self.power), "samples": len(self.sclk)}


def sustained_block(ts, loader, args, seconds, step0, barrier):
    """>= `seconds` of back-to-back steps in the same process, after the timed block (the driver's K steps are a 0.2 s burst; the
    matrix kernels are power-limited and settle at a lower clock within a second or two): samples/s over the block + the observed
    engine clock / socket power."""
    nb = len(loader)
    barrier()
    n, t0 = 0, time.perf_counter()
    with ClockWatch() as cw:
        while True:
            for _ in range(50):
                ts.stage(loader, (step0 + n) % nb)
                ts()
                n += 1
            import torch
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= seconds:
                break
    barrier()
    dt = time.perf_counter() - t0
    out = {"sustained_samples_per_s": round(args.bs * n / dt, 1), "sustained_ms_per_step": round(dt / n * 1e3, 3),
           "sustained_steps": n, "sustained_seconds": round(dt, 2)}
    out.update(cw.summary())
    return out


def eval_forward(args, model, static, steps=30, warmup=5, want_roofline=True):
    """BASELINE configs[1]: HO3Dv2 clasbased eval, ResNet-34, bs = 64 on one GPU -- the eval-mode forward (conv stack with the
    BatchNorms folded into the 3x3 epilogues, soft-argmax integral, pose assembly) of train/submit_reload.py:26-79 on a batch that is
    already resident in HBM, replayed as a hipGraph.  Returns value (samples/s), ms, and the forward conv stack's MFMA roofline."""
    import torch
    from artiboost_amd import kernels as K
    from artiboost_amd.train import CAPTURE_MODE
    batch = {k: v for k, v in static.items() if not k.startswith("_")}
    was_training = model.training
    model.eval()
    hb = model.model_list[0]
    seg = hb.segment_graphs
    hb.segment_graphs = False                       # one whole-forward graph below instead of the drop-in loop's segment graphs
    try:
        with torch.no_grad():
            for _ in range(2):
                model(batch)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                out = model(batch)
            for _ in range(warmup):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            res = {"metric": f"eval samples/sec (fwd only) {args.size}x{args.size} bs={args.bs}", "value": round(args.bs / dt, 1), "unit": "samples/s",
                   "ms_per_batch": round(dt * 1e3, 3), "steps": steps,
                   "workload": f"HO3Dv2 clasbased eval forward (HybridBaseline/ResNet-34, eval-mode BatchNorm folded into the 3x3 conv epilogues, "
                               f"soft-argmax, pose assembly), bs {args.bs}, {args.size}x{args.size}, batch resident in HBM, graph replay",
                   "finite": bool(torch.isfinite(out["HybridBaseline"]["joints_3d_abs"]).all())}
            if want_roofline:
                names = ["conv2d_fwd_x3", "conv2d_stem_fwd_x3", "conv2d_dgrad_x3", "conv2d_fwd_x3_evalbn", "conv2d_fwd_x3_affine", "conv2d_dgrad_x3_affine",
                         "conv2d_fwd", "conv2d_stem_fwd", "conv2d_dgrad"]
                orig = {n: getattr(K, n) for n in names}
                spans = []

                def wrap(fn):
                    def f(*a, **k):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); r = fn(*a, **k); e1.record()
                        spans.append((e0, e1))
                        return r
                    return f
                tots = []
                try:
                    for n in names:
                        setattr(K, n, wrap(orig[n]))
                    for it in range(4):          # iteration 0 not counted (first-use allocations), median of the other three
                        spans.clear()
                        torch.cuda._sleep(60_000_000)
                        model(batch)
                        empties = []
                        for _ in range(16):
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(); e1.record(); empties.append((e0, e1))
                        torch.cuda.synchronize()
                        em = sorted(a.elapsed_time(b) for a, b in empties)[8]
                        if it:
                            tots.append(sum(max(a.elapsed_time(b) - em, 0.0) for a, b in spans))
                        nl = len(spans)
                finally:
                    for n in names:
                        setattr(K, n, orig[n])
                conv_ms = sorted(tots)[1]
                fl = GFLOP_FWD_PER_SAMPLE.get(args.size, 10.698) * 1e9 * args.bs
                ach = fl / (conv_ms * 1e-3) / 1e12
                res["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                                   "frac": round(ach / MFMA_PEAK_TFLOPS[args.dtype], 4), "traffic": None,
                                   "kernel": "forward conv stack, eval-mode BatchNorm folded into every epilogue (conv3x3_kernel<..,X3=3>, conv_gemm2_kernel with the "
                                             "affine epilogue, stem_halo_x3_kernel)",
                                   "conv_ms_per_batch": round(conv_ms, 3), "conv_launches": nl}
    finally:
        hb.segment_graphs = seg
        model.train(was_training)
    return res


def cpu_baseline(args, cfg):
    """The CPU oracle of the same step on this box's host cores: C software renderer (OpenMP over samples) + torch-CPU
    fp32 HybridBaseline forward/loss/backward/clip+Adam.  Bounded sample."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gen_scene
    import learner_oracle as lo
    import render_oracle as ro
    from artiboost_amd.assets import SceneAssets
    # 32 threads: torch's CPU convolutions stop scaling (and collapse under oversubscription) well before the 256
    # hardware threads of the GPU box; `cores` reports the threads actually used
    cores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    n = args.cpu_samples
    assets = SceneAssets(args.dataset, seed=1)
    holder = ro.SceneHolder(assets)
    params = lo.fill_params(lo.param_shapes(22, 28), seed=1)
    leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in params.items()}
    iters = max(1, args.cpu_iters)
    names, ms, vs = None, None, None
    t_render = t_learn = 0.0

    def learner_step(batch, it, state):
        for v in leaf.values():
            if getattr(v, "grad", None) is not None:
                v.grad = None
        preds = lo.hybrid_forward(leaf, batch, [args.size, args.size], 22, 28, 0, training=True)
        total, _, _ = lo.criterion(preds, batch)
        total.backward()
        if state.get("names") is None:
            state["names"] = [k for k, v in leaf.items() if v.dtype.is_floating_point and getattr(v, "grad", None) is not None]
            state["ms"] = [torch.zeros_like(leaf[k]) for k in state["names"]]
            state["vs"] = [torch.zeros_like(leaf[k]) for k in state["names"]]
        lo.clip_and_adam([leaf[k].detach() for k in state["names"]], [leaf[k].grad for k in state["names"]], state["ms"], state["vs"], it + 1)

    # thread-count calibration (8 / 16 / 32 / 64): one learner step of the SAME batch size at each candidate, the fastest is used for the measured
    # steps (the box has 64 cores / 256 hardware threads; torch's CPU convolutions at batch 64 stop scaling well before that)
    calib = {}
    if not getattr(args, "cpu_threads_fixed", False) and (os.cpu_count() or 1) >= 64:
        sc0 = gen_scene.make_samples(assets, n, 99, out_res=(args.size, args.size))
        img0, _, _ = holder.render_batch(sc0["samples"], sc0["hand_verts"], sc0["order"], sc0["factor"], sc0["inv_affine"], args.size, args.size,
                                         blur=sc0["blur"])
        b0 = {"image": torch.from_numpy(img0)}
        for k in ("root_joint", "cam_intr", "corners_can", "joints_3d", "corners_3d", "joints_vis", "corners_vis"):
            b0[k] = torch.from_numpy(np.stack([g[k] for g in sc0["gt"]]).astype(np.float32))
        st0 = {}
        for nt in (8, 16, 32, 64):
            torch.set_num_threads(nt)
            t0 = time.time()
            learner_step(b0, 0, st0)
            calib[nt] = round(n / (time.time() - t0), 2)
        cores = max(calib, key=calib.get)
        torch.set_num_threads(cores)
        os.environ["OMP_NUM_THREADS"] = str(cores)
        leaf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in params.items()}
    for it in range(iters):                      # `iters` whole steps: render a fresh batch of n samples, then one optimizer step on it
        sc = gen_scene.make_samples(assets, n, 1 + it, out_res=(args.size, args.size))
        t0 = time.time()
        img, _, _ = holder.render_batch(sc["samples"], sc["hand_verts"], sc["order"], sc["factor"], sc["inv_affine"], args.size, args.size,
                                        blur=sc["blur"])
        t_render += time.time() - t0
        gt = sc["gt"]
        batch = {"image": torch.from_numpy(img)}
        for k in ("root_joint", "cam_intr", "corners_can", "joints_3d", "corners_3d", "joints_vis", "corners_vis"):
            batch[k] = torch.from_numpy(np.stack([g[k] for g in gt]).astype(np.float32))
        t0 = time.time()
        if names is None:
            names = {}
        learner_step(batch, it, names)
        t_learn += time.time() - t0
    tot = t_render + t_learn
    # cross-check against BASELINE.md section 2 (the IMPORTED reference measured 19 samples/s fwd+bwd at bs 8, 224^2 on 8 cores /
    # 8 torch threads): the same restatement at that configuration on this host, so that "same work" can be judged at equal
    # geometry and thread count instead of across two batch sizes, two image sizes and two machines
    xc = None
    if not getattr(args, "no_cpu_crosscheck", False) and args.size == 256:
        try:
            torch.set_num_threads(min(8, cores))
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from gen_batch import make_batch
            b8 = make_batch(8, 224, 3)
            lf = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in params.items()}
            ts_ = []
            for it in range(3):
                t0 = time.time()
                for v in lf.values():
                    if getattr(v, "grad", None) is not None:
                        v.grad = None
                pr = lo.hybrid_forward(lf, b8, [224, 224], 22, 28, 0, training=True)
                tt, _, _ = lo.criterion(pr, b8)
                tt.backward()
                ts_.append(time.time() - t0)
            xc = {"samples_per_s": round(8 / min(ts_[1:]), 2), "config": f"oracle fwd+loss+bwd, bs 8, 224x224, {min(8, cores)} torch threads",
                  "imported_reference_on_the_build_container": "19 samples/s (BASELINE.md section 2: 8 cores, 8 threads)"}
        except Exception as e:   # noqa: BLE001
            xc = {"error": repr(e)}
        torch.set_num_threads(cores)
    return {"value": round(n * iters / tot, 3), "unit": "samples/s", "cores": cores, "kind": "port", "host": host_info(), "crosscheck_bs8_224": xc,
            "thread_calibration_samples_per_s": calib or None,
            "scaling_note": "torch-CPU fp32 convolutions at batch 64 x 256^2 do not scale past ~32 threads on this host (BatchNorm / elementwise "
                            "passes over 268 MB activations are DRAM-bound, oneDNN's convolutions saturate, more threads oversubscribe); "
                            "`crosscheck_bs8_224` repeats BASELINE.md section 2's configuration with the same restatement for an equal-geometry comparison",
            "sample": f"{iters} whole steps of the same workload at per-step batch {n} ({args.dataset}-like CCV samples, {args.size}x{args.size}): "
                      f"each step = C oracle render of a fresh batch (OpenMP, {t_render / iters:.2f}s) + torch-CPU fp32 HybridBaseline "
                      f"fwd+loss+bwd+clip/Adam ({t_learn / iters:.2f}s); {tot:.1f}s of CPU work in total",
            "render_samples_per_s": round(n * iters / t_render, 2), "learner_samples_per_s": round(n * iters / t_learn, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16x3", choices=["bf16x3", "f32", "bf16"],
                    help="bf16x3 (default) = the reference's fp32-grade precision on split-bf16 MFMA; f32 = exact-f32 MFMA; "
                         "bf16 = reduced precision (not a parity configuration)")
    ap.add_argument("--dataset", default="HO3D", choices=["HO3D", "DexYCB"])
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--image-plane", dest="image_plane", choices=["u8n", "f32"], default=DEFAULT_IMAGE_PLANE,
                    help="bf16x3: what the loader writes for the stem -- u8n: ONE bf16 plane of the odd integers 2v-255 (exact; two MFMA passes); "
                         "f32: the fp32 image, split into (hi, lo) planes by a pass of its own (three passes)")
    ap.add_argument("--pipeline", action="store_true",
                    help="render batch i+1 on a side stream while step i learns (measured slower on one GPU: the conv "
                         "kernels already fill the chip, co-scheduling the rasteriser only evicts their workgroups)")
    ap.add_argument("--pipeline-opt", action="store_true",
                    help="render batch i+1 on a side stream while step i's all-reduce and clip+Adam run")
    ap.add_argument("--no-render-overlap", action="store_true",
                    help="N > 1 only: keep the render of the next batch on the compute stream instead of overlapping it with the "
                         "gradient all-reduce and the optimizer")
    ap.add_argument("--cpu-samples", type=int, default=64)
    ap.add_argument("--cpu-iters", type=int, default=4)
    ap.add_argument("--dry-launch", action="store_true",
                    help="only start the ranks, form the process group and print the line skeleton (launcher test; no GPU work)")
    ap.add_argument("--allow-shared-devices", action="store_true",
                    help="let several ranks share one GPU over gloo (dry runs of the multi-rank schedule on a 1-GPU box)")
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--cpu-threads-fixed", action="store_true", help="use --cpu-threads as given (no 16/32/64 calibration step)")
    ap.add_argument("--no-cpu-crosscheck", action="store_true")
    ap.add_argument("--eval", action="store_true",
                    help="BASELINE configs[1] instead of the training step: eval-mode forward only (bs 64, conv + soft-argmax integral kernels), "
                         "with its own roofline object")
    ap.add_argument("--sustain", type=float, default=3.0,
                    help="seconds of the sustained block run after the timed steps (extra keys of the line; 0 = off)")
    ap.add_argument("--no-eval-leg", action="store_true", help="skip the configs[1] eval-forward sub-object of the default line")
    ap.add_argument("--no-dexycb-leg", action="store_true", help="skip the configs[4]-on-one-GPU sub-object of the default line")
    ap.add_argument("--rccl-single-rank", action="store_true",
                    help="N = 1 only: run the multi-rank schedule (three backward graphs, SUM all-reduces + 1 / world on the comm stream, render "
                         "overlap) over a ONE-rank RCCL group -- the part of the RCCL path a 1-GPU box can execute; not the headline")
    ap.add_argument("--no-rccl-leg", action="store_true", help="skip the one-rank RCCL schedule sub-object of the default line")
    ap.add_argument("--no-dropin-leg", action="store_true", help="skip the reference-shaped epoch_pass loop sub-object of the default line")
    ap.add_argument("--no-mixed-leg", action="store_true", help="skip the mixed real + synthetic training-step sub-object of the default line")
    ap.add_argument("--no-jpeg-leg", action="store_true", help="skip the real-frame JPEG decode sub-object of the default line")
    ap.add_argument("--no-study-leg", action="store_true", help="skip the one-pass weight-gradient study sub-object of the default line")
    ap.add_argument("--wgrad-1pass", action="store_true",
                    help="PRECISION STUDY, not a parity configuration (DESIGN 12.1): weight gradients from the hi planes only (one bf16 MFMA "
                         "pass instead of three); the line's dtype says so")
    args = ap.parse_args()

    # `python bench.py --gpus N` is the whole multi-GPU command (the reference's is `train_artiboost.py --gpu_id 0,1,..`,
    # train/train_artiboost.py:131,249-257): without a launcher around it, start one rank per GPU under torch.distributed.run
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    if args.wgrad_1pass:
        os.environ["AB_WGRAD_1PASS"] = "1"          # read when artiboost_amd.hybridnet is imported (below)
    import torch
    if args.dry_launch:
        world = int(os.environ.get("WORLD_SIZE", 1))
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.distributed.init_process_group("gloo")
            t = torch.ones(1)
            torch.distributed.all_reduce(t)
            world = int(t.item())
            rank0 = torch.distributed.get_rank() == 0
            torch.distributed.destroy_process_group()
        else:
            rank0 = True
        if rank0:
            print(json.dumps({"dry_launch": True, "n_gpus": world, "gpus_arg": args.gpus}), flush=True)
        return
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    ngpu = torch.cuda.device_count()
    # one rank per GPU over RCCL; if there are fewer devices than ranks (dry runs of the multi-rank path on a 1-GPU box)
    # the ranks share devices and the collectives go through gloo
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    shared = world > ngpu
    if shared and not args.allow_shared_devices:
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, found {ngpu} "
                         f"(--allow-shared-devices runs the ranks on shared devices over gloo: a schedule dry run, not a measurement)")
    local = local % max(ngpu, 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if shared:
            torch.distributed.init_process_group("gloo")
        else:
            from artiboost_amd.train import rccl_env_defaults
            rccl_env_defaults()
            torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    if args.rccl_single_rank:
        if world != 1:
            raise SystemExit("bench.py: --rccl-single-rank is an N = 1 mode")
        import socket
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["AB_DDP_SINGLE_RANK"] = "1"
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        from artiboost_amd.train import rccl_env_defaults
        rccl_env_defaults()
        torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                             device_id=torch.device(device))
        args.no_eval_leg = args.no_dexycb_leg = args.no_study_leg = args.no_jpeg_leg = args.no_mixed_leg = args.no_rccl_leg = args.no_dropin_leg = args.no_cpu_baseline = True
    cfg, model, crit, opt, loader, ts, static = build_everything(args, rank, world, device)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.eval:            # BASELINE configs[1]: forward only
        loader.render_into(static)
        torch.cuda.synchronize()
        res = eval_forward(args, model, static, steps=args.steps, warmup=args.warmup)
        if rank == 0:
            line = {"metric": res["metric"], "value": res["value"], "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                    "ms_per_step": res["ms_per_batch"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
                    "dtype_note": DTYPE_NOTE[args.dtype], "data": "synthetic (one rendered batch of seeded stand-in scenes; random-init weights)",
                    "config": {"workload": res["workload"], "global_batch": args.bs * world, "image": args.size, "parallelism": f"dp{world}", "graph": True},
                    "roofline": res.get("roofline"), "cpu_baseline": None}
            print(json.dumps(line), flush=True)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    nb = len(loader)
    ts.prime(loader, 0)
    for i in range(args.warmup):
        ts.stage(loader, i % nb)
        ts()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ts.stage(loader, (args.warmup + i) % nb)
        ts()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    ms = dt / args.steps * 1e3
    if world > 1:
        world = torch.distributed.get_world_size()
    value = args.bs * world * args.steps / dt
    sustained = None
    if args.sustain > 0:      # every rank runs it (the steps contain the gradient all-reduce)
        sustained = sustained_block(ts, loader, args, args.sustain, args.warmup + args.steps, barrier)
        if world > 1:
            sustained["sustained_samples_per_s"] = round(sustained["sustained_samples_per_s"] * world, 1)
    out = None
    if rank == 0:
        losses = ts.out[1].float().cpu().tolist() if ts.fused is not None else []
        roof = None
        try:
            conv_ms_eager, nlaunch = conv_kernel_time_ms(ts, loader, static)
            conv_ms, graph_err, boundary_us = conv_ms_eager, None, None
            if not args.eager and ts.use_graph and not ts.split:
                try:          # the number the roofline is quoted on: the same launches timed inside the graph replay
                    conv_ms, nlaunch, boundary_us = conv_kernel_time_graph_ms(ts, loader)
                except Exception as e:   # noqa: BLE001
                    graph_err = repr(e)
            flops = GFLOP_FWD_BWD_PER_SAMPLE.get(args.size, 31.785) * 1e9 * args.bs
            ach = flops / (conv_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_PEAK_TFLOPS[args.dtype], 4), "traffic": pmc_traffic(args),
                    "traffic_unit": f"HBM bytes per step over the conv-stack launches (PMC passes of this precision: profiles/{PMC_FILE[args.dtype]}; "
                                    "algorithmic bytes of the stack in the same file)",
                    "kernel": "implicit-GEMM conv stack: conv3x3_kernel / conv3x3r_kernel / conv2x2_kernel / convp_kernel / conv_gemm2_kernel / gemm_rw_kernel / "
                              "stem_halo_x3_kernel (fwd, dgrad) + wgrad3x3_kernel / wgrad_gemm2_kernel / wgrad_reduce (weight grad)",
                    "conv_ms_per_step": round(conv_ms, 3), "conv_launches_per_step": nlaunch,
                    "conv_ms_source": ("graph replay: wall-clock stamps (ab_wall_stamp) captured around every conv-stack call, the leading stamp "
                                       "launch of each bracket (= the difference of back-to-back stamps) removed" if boundary_us is not None else "eager HIP events"),
                    "conv_ms_per_step_eager": round(conv_ms_eager, 3), "stamp_boundary_us": None if boundary_us is None else round(boundary_us, 2),
                    "conv_ms_instrument_note": ("the stamped replay reads HIGH: the ~113 one-thread stamp launches break kernel-to-kernel chaining; against "
                                                "the rocprofv3 kernel trace of the same build it read +3.5 % in round 5 (6.749 vs 6.518 ms) and "
                                                "+3.4 % in round 6 (profiles/round6_*_step_trace.txt) -- `frac` is a lower bound by that much")}
            if graph_err:
                roof["graph_stamp_error"] = graph_err
        except Exception as e:   # noqa: BLE001
            roof = {"bound": "mfma", "achieved": None, "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s", "frac": None,
                    "traffic": None, "error": repr(e)}
        base = None
        ev = None
        if world == 1 and not args.no_eval_leg:
            try:
                ev = eval_forward(args, model, static)
            except Exception as e:   # noqa: BLE001
                ev = {"error": repr(e)}
        hbm = None
        try:
            hbm = hbm_rooflines(args, loader, static, model)
        except Exception as e:   # noqa: BLE001
            hbm = {"error": repr(e)}
        dex = None
        if world == 1 and args.dataset == "HO3D" and not args.no_dexycb_leg and not args.eager:
            try:
                dex = dexycb_leg(args, device)
            except Exception as e:   # noqa: BLE001
                dex = {"error": repr(e)}
        study = None
        if world == 1 and args.dtype == "bf16x3" and not args.no_study_leg and not args.eager and not args.wgrad_1pass:
            try:
                study = wgrad_1pass_leg(args, device)
            except Exception as e:   # noqa: BLE001
                study = {"error": repr(e)}
        jpg = None
        if world == 1 and not args.no_jpeg_leg and not args.eager:
            try:
                jpg = jpeg_leg(device)
            except Exception as e:   # noqa: BLE001
                jpg = {"error": repr(e)}
        mixed = None
        if world == 1 and not args.no_mixed_leg and not args.eager and args.dtype == "bf16x3" and args.bs == 64:
            try:
                mixed = mixed_leg(args)
            except Exception as e:   # noqa: BLE001
                mixed = {"error": repr(e)}
        dropin = None
        if world == 1 and not args.no_dropin_leg and not args.eager and args.dtype == "bf16x3" and args.dataset == "HO3D":
            try:
                torch.cuda.synchronize()
                dropin = dropin_leg(args, ms)
            except Exception as e:   # noqa: BLE001
                dropin = {"error": repr(e)}
        rccl1 = None
        if world == 1 and not args.no_rccl_leg and not args.eager and not args.rccl_single_rank:
            try:
                torch.cuda.synchronize()
                rccl1 = rccl_leg(args, ms)
            except Exception as e:   # noqa: BLE001
                rccl1 = {"error": repr(e)}
        roof["peak_note"] = ("dense bf16 MFMA peak / 3 passes" if args.dtype == "bf16x3" else "dense MFMA peak of the operand type")
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline(args, cfg)
        out = {"metric": f"synth samples/sec (render+fwd+bwd) {args.size}x{args.size} bs={args.bs}", "value": round(value, 2), "unit": "samples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype + ("+wgrad_bf16_1pass" if args.wgrad_1pass else ""),
               "dtype_note": DTYPE_NOTE[args.dtype] + ("; STUDY: weight gradients from bf16 (hi-plane) operands in one MFMA pass -- below the "
                                                        "reference's operand precision, not a parity configuration" if args.wgrad_1pass else ""),
               "data": "synthetic (seeded stand-in meshes/textures/grasps; random-init weights)",
               "config": {"workload": f"train_artiboost HO3Dv2-clasbased (HybridBaseline/ResNet-34, 22x28x{args.size // 8}x{args.size // 8} heat-map) "
                                      f"+ online CCV render 512->{args.size}, per-GPU batch {args.bs}, {args.dataset}-like objects",
                          "global_batch": args.bs * world, "image": args.size,
                          "parallelism": f"dp{world}" + ("+multi_rank_schedule_over_one_rank_rccl" if args.rccl_single_rank else ""),
                          "graph": not args.eager, "shared_devices": bool(shared),
                          "render_overlap": bool(getattr(args, "render_overlap", False))},
               "final_loss": losses[5] if losses else None,
               "roofline": roof, "cpu_baseline": base,
               "roofline_hbm_kernels": hbm,            # the HBM-bound halves (render chain, soft-argmax head) against the 8 TB/s roof
               "sustained": sustained,                 # same process, >= --sustain seconds after the timed block (+ observed sclk / power)
               "configs1_eval_forward": ev,            # BASELINE configs[1] (forward only) with its own roofline; `bench.py --eval` prints it as the line
               "configs4_dexycb_1gpu": dex,
               "dropin_epoch_pass": dropin,            # the reference's own loop through the anakin.* aliases (the drop-in boundary)
               "rccl_one_rank_schedule": rccl1,        # the N > 1 code path with the real collective over one rank (the box has one GPU)
               "mixed_real_synth_step": mixed,         # the reference's MixedDataset batch (real .jpg frames + synthetic) as one training step
               "real_half_jpeg_decode": jpg,           # .jpg files -> frames on the device, beside Pillow on this host
               "study_wgrad_bf16_1pass": study}        # precision / speed study beside the headline (one-pass weight gradients), see its note            # BASELINE configs[4]'s per-GPU step (DexYCB scenes + SymCornerLoss) on this one GPU
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
    if world > 1 or args.rccl_single_rank:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

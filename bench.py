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


def ref_cfg(size):
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"] = [size, size]
    cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [size // 8, size // 8]
    cfg["ARCH"]["BACKBONE"]["PRETRAINED"] = False       # ImageNet weights are a download; random init (stated in `data`)
    return cfg


def build_everything(args, rank, world, device):
    import torch
    from artiboost_amd import registry as R
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.synth import ArtiBoostLoader
    from artiboost_amd.train import TrainStep
    cfg = ref_cfg(args.size)
    arch_cfg = dict(cfg["ARCH"], COMPUTE_DTYPE=args.dtype, DEVICE=device, INIT_SEED=cfg["TRAIN"]["MANUAL_SEED"])
    model = Arch({"ARCH": arch_cfg}, R.build_arch_model_list(arch_cfg, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=cfg["TRAIN"]["LR"], max_norm=cfg["TRAIN"]["GRAD_CLIP"], model=hb)
    assets = SceneAssets(args.dataset, seed=1)
    mgr = dict(cfg["MANAGER"], EPOCH=cfg["TRAIN"]["EPOCH"])
    synth_len = args.bs * world * max(args.steps + args.warmup + 2, 4)
    loader = ArtiBoostLoader.from_assets(assets, mgr, cfg["DATA_PRESET"], args.bs, synth_len, device=device,
                             compute_dtype=hb.net.dtype, random_seed=cfg["TRAIN"]["MANUAL_SEED"], rank=rank, world_size=world)
    loader.prepare()
    static = loader.new_static_batch()
    loader.load_batch(static, 0)
    group = torch.distributed.group.WORLD if world > 1 else None
    model.train()
    # N > 1: the north-star schedule -- batch t+1 is rendered on a second stream while step t's last gradient range is
    # all-reduced and its clip + Adam runs (on one GPU that concurrency measured slower than the single queue, so it stays off)
    overlap = args.pipeline_opt or (world > 1 and not args.no_render_overlap)
    ts = TrainStep(model, crit, opt, static, use_graph=not args.eager, dist_group=group, renderer=loader,
                   pipeline_render=("opt" if overlap else args.pipeline))
    args.render_overlap = bool(overlap)
    ts.static = static
    return cfg, model, crit, opt, loader, ts, static


def conv_kernel_time_ms(ts, loader, static, iters=3):
    """Average per-step time of the conv-stack MFMA kernels, measured with HIP events on the compute stream by
    running the step eagerly with events around every conv launch (kernels.py hooks); agrees with the per-kernel
    durations of profiles/round1_*_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this same command)."""
    import torch
    from artiboost_amd import kernels as K
    names = ["conv2d_fwd", "conv2d_stem_fwd", "conv2d_dgrad", "conv2d_wgrad", "conv2d_stem_wgrad",
             "conv2d_fwd_x3", "conv2d_stem_fwd_x3", "conv2d_dgrad_x3", "conv2d_wgrad_x3", "conv2d_stem_wgrad_x3"]
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

    total = 0.0
    try:
        for n in names:
            setattr(K, n, wrap(orig[n]))
        for it in range(iters):
            spans.clear()
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
            total += sum(max(a.elapsed_time(b) - empty_ms, 0.0) for a, b in spans)
            nl = len(spans)
    finally:
        for n in names:
            setattr(K, n, orig[n])
    return total / iters, nl


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


PMC_FILE = {"bf16": "round1_pmc_hbm_traffic.json", "bf16x3": "round2_h_pmc_hbm_traffic.json", "f32": "none"}


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
        for v in leaf.values():
            if getattr(v, "grad", None) is not None:
                v.grad = None
        preds = lo.hybrid_forward(leaf, batch, [args.size, args.size], 22, 28, 0, training=True)
        total, _, _ = lo.criterion(preds, batch)
        total.backward()
        if names is None:
            names = [k for k, v in leaf.items() if v.dtype.is_floating_point and getattr(v, "grad", None) is not None]
            ms = [torch.zeros_like(leaf[k]) for k in names]
            vs = [torch.zeros_like(leaf[k]) for k in names]
        lo.clip_and_adam([leaf[k].detach() for k in names], [leaf[k].grad for k in names], ms, vs, it + 1)
        t_learn += time.time() - t0
    tot = t_render + t_learn
    return {"value": round(n * iters / tot, 3), "unit": "samples/s", "cores": cores, "kind": "port",
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
            torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    cfg, model, crit, opt, loader, ts, static = build_everything(args, rank, world, device)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

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
    out = None
    if rank == 0:
        losses = ts.out[1].float().cpu().tolist() if ts.fused is not None else []
        roof = None
        try:
            conv_ms, nlaunch = conv_kernel_time_ms(ts, loader, static)
            flops = GFLOP_FWD_BWD_PER_SAMPLE.get(args.size, 31.785) * 1e9 * args.bs
            ach = flops / (conv_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                    "frac": round(ach / MFMA_PEAK_TFLOPS[args.dtype], 4), "traffic": pmc_traffic(args),
                    "traffic_unit": f"HBM bytes per step over the conv-stack launches (PMC passes of this precision: profiles/{PMC_FILE[args.dtype]}; "
                                    "algorithmic bytes of the stack in the same file)",
                    "kernel": "implicit-GEMM conv stack: conv3x3_kernel / conv_gemm2_kernel / conv_gemm_kernel (fwd, dgrad) + "
                              "wgrad3x3_kernel / wgrad_gemm2_kernel / wgrad_reduce (weight grad)",
                    "conv_ms_per_step": round(conv_ms, 3), "conv_launches_per_step": nlaunch}
        except Exception as e:   # noqa: BLE001
            roof = {"bound": "mfma", "achieved": None, "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s", "frac": None,
                    "traffic": None, "error": repr(e)}
        base = None
        roof["peak_note"] = ("dense bf16 MFMA peak / 3 passes" if args.dtype == "bf16x3" else "dense MFMA peak of the operand type")
        if world == 1 and not args.no_cpu_baseline:
            base = cpu_baseline(args, cfg)
        out = {"metric": f"synth samples/sec (render+fwd+bwd) {args.size}x{args.size} bs={args.bs}", "value": round(value, 2), "unit": "samples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
               "dtype_note": DTYPE_NOTE[args.dtype],
               "data": "synthetic (seeded stand-in meshes/textures/grasps; random-init weights)",
               "config": {"workload": f"train_artiboost HO3Dv2-clasbased (HybridBaseline/ResNet-34, 22x28x{args.size // 8}x{args.size // 8} heat-map) "
                                      f"+ online CCV render 512->{args.size}, per-GPU batch {args.bs}, {args.dataset}-like objects",
                          "global_batch": args.bs * world, "image": args.size, "parallelism": f"dp{world}",
                          "graph": not args.eager, "shared_devices": bool(shared),
                          "render_overlap": bool(getattr(args, "render_overlap", False))},
               "final_loss": losses[5] if losses else None,
               "roofline": roof, "cpu_baseline": base}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

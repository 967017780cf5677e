#!/usr/bin/env python
"""train/train_artiboost.py of the reference on this build: epochs of online-synthesised batches through the fused,
graph-replayed step, the per-(object, view, grasp) validation metric fed every step, and the mining update of the CCV
sampling weights at the end of every epoch (artiboost_loader.step_eval).  Single GPU, or one process per GPU:

    python tools/train_artiboost.py --epochs 3 --synth-len 2048
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_artiboost.py ...

Real frames (HO3D / DexYCB) are downloads: this driver trains on the synthetic half only (the reference with an empty real
set); artiboost_amd.realdata.MixedLoader mixes in a real source when one is available.  --dump DIR writes the mining state
per epoch in the reference's own file layout (ccv_cache.record_artiboost_loader); --resume-epoch N reads it back."""
import argparse
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default=os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml"))
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--synth-len", type=int, default=1024, help="synthetic samples per epoch over all ranks (SYNTH_FACTOR x len(real) in the reference)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dataset", default="", help="asset set: HO3D (4 objects) or DexYCB (21); default: from the config's OBJ_ORIGIN_DATASET")
    ap.add_argument("--dump", default="")
    ap.add_argument("--resume-epoch", type=int, default=0)
    ap.add_argument("--dtype", default="bf16x3", choices=["bf16x3", "f32", "bf16"], help="bf16x3 = the reference's fp32-grade precision")
    ap.add_argument("--per-step-eval", action="store_true", help="feed the evaluator after every batch as the reference does "
                    "(a host synchronisation per step) instead of once per epoch from device-side records")
    args = ap.parse_args()

    import torch
    import yaml
    from artiboost_amd import ccv_cache, registry as R
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.metrics import Evaluator
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.synth import ArtiBoostLoader
    from artiboost_amd.train import DeferredEpochMetrics, TrainStep

    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from artiboost_amd.train import rccl_env_defaults
        rccl_env_defaults()
        torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    dev = f"cuda:{local}"
    cfg = yaml.safe_load(open(args.cfg))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [args.size, args.size], [args.size // 8, args.size // 8]
    cfg["ARCH"]["BACKBONE"]["PRETRAINED"] = False
    arch = dict(cfg["ARCH"], COMPUTE_DTYPE=args.dtype, DEVICE=dev, INIT_SEED=cfg["TRAIN"]["MANUAL_SEED"])
    model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    evaluator = Evaluator(cfg, R.build_evaluator_metric_list(cfg["EVALUATOR"], preset_cfg=cfg["DATA_PRESET"]))
    hb = model.model_list[0]
    opt = FusedClipAdam(model.models_params, lr=cfg["TRAIN"]["LR"], max_norm=cfg["TRAIN"]["GRAD_CLIP"], model=hb)
    dataset = args.dataset or cfg["MANAGER"].get("OBJ_ENGINE", {}).get("OBJ_ORIGIN_DATASET", "HO3D")
    loader = ArtiBoostLoader.from_assets(SceneAssets(dataset, seed=1), dict(cfg["MANAGER"], EPOCH=args.epochs), cfg["DATA_PRESET"], args.bs,
                             args.synth_len, device=dev, compute_dtype=hb.net.dtype, random_seed=cfg["TRAIN"]["MANUAL_SEED"],
                             rank=rank, world_size=world)
    ckpt = os.path.join(args.dump, "checkpoints", "checkpoint") if args.dump else ""
    if args.resume_epoch:        # utils/io_utils.py:19-44,73-93: <model type>.pth.tar + train_param.pth.tar, and the mining state
        hb.load_state_dict(torch.load(os.path.join(ckpt, "HybridBaseline.pth.tar"), map_location=dev))
        opt.load_state_dict(torch.load(os.path.join(ckpt, "train_param.pth.tar"), map_location=dev)["optimizer"])
        ccv_cache.resume_artiboost_loader(loader, args.resume_epoch, args.dump)
        rng_file = os.path.join(ckpt, "loader_rng.pkl")      # the loader's generator states + the host RNG streams of the loss draws
        if os.path.exists(rng_file):
            import pickle
            import random
            import numpy as np
            with open(rng_file, "rb") as f:
                st = pickle.load(f)
            loader.rng.bit_generator.state = st["numpy"]
            loader.torch_gen.set_state(st["torch"])
            random.setstate(st["py"]); np.random.set_state(st["np_global"]); torch.set_rng_state(st["torch_global"])
    model.train()
    ts = rec = None
    for epoch in range(args.resume_epoch, args.epochs):
        loader.prepare()                                            # sample CCV triplets by weight, generate the epoch's poses
        evaluator.reset_all()
        if ts is None:
            static = loader.new_static_batch()
            loader.load_batch(static, 0)
            ts = TrainStep(model, crit, opt, static, use_graph=True, renderer=loader,
                           dist_group=torch.distributed.group.WORLD if world > 1 else None)
            rec = None if args.per_step_eval else DeferredEpochMetrics(ts, len(loader), evaluator)
        t0 = time.time()
        for bi in range(len(loader)):
            ts.stage(loader, bi)
            preds, losses, _ = ts()                                 # render -> forward -> losses -> backward -> [all-reduce] -> clip + Adam
            if rec is not None:
                rec.collect()
            else:
                evaluator.feed_all(ts.predictions(), ts.static, ts.fused.losses_dict() if ts.fused is not None else losses)
        if rec is not None:
            rec.flush(evaluator)
        torch.cuda.synchronize()
        dt = time.time() - t0
        loader.step_eval(epoch, evaluator)                          # mining: re-weight the CCV space from this epoch's errors
        if rank == 0:
            w = loader.sample_weight_map
            print(f"epoch {epoch}: {len(loader) * args.bs * world / dt:8.0f} samples/s | {evaluator} | "
                  f"weights min {float(w.min()):.2f} max {float(w.max()):.2f} | explored {float(loader.occurence_map.float().mean()):.3f}", flush=True)
            if args.dump:
                ccv_cache.record_artiboost_loader(loader, epoch, args.dump)
                os.makedirs(ckpt, exist_ok=True)
                torch.save(hb.state_dict(), os.path.join(ckpt, "HybridBaseline.pth.tar"))          # the reference's keys and layouts
                torch.save({"epoch": epoch + 1, "optimizer": opt.state_dict(), "scheduler": {}}, os.path.join(ckpt, "train_param.pth.tar"))
                import pickle
                import random
                import numpy as np
                with open(os.path.join(ckpt, "loader_rng.pkl"), "wb") as f:      # without these a resumed run replays epoch 0's draws
                    pickle.dump({"numpy": loader.rng.bit_generator.state, "torch": loader.torch_gen.get_state(), "py": random.getstate(),
                                 "np_global": np.random.get_state(), "torch_global": torch.get_rng_state()}, f)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

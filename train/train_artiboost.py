#!/usr/bin/env python
"""The reference's training command (train/train_artiboost.py of lixiny/ArtiBoost) with the SAME command line,

    python train/train_artiboost.py --cfg config/ho3dv2_clasbased_artiboost_mi355x.yaml \\
        --gpu_id 0,1,2,3 --gpu_render_id 0,1,2,3 --batch_size 256 --exp_id my_run [--resume exp/<dir>] [--snapshot 50]

on the MI355X path.  The objects are built through the `anakin.*` import paths and the reference's keyword signatures
(train_artiboost.py:108-190: Recorder, Summarizer, builder.build_arch_model_list, Arch, build_optimizer / build_scheduler,
Criterion, Evaluator, builder.build_dataset, ArtiBoostLoader) and the epoch follows the reference's order (prepare -> epoch pass ->
scheduler.step -> step_eval -> recorder); what differs is HOW a step is issued and how GPUs are used:

* the step (render -> forward -> losses -> backward -> clip -> Adam) replays as hipGraphs (`artiboost_amd.train.TrainStep`) and
  the evaluator is fed once per epoch from device-side records -- the same updates and the same metrics as the reference's
  per-batch Python loop (tests/test_gpu_synth.py), without a host round trip per step;
* `--gpu_id a,b,c,..`: where the reference wraps the model in nn.DataParallel (one process, batch split over the GPUs,
  train_artiboost.py:131,249-257), this script starts ONE PROCESS PER GPU under torch.distributed.run: each rank renders and
  learns its share of `--batch_size` (batch_size / N samples per step; per-rank BatchNorm statistics as under DataParallel), the
  flat gradient is averaged with RCCL while the backward is still running, every rank applies the same mining update.
  `--gpu_render_id` is accepted and unused: rendering is in-process on the training GPU.

Extra flags: --dtype {bf16x3,f32,bf16}, --synth_len N (samples per epoch when the real set is absent), --size S (square image),
--no_refiner (drop the MANAGER.REFINER block: its GrabNet checkpoint is a download).
--dry-launch: parse, start the ranks (gloo), report the world size and exit (the launcher test)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _pop_flag(argv, name, has_value=True, default=None):
    """Remove `--name [value]` from argv (the reference's parser must not see this script's own flags)."""
    if name not in argv:
        return default
    i = argv.index(name)
    if not has_value:
        del argv[i]
        return True
    val = argv[i + 1]
    del argv[i:i + 2]
    return val


def _gpu_ids(argv):
    if "--gpu_id" in argv:
        return [g for g in argv[argv.index("--gpu_id") + 1].split(",") if g != ""]
    return []


def main():
    argv = sys.argv[1:]
    dry = _pop_flag(argv, "--dry-launch", has_value=False, default=False)
    dtype = _pop_flag(argv, "--dtype", default="bf16x3")
    synth_len = _pop_flag(argv, "--synth_len")
    size = _pop_flag(argv, "--size")
    no_refiner = _pop_flag(argv, "--no_refiner", has_value=False, default=False)
    shared = _pop_flag(argv, "--allow-shared-devices", has_value=False, default=False)      # ranks over gloo on one device: tests only
    gpus = _gpu_ids(argv)
    if len(gpus) > 1 and "WORLD_SIZE" not in os.environ:
        # one process per listed GPU (train_artiboost.py:249-257 sets CUDA_VISIBLE_DEVICES the same way for its DataParallel)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if not shared:
            env["CUDA_VISIBLE_DEVICES"] = ",".join(gpus)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={len(gpus)}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if dry:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.distributed.init_process_group("gloo")
            t = torch.ones(1)
            torch.distributed.all_reduce(t)
            world = int(t.item())
            torch.distributed.destroy_process_group()
        if rank == 0:
            import json
            print(json.dumps({"dry_launch": True, "n_gpus": world, "gpu_id": gpus}), flush=True)
        return
    # the launcher (or the single-GPU run) already narrowed CUDA_VISIBLE_DEVICES: the reference's parser must not narrow it again
    if world > 1 and "--gpu_id" in argv:
        i = argv.index("--gpu_id")
        del argv[i:i + 2]
    sys.argv = [sys.argv[0]] + argv

    ngpu = torch.cuda.device_count()
    if world > ngpu and not shared:
        raise SystemExit(f"train_artiboost.py: {world} ranks but {ngpu} visible GPUs")
    local = local % max(ngpu, 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world > ngpu:
            torch.distributed.init_process_group("gloo")
        else:
            from artiboost_amd.train import rccl_env_defaults
            rccl_env_defaults()
            torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))

    from anakin.artiboost import ArtiBoostLoader
    from anakin.criterions.criterion import Criterion
    from anakin.datasets.hodata import ho_collate
    from anakin.metrics.evaluator import Evaluator
    from anakin.models.arch import Arch
    from anakin.opt import arg, cfg
    from anakin.opt_extra import data_generation_manager_parse
    from anakin.utils import builder
    from anakin.utils.misc import TrainMode
    from anakin.utils.netutils import build_optimizer, build_scheduler
    from anakin.utils.recorder import Recorder
    from anakin.utils.summarizer import Summarizer
    from artiboost_amd.train import DeferredEpochMetrics, TrainStep

    import random
    import numpy as np
    seed = cfg["TRAIN"]["MANUAL_SEED"]                                     # set_all_seeds (train_artiboost.py:240)
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    time_f = time.time()
    dev = f"cuda:{local}"
    arg.device = dev
    if arg.batch_size % world:
        raise SystemExit(f"--batch_size {arg.batch_size} does not split over {world} GPUs")
    per_rank = arg.batch_size // world                                      # DataParallel splits the batch the same way
    if size:
        cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [int(size)] * 2, [int(size) // 8] * 2
    if synth_len:
        cfg["MANAGER"]["SYNTH_LEN"] = int(synth_len)
    if no_refiner:
        cfg["MANAGER"].pop("REFINER", None)      # the GrabNet checkpoint (assets/GrabNet/refinenet.pt) is a download
    arch = cfg["ARCH"] if isinstance(cfg["ARCH"], dict) else cfg["ARCH"][0]
    arch.update(COMPUTE_DTYPE=dtype, DEVICE=dev, INIT_SEED=seed)

    recorder = Recorder(arg.exp_id, cfg, rank=rank, time_f=time_f)
    summarizer = Summarizer(arg.exp_id, cfg, rank=rank, time_f=time_f)
    model = Arch(cfg, model_list=builder.build_arch_model_list(cfg["ARCH"], preset_cfg=cfg["DATA_PRESET"]))
    recorder.record_arch_graph(model)
    optimizer = build_optimizer(model.models_params, **cfg["TRAIN"])
    scheduler = build_scheduler(optimizer, **cfg["TRAIN"])
    grad_clip = cfg["TRAIN"].get("GRAD_CLIP")
    if hasattr(optimizer, "max_norm"):
        optimizer.max_norm = grad_clip          # the reference's clip_grad_norm_ (train_artiboost.py:91-92), inside the fused clip + Adam pass
    criterion = Criterion(cfg, loss_list=builder.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    evaluator = Evaluator(cfg, metrics_list=builder.build_evaluator_metric_list(cfg["EVALUATOR"], preset_cfg=cfg["DATA_PRESET"]))
    cfg["MANAGER"].update({"VAL_FREQ": cfg["TRAIN"].get("EVAL_FREQ", 5), "VAL_START_EPOCH": cfg["TRAIN"].get("VAL_START_EPOCH", 0),
                           "EPOCH": cfg["TRAIN"]["EPOCH"]})
    train_data = builder.build_dataset(cfg["DATASET"]["TRAIN"], preset_cfg=cfg["DATA_PRESET"])
    arg_extra = data_generation_manager_parse()
    # A real training set that is present (DATA_ROOT holds the download): the reference's MixedDataset (mixed_dataset.py:5-37) -- every batch
    # holds its share of real frames (decoded and augmented on the device) and of the epoch's synthetic samples.  The synthetic loader then
    # renders only ITS share of the per-rank batch; realdata.MixedLoader assembles the batches.
    from artiboost_amd.realdata import MixedLoader, RealBatcher, ThreadedPrefetcher
    real_len = len(train_data)
    synth_share = per_rank
    if real_len > 0:
        total_synth = int(synth_len) if synth_len else int(cfg["MANAGER"].get("SYNTH_FACTOR", 0.0) * real_len)
        synth_share = MixedLoader.n_synth_for(per_rank, real_len, total_synth) if total_synth > 0 else 0
        if synth_share == 0:
            raise SystemExit("a real training set with no synthetic share (SYNTH_FACTOR 0): not the ArtiBoost loop; use a plain training script")
    loader = ArtiBoostLoader(train_data, arg=arg, arg_extra=arg_extra, cfg=cfg["MANAGER"], cfg_dataset=cfg["DATASET"],
                             cfg_preset=cfg["DATA_PRESET"], time_f=time_f, batch_size=synth_share, shuffle=True,
                             **({"synth_len": total_synth} if real_len > 0 else {}),
                             num_workers=int(arg.workers), pin_memory=True, drop_last=arg.drop_last, collate_fn=ho_collate,
                             random_seed=seed, rank=rank, world_size=world,
                             compute_dtype=("u8n" if getattr(getattr(model.model_list[0], "net", None), "x3", False) and os.environ.get("AB_IMAGE_PLANE", "u8n") == "u8n"
                                            else getattr(getattr(model.model_list[0], "net", None), "dtype", torch.float32)))
    mixed = None
    if real_len > 0:
        tr = cfg["DATASET"]["TRAIN"]
        mixed = MixedLoader(RealBatcher(train_data, cfg["DATA_PRESET"], aug=bool(tr.get("AUG", False)), aug_param=tr.get("AUG_PARAM") or None, device=dev,
                                        compute_dtype=("u8n" if loader.image_plane == "u8n" else loader.dtype), seed=seed, num_workers=int(arg.workers) or None),
                            loader, per_rank, seed=seed, rank=rank, world_size=world, want_chw=False, reuse_buffers=4)      # TrainStep copies a batch in
    epoch0 = 0
    if arg.resume:
        epoch0 = recorder.resume_checkpoints(model, optimizer, scheduler, arg.resume, resume_epoch=arg.resume_epoch or None)
        recorder.resume_artiboost_loader(loader, epoch0, arg.resume)

    # ---- the TEST pass of the reference (train_artiboost.py:112-122,224-240): eval-mode epoch over DATASET.TEST every
    # --test_freq epochs, and once (then exit) under --evaluate.  HO3D / DexYCB are downloads: when the real test set is absent
    # the pass runs over a VAL-mode synthetic epoch instead (OVGSet.val(), ovg_set.py:108-118: uniform over the admissible CCV
    # triplets, no augmentation re-weighting) and says so in its record.
    test_data = builder.build_dataset(cfg["DATASET"]["TEST"], preset_cfg=cfg["DATA_PRESET"])
    test_state = {}

    def test_pass(epoch_idx):
        model.eval()
        evaluator.reset_all()
        n = 0
        with torch.no_grad():
            if len(test_data) > 0:
                from artiboost_amd.realdata import RealBatcher
                if "real" not in test_state:
                    test_state["real"] = RealBatcher(test_data, cfg["DATA_PRESET"], aug=False, device=dev,
                                                    compute_dtype=("u8n" if loader.image_plane == "u8n" else loader.dtype), seed=seed)
                rb = test_state["real"]
                # shuffle=True, drop_last=False (train_artiboost.py:113-121).  Every rank evaluates the WHOLE test set (the reference's
                # DataParallel process does): no cross-rank reduction of the evaluator is needed and rank 0's record covers every frame;
                # the order comes from a dedicated generator, not from the global numpy state the ranks may have advanced differently
                idxs = np.random.default_rng(seed + 7919 * (epoch_idx + 1)).permutation(len(test_data))
                source = (rb.batch(idxs[i:i + per_rank].tolist()) for i in range(0, len(idxs), per_rank))
                what = f"{len(test_data)} frames of DATASET.TEST ({cfg['DATASET']['TEST']['TYPE']})"
            else:
                saved = loader.epoch, loader.cursor, loader.synth_len
                admissible = int((~loader.blacklist_map).sum())
                loader.synth_len = min(loader.synth_len, admissible // (per_rank * world) * per_rank * world)
                loader.prepare(is_train=False)
                source = iter(loader)                                             # rendered batches with the reference's keys, one at a time
                what = (f"DATASET.TEST ({cfg['DATASET']['TEST']['TYPE']}) is absent (a download): {len(loader) * per_rank} synthetic "
                        f"val-mode CCV samples per rank instead")
            for batch in source:
                pd = model(batch)
                predicts = {}
                for key in pd:
                    predicts.update(pd[key])
                _, losses = criterion.compute_losses(predicts, batch)
                evaluator.feed_all(predicts, batch, losses)
                n += int(batch["root_joint"].shape[0])
            if len(test_data) == 0:
                loader.epoch, loader.cursor, loader.synth_len = saved
        recorder.record_evaluator(evaluator, epoch_idx, TrainMode.TEST)
        summarizer.summarize_evaluator(evaluator, epoch_idx, train_mode=TrainMode.TEST)
        if rank == 0:
            print(f"test  {epoch_idx}: {n} samples per rank | {what} | {evaluator}", flush=True)

    n_epochs = cfg["TRAIN"]["EPOCH"]
    if arg.evaluate:
        n_epochs = max(epoch0, 0) + 1             # "enter into the train loop" once (train_artiboost.py:144-145)
        epoch0 = n_epochs - 1
    ts = rec = None
    for epoch_idx in range(epoch0, n_epochs):
        if not arg.evaluate:
            loader.prepare()
            if len(loader) == 0:
                raise SystemExit("empty epoch: SYNTH_LEN / the real set give fewer samples than one batch per rank")
            model.train()
            evaluator.reset_all()
            if mixed is not None:
                # real + synthetic batches: assembled two batches ahead by a worker thread on its own stream (the role of the reference's
                # DataLoader worker processes: host-side ground truth / parsing / planning AND the device-side decode / augment / render),
                # the graph-replayed step copies each into its static inputs
                mixed.update()
                t0 = time.time()
                nb = 0
                for batch in ThreadedPrefetcher(mixed, depth=2):
                    if ts is None:
                        ts = TrainStep(model, criterion, optimizer, {k: v.clone() for k, v in batch.items()}, use_graph=True, renderer=None,
                                       dist_group=torch.distributed.group.WORLD if world > 1 else None)
                        rec = DeferredEpochMetrics(ts, len(mixed), evaluator) if ts.fused is not None else None
                    preds, losses, _ = ts(batch)
                    nb += 1
                    if rec is not None:
                        rec.collect()
                    else:
                        evaluator.feed_all(ts.predictions(), ts.static, losses)
                        summarizer.summarize_losses(ts.fused.losses_dict() if ts.fused is not None else losses)
                if rec is not None:
                    rec.flush(evaluator, summarizer=summarizer)
                torch.cuda.synchronize()
                dt = time.time() - t0
                scheduler.step()
                loader.step_eval(epoch_idx=epoch_idx, evaluator=evaluator)
                recorder.record_checkpoints(model, optimizer, scheduler, epoch_idx, arg.snapshot)
                recorder.record_evaluator(evaluator, epoch_idx, TrainMode.TRAIN)
                summarizer.summarize_evaluator(evaluator, epoch_idx, train_mode=TrainMode.TRAIN)
                recorder.record_artiboost_loader(loader, epoch_idx)
                if rank == 0:
                    print(f"epoch {epoch_idx}: {nb * per_rank * world / dt:8.0f} samples/s on {world} GPU(s), {mixed.n_real} real + {mixed.n_synth} synthetic "
                          f"per batch of {per_rank} | {evaluator}", flush=True)
                if arg.test_freq > 0 and epoch_idx % arg.test_freq == arg.test_freq - 1:
                    test_pass(epoch_idx)
                continue
            if ts is None:
                static = loader.new_static_batch()
                loader.load_batch(static, 0)
                ts = TrainStep(model, criterion, optimizer, static, use_graph=True, renderer=loader,
                               dist_group=torch.distributed.group.WORLD if world > 1 else None,
                               pipeline_render="opt" if world > 1 else False)
                rec = DeferredEpochMetrics(ts, len(loader), evaluator) if ts.fused is not None else None
            t0 = time.time()
            ts.prime(loader, 0)
            for bi in range(len(loader)):
                ts.stage(loader, bi)
                preds, losses, _ = ts()
                if rec is not None:
                    rec.collect()
                else:
                    evaluator.feed_all(ts.predictions(), ts.static, losses)
                    summarizer.summarize_losses(ts.fused.losses_dict() if ts.fused is not None else losses)
            if rec is not None:
                rec.flush(evaluator, summarizer=summarizer)      # per-step loss scalars (epoch_pass: summarizer.summarize_losses) from the records
            torch.cuda.synchronize()
            dt = time.time() - t0
            scheduler.step()
            loader.step_eval(epoch_idx=epoch_idx, evaluator=evaluator)
            recorder.record_checkpoints(model, optimizer, scheduler, epoch_idx, arg.snapshot)
            recorder.record_evaluator(evaluator, epoch_idx, TrainMode.TRAIN)
            summarizer.summarize_evaluator(evaluator, epoch_idx, train_mode=TrainMode.TRAIN)
            recorder.record_artiboost_loader(loader, epoch_idx)
            if rank == 0:
                print(f"epoch {epoch_idx}: {len(loader) * per_rank * world / dt:8.0f} samples/s on {world} GPU(s) | {evaluator}", flush=True)
        if arg.evaluate or (arg.test_freq > 0 and epoch_idx % arg.test_freq == arg.test_freq - 1):
            test_pass(epoch_idx)
            if arg.evaluate:
                break
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""Times the REFERENCE-SHAPED training loop (train/train_artiboost.py:46-105 epoch_pass: `for batch in artiboost_loader` ->
arch_model(batch) -> compute_losses -> feed_all -> zero_grad -> backward -> clip_grad_norm_ -> optimizer.step), built through
the `anakin.*` import paths and the reference's keyword signatures, at the benchmark geometry (B = 64, 256 x 256).

This is what a user of the reference gets without touching their loop; `bench.py` times the same work issued as replayed
hipGraphs through `TrainStep`.     python tools/bench_dropin.py [--steps 30] [--batch_size 64] [--size 256]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--no-feed", action="store_true", help="skip evaluator.feed_all (a host read per step in the reference)")
    a = ap.parse_args()
    sys.argv = ["train_artiboost.py", "--cfg", os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml"),
                "--batch_size", str(a.batch), "--gpu_render_id", "0"]
    from anakin.artiboost import ArtiBoostLoader
    from anakin.criterions.criterion import Criterion
    from anakin.datasets.hodata import ho_collate
    from anakin.metrics.evaluator import Evaluator
    from anakin.models.arch import Arch
    from anakin.opt import arg, cfg
    from anakin.opt_extra import data_generation_manager_parse
    from anakin.utils import builder
    from anakin.utils.netutils import build_optimizer

    t0 = time.time()
    import random
    import numpy as np
    seed = cfg["TRAIN"]["MANUAL_SEED"]                    # set_all_seeds of train_artiboost.py:240
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [a.size, a.size], [a.size // 8, a.size // 8]
    cfg["MANAGER"]["SYNTH_LEN"] = a.batch * (a.steps + a.warmup)
    model = Arch(cfg, model_list=builder.build_arch_model_list(cfg["ARCH"], preset_cfg=cfg["DATA_PRESET"])).to(arg.device)
    optimizer = build_optimizer(model.models_params, **cfg["TRAIN"])
    grad_clip = cfg["TRAIN"].get("GRAD_CLIP")
    criterion = Criterion(cfg, loss_list=builder.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    evaluator = Evaluator(cfg, metrics_list=builder.build_evaluator_metric_list(cfg["EVALUATOR"], preset_cfg=cfg["DATA_PRESET"]))
    cfg["MANAGER"].update({"VAL_FREQ": 5, "VAL_START_EPOCH": 15, "EPOCH": 1})
    train_data = builder.build_dataset(cfg["DATASET"]["TRAIN"], preset_cfg=cfg["DATA_PRESET"])
    loader = ArtiBoostLoader(train_data, arg=arg, arg_extra=data_generation_manager_parse(), cfg=cfg["MANAGER"], cfg_dataset=cfg["DATASET"],
                             cfg_preset=cfg["DATA_PRESET"], time_f=t0, batch_size=arg.batch_size, shuffle=True, num_workers=0,
                             pin_memory=True, drop_last=True, collate_fn=ho_collate, random_seed=cfg["TRAIN"]["MANUAL_SEED"])
    loader.prepare()
    model.train()
    evaluator.reset_all()
    t_start, n, last, bar = None, 0, None, ""
    for batch_idx, batch in enumerate(loader):
        if batch_idx == a.warmup:
            torch.cuda.synchronize()
            t_start = time.perf_counter()
        predicts = {}
        for v in model(batch).values():
            predicts.update(v)
        final_loss, losses = criterion.compute_losses(predicts, batch)
        if not a.no_feed:
            evaluator.feed_all(predicts, batch, losses)
        optimizer.zero_grad()
        final_loss.backward()
        if grad_clip is not None:
            torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
        optimizer.step()
        last = final_loss
        if not a.no_feed:
            bar = f"{evaluator}"              # the progress string of train_artiboost.py:105 (etqdm description), every iteration
        if batch_idx >= a.warmup:
            n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t_start
    measures = evaluator.get_measures_all_striped() if not a.no_feed else {}      # the epoch-end read: every batch fed
    print(json.dumps({"loop": "reference epoch_pass (eager, drop-in)", "batch": a.batch, "size": a.size, "steps": n,
                      "ms_per_step": round(dt / n * 1e3, 3), "samples_per_s": round(a.batch * n / dt, 1),
                      "feed_all": not a.no_feed, "segment_graphs": os.environ.get("AB_SEGMENT_GRAPHS", "1") != "0",
                      "image_plane": str(getattr(loader, "image_plane", None)), "progress": bar[:120],
                      "epoch_final_loss_mean": measures.get("LossesMetric", {}).get("final_loss"), "final_loss": float(last)}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""The reference's evaluation / submission command (train/submit_reload.py of lixiny/ArtiBoost) with the same command line:

    python train/submit_reload.py --cfg config/eval_ho3dv2_regbased_artiboost_cpu.yaml --batch_size 8 --submit_dump [--gpu_id 0]

Objects are built through the `anakin.*` import paths in the reference's order (submit_reload.py:26-79): Recorder(eval_only),
SubmitEpochPass.build(arg.submit_dataset), builder.build_dataset(TEST), Arch, Criterion, Evaluator, then one eval-mode pass that
feeds the evaluator and writes the HO3D CodaLab prediction file next to the evaluation record.

The HO3D / DexYCB test sets are downloads: with `./data` absent the TEST dataset is empty and the pass writes an empty
prediction file.  `--random_frames N` runs the same plumbing over N seeded stand-in frames instead, `--ignore_pretrained` clears
ARCH.PRETRAINED (the eval YAMLs name downloaded checkpoints) -- with both, BASELINE.json configs[0] (regbased HOPRegNet, CPU, batch
size 8, forward only; the reference's own config_eval/eval_ho3dv2_regbased_artiboost.yaml) runs end to end without data."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _random_batches(n, bs, size, seed):
    """Seeded stand-in frames with the keys HOdata yields for a test frame (anakin/datasets/hodata.py:315-450)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    out = []
    for s in range(0, n, bs):
        b = min(bs, n - s)
        root = torch.tensor([0.0, 0.0, 0.6]) + 0.05 * torch.randn((b, 3), generator=g)
        K = torch.tensor([[617.0, 0, size[0] / 2], [0, 617.0, size[1] / 2], [0, 0, 1.0]]).repeat(b, 1, 1)
        j3, c3 = 0.05 * torch.randn((b, 21, 3), generator=g), 0.05 * torch.randn((b, 8, 3), generator=g)

        def proj(p):        # pinhole projection of root-relative points
            q = torch.matmul(K, (p + root[:, None]).permute(0, 2, 1)).permute(0, 2, 1)
            return q[..., :2] / q[..., 2:3]
        out.append({"image": torch.rand((b, 3, size[1], size[0]), generator=g) - 0.5, "cam_intr": K,
                    "root_joint": root, "corners_can": 0.05 * (torch.rand((b, 8, 3), generator=g) * 2 - 1),
                    "joints_3d": j3, "corners_3d": c3, "joints_2d": proj(j3), "corners_2d": proj(c3),
                    "joints_vis": torch.ones(b, 21), "corners_vis": torch.ones(b, 8),
                    "is_synth": torch.zeros(b, dtype=torch.bool), "obj_idx": torch.ones(b, dtype=torch.long)})
    return out


def main():
    argv = sys.argv[1:]
    nrand = 0
    if "--random_frames" in argv:
        i = argv.index("--random_frames")
        nrand = int(argv[i + 1])
        del argv[i:i + 2]
    ignore_pretrained = "--ignore_pretrained" in argv       # ARCH.PRETRAINED of the eval YAMLs names a downloaded checkpoint
    if ignore_pretrained:
        argv.remove("--ignore_pretrained")
    sys.argv = [sys.argv[0]] + argv

    import random
    import numpy as np
    import torch
    from anakin.criterions.criterion import Criterion
    from anakin.datasets.hodata import ho_collate
    from anakin.metrics.evaluator import Evaluator
    from anakin.models.arch import Arch
    from anakin.opt import arg, cfg
    from anakin.submit import SubmitEpochPass
    from anakin.utils import builder
    from anakin.utils.misc import TrainMode
    from anakin.utils.recorder import Recorder

    if ignore_pretrained:
        for a in (cfg["ARCH"] if isinstance(cfg["ARCH"], list) else [cfg["ARCH"]]):
            a["PRETRAINED"] = ""
    time_f = time.time()
    seed = cfg.get("TRAIN", {}).get("MANUAL_SEED", 1)
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    cfg_name = os.path.basename(arg.cfg).split(".")[0]
    recorder = Recorder(f"submit_{cfg_name}", cfg, rank=0, time_f=time_f, eval_only=True)
    submit_epoch_pass = SubmitEpochPass.build(arg.submit_dataset, cfg={"DUMP": bool(arg.submit_dump), "TRUE_ROOT": bool(arg.true_root)})
    batch_size = arg.batch_size or cfg.get("TRAIN", {}).get("BATCH_SIZE", 8)
    if nrand:
        test_loader = _random_batches(nrand, batch_size, cfg["DATA_PRESET"]["IMAGE_SIZE"], seed)
    else:
        test_data = builder.build_dataset(cfg["DATASET"]["TEST"], preset_cfg=cfg["DATA_PRESET"])
        test_loader = torch.utils.data.DataLoader(test_data, batch_size=batch_size, shuffle=False, num_workers=0, drop_last=False,
                                                  collate_fn=ho_collate) if len(test_data) else []
    model = Arch(cfg, model_list=builder.build_arch_model_list(cfg["ARCH"], preset_cfg=cfg["DATA_PRESET"]))
    if arg.resume:        # <exp>/checkpoints/checkpoint/<ModelType>.pth.tar (utils/io_utils.py:46-70)
        for m in model.model_list:
            path = os.path.join(arg.resume, "checkpoints", "checkpoint", f"{type(m).__name__}.pth.tar")
            m.load_state_dict(torch.load(path, map_location="cpu"))
    criterion = Criterion(cfg, loss_list=builder.build_criterion_loss_list(cfg.get("CRITERION", []), cfg["DATA_PRESET"], LAMBDAS=cfg.get("LAMBDAS", [])))
    evaluator = Evaluator(cfg, metrics_list=builder.build_evaluator_metric_list(cfg["EVALUATOR"], cfg["DATA_PRESET"], arg=arg))
    name = cfg_name + ("_trueroot" if arg.true_root else "") + "_SUBMIT" + (".json" if not arg.resume_epoch else f"_epoch{arg.resume_epoch}.json")
    dump_path = os.path.join(recorder.dump_path, name)
    t0 = time.time()
    with torch.no_grad():
        model.eval()
        joints = submit_epoch_pass(epoch_idx=0, data_loader=test_loader, arch_model=model, criterion=criterion if cfg.get("CRITERION") else None,
                                   evaluator=evaluator, rank=0, dump_path=dump_path, draw_path=os.path.join(recorder.dump_path, "rendered_image"))
    dt = time.time() - t0
    recorder.record_evaluator(evaluator, 0, TrainMode.TEST)
    dev = next(model.parameters()).device
    print(f"submit: {len(joints)} frames on {dev} in {dt:.2f} s ({len(joints) / max(dt, 1e-9):.1f} frames/s) | {evaluator} | {dump_path if arg.submit_dump else 'no dump'}")


if __name__ == "__main__":
    main()

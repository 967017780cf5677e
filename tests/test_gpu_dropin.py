"""The reference's training loop, constructed ONLY through the reference's import paths and keyword signatures
(train/train_artiboost.py:9-22 imports, :108-190 construction, :46-105 epoch_pass, :205-237 epoch loop), running on the
HIP path: `for batch in artiboost_loader` -> arch_model(batch) -> compute_losses -> feed_all -> zero_grad -> backward ->
clip_grad_norm_ -> optimizer.step, two epochs with step_eval, then checkpoint + resume through the Recorder."""
import os
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("segment_graphs", ["1", "0"])
def test_reference_epoch_pass_sequence_two_epochs(tmp_path, monkeypatch, segment_graphs):
    # "0": the loop issued kernel by kernel through torch.ops.artiboost_hip.* (no hipGraph inside the model): same sequence, same checks
    monkeypatch.setenv("AB_SEGMENT_GRAPHS", segment_graphs)
    monkeypatch.setattr(sys, "argv", ["train_artiboost.py", "--cfg", os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml"),
                                      "--batch_size", "8", "--gpu_render_id", "0", "--exp_id", "default"])
    for m in [k for k in sys.modules if k == "anakin.opt" or k == "anakin.opt_extra"]:
        del sys.modules[m]
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

    time_f = time.time()
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [64, 64], [8, 8]      # small geometry: a test, not a benchmark
    cfg["MANAGER"]["SYNTH_LEN"] = 32
    cfg["TRAIN"]["EPOCH"] = 2
    recorder = Recorder(arg.exp_id, cfg, root_path=str(tmp_path / "exp"), time_f=time_f)
    summarizer = Summarizer(arg.exp_id, cfg, tensorboard_path=str(tmp_path / "runs"), time_f=time_f)
    # ---- model / optimizer / scheduler (train_artiboost.py:127-137)
    model_list = builder.build_arch_model_list(cfg["ARCH"], preset_cfg=cfg["DATA_PRESET"])
    model = Arch(cfg, model_list=model_list)
    recorder.record_arch_graph(model)
    model = torch.nn.DataParallel(model).to(arg.device)                       # train_artiboost.py:131, as written there
    optimizer = build_optimizer(model.module.models_params if hasattr(model, "module") else model.models_params, **cfg["TRAIN"])
    scheduler = build_scheduler(optimizer, **cfg["TRAIN"])
    grad_clip = cfg["TRAIN"].get("GRAD_CLIP")
    # ---- criterion / evaluator (:148-160)
    criterion = Criterion(cfg, loss_list=builder.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    evaluator = Evaluator(cfg, metrics_list=builder.build_evaluator_metric_list(cfg["EVALUATOR"], preset_cfg=cfg["DATA_PRESET"]))
    # ---- artiboost (:163-190)
    cfg["MANAGER"].update({"VAL_FREQ": cfg["TRAIN"]["EVAL_FREQ"], "VAL_START_EPOCH": cfg["TRAIN"]["VAL_START_EPOCH"], "EPOCH": cfg["TRAIN"]["EPOCH"]})
    train_data = builder.build_dataset(cfg["DATASET"]["TRAIN"], preset_cfg=cfg["DATA_PRESET"])
    assert len(train_data) == 0            # ./data is a download: the real share of the mix is empty here
    arg_extra = data_generation_manager_parse()
    artiboost_loader = ArtiBoostLoader(train_data, arg=arg, arg_extra=arg_extra, cfg=cfg["MANAGER"], cfg_dataset=cfg["DATASET"],
                                       cfg_preset=cfg["DATA_PRESET"], time_f=time_f, batch_size=arg.batch_size, shuffle=True,
                                       num_workers=int(arg.workers), pin_memory=True, drop_last=arg.drop_last, collate_fn=ho_collate,
                                       random_seed=cfg["TRAIN"]["MANUAL_SEED"])
    w0 = artiboost_loader.sample_weight_map.clone()
    first_losses = []
    for epoch_idx in range(cfg["TRAIN"]["EPOCH"]):
        artiboost_loader.prepare()
        model.train()
        evaluator.reset_all()
        nb = 0
        for batch_idx, batch in enumerate(artiboost_loader):
            predict_arch_dict = model(batch)
            predicts = {}
            for key in predict_arch_dict.keys():
                predicts.update(predict_arch_dict[key])
            final_loss, losses = criterion.compute_losses(predicts, batch)
            evaluator.feed_all(predicts, batch, losses)
            summarizer.summarize_losses(losses)
            optimizer.zero_grad()
            final_loss.backward()
            if grad_clip is not None:
                total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
                assert torch.isfinite(total_norm)
            optimizer.step()
            optimizer.zero_grad()
            first_losses.append(float(final_loss))
            nb += 1
        assert nb == len(artiboost_loader) == 4
        scheduler.step()
        artiboost_loader.step_eval(epoch_idx=epoch_idx, evaluator=evaluator)
        recorder.record_checkpoints(model, optimizer, scheduler, epoch_idx, arg.snapshot)
        recorder.record_evaluator(evaluator, epoch_idx, TrainMode.TRAIN)
        summarizer.summarize_evaluator(evaluator, epoch_idx, train_mode=TrainMode.TRAIN)
        recorder.record_artiboost_loader(artiboost_loader, epoch_idx)
    assert all(np.isfinite(first_losses)) and len(first_losses) == 8
    assert not torch.equal(artiboost_loader.sample_weight_map, w0)          # mining re-weighted the visited CCV cells
    assert type(optimizer).__name__ == "FusedClipAdam" and optimizer.max_norm is None
    # ---- resume (train_artiboost.py:143-146,192-194) into fresh objects: identical weights, optimizer step, mining state
    model2 = Arch(cfg, model_list=builder.build_arch_model_list(cfg["ARCH"], preset_cfg=cfg["DATA_PRESET"])).to(arg.device)
    optimizer2 = build_optimizer(model2.models_params, **cfg["TRAIN"])
    scheduler2 = build_scheduler(optimizer2, **cfg["TRAIN"])
    epoch = recorder.resume_checkpoints(model2, optimizer2, scheduler2, recorder.dump_path)
    assert epoch == 2
    sd1, sd2 = model.module.model_list[0].state_dict(), model2.model_list[0].state_dict()
    assert all(torch.equal(sd1[k], sd2[k]) for k in sd1)
    loader2 = ArtiBoostLoader(train_data, arg=arg, arg_extra=arg_extra, cfg=cfg["MANAGER"], cfg_dataset=cfg["DATASET"],
                              cfg_preset=cfg["DATA_PRESET"], time_f=time_f, batch_size=arg.batch_size, random_seed=cfg["TRAIN"]["MANUAL_SEED"])
    recorder.resume_artiboost_loader(loader2, epoch, recorder.dump_path)
    assert torch.equal(loader2.sample_weight_map, artiboost_loader.sample_weight_map)
    assert torch.equal(loader2.occurence_map, artiboost_loader.occurence_map)
    assert os.path.exists(os.path.join(recorder.dump_path, "checkpoints", "checkpoint", "random_state.pkl"))


def test_train_script_with_the_reference_command_line(tmp_path):
    """train/train_artiboost.py -- the reference's command line on the graph-replayed step: two epochs, checkpoint, resume."""
    import subprocess
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["TRAIN"]["EPOCH"] = 2
    # the metric list of the reference's shipped YAML (config/ho3dv2_clasbased_jlol_artiboost2.yaml): PCK and the 2-D visual metric
    # are fed from TrainStep.predictions() after every step, the rest once per epoch from device-side records
    cfg["EVALUATOR"] += [{"TYPE": "Hand3DPCKMetric", "VAL_MIN": 0.0, "VAL_MAX": 0.05, "STEPS": 20},
                         {"TYPE": "Obj3DPCKMetric", "VAL_MIN": 0.0, "VAL_MAX": 0.05, "STEPS": 20}, {"TYPE": "Vis2DMetric", "NCOL": 2, "NROW": 2}]
    y = tmp_path / "cfg.yaml"
    y.write_text(yaml.dump(cfg))
    cmd = [sys.executable, os.path.join(ROOT, "train", "train_artiboost.py"), "--cfg", str(y), "--gpu_id", "0", "--gpu_render_id", "0",
           "--batch_size", "8", "--exp_id", "t", "--snapshot", "1", "--synth_len", "32", "--size", "64", "--test_freq", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("epoch ")]
    assert len(lines) == 2 and "final_loss" in lines[-1] and "hand3d pck" in lines[-1]
    # the TEST pass of train_artiboost.py:224-240 ran after epoch 1 (--test_freq 2) and left its record
    tests_ = [l for l in out.stdout.splitlines() if l.startswith("test ")]
    assert len(tests_) == 1 and "final_loss" in tests_[0]
    exp = [d for d in os.listdir(tmp_path / "exp") if d.startswith("t_")]
    assert len(exp) == 1
    assert (tmp_path / "exp" / exp[0] / "evaluations" / "test_eval.txt").exists()
    ck = tmp_path / "exp" / exp[0] / "checkpoints" / "checkpoint"
    assert (ck / "HybridBaseline.pth.tar").exists() and (ck / "train_param.pth.tar").exists()
    # resume: nothing left to train (epoch 2 of 2), but every piece must load
    out = subprocess.run(cmd + ["--resume", str(tmp_path / "exp" / exp[0])], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    # --evaluate --resume: ONE test pass, no training, the checkpoint it read is left alone (train_artiboost.py:143-146,207,237-240)
    before = (ck / "HybridBaseline.pth.tar").read_bytes()
    out = subprocess.run(cmd + ["--resume", str(tmp_path / "exp" / exp[0]), "--evaluate", "--exp_id", "ev"], capture_output=True, text=True,
                         timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("epoch ")]
    assert len([l for l in out.stdout.splitlines() if l.startswith("test ")]) == 1
    assert (ck / "HybridBaseline.pth.tar").read_bytes() == before


def test_train_script_dtype_flag_reaches_the_loader(tmp_path):
    """`--dtype bf16`: the loader's image buffer must be built in the network's dtype (a mismatch made the stem read bf16 weights
    as fp32: finite garbage).  One epoch in bf16 trains to a loss in the range of the bf16x3 run on the same seeds."""
    import re
    import subprocess
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["TRAIN"]["EPOCH"] = 1
    y = tmp_path / "cfg.yaml"
    y.write_text(yaml.dump(cfg))
    vals = {}
    for dt in ("bf16x3", "bf16"):
        cmd = [sys.executable, os.path.join(ROOT, "train", "train_artiboost.py"), "--cfg", str(y),
               "--gpu_id", "0", "--batch_size", "8", "--exp_id", dt, "--synth_len", "32", "--size", "64", "--dtype", dt, "--no_refiner"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
        assert out.returncode == 0, out.stderr[-3000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("epoch 0")][0]
        vals[dt] = float(re.search(r"final_loss[^0-9]*([0-9.eE+-]+)", line).group(1))
    assert np.isfinite(list(vals.values())).all()
    assert abs(vals["bf16"] - vals["bf16x3"]) <= 0.05 * vals["bf16x3"], vals


def test_train_script_two_ranks_on_a_shared_device(tmp_path):
    """`--gpu_id 0,0 --allow-shared-devices`: the script launches two ranks (gloo, one GPU) -- per-rank share of --batch_size, rank
    slices of the epoch, averaged flat gradient with the staged backward, one mining update from all ranks' errors."""
    import subprocess
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["TRAIN"]["EPOCH"] = 2
    y = tmp_path / "cfg.yaml"
    y.write_text(yaml.dump(cfg))
    cmd = [sys.executable, os.path.join(ROOT, "train", "train_artiboost.py"), "--cfg", str(y), "--gpu_id", "0,0", "--allow-shared-devices",
           "--batch_size", "16", "--exp_id", "two", "--synth_len", "64", "--size", "64"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("epoch ")]
    assert len(lines) == 2 and "on 2 GPU(s)" in lines[-1] and "final_loss" in lines[-1]
    exp = [d for d in os.listdir(tmp_path / "exp") if d.startswith("two_")]
    assert len(exp) == 1            # only rank 0 records


def test_train_script_mixes_a_real_ho3d_download_with_the_synthetic_share(tmp_path):
    """train/train_artiboost.py with DATA_ROOT pointing at a (miniature) HO3D v2 download: the reference's MixedDataset loop -- every batch
    holds real frames (read by datasets.HO3D, decoded from their .png files and augmented on the device) and synthetic samples rendered for
    the same step; two epochs, the TEST pass over the real evaluation split, checkpoint."""
    import subprocess
    import yaml
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ho3d_fake_tree as T
    data = tmp_path / "data"
    T.build(str(data), seed=7)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["TRAIN"]["EPOCH"] = 2
    cfg["DATASET"]["TRAIN"]["DATA_ROOT"] = cfg["DATASET"]["TEST"]["DATA_ROOT"] = str(data)
    cfg["DATA_PRESET"].update(USE_CACHE=True, FILTER_NO_CONTACT=False, FILTER_THRESH=0.0)
    y = tmp_path / "cfg.yaml"
    y.write_text(yaml.dump(cfg))
    cmd = [sys.executable, os.path.join(ROOT, "train", "train_artiboost.py"), "--cfg", str(y), "--gpu_id", "0", "--gpu_render_id", "0",
           "--batch_size", "8", "--exp_id", "m", "--snapshot", "1", "--synth_len", "8", "--size", "64", "--test_freq", "2", "--workers", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("epoch ")]
    assert len(lines) == 2 and "4 real + 4 synthetic per batch of 8" in lines[-1] and "final_loss" in lines[-1]
    tests_ = [l for l in out.stdout.splitlines() if l.startswith("test ")]
    assert len(tests_) == 1 and "5 frames of DATASET.TEST (HO3D)" in tests_[0]
    exp = [d for d in os.listdir(tmp_path / "exp") if d.startswith("m_")]
    assert (tmp_path / "exp" / exp[0] / "checkpoints" / "checkpoint" / "HybridBaseline.pth.tar").exists()
    assert os.path.isdir(tmp_path / "common" / "cache" / "HO3D")

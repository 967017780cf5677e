"""Recorder / Summarizer with the reference's names, call contract and on-disk layout (anakin/utils/recorder.py:28-226,
anakin/utils/io_utils.py:19-93, anakin/utils/summarizer.py:12-60), as called by train/train_artiboost.py:108-237.

Files under `<root>/<exp_id>_<timestamp>/`:
  dump_cfg.yaml
  checkpoints/checkpoint/<ModelType>.pth.tar     state_dict with the reference's keys and tensor layouts
  checkpoints/checkpoint/train_param.pth.tar     {"epoch", "optimizer", "scheduler"}
  checkpoints/checkpoint/random_state.pkl        RandomState namedtuple (python / numpy / torch / torch.cuda streams)
                                                 + the loader's own generators under checkpoints/checkpoint/loader_rng.pkl
  evaluations/{train,val,test}_eval.txt
  artiboost/sample_weight/<epoch>_train.pkl, artiboost/occurence_map/<epoch>.pkl, artiboost/shutdown   (ccv_cache.py)
Host-side bookkeeping only; nothing here touches the hot path."""
import os
import pickle
import random
import shutil
import sys
import time
from collections import namedtuple
from pprint import pformat

import numpy as np
import torch
import yaml

from . import ccv_cache
from .registry import TrainMode

# anakin/utils/misc.py: the pickled class of random_state.pkl (the `anakin.utils.misc` alias module re-exports it, so files
# written here and by the reference unpickle on either side)
RandomState = namedtuple("RandomState", ["torch_rng_state", "torch_cuda_rng_state", "torch_cuda_rng_state_all",
                                         "numpy_rng_state", "random_rng_state"])
RandomState.__module__ = "anakin.utils.misc"

_PREFIX = {TrainMode.TRAIN: "train", TrainMode.VAL: "val", TrainMode.TEST: "test"}


def _models(model):
    return model.module.model_list if hasattr(model, "module") else model.model_list


class Recorder:
    def __init__(self, exp_id, cfg, root_path="./exp", rank=None, time_f=None, eval_only=False):
        self.timestamp = time.strftime("%Y_%m%d_%H%M_%S", time.localtime(time_f if time_f else time.time()))
        self.exp_id, self.cfg, self.rank, self.eval_only = exp_id, cfg, rank, eval_only
        self.dump_path = os.path.join(root_path, f"{exp_id}_{self.timestamp}")
        if not self.rank:
            os.makedirs(self.dump_path, exist_ok=True)
            with open(os.path.join(self.dump_path, "dump_cfg.yaml"), "w") as f:
                yaml.dump(self.cfg, f, Dumper=yaml.Dumper, sort_keys=False)
            with open(os.path.join(self.dump_path, "command.txt"), "w") as f:
                f.write(" ".join(sys.argv) + "\n")

    # ---- checkpoints (io_utils.save_states / load_arch / load_train_param / load_random_state)
    def record_checkpoints(self, model, optimizer, scheduler, epoch, snapshot):
        if self.rank:
            return
        root = os.path.join(self.dump_path, "checkpoints")
        fold = os.path.join(root, "checkpoint")
        os.makedirs(fold, exist_ok=True)
        for m in _models(model):
            torch.save(m.state_dict(), os.path.join(fold, f"{type(m).__name__}.pth.tar"))
        cuda = torch.cuda.is_available()
        rs = RandomState(torch_rng_state=torch.get_rng_state(),
                         torch_cuda_rng_state=torch.cuda.get_rng_state() if cuda else None,
                         torch_cuda_rng_state_all=torch.cuda.get_rng_state_all() if cuda else None,
                         numpy_rng_state=np.random.get_state(), random_rng_state=random.getstate())
        with open(os.path.join(fold, "random_state.pkl"), "wb") as f:
            pickle.dump(rs, f)
        torch.save({"epoch": epoch + 1, "optimizer": optimizer.state_dict(), "scheduler": scheduler.state_dict()},
                   os.path.join(fold, "train_param.pth.tar"))
        if snapshot and (epoch + 1) % snapshot == 0:
            dst = os.path.join(root, f"checkpoint_{epoch + 1}")
            shutil.rmtree(dst, ignore_errors=True)
            shutil.copytree(fold, dst)

    def resume_checkpoints(self, model, optimizer, scheduler, resume_path, resume_epoch=None):
        fold = os.path.join(resume_path, "checkpoints", f"checkpoint_{resume_epoch}" if resume_epoch else "checkpoint")
        par = torch.load(os.path.join(fold, "train_param.pth.tar"), map_location="cpu", weights_only=False)
        optimizer.load_state_dict(par["optimizer"])
        scheduler.load_state_dict(par["scheduler"])
        rsp = os.path.join(fold, "random_state.pkl")
        if os.path.exists(rsp):
            with open(rsp, "rb") as f:
                rs = pickle.load(f)
            random.setstate(rs.random_rng_state)
            np.random.set_state(rs.numpy_rng_state)
            torch.set_rng_state(rs.torch_rng_state)
            if torch.cuda.is_available() and rs.torch_cuda_rng_state is not None:
                torch.cuda.set_rng_state(rs.torch_cuda_rng_state)
        for m in _models(model):
            sd = torch.load(os.path.join(fold, f"{type(m).__name__}.pth.tar"), map_location="cpu", weights_only=False)
            if sd and next(iter(sd)).startswith("module."):
                sd = {k.split(".", 1)[1]: v for k, v in sd.items()}
            m.load_state_dict(sd)
        return par["epoch"]

    # ---- evaluations
    def record_evaluator(self, evaluator, epoch, train_mode):
        if self.rank:
            return
        path = os.path.join(self.dump_path, "evaluations")
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, f"{_PREFIX[train_mode]}_eval.txt"), "a") as f:
            f.write(f"Epoch {epoch} evaluator msg:\n {pformat(evaluator.get_measures_all_striped())}\n\n")

    def record_arch_graph(self, model):
        """recorder.py:171-177 draws the model DAG with networkx / matplotlib; here the edge list goes to arch.txt."""
        if self.rank:
            return
        m = model.module if hasattr(model, "module") else model
        with open(os.path.join(self.dump_path, "arch.txt"), "w") as f:
            for name, v in getattr(m, "models", {}).items():
                f.write(f"{name} <- {v['previous']}\n")

    # ---- CCV mining state (+ the loader's generator states, which the reference keeps inside its global RNG streams)
    def record_artiboost_loader(self, artiboost_loader, epoch):
        if self.rank:
            return
        ccv_cache.record_artiboost_loader(artiboost_loader, epoch, self.dump_path)
        fold = os.path.join(self.dump_path, "checkpoints", "checkpoint")
        os.makedirs(fold, exist_ok=True)
        with open(os.path.join(fold, "loader_rng.pkl"), "wb") as f:
            pickle.dump({"numpy": artiboost_loader.rng.bit_generator.state, "torch": artiboost_loader.torch_gen.get_state()}, f)

    def resume_artiboost_loader(self, artiboost_loader, resume_epoch, resume_path):
        ccv_cache.resume_artiboost_loader(artiboost_loader, resume_epoch, resume_path)
        p = os.path.join(resume_path, "checkpoints", "checkpoint", "loader_rng.pkl")
        if os.path.exists(p):
            with open(p, "rb") as f:
                st = pickle.load(f)
            artiboost_loader.rng.bit_generator.state = st["numpy"]
            artiboost_loader.torch_gen.set_state(st["torch"])


class Summarizer:
    """TensorBoard scalars when torch.utils.tensorboard is importable (it needs the `tensorboard` wheel); a jsonl file of
    the same records otherwise."""

    def __init__(self, exp_id, cfg, tensorboard_path="./runs", rank=None, time_f=None):
        self.timestamp = time.strftime("%Y_%m%d_%H%M_%S", time.localtime(time_f if time_f else time.time()))
        self.exp_id, self.cfg, self.rank, self._n_iter = exp_id, cfg, rank, 0
        self.tb_writer = self._fallback = None
        if not self.rank:
            path = os.path.join(tensorboard_path, f"{exp_id}_{self.timestamp}")
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.tb_writer = SummaryWriter(path)
            except Exception:       # noqa: BLE001  (tensorboard missing)
                os.makedirs(path, exist_ok=True)
                self._fallback = open(os.path.join(path, "scalars.jsonl"), "a")

    def _scalar(self, tag, value, step):
        if self.tb_writer is not None:
            self.tb_writer.add_scalar(tag, value, step)
        elif self._fallback is not None:
            import json
            self._fallback.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
            self._fallback.flush()

    def summarize_evaluator(self, evaluator, epoch, train_mode):
        if self.rank:
            return
        for k, v in evaluator.get_measures_all_striped(return_losses=False).items():
            for k_, v_ in (v.items() if isinstance(v, dict) else [("", v)]):
                self._scalar(f"{k}/{_PREFIX[train_mode]}/{k_}".rstrip("/"), v_, epoch)
        # summarizer.py:41-47: the Vis* metrics' images (BGR arrays) -> TensorBoard, or PNG files next to the scalars
        for k, img in (evaluator.dump_images() if hasattr(evaluator, "dump_images") else {}).items():
            if img is None:
                continue
            if self.tb_writer is not None:
                self.tb_writer.add_image(f"{k}/{_PREFIX[train_mode]}", img[:, :, ::-1].copy(), epoch, dataformats="HWC")
            elif self._fallback is not None:
                from PIL import Image
                Image.fromarray(img[:, :, ::-1].copy()).save(os.path.join(os.path.dirname(self._fallback.name),
                                                                            f"{k}_{_PREFIX[train_mode]}_{epoch}.png"))

    def summarize_losses(self, losses):
        if self.rank:
            return
        for k, v in losses.items():
            if v is not None:
                self._scalar("Loss" if k == "final_loss" else f"Losses/{k}", float(v), self._n_iter)
        self._n_iter += 1

    def clear_summarizer(self):
        self._n_iter = 0

"""Command line of the reference (anakin/opt.py:9-77, anakin/opt_extra.py:6-19): `parse()` returns the (arg, cfg,
custom_arg_string) triple that `from anakin.opt import arg, cfg` provides at import time."""
import argparse
import os

import torch

from .registry import update_config


def build_parser():
    p = argparse.ArgumentParser(description="ANAKIN SKYWALKER")
    p.add_argument("--vis_toc", type=float, default=5)
    p.add_argument("--cfg", help="experiment configure file name", type=str, default=None)
    p.add_argument("--exp_id", default="default", type=str, help="Experiment ID")
    p.add_argument("--resume", help="resume training from exp", type=str, default=None)
    p.add_argument("--workers", help="worker number from data loader", type=int, default=20)
    p.add_argument("--batch_size", help="batch size of exp, will replace bs in cfg file if is given", type=int, default=None)
    p.add_argument("--evaluate", help="evaluate the network (ignore training)", action="store_true")
    p.add_argument("--gpu_id", type=str, default=None, help="override enviroment var CUDA_VISIBLE_DEVICES")
    p.add_argument("--snapshot", default=50, type=int, help="How often to take a snapshot of the model (0 = never)")
    p.add_argument("--test_freq", type=int, default=5, help="How often to test, 1 for always -1 for never")
    p.add_argument("--gpu_render_port", type=str, default="34567")
    p.add_argument("--resume_epoch", help="resume from the given epoch", type=int, default=0)
    p.add_argument("--submit_dataset", type=str, default="hodata")
    p.add_argument("--filter_unseen_obj_idxs", type=int, nargs="+", default=[])
    p.add_argument("--true_root", action="store_true", help="use GT hand root")
    p.add_argument("--true_bone_scale", action="store_true", help="use GT bone length")
    p.add_argument("--submit_dump", action="store_true", help="whether to save json for benchmark")
    p.add_argument("--postprocess_fit_mesh", action="store_true")
    p.add_argument("--postprocess_fit_mesh_ik", type=str, choices=["iknet", "iksolver"], default="iknet")
    p.add_argument("--postprocess_fit_mesh_use_fitted_joints", action="store_true")
    p.add_argument("--use_pseudo_hand_root", action="store_true")
    p.add_argument("--postprocess_draw", action="store_true")
    p.add_argument("--postprocess_draw_path", type=str, default=None)
    return p


def parse(argv=None):
    arg, custom = build_parser().parse_known_args(argv)
    if arg.resume:
        cfg = update_config(os.path.join(arg.resume, "dump_cfg.yaml"))
    else:
        cfg = update_config(arg.cfg) if arg.cfg else dict()
        cfg["FILE_NAME"] = arg.cfg
    if arg.gpu_id is not None:
        os.environ["CUDA_VISIBLE_DEVICES"] = arg.gpu_id
    arg.device = "cuda" if torch.cuda.is_available() else "cpu"
    if "TRAIN" in cfg:
        if arg.batch_size:
            cfg["TRAIN"]["BATCH_SIZE"] = arg.batch_size
        else:
            arg.batch_size = cfg["TRAIN"]["BATCH_SIZE"]
        arg.drop_last = cfg["TRAIN"].get("DROP_LAST", True)
    arg.gpus = list(range(torch.cuda.device_count()))
    return arg, cfg, custom


def data_generation_manager_parse(custom_arg_string=None):
    """opt_extra.py:6-19.  --gpu_render_id names the GPUs of the reference's render servers; rendering is in-process here, so
    it is optional instead of required."""
    p = argparse.ArgumentParser()
    p.add_argument("--opg_batch_size", type=int, default=256)
    p.add_argument("--opg_num_workers", type=int, default=20)
    p.add_argument("--gpu_render_id", type=str, default="0")
    p.add_argument("--synth_root", type=str, default="/dev/shm/anakin")
    a, _ = p.parse_known_args(custom_arg_string)
    a.ovg_batch_size, a.ovg_num_workers = a.opg_batch_size, a.opg_num_workers
    return a

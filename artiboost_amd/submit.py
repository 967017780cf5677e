"""Evaluation / submission pass: anakin/submit/hodata_submit_epoch_pass.py:21-156 and
submit_epoch_pass.py -- eval-mode forward over a loader, metrics, and the HO3D-v2 CodaLab prediction file.

The hand-mesh fitting (`fit_mesh`: an IK network + licensed MANO assets) and the matplotlib drawings of the reference
are not part of this build; without fitting the reference writes zero vertices, as here."""
import json
import os
import zipfile

import numpy as np
import torch


class HOSubmitEpochPass:
    """SubmitEpochPass.reg("hodata").  cfg: {"DUMP": bool, "TRUE_ROOT": bool} (arg.true_root in the reference)."""

    def __init__(self, cfg=None):
        cfg = cfg or {}
        self.dump = cfg.get("DUMP", True)
        self.true_root = cfg.get("TRUE_ROOT", False)

    @staticmethod
    def get_order_idxs():
        reorder_idxs = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]
        return reorder_idxs, np.argsort(reorder_idxs)

    @staticmethod
    def dump_json(pred_out_path, xyz_pred_list, verts_pred_list, codalab=True):
        """hodata_submit_epoch_pass.py:34-56: [joints, verts] rounded to 5 decimals; zipped for CodaLab (zipfile instead
        of the `zip` binary, same single flat entry as `zip -j`)."""
        def roundall(rows):
            return [[round(val, 5) for val in row] for row in rows]
        xyz = [roundall(x.tolist()) for x in xyz_pred_list]
        verts = [roundall(x.tolist()) for x in verts_pred_list]
        with open(pred_out_path, "w") as fo:
            json.dump([xyz, verts], fo)
        if codalab:
            with zipfile.ZipFile(pred_out_path.replace(".json", ".zip"), "w", zipfile.ZIP_DEFLATED) as z:
                z.write(pred_out_path, os.path.basename(pred_out_path))

    def __call__(self, epoch_idx, data_loader, arch_model, criterion=None, evaluator=None, rank=0, dump_path=None, draw_path=None):
        arch_model.eval()
        if evaluator:
            evaluator.reset_all()
        res_joints, res_verts = [], []
        _, unorder = self.get_order_idxs()
        with torch.no_grad():
            for batch in data_loader:
                predicts = {}
                for preds in arch_model(batch).values():
                    predicts.update(preds)
                if criterion:
                    _, losses = criterion.compute_losses(predicts, batch)
                else:
                    losses = {}
                if self.true_root:
                    predicts["joints_3d_abs"][:, 0] = batch["root_joint"].to(predicts["joints_3d_abs"].device)
                if evaluator:
                    evaluator.feed_all(predicts, batch, losses)
                # HO3D submission convention (hodata_submit_epoch_pass.py:141-145): undo the joint reorder, OpenGL axes
                pj = predicts["joints_3d_abs"].detach().cpu()[:, unorder].clone()
                pj[:, :, 0] = -pj[:, :, 0]
                joints = [-val.numpy()[0] for val in pj.split(1)]
                res_joints.extend(joints)
                res_verts.extend([np.zeros((778, 3))] * len(joints))
        if self.dump and dump_path:
            self.dump_json(dump_path, res_joints, res_verts, codalab=True)
        return res_joints


class SubmitEpochPass:
    """anakin/submit/submit_epoch_pass.py: `SubmitEpochPass.build(arg.submit_dataset, cfg=None)` (train/submit_reload.py:38)."""
    _types = {"hodata": HOSubmitEpochPass}

    @classmethod
    def build(cls, type, cfg=None):
        if type not in cls._types:
            raise NotImplementedError(f"SubmitEpochPass of type {type} is not implemented")
        return cls._types[type](cfg)

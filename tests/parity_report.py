"""Deviation of each precision mode from the reference goldens (tests/golden/learner_*.npz): max abs error of the
predictions, relative error of the losses, worst gradient-norm ratio.  Run on the GPU box:  python tests/parity_report.py"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):   # test infrastructure: may use oracle/
    sys.path.insert(0, p)
from test_gpu_learner import build          # noqa: E402
from gen_batch import make_batch            # noqa: E402

for tag in ("g224", "g256"):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"learner_{tag}.npz"))
    size, heat, depth, B, seed = [int(x) for x in g["meta"]]
    for dtype in sys.argv[1:] or ["f32", "bf16x3", "bf16"]:
        model, crit, params = build(size, heat, dtype, seed)
        hb = model.model_list[0]
        batch = make_batch(B, size, seed + 100)
        model.eval()
        with torch.no_grad():
            preds = model(batch)["HybridBaseline"]
        ev = {k: float(np.abs(preds[k].cpu().numpy() - g[f"eval.pred.{k}"]).max())
              for k in ("joints_3d_abs", "corners_3d_abs", "2d_uvd", "box_rot_rotmat")}
        model.train()
        preds = model(batch)["HybridBaseline"]
        tr = {k: float(np.abs(preds[k].detach().cpu().numpy() - g[f"train.pred.{k}"]).max())
              for k in ("joints_3d_abs", "corners_3d_abs", "2d_uvd", "box_rot_rotmat")}
        random.seed(seed + 7)
        torch.manual_seed(seed + 7)
        total, losses = crit.compute_losses(preds, batch)
        lr = {k: float(abs(losses[k].detach().cpu().numpy().reshape(-1)[0] / g[f"loss.{k}"].reshape(-1)[0] - 1))
              for k in ("joints_3d_loss", "corners_3d_loss", "joint_ord_loss", "part_ord_loss", "scene_ord_loss", "final_loss")}
        total.backward()
        grads = hb.store.reference_state_dict(grads=True)
        ref = dict(zip([str(n) for n in g["grad.names"]], g["grad.norms"]))
        gr = max(abs(float(grads[n].norm()) / r - 1) for n, r in ref.items() if r > 0)
        print(f"{tag} {dtype:7s} eval " + " ".join(f"{k}={v:.2e}" for k, v in ev.items()))
        print(f"{tag} {dtype:7s} train " + " ".join(f"{k}={v:.2e}" for k, v in tr.items()))
        print(f"{tag} {dtype:7s} loss-rel " + " ".join(f"{k}={v:.2e}" for k, v in lr.items()) + f"  worst-gradnorm-rel={gr:.2e}")

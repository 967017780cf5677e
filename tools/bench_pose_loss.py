"""Fused pose assembly + criterion + backward kernel (ab_pose_loss) at B = 64: time per call and a checksum of its outputs."""
import os
import random
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gen_batch import make_batch
from artiboost_amd import registry as R
from artiboost_amd.criterions import Criterion, FusedPoseCriterion

B, size = 64, 256
cfgc = [{"TYPE": "JointsLoss", "LAMBDA_JOINTS_3D": 1.0, "LAMBDA_CORNERS_3D": 0.2}, {"TYPE": "HandOrdLoss"}, {"TYPE": "SceneOrdLoss"}]
crit = Criterion({"LAMBDAS": [0.5, 0.2, 0.1]}, R.build_criterion_loss_list(cfgc, preset_cfg={}, LAMBDAS=[0.5, 0.2, 0.1]))
fused = FusedPoseCriterion(crit, [size, size], 0)
random.seed(3)
torch.manual_seed(3)
fused.draw(torch.device("cuda"))
tb = {k: v.cuda() for k, v in make_batch(B, size, 11).items()}
g = torch.Generator().manual_seed(0)
kp3d = torch.rand(B, 22, 3, generator=g).cuda()
box = torch.zeros(B, 64).cuda()
box[:, :6] = torch.randn(B, 6, generator=g).cuda()
for _ in range(5):
    o = fused(kp3d, box, 64, tb)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    fused(kp3d, box, 64, tb)
e1.record()
torch.cuda.synchronize()
ck = float(o["g_kp3d"].double().abs().sum() + o["g_box6d"].double().abs().sum() + o["sample_part"].double().abs().sum())
print(f"pose_loss + finalize: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per call   checksum {ck:.9e}")

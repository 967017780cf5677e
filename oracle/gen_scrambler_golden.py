"""Golden vectors of the reference's RandomScrambler (anakin/artiboost/scrambler.py:65-81), produced by RUNNING the reference class in
the build container (manotorch / pytorch3d only stubbed at import: forward() uses torch alone).  The draws the class takes from
torch's global RNG are recorded by replaying the same two Normal.sample calls after the same seed.

    python oracle/gen_scrambler_golden.py        ->  tests/golden/scrambler.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402


def main():
    ref_import.load_control_plane()
    from anakin.artiboost.scrambler import Scrambler
    from torch.distributions.normal import Normal
    cfg = {"TYPE": "random", "HAND_TSL_SIGMA": 0.01, "HAND_POSE_SIGMA": 0.1}
    scr = Scrambler.build(cfg["TYPE"], cfg)
    g = torch.Generator().manual_seed(5)
    B = 6
    pose = (0.4 * torch.randn((B, 48), generator=g)).float()
    pose[0, 3:6] = 0.0                        # a zero joint rotation: the axis is 0 / clip(0, 1e-7)
    tsl = (0.05 * torch.randn((B, 3), generator=g)).float()
    out = {"cfg.tsl_sigma": cfg["HAND_TSL_SIGMA"], "cfg.pose_sigma": cfg["HAND_POSE_SIGMA"], "in.hand_pose": pose.numpy(), "in.hand_tsl": tsl.numpy()}
    for seed in (11, 12):
        torch.manual_seed(seed)
        r = scr({"hand_pose": pose.clone(), "hand_tsl": tsl.clone()})
        torch.manual_seed(seed)               # the same draws, in the class's order (scrambler.py:77-78)
        rand_tsl = Normal(torch.tensor(0.0), torch.tensor(cfg["HAND_TSL_SIGMA"])).sample((B, 3))
        rand_ang = Normal(torch.tensor(0.0), torch.tensor(cfg["HAND_POSE_SIGMA"])).sample((B, 16))
        out[f"s{seed}.rand_tsl"], out[f"s{seed}.rand_angle"] = rand_tsl.numpy(), rand_ang.numpy()
        out[f"s{seed}.hand_pose"], out[f"s{seed}.hand_tsl"] = r["hand_pose"].numpy(), r["hand_tsl"].numpy()
    dst = os.path.join(HERE, "..", "tests", "golden", "scrambler.npz")
    np.savez_compressed(dst, **out)
    print("wrote", os.path.normpath(dst), {k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()

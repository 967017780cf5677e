"""TEST INFRASTRUCTURE ONLY -- seeded CCV samples (poses via oracle/pose_oracle.py, render/augment draws) for the
render parity tests and the CPU baseline."""
import numpy as np

import pose_oracle as po
import render_oracle as ro


def make_samples(assets, B, seed, out_res=(256, 256), K=None, render=512):
    rng = np.random.default_rng(seed)
    K = np.array([[435.0, 0, 256.0], [0, 435.0, 256.0], [0, 0, 1.0]]) if K is None else K
    n_obj = assets.n_obj
    obj_id = rng.integers(0, n_obj, B)
    persp = rng.integers(0, 288, B)
    gpose = np.clip(0.3 * rng.standard_normal((B, 48)), -1.2, 1.2)
    gshape = np.zeros((B, 10))
    gtsl = rng.uniform(-0.05, 0.05, (B, 3))
    Rp = np.stack([po.perspective_from_id(int(p), rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5)) for p in persp])
    free = rng.uniform(0, 2 * np.pi, B)
    Tf = np.tile(np.eye(4), (B, 1, 1))
    Tf[:, 0, 0] = np.cos(free); Tf[:, 0, 1] = -np.sin(free); Tf[:, 1, 0] = np.sin(free); Tf[:, 1, 1] = np.cos(free)
    z = np.zeros((B, 3)); z[:, 2] = rng.uniform(0.45, 0.55, B)
    obj_pose, verts, joints = po.pose_generator(assets.hand, gpose, gshape, gtsl, Rp, Tf, z,
                                                rand_pose_angle=0.1 * rng.standard_normal((B, 16)),
                                                rand_tsl=0.01 * rng.standard_normal((B, 3)))
    samples = np.zeros(B, ro.SAMPLE_DTYPE)
    samples["obj_id"] = obj_id
    samples["hand_tex_id"] = rng.integers(0, assets.hand_tex.shape[0], B)
    samples["bg_id"] = rng.integers(0, assets.backgrounds.shape[0], B)
    bgs = assets.backgrounds.shape[1]
    ch = rng.integers(render, bgs + 1, B)       # get_rand_bg (renderer.py:125-136), square backgrounds
    samples["bg_w"] = ch; samples["bg_h"] = ch
    samples["bg_x0"] = [rng.integers(0, bgs - c + 1) for c in ch]
    samples["bg_y0"] = [rng.integers(0, bgs - c + 1) for c in ch]
    samples["light"] = rng.uniform(1.0, 5.0, B)
    samples["obj_pose"] = obj_pose.reshape(B, 16).astype(np.float32)
    order = np.stack([rng.permutation(4) for _ in range(B)]).astype(np.int32)
    lo_hi = {0: (0.9, 1.1), 1: (0.9, 1.1), 2: (-0.075, 0.075), 3: (0.9, 1.1)}
    factor = np.array([[rng.uniform(*lo_hi[int(o)]) for o in row] for row in order], np.float32)
    gts, inv = [], []
    for b in range(B):
        draws = dict(center=rng.uniform(-1, 1, 2), scale=rng.normal(0, 0.1 / 3), rot=rng.uniform(-0.2 * np.pi, 0.2 * np.pi))
        gt = po.assemble_sample_gt(K, joints[b], obj_pose[b], assets.corners_can[obj_id[b]], list(out_res), draws,
                                   raw_size=[render, render])
        gts.append(gt)
        inv.append(np.linalg.inv(np.vstack([gt["affine"][:2], [0, 0, 1]]).astype(np.float64))[:2].reshape(-1))
    blur = (0.1 * rng.uniform(0, 1, B)).astype(np.float32)          # rendered_dataset.py:257 (blur_radius 0.1, :67)
    blur[::3] = np.float32(0.1) - blur[::3] * np.float32(0.2)        # keep radii > 0.077 (where the blur acts) in every batch
    return dict(samples=samples, hand_verts=verts.astype(np.float32), joints=joints, obj_pose=obj_pose, order=order,
                factor=factor, inv_affine=np.asarray(inv, np.float32), gt=gts, obj_id=obj_id, persp=persp, blur=blur)

"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the grasp refiner:
anakin/artiboost/refiner.py  HORefiner.forward (:181-224), _RefineNet.forward (:252-283), ResBlock.forward (:305-319),
CRot2rotmat (:86-97), parms_decode (:100-105), point2point_signed (:21-83).

Pinned by tests/golden/refiner.npz, produced by oracle/gen_golden.py from the REAL reference classes (imported in the build
container).  Three third-party pieces the reference calls are absent there and enter the golden run as stand-ins
(parity unpinned at those boundaries, as everywhere in this build): chamfer_distance (un-pinned git dependency; brute-force
nearest neighbour), manotorch.ManoLayer (pose_oracle.mano_lbs on the synthetic MANO-shaped model) and pytorch3d's rotation
conversions (pose_oracle.rotmat_to_aa / aa_to_rotmat).  GrabNet's refinenet.pt is a download: weights are a seeded fill.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module."""
import math

import numpy as np
import torch

import pose_oracle as po
from learner_oracle import _key_seed

IN_SIZE, H_SIZE, N_NEURONS = 778 + 16 * 6 + 3, 512, 256


def param_shapes():
    """state_dict of _RefineNet (refiner.py:229-241) without the MANO layer's buffers."""
    out = []

    def bn(prefix, n):
        out.extend([(f"{prefix}.weight", (n,)), (f"{prefix}.bias", (n,)), (f"{prefix}.running_mean", (n,)),
                    (f"{prefix}.running_var", (n,)), (f"{prefix}.num_batches_tracked", ())])

    def lin(prefix, n, k):
        out.extend([(f"{prefix}.weight", (n, k)), (f"{prefix}.bias", (n,))])

    bn("bn1", 778)
    for name, fin in (("rb1", IN_SIZE), ("rb2", IN_SIZE + H_SIZE), ("rb3", IN_SIZE + H_SIZE)):
        lin(f"{name}.fc1", N_NEURONS, fin); bn(f"{name}.bn1", N_NEURONS)
        lin(f"{name}.fc2", H_SIZE, N_NEURONS); bn(f"{name}.bn2", H_SIZE)
        lin(f"{name}.fc3", H_SIZE, fin)
    lin("out_p", 16 * 6, H_SIZE)
    lin("out_t", 3, H_SIZE)
    return out


def fill_params(seed=1):
    """Deterministic name-keyed fill (a stand-in for the GrabNet checkpoint): linear weights ~ N(0, 1/fan_in), the two
    output heads scaled down so that three refinement iterations stay a perturbation of the input grasp."""
    params = {}
    for name, shp in param_shapes():
        g = torch.Generator().manual_seed(_key_seed("refiner." + name, seed))
        if name.endswith("num_batches_tracked"):
            params[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            params[name] = 0.02 * torch.randn(shp, generator=g) + (0.05 if name.startswith("bn1.") else 0.0)
        elif name.endswith("running_var"):
            params[name] = (0.5 + torch.rand(shp, generator=g)) * (0.002 if name.startswith("bn1.") else 1.0)
        elif ".bn" in name or name.startswith("bn1."):
            params[name] = 0.5 + torch.rand(shp, generator=g) if name.endswith(".weight") else 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            params[name] = (0.002 if name.startswith("out_") else 0.05) * torch.randn(shp, generator=g)
        else:
            gain = 0.002 if name.startswith("out_") else 1.0
            params[name] = gain * math.sqrt(1.0 / shp[1]) * torch.randn(shp, generator=g)
    return params


def nearest_dist(x, y):
    """point2point_signed(x, y) with y_normals None (refiner.py:21-83): ||x_i - y_nn(i)||, nn by squared distance
    (dx*dx + dy*dy) + dz*dz in fp32, first minimum.  x (B,P1,3), y (B,P2,3) float32 -> dist (B,P1) f32, idx (B,P1) i32."""
    x = np.asarray(x, np.float32); y = np.asarray(y, np.float32)
    B, P1, _ = x.shape
    dist = np.empty((B, P1), np.float32); idx = np.empty((B, P1), np.int32)
    for b in range(B):
        dx = x[b, :, None, 0] - y[b, None, :, 0]
        dy = x[b, :, None, 1] - y[b, None, :, 1]
        dz = x[b, :, None, 2] - y[b, None, :, 2]
        d2 = (dx * dx + dy * dy) + dz * dz
        idx[b] = np.argmin(d2, axis=1)
        dist[b] = np.sqrt(d2[np.arange(P1), idx[b]])
    return dist, idx


def rotate_points(rot, pts):
    """verts_object = (obj_rot @ pts^T)^T (refiner.py:196-199), evaluated as (r0*x + r1*y) + r2*z in fp32."""
    r = np.asarray(rot, np.float32); p = np.asarray(pts, np.float32)
    return np.stack([(r[:, None, k, 0] * p[..., 0] + r[:, None, k, 1] * p[..., 1]) + r[:, None, k, 2] * p[..., 2] for k in range(3)], -1)


def crot2rotmat(pose):
    """CRot2rotmat (refiner.py:86-97): (N*?,6)-> (N,3,3) from the first two columns, Gram-Schmidt + cross."""
    x = pose.reshape(-1, 3, 2)
    b1 = torch.nn.functional.normalize(x[:, :, 0], dim=1)
    dot = torch.sum(b1 * x[:, :, 1], dim=1, keepdim=True)
    b2 = torch.nn.functional.normalize(x[:, :, 1] - dot * b1, dim=-1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack([b1, b2, b3], dim=-1)


def _bn(p, prefix, x, eps=1e-5):
    return (x - p[prefix + ".running_mean"]) / torch.sqrt(p[prefix + ".running_var"] + eps) * p[prefix + ".weight"] + p[prefix + ".bias"]


def _lin(p, prefix, x):
    return x @ p[prefix + ".weight"].t() + p[prefix + ".bias"]


def res_block(p, name, x):
    """ResBlock.forward (refiner.py:305-319), Fin != Fout, eval mode."""
    ll = lambda t: torch.nn.functional.leaky_relu(t, 0.2)   # noqa: E731
    xin = ll(_lin(p, name + ".fc3", x))
    h = ll(_bn(p, name + ".bn1", _lin(p, name + ".fc1", x)))
    h = _bn(p, name + ".bn2", _lin(p, name + ".fc2", h))
    return ll(xin + h)


def parms_decode(pose_crot, trans):
    """refiner.py:100-105: 6-D rotations -> matrices -> axis-angle (pytorch3d semantics, pose_oracle.rotmat_to_aa)."""
    bs = trans.shape[0]
    R = crot2rotmat(pose_crot)
    aa = torch.from_numpy(po.rotmat_to_aa(R.double().numpy())).float().reshape(bs, -1)
    return aa, trans


def refine_net(p, h2o, rel_rotmat, tsl, glob_rotmat, verts_object, mano_fn, n_iters=3):
    """_RefineNet.forward (refiner.py:252-283), eval mode (BatchNorm running stats, Dropout off)."""
    bs = h2o.shape[0]
    init_pose = torch.cat([glob_rotmat[..., :2].reshape(bs, -1), rel_rotmat[..., :2].reshape(bs, -1)], dim=1)
    init_trans = tsl
    for i in range(n_iters):
        if i != 0:
            aa, tr = parms_decode(init_pose, init_trans)
            verts = mano_fn(aa)[0] + tr[:, None]
            h2o = torch.from_numpy(nearest_dist(verts.numpy(), verts_object)[0])
        h = _bn(p, "bn1", h2o)
        x0 = torch.cat([h, init_pose, init_trans], dim=1)
        x = res_block(p, "rb1", x0)
        x = res_block(p, "rb2", torch.cat([x, x0], dim=1))
        x = res_block(p, "rb3", torch.cat([x, x0], dim=1))
        init_trans = init_trans + _lin(p, "out_t", x)
        init_pose = init_pose + _lin(p, "out_p", x)
    return parms_decode(init_pose, init_trans)


def mano_fn_of(model):
    """ManoLayer(rot_mode='axisang', center_idx=None, flat_hand_mean=True)(pose) with zero betas (refiner.py:138,193,216)."""
    def fn(pose):
        v, j, _ = po.mano_lbs(model, pose.double().numpy(), np.zeros((pose.shape[0], 10)))
        return torch.from_numpy(v).float(), torch.from_numpy(j).float()
    return fn


def ho_refiner(p, model, hand_pose, hand_tsl, obj_rot, obj_points, n_iters=3):
    """HORefiner.forward (refiner.py:181-224).  hand_pose (B,48), hand_tsl (B,3), obj_rot (B,3,3) torch f32;
    obj_points (B,P2,3) = resampled_objs[obj_idx] (numpy).  -> dict hand_verts, joints, hand_pose, hand_tsl."""
    bs = hand_pose.shape[0]
    mano_fn = mano_fn_of(model)
    rotm = torch.from_numpy(po.aa_to_rotmat(hand_pose.reshape(bs, -1, 3).double().numpy())).float()
    verts = mano_fn(hand_pose)[0] + hand_tsl[:, None]
    verts_object = rotate_points(obj_rot.numpy(), obj_points)
    h2o = torch.from_numpy(nearest_dist(verts.numpy(), verts_object)[0]).abs()
    pose, tsl = refine_net(p, h2o, rotm[:, 1:], hand_tsl, rotm[:, 0], verts_object, mano_fn, n_iters)
    v, j = mano_fn(pose)
    return {"hand_verts": v + tsl[:, None], "joints": j + tsl[:, None], "hand_pose": pose, "hand_tsl": tsl}


def make_inputs(assets, B, seed, n_points=10000):
    """Seeded refiner inputs in the scale of the pose generator's output (preprocessor.py:62-80)."""
    rng = np.random.default_rng(seed)
    pose = np.clip(0.3 * rng.standard_normal((B, 48)), -1.2, 1.2).astype(np.float32)
    tsl = rng.uniform(-0.05, 0.05, (B, 3)).astype(np.float32)
    rot = po.aa_to_rotmat(rng.standard_normal((B, 3))).astype(np.float32)
    obj_idx = rng.integers(0, assets.n_obj, B)
    return pose, tsl, rot, obj_idx

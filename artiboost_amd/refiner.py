"""Grasp refiner on the GPU: anakin/artiboost/refiner.py.

`HORefiner` (registry key "hand_obj") and `NullRefine` ("null") keep the reference's names and call contract
(`Refiner.build(type, cfg)`, `setup(...)`, `forward(inp, obj) -> {hand_verts, joints, hand_pose, hand_tsl}`), batched on
device tensors: the nearest-point distances come from `ab_nearest_dist` (replacing the third-party chamfer_distance CUDA
extension and the rotated-object temporary), every layer of the RefineNet MLP is one `ab_linear_fused` launch (eval-mode
BatchNorm1d folded into a per-column affine, residual and LeakyReLU in the epilogue), MANO is `ab_mano_lbs`.
Inference only, as in the reference (`refine_net.eval()`, refiner.py:161)."""
import os

import numpy as np
import torch

from . import _lib as L

IN_SIZE, H_SIZE, N_NEURONS = 778 + 16 * 6 + 3, 512, 256
PAD0, PAD1 = 880, 1392          # feature widths padded to a multiple of 4 (877, 512 + 877 = 1389)


def nearest_dist(x, ypts, obj_idx=None, rot=None, scale=None, shift=None, out=None, want_idx=False):
    """min_j ||x[b,i] - rot[b] ypts[obj_idx[b], j]|| -> [B,P1] (optionally written into `out`, a [B, >=P1] row-pitched
    view) and, with want_idx, the arg-min indices (int32)."""
    B, P1, _ = x.shape
    P2 = ypts.shape[1]
    if out is None:
        out = torch.empty((B, P1), dtype=torch.float32, device=x.device)
    assert out.stride(1) == 1 and out.shape[0] == B
    idx = torch.empty((B, P1), dtype=torch.int32, device=x.device) if want_idx else None
    L.check(L.lib().ab_nearest_dist(L.ptr(x), L.ptr(ypts), L.ptr(obj_idx), L.ptr(rot), L.i(B), L.i(P1), L.i(P2), L.ptr(scale),
                                    L.ptr(shift), ctypes_ptr(out), L.i(out.stride(0)), L.ptr(idx), L.stream()), "ab_nearest_dist")
    return (out, idx) if want_idx else out


def ctypes_ptr(t):
    """A (possibly row-pitched) device view as a buffer argument (the pitch is passed next to it)."""
    return L.view_ptr(t)


def linear_fused(x, w, bias=None, scale=None, shift=None, residual=None, act=0, slope=0.2, out=None):
    """act(((x @ w^T + bias) * scale + shift) + residual); x [M,K] contiguous, w [N,K]; out / residual may be row-pitched views."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    ldr = residual.stride(0) if residual is not None else 0
    L.check(L.lib().ab_linear_fused(L.ptr(x), L.ptr(w), L.ptr(bias), L.ptr(scale), L.ptr(shift),
                                    ctypes_ptr(residual) if residual is not None else L.ptr(None), L.i(ldr), L.i(M), L.i(N), L.i(K),
                                    L.i(act), L.f(slope), ctypes_ptr(out), L.i(out.stride(0)), L.stream()), "ab_linear_fused")
    return out


def crot2rotmat(pose):
    """CRot2rotmat (refiner.py:86-97)."""
    x = pose.reshape(-1, 3, 2)
    b1 = torch.nn.functional.normalize(x[:, :, 0], dim=1)
    dot = torch.sum(b1 * x[:, :, 1], dim=1, keepdim=True)
    b2 = torch.nn.functional.normalize(x[:, :, 1] - dot * b1, dim=-1)
    return torch.stack([b1, b2, torch.cross(b1, b2, dim=1)], dim=-1)


class RefineNet:
    """_RefineNet (refiner.py:227-283) in eval mode over device buffers.  `load_state_dict` takes the reference's keys
    (GrabNet's refinenet.pt); BatchNorm1d layers are folded into (scale, shift) there."""

    def __init__(self, mano, n_iters=3, device="cuda"):
        self.mano, self.n_iters, self.dev = mano, n_iters, torch.device(device)
        self.p = None

    @staticmethod
    def _fold(sd, prefix, eps=1e-5):
        s = sd[prefix + ".weight"].double() / torch.sqrt(sd[prefix + ".running_var"].double() + eps)
        return s.float(), (sd[prefix + ".bias"].double() - sd[prefix + ".running_mean"].double() * s).float()

    def load_state_dict(self, sd, strict=False):
        sd = {k: v.detach().cpu().float() if v.dtype.is_floating_point else v for k, v in sd.items()}
        p = {}
        p["bn1.scale"], p["bn1.shift"] = self._fold(sd, "bn1")
        for name, fin, pad in (("rb1", IN_SIZE, PAD0), ("rb2", IN_SIZE + H_SIZE, PAD1), ("rb3", IN_SIZE + H_SIZE, PAD1)):
            for fc in ("fc1", "fc3"):
                w = torch.zeros((sd[f"{name}.{fc}.weight"].shape[0], pad))
                w[:, :fin] = sd[f"{name}.{fc}.weight"]
                p[f"{name}.{fc}.w"], p[f"{name}.{fc}.b"] = w, sd[f"{name}.{fc}.bias"]
            p[f"{name}.fc2.w"], p[f"{name}.fc2.b"] = sd[f"{name}.fc2.weight"], sd[f"{name}.fc2.bias"]
            p[f"{name}.bn1.scale"], p[f"{name}.bn1.shift"] = self._fold(sd, f"{name}.bn1")
            p[f"{name}.bn2.scale"], p[f"{name}.bn2.shift"] = self._fold(sd, f"{name}.bn2")
        p["out.w"] = torch.cat([sd["out_p.weight"], sd["out_t.weight"]])           # one launch for both heads: [99, 512]
        p["out.b"] = torch.cat([sd["out_p.bias"], sd["out_t.bias"]])
        self.p = {k: v.contiguous().to(self.dev) for k, v in p.items()}

    def _res_block(self, name, x, out):
        p = self.p
        xin = linear_fused(x, p[f"{name}.fc3.w"], p[f"{name}.fc3.b"], act=2)
        h = linear_fused(x, p[f"{name}.fc1.w"], p[f"{name}.fc1.b"], p[f"{name}.bn1.scale"], p[f"{name}.bn1.shift"], act=2)
        return linear_fused(h, p[f"{name}.fc2.w"], p[f"{name}.fc2.b"], p[f"{name}.bn2.scale"], p[f"{name}.bn2.shift"],
                            residual=xin, act=2, out=out)

    def decode(self, pose_crot, trans):
        """parms_decode (refiner.py:100-105)."""
        from .synth import rotmat_to_aa
        return rotmat_to_aa(crot2rotmat(pose_crot)).reshape(trans.shape[0], -1), trans

    def __call__(self, verts0, rel_rotmat, tsl, glob_rotmat, ypts, obj_idx, obj_rot):
        """verts0: the hand vertices of iteration 0 (their distances are taken here, with |.| a no-op on a norm);
        ypts/obj_idx/obj_rot: the resampled object point table, row selector and rotation (verts_object is never built)."""
        p, bs = self.p, tsl.shape[0]
        x0 = torch.zeros((bs, PAD0), dtype=torch.float32, device=self.dev)
        xa = torch.zeros((bs, PAD1), dtype=torch.float32, device=self.dev)     # cat([X, X0]) of rb2
        xb = torch.zeros((bs, PAD1), dtype=torch.float32, device=self.dev)     # cat([X, X0]) of rb3
        x0[:, 778:874] = torch.cat([glob_rotmat[..., :2].reshape(bs, -1), rel_rotmat[..., :2].reshape(bs, -1)], dim=1)
        x0[:, 874:877] = tsl
        zeros10 = torch.zeros((bs, 10), dtype=torch.float32, device=self.dev)
        verts = verts0
        for i in range(self.n_iters):
            if i != 0:
                aa, tr = self.decode(x0[:, 778:874], x0[:, 874:877])
                verts = (self.mano(aa, zeros10)[0] + tr[:, None]).contiguous()
            nearest_dist(verts, ypts, obj_idx, obj_rot, p["bn1.scale"], p["bn1.shift"], out=x0[:, :778])
            xa[:, H_SIZE:H_SIZE + IN_SIZE] = x0[:, :IN_SIZE]
            xb[:, H_SIZE:H_SIZE + IN_SIZE] = x0[:, :IN_SIZE]
            self._res_block("rb1", x0, xa[:, :H_SIZE])
            self._res_block("rb2", xa, xb[:, :H_SIZE])
            x = self._res_block("rb3", xb, None)
            x0[:, 778:877] += linear_fused(x, p["out.w"], p["out.b"])
        return self.decode(x0[:, 778:874], x0[:, 874:877].clone())


class NullRefine:
    """Registry key "null" (refiner.py:116-148): decode the grasp, no refinement."""

    def __init__(self, cfg=None, mano=None, device="cuda"):
        self.mano, self.dev = mano, torch.device(device)

    def setup(self, resampled_objs=None):
        pass

    def __call__(self, inp, obj_idx=None):
        pose, tsl = inp["hand_pose"], inp["hand_tsl"]
        v, j, _ = self.mano(pose, torch.zeros((pose.shape[0], 10), dtype=torch.float32, device=self.dev))
        return {"hand_verts": v + tsl[:, None], "joints": j + tsl[:, None], "hand_pose": pose, "hand_tsl": tsl}

    forward = __call__


class HORefiner:
    """Registry key "hand_obj" (refiner.py:151-224).  cfg: {"PRETRAINED": path to GrabNet's refinenet.pt, "ITERS": n}.
    A missing checkpoint is an error unless cfg["ALLOW_RANDOM_INIT"] is set (benchmarks / tests: seeded stand-in weights
    must then be given through load_state_dict)."""

    def __init__(self, cfg, mano, device="cuda"):
        self.dev = torch.device(device)
        self.mano = mano
        self.refine_net = RefineNet(mano, n_iters=int(cfg.get("ITERS", 3)), device=device)
        ckp = cfg.get("PRETRAINED", "")
        if ckp and os.path.exists(ckp):
            self.refine_net.load_state_dict(torch.load(ckp, map_location="cpu"), strict=False)
        elif not cfg.get("ALLOW_RANDOM_INIT", False):
            raise FileNotFoundError(f"refiner checkpoint {ckp!r} not found (GrabNet refinenet.pt is a download)")
        self.resampled = None

    def load_state_dict(self, sd, strict=False):
        self.refine_net.load_state_dict(sd, strict)

    def setup(self, resampled_objs):
        """resampled_objs: float32 [n_obj, n_points, 3] (assets.resample_objects == HORefiner.resample_obj per mesh)."""
        self.resampled = torch.from_numpy(np.ascontiguousarray(resampled_objs, np.float32)).to(self.dev)

    def __call__(self, inp, obj_idx):
        """inp: hand_pose [B,48], hand_tsl [B,3], obj_rot [B,3,3] (device f32); obj_idx int64 [B] rows of the point table
        (the reference passes object names and looks the rows up, refiner.py:195)."""
        from .synth import aa_to_rotmat
        pose, tsl, obj_rot = inp["hand_pose"].contiguous(), inp["hand_tsl"].contiguous(), inp["obj_rot"].contiguous()
        bs = pose.shape[0]
        rotm = aa_to_rotmat(pose.reshape(bs, -1, 3))
        zeros10 = torch.zeros((bs, 10), dtype=torch.float32, device=self.dev)
        verts = (self.mano(pose, zeros10)[0] + tsl[:, None]).contiguous()
        obj_idx = obj_idx.to(self.dev, torch.int64).contiguous()
        fpose, ftsl = self.refine_net(verts, rotm[:, 1:], tsl, rotm[:, 0], self.resampled, obj_idx, obj_rot)
        v, j, _ = self.mano(fpose.contiguous(), zeros10)
        return {"hand_verts": v + ftsl[:, None], "joints": j + ftsl[:, None], "hand_pose": fpose, "hand_tsl": ftsl}

    forward = __call__


class Refiner:
    """Refiner.build(type, cfg) (refiner.py:108-114)."""
    build_mapping = {"null": NullRefine, "hand_obj": HORefiner}

    @staticmethod
    def build(type, *args, **kwargs):
        return Refiner.build_mapping[type](*args, **kwargs)

"""Loss plugin classes with the reference's names, config keys and RNG behaviour
(anakin/criterions/criterion.py:8-67, jointloss.py:14-67, ordinal.py:17-306, symcornerloss.py:18-102).

The ordinal losses draw from the same global RNGs in the same order as the reference (torch.rand for the virtual
view vectors, random.shuffle for the pair subsets), so seeding `random` and `torch` identically reproduces the
reference's loss values.  Tensors here are tiny ((B,21,3)-sized); the arithmetic runs as device torch ops."""
import os
import random
from itertools import combinations, product
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .registry import CONST, LOSS, Queries, camel_to_snake


class TensorLoss(object):
    def __init__(self):
        self.output_key = f"{camel_to_snake(type(self).__name__)}_output"

    def __call__(self, preds: Dict, targs: Dict, **kwargs) -> Tuple[torch.Tensor, Dict]:
        dev = None
        for v in preds.values():
            if isinstance(v, torch.Tensor):
                dev = v.device
                break
        if dev is None:
            raise RuntimeError("Cannot found valid Tensor with device")
        return torch.zeros(1, dtype=torch.float32, device=dev), {}


class Criterion(TensorLoss):
    def __init__(self, cfg: Dict, loss_list: List[TensorLoss]) -> None:
        super().__init__()
        self._loss_list = loss_list
        lambdas = list(cfg["LAMBDAS"])
        self._loss_lambdas = {type(l).__name__: lambdas[i] for i, l in enumerate(loss_list)}
        # predictions that come straight from the HIP model carry a link to its raw outputs; with it compute_losses runs
        # the fused pose/loss kernel instead of the registry losses' torch ops.  FUSED_CRITERION: false / AB_FUSED_CRITERION=0: off
        self.fused_route = bool(cfg.get("FUSED_CRITERION", os.environ.get("AB_FUSED_CRITERION", "1") != "0"))

    @property
    def loss_list(self):
        return self._loss_list

    @property
    def loss_lambdas(self):
        return self._loss_lambdas

    def draw(self, dev):
        """Refresh the random draws of every loss (host RNG, reference order) into their device buffers.  On a GPU the
        draws of all losses are packed into one pinned staging tensor and shipped with a single H2D copy."""
        dev = torch.device(dev)
        arena = None
        if dev.type == "cuda":
            arena = getattr(self, "_draw_arena", None)
            if arena is None:
                arena = self._draw_arena = _DrawArena()
            arena.begin()
        for loss in self.loss_list:
            if hasattr(loss, "draw"):
                if hasattr(loss, "draws"):
                    loss.draws.arena = arena
                loss.draw(dev)
        if arena is not None:
            arena.flush(dev)
            for loss in self.loss_list:          # a loss called on its own (not through this method) copies directly
                if hasattr(loss, "draws"):
                    loss.draws.arena = None

    def freeze_draws(self, frozen=True):
        for loss in self.loss_list:
            if hasattr(loss, "draws"):
                loss.draws.frozen = frozen

    # ---- the fused route of the reference-shaped loop ----------------------------------------------------------------
    _FUSED_TARGETS = (Queries.ROOT_JOINT, Queries.CAM_INTR, Queries.CORNERS_CAN, Queries.JOINTS_3D, Queries.CORNERS_3D,
                      Queries.JOINTS_VIS, Queries.CORNERS_VIS)

    def _fused_for(self, link):
        """FusedPoseCriterion of this loss list for the model geometry in `link`, or None when a loss is outside the kernel."""
        key = (tuple(link["inp_res"]), link["center_idx"])
        cache = self.__dict__.setdefault("_fused_cache", {})
        if key not in cache:
            try:
                cache[key] = FusedPoseCriterion(self, link["inp_res"], link["center_idx"])
            except NotImplementedError:
                cache[key] = None
        return cache[key]

    def _compute_losses_fused(self, link, targs):
        """One HIP kernel (csrc/pose_loss.hip) for pose assembly + all losses + their gradients, behind `compute_losses`:
        the ~350 small torch kernels per step of the registry losses and of their autograd backward disappear, and the
        update becomes deterministic (the autograd backward of index_select accumulates with float atomics)."""
        fused = self._fused_for(link)
        if fused is None:
            return None
        kp3d, box6d = link["kp3d"], link["box6d"]
        dev = kp3d.device
        t = {}
        for k in self._FUSED_TARGETS:
            v = targs[k]
            if not torch.is_tensor(v):
                return None
            t[k] = v.to(device=dev, dtype=torch.float32, non_blocking=True).contiguous()
        if fused.sym is not None:                            # SymCornerLoss: the object's class index and pose
            for k, dt in ((Queries.OBJ_IDX, torch.int64), (Queries.OBJ_TRANSF, torch.float32)):
                v = targs.get(k)
                if not torch.is_tensor(v):
                    return None
                t[k] = v.to(device=dev, dtype=dt, non_blocking=True).contiguous()
        if not all(l.draws.frozen for l in self.loss_list if hasattr(l, "draws")):
            self.draw(dev)                                   # reference RNG order: the losses' draws in list order
        total, vals, sym = _FusedLossFn.apply(kp3d, box6d, fused, t)
        out = {}
        for loss in self.loss_list:                          # the entries each registry class reports
            if isinstance(loss, JointsLoss):
                out["joints_3d_loss"] = vals[0] if loss.lambda_joints_3d else None
                out["corners_3d_loss"] = vals[1] if loss.lambda_corners_3d else None
                out[loss.output_key] = loss.lambda_joints_3d * vals[0] + loss.lambda_corners_3d * vals[1]
            elif isinstance(loss, HandOrdLoss):
                out["joint_ord_loss"], out["part_ord_loss"] = vals[2], vals[3]
                out[loss.output_key] = loss.lambda_joint_lev * vals[2] + loss.lambda_part_lev * vals[3]
            elif isinstance(loss, SceneOrdLoss):
                out["scene_ord_loss"] = vals[4]
            elif isinstance(loss, SymCornerLoss):
                out["sym_corners_3d_loss"] = sym[0] if loss.lambda_sym_corners_3d else None
                out[loss.output_key] = loss.lambda_sym_corners_3d * sym[0]
        out["final_loss"] = total
        return total, out

    def compute_losses(self, preds: Dict, targs: Dict, **kwargs):
        link = getattr(preds.get("joints_3d_abs"), "_ab_fuse", None) if self.fused_route else None
        if link is not None and torch.is_grad_enabled():
            r = self._compute_losses_fused(link, targs)
            if r is not None:
                return r
        total, out = super().__call__(preds, targs, **kwargs)
        for loss in self.loss_list:
            fl, d = loss(preds, targs, **kwargs)
            total = total + self.loss_lambdas[type(loss).__name__] * fl
            out.update(d)
        assert "final_loss" not in out
        out["final_loss"] = total
        return total, out


def _abs_masked(preds, targs, dev):
    root = targs[Queries.ROOT_JOINT].to(dev)
    jv = targs[Queries.JOINTS_VIS].to(dev)[..., None]
    cv = targs[Queries.CORNERS_VIS].to(dev)[..., None]
    jt = (targs[Queries.JOINTS_3D].to(dev) + root[:, None]) * jv
    ct = (targs[Queries.CORNERS_3D].to(dev) + root[:, None]) * cv
    return preds["joints_3d_abs"] * jv, jt, preds["corners_3d_abs"] * cv, ct


@LOSS.register_module
class JointsLoss(TensorLoss):
    def __init__(self, **cfg):
        super().__init__()
        self.lambda_joints_3d = cfg.get("LAMBDA_JOINTS_3D", 0.0)
        self.lambda_corners_3d = cfg.get("LAMBDA_CORNERS_3D", 0.0)

    def __call__(self, preds, targs, **kwargs):
        final_loss, losses = super().__call__(preds, targs, **kwargs)
        jp, jt, cp, ct = _abs_masked(preds, targs, final_loss.device)
        lj = F.mse_loss(jp, jt) if self.lambda_joints_3d else None
        lc = F.mse_loss(cp, ct) if self.lambda_corners_3d else None
        if lj is not None:
            final_loss = final_loss + self.lambda_joints_3d * lj
        if lc is not None:
            final_loss = final_loss + self.lambda_corners_3d * lc
        losses["joints_3d_loss"] = lj
        losses["corners_3d_loss"] = lc
        losses[self.output_key] = final_loss
        return final_loss, losses


@LOSS.register_module
class ManoLoss(TensorLoss):
    """anakin/criterions/honetloss.py:12-73: MANO shape / pose regularisers and (optional) joint / vertex supervision of the
    regression-based model (config_eval/eval_ho3dv2_regbased_artiboost.yaml)."""

    def __init__(self, **cfg):
        super().__init__()
        self.lambda_joints_3d = float(cfg["LAMBDA_JOINTS_3D"])
        self.lambda_hand_verts_3d = float(cfg["LAMBDA_HAND_VERTS_3D"])
        self.lambda_shape_reg = float(cfg["LAMBDA_SHAPE_REG"])
        self.lambda_pose_reg = float(cfg["LAMBDA_POSE_REG"])

    def __call__(self, preds, targs, **kwargs):
        final_loss, losses = super().__call__(preds, targs, **kwargs)
        shape_reg = pose_reg = lj = lv = None
        if self.lambda_shape_reg:
            shape_reg = preds["mano_shape"].pow(2).mean()
            final_loss = final_loss + self.lambda_shape_reg * shape_reg
        if self.lambda_pose_reg:
            pose_reg = preds["mano_pca_pose"][:, 3:].pow(2).mean()            # the root rotation is not regularised
            final_loss = final_loss + self.lambda_pose_reg * pose_reg
        if self.lambda_joints_3d and Queries.JOINTS_3D in targs:
            p = preds["joints_3d_abs"]
            lj = F.mse_loss(p, targs[Queries.JOINTS_3D].to(p.device) + targs[Queries.ROOT_JOINT].to(p.device).unsqueeze(1))
            final_loss = final_loss + self.lambda_joints_3d * lj
        if self.lambda_hand_verts_3d and "hand_verts_3d" in targs:
            p = preds["hand_verts_3d_abs"]
            lv = F.mse_loss(p, targs["hand_verts_3d"].to(p.device) + targs[Queries.ROOT_JOINT].to(p.device).unsqueeze(1))
            final_loss = final_loss + self.lambda_hand_verts_3d * lv
        losses.update(mano_shape=shape_reg, mano_pca_pose=pose_reg, joints_3d_loss=lj, hand_verts_3d_loss=lv)
        return final_loss, losses


def sample_view_vectors(n_virtual_views=20):
    """ordinal.py:59-71 (CPU draws from the global torch RNG: rand(n) for theta, then rand(n) for u)."""
    cam_vec = torch.Tensor([0.0, 0.0, 1.0]).unsqueeze(0)
    theta = torch.rand(n_virtual_views) * 2.0 * np.pi
    u = torch.rand(n_virtual_views)
    s = torch.sqrt(1.0 - u ** 2)
    nv = torch.cat([(s * torch.cos(theta)).unsqueeze(1), (s * torch.sin(theta)).unsqueeze(1), u.unsqueeze(1)], dim=1)
    return torch.cat([cam_vec, nv], dim=0)


def _shuffled_third(n):
    idx = list(range(n))
    random.shuffle(idx)
    return idx[: n // 3]


class _DrawArena:
    """One persistent device byte buffer for every per-step draw of a Criterion: `add` records (owner, name, cpu tensor) in
    call order, `flush` packs them (16-byte aligned slots, layout fixed by the first step), pins the pack and issues ONE
    async H2D copy; the owners' `bufs[name]` are typed views into the device buffer (stable addresses for hipGraphs)."""

    def __init__(self):
        self.items, self.layout, self.dev_buf = [], None, None

    def begin(self):
        self.items = []

    def add(self, owner, name, cpu_tensor):
        self.items.append((owner, name, cpu_tensor.contiguous()))

    def flush(self, dev):
        sig = tuple((id(o), n, tuple(t.shape), t.dtype) for o, n, t in self.items)
        if self.layout is None or self.layout[0] != sig:
            offs, off = [], 0
            for _, _, t in self.items:
                offs.append(off)
                off += (t.numel() * t.element_size() + 15) // 16 * 16
            self.layout = (sig, offs, off)
            self.dev_buf = torch.empty(max(off, 16), dtype=torch.uint8, device=dev)
            self.views = [self.dev_buf[b:b + t.numel() * t.element_size()].view(t.dtype).view(t.shape) for (_, _, t), b in zip(self.items, offs)]
        for (o, n, _), v in zip(self.items, self.views):      # (re)bind every step: a loss called on its own in between
            o.bufs[n] = v                                       # (registry route) works on private buffers, see _Draws.put
        _, offs, total = self.layout
        pack = torch.empty(max(total, 16), dtype=torch.uint8)
        for (_, _, t), b in zip(self.items, offs):
            nb = t.numel() * t.element_size()
            pack[b:b + nb] = t.reshape(-1).view(torch.uint8)
        self.dev_buf.copy_(pack.pin_memory(), non_blocking=True)      # a fresh pinned block per step: no reuse hazard


class _Draws:
    """Per-call random draws of the ordinal losses, kept in persistent device buffers so that the arithmetic can be
    captured in a hipGraph: `draw()` (host RNG, same order as the reference) refreshes the buffers with async H2D
    copies OUTSIDE the graph; the captured ops only read them."""

    def __init__(self):
        self.dev = None
        self.bufs = {}
        self.frozen = False     # True while a captured graph owns the call: __call__ must not draw
        self.arena = None       # set by Criterion.draw: pack all draws of the step into one H2D copy

    def put(self, name, cpu_tensor, dev):
        if self.arena is not None:
            self.arena.add(self, name, cpu_tensor)
            return None
        b = self.bufs.get(name)
        # buffers are persistent (a captured hipGraph holds their addresses) -- but never a view of the arena here: the
        # arena's views share one version counter, and an in-place refresh of one would invalidate what autograd saved of another
        if b is None or b.shape != cpu_tensor.shape or b._base is not None:
            b = torch.empty(cpu_tensor.shape, dtype=cpu_tensor.dtype, device=dev)
            self.bufs[name] = b
        src = cpu_tensor.pin_memory() if dev.type == "cuda" else cpu_tensor
        b.copy_(src, non_blocking=True)
        return b


def _ord(a, b, views):      # jointlevel_ordinal_relation (ordinal.py:39-56)
    return torch.einsum("bpk,vk->bpv", a - b, views)


@LOSS.register_module
class HandOrdLoss(TensorLoss):
    def __init__(self, **cfg):
        super().__init__()
        self.lambda_part_lev = float(cfg.get("LAMBDA_PART_LEVEL", 1.0))
        self.lambda_joint_lev = float(cfg.get("LAMBDA_JOINTS_LEVEL", 1.0))
        self.n_virtual_views = int(cfg.get("N_VIRTUAL_VIEWS", 20))
        self.joint_pairs_idx = list(combinations(range(CONST.NUM_JOINTS), 2))
        self.parts_pairs_idx = list(combinations(range(CONST.NUM_JOINTS - 1), 2))
        self.draws = _Draws()

    def draw(self, dev):
        """RNG order of the reference (ordinal.py:159,165-166,202-203): views, joint-pair shuffle, part-pair shuffle."""
        d = self.draws
        d.put("views", sample_view_vectors(self.n_virtual_views), dev)
        sel = _shuffled_third(len(self.joint_pairs_idx))
        d.put("j0", torch.tensor([self.joint_pairs_idx[i][0] for i in sel]), dev)
        d.put("j1", torch.tensor([self.joint_pairs_idx[i][1] for i in sel]), dev)
        sel = _shuffled_third(len(self.parts_pairs_idx))
        d.put("p0", torch.tensor([self.parts_pairs_idx[i][0] for i in sel]), dev)
        d.put("p1", torch.tensor([self.parts_pairs_idx[i][1] for i in sel]), dev)

    def __call__(self, preds, targs, **kwargs):
        final_loss, losses = super().__call__(preds, targs, **kwargs)
        dev = final_loss.device
        jp, jt, _, _ = _abs_masked(preds, targs, dev)
        if not self.draws.frozen:
            self.draw(dev)
        b = self.draws.bufs
        views, i0, i1 = b["views"], b["j0"], b["j1"]
        gt = _ord(jt.index_select(1, i0), jt.index_select(1, i1), views)
        pr = _ord(jp.index_select(1, i0), jp.index_select(1, i1), views)
        joint_ord_loss = torch.log(1.0 + F.relu(-1.0 * torch.sign(gt) * pr)).mean()
        final_loss = final_loss + self.lambda_joint_lev * joint_ord_loss
        losses["joint_ord_loss"] = joint_ord_loss

        def parts(j):   # joints_2_part_pairs (ordinal.py:98-121)
            return (j - j[:, CONST.JOINTS_IDX_PARENTS])[:, 1:]

        pp, pt = parts(jp), parts(jt)
        a0, a1 = b["p0"], b["p1"]
        gt = torch.einsum("bpk,vk->bpv", torch.cross(pt.index_select(1, a0), pt.index_select(1, a1), dim=-1), views)
        pr = torch.einsum("bpk,vk->bpv", torch.cross(pp.index_select(1, a0), pp.index_select(1, a1), dim=-1), views)
        part_ord_loss = F.relu(-1.0 * torch.sign(gt) * pr).mean()
        final_loss = final_loss + self.lambda_part_lev * part_ord_loss
        losses["part_ord_loss"] = part_ord_loss
        losses[self.output_key] = final_loss
        return final_loss, losses


@LOSS.register_module
class SceneOrdLoss(TensorLoss):
    def __init__(self, **cfg):
        super().__init__()
        self.lambda_scene_lev = float(cfg.get("LAMBDA_SCENE_LEVEL", 1.0))
        self.n_virtual_views = int(cfg.get("N_VIRTUAL_VIEWS", 40))
        self.ho_pairs_idx = list(product(range(CONST.NUM_JOINTS), range(CONST.NUM_CORNERS)))
        self.draws = _Draws()

    def draw(self, dev):
        d = self.draws
        d.put("views", sample_view_vectors(self.n_virtual_views), dev)
        sel = _shuffled_third(len(self.ho_pairs_idx))
        d.put("i0", torch.tensor([self.ho_pairs_idx[i][0] for i in sel]), dev)
        d.put("i1", torch.tensor([self.ho_pairs_idx[i][1] for i in sel]), dev)

    def __call__(self, preds, targs, **kwargs):
        final_loss, losses = super().__call__(preds, targs, **kwargs)
        dev = final_loss.device
        jp, jt, cp, ct = _abs_masked(preds, targs, dev)
        if not self.draws.frozen:
            self.draw(dev)
        b = self.draws.bufs
        views, i0, i1 = b["views"], b["i0"], b["i1"]
        gt = _ord(jt.index_select(1, i0), ct.index_select(1, i1), views)
        pr = _ord(jp.index_select(1, i0), cp.index_select(1, i1), views)
        scene_ord_loss = torch.log(1.0 + F.relu(-1.0 * torch.sign(gt) * pr)).mean()
        final_loss = final_loss + self.lambda_scene_lev * scene_ord_loss
        losses["scene_ord_loss"] = scene_ord_loss
        return final_loss, losses


def get_symmetry_transformations(model_info, max_sym_disc_step):
    """anakin/utils/bop_toolkit/bop_misc.py:18-65 (BOP toolkit): discrete x discretised-continuous symmetries."""
    trans_disc = [{"R": np.eye(3), "t": np.array([[0, 0, 0]]).T}]
    for sym in model_info.get("symmetries_discrete", []):
        s = np.reshape(sym, (4, 4))
        trans_disc.append({"R": s[:3, :3], "t": s[:3, 3].reshape((3, 1))})
    trans_cont = []
    for sym in model_info.get("symmetries_continuous", []):
        axis = np.array(sym["axis"], dtype=np.float64)
        offset = np.array(sym["offset"], dtype=np.float64).reshape((3, 1))
        steps = int(np.ceil(np.pi / max_sym_disc_step))
        step = 2.0 * np.pi / steps
        axis_n = axis / np.linalg.norm(axis)
        for i in range(1, steps):
            a = i * step
            K_ = np.array([[0, -axis_n[2], axis_n[1]], [axis_n[2], 0, -axis_n[0]], [-axis_n[1], axis_n[0], 0]])
            R = np.cos(a) * np.eye(3) + (1 - np.cos(a)) * np.outer(axis_n, axis_n) + np.sin(a) * K_
            trans_cont.append({"R": R, "t": -R.dot(offset) + offset})
    trans = []
    for td in trans_disc:
        if trans_cont:
            for tc in trans_cont:
                trans.append({"R": tc["R"].dot(td["R"]), "t": tc["R"].dot(td["t"]) + tc["t"]})
        else:
            trans.append(td)
    return trans


@LOSS.register_module
class SymCornerLoss(TensorLoss):
    def __init__(self, **cfg):
        super().__init__()
        self.lambda_sym_corners_3d = cfg.get("LAMBDA_SYM_CORNERS_3D", 0.0)
        info = cfg.get("MODEL_INFO")
        if info is None:
            import json
            info = json.load(open(cfg["MODEL_INFO_PATH"], "r"))
        step = cfg.get("MAX_SYM_DISC_STEP", 0.01)
        if cfg.get("USE_HO3D_YCB", False):
            raise NotImplementedError("USE_HO3D_YCB")
        syms = [get_symmetry_transformations(info[str(i)], step) for i in range(1, len(info) + 1)]
        kmax = max(len(s) for s in syms)
        R = np.tile(np.eye(3), (len(syms), kmax, 1, 1))
        t = np.zeros((len(syms), kmax, 3, 1))
        for i, s in enumerate(syms):
            for k, tr in enumerate(s):
                R[i, k] = tr["R"]
                t[i, k] = tr["t"] / 1000.0
        self.R = torch.Tensor(R)
        self.t = torch.Tensor(t)

    def __call__(self, preds, targs, **kwargs):
        final_loss, losses = super().__call__(preds, targs, **kwargs)
        dev = final_loss.device
        loss = None
        if self.lambda_sym_corners_3d:
            obj = (targs[Queries.OBJ_IDX] - 1).long().to(dev)
            Rs, ts = self.R.to(dev)[obj], self.t.to(dev)[obj]
            T = targs[Queries.OBJ_TRANSF].to(dev)
            can = targs[Queries.CORNERS_CAN].to(dev).permute(0, 2, 1)[:, None]
            sym_can = torch.matmul(Rs, can) + ts
            gt = (torch.matmul(T[:, None, :3, :3], sym_can) + T[:, None, :3, 3:]).permute(0, 1, 3, 2)
            vis = targs[Queries.CORNERS_VIS].to(dev)
            pred = (preds["corners_3d_abs"] * vis[..., None])[:, None]
            gt = gt * vis[:, None, :, None]
            loss = ((gt - pred) ** 2).mean(-1).mean(-1).min(dim=-1)[0].mean()
            final_loss = final_loss + self.lambda_sym_corners_3d * loss
        losses["sym_corners_3d_loss"] = loss
        losses[self.output_key] = final_loss
        return final_loss, losses


class _FusedLossFn(torch.autograd.Function):
    """final_loss as a function of the network's raw outputs; the kernel has already produced its gradient."""

    @staticmethod
    def forward(ctx, kp3d, box6d, fused, targs):
        o = fused(kp3d.detach().contiguous(), box6d.detach().contiguous(), box6d.shape[-1], targs, backward=True)
        vals, sym = o["losses"].clone(), o["sym_loss"].clone()
        ctx.save_for_backward(o["g_kp3d"].clone(), o["g_box6d"].clone())
        ctx.mark_non_differentiable(vals, sym)
        return vals[5].clone(), vals, sym

    @staticmethod
    def backward(ctx, g_total, g_vals, g_sym):
        gk, gb = ctx.saved_tensors
        return gk * g_total, gb * g_total, None, None


class FusedPoseCriterion:
    """Pose assembly + JointsLoss + HandOrdLoss + SceneOrdLoss + per-sample EPE + backward in ONE HIP kernel
    (csrc/pose_loss.hip).  Built from a `Criterion` whose loss list is [JointsLoss, HandOrdLoss, SceneOrdLoss] (any
    subset); uses that criterion's draw buffers so the RNG behaviour is the reference's."""

    def __init__(self, criterion: Criterion, inp_res, center_idx=0):
        import ctypes
        self.crit = criterion
        self.inp_res = inp_res
        self.center_idx = center_idx
        w = [0.0] * 8
        self.hand = self.scene = self.sym = None
        self.sym_weight = 0.0
        self._sym_dev = None
        for loss in criterion.loss_list:
            lam = criterion.loss_lambdas[type(loss).__name__]
            if isinstance(loss, JointsLoss):
                w[0], w[1], w[5] = float(loss.lambda_joints_3d), float(loss.lambda_corners_3d), float(lam)
            elif isinstance(loss, HandOrdLoss):
                w[2], w[3], w[6] = loss.lambda_joint_lev, loss.lambda_part_lev, float(lam)
                self.hand = loss
            elif isinstance(loss, SceneOrdLoss):
                w[4], w[7] = loss.lambda_scene_lev, float(lam)
                self.scene = loss
            elif isinstance(loss, SymCornerLoss):
                self.sym = loss
                self.sym_weight = float(lam)
            else:
                raise NotImplementedError(f"{type(loss).__name__} is not part of the fused criterion")
        self.weights = (ctypes.c_float * 8)(*w)
        self.out = None

    def draw(self, dev):
        self.crit.draw(dev)

    def _alloc(self, B, dev):
        z = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)   # noqa: E731
        self.out = dict(joints_3d_abs=z(B, 21, 3), corners_3d_abs=z(B, 8, 3), box_rot_rotmat=z(B, 3, 3),
                        uvd2d=z(B, 30, 3), sample_part=z(B, 8), losses=z(8), g_kp3d=z(B, 22, 3), g_box6d=z(B, 6),
                        sym_loss=torch.zeros(1, dtype=torch.float32, device=dev))

    def __call__(self, kp3d, box6d_buf, box_stride, targs, backward=True):
        """kp3d [B,22,3] f32; box6d_buf: f32 buffer whose rows (pitch box_stride) start with the 6-D rotation."""
        from . import _lib as L
        B, dev = kp3d.shape[0], kp3d.device
        if self.out is None or self.out["g_kp3d"].shape[0] != B:
            self._alloc(B, dev)
        o = self.out
        hb = self.hand.draws.bufs if self.hand is not None else {}
        sb = self.scene.draws.bufs if self.scene is not None else {}
        nvh = hb["views"].shape[0] if hb else 0
        nvs = sb["views"].shape[0] if sb else 0
        t = lambda k: targs[k]   # noqa: E731
        symp = None
        if self.sym is not None and self.sym.lambda_sym_corners_3d:
            import ctypes
            if self._sym_dev is None or self._sym_dev[0].device != dev:
                self._sym_dev = (self.sym.R.to(dev).contiguous(), self.sym.t.to(dev).reshape(self.sym.t.shape[0], -1, 3).contiguous())
            Rd, td = self._sym_dev
            obj = t(Queries.OBJ_IDX)
            if obj.dtype != torch.int64:
                raise TypeError("obj_idx must be int64")
            self._sym_struct = L.SymCorner(L.addr(Rd), L.addr(td), int(Rd.shape[1]), L.addr(obj), L.addr(t(Queries.OBJ_TRANSF)),
                                           float(self.sym.lambda_sym_corners_3d), float(self.sym_weight), L.addr(o["sym_loss"]))
            symp = ctypes.byref(self._sym_struct)
        L.check(L.lib().ab_pose_loss_sym(
            L.ptr(kp3d), L.ptr(box6d_buf), L.i(box_stride), L.ptr(t(Queries.ROOT_JOINT)), L.ptr(t(Queries.CAM_INTR)),
            L.ptr(t(Queries.CORNERS_CAN)), L.ptr(t(Queries.JOINTS_3D)), L.ptr(t(Queries.CORNERS_3D)),
            L.ptr(t(Queries.JOINTS_VIS)), L.ptr(t(Queries.CORNERS_VIS)),
            L.ptr(hb.get("views")), L.i(nvh), L.ptr(hb.get("j0")), L.ptr(hb.get("j1")), L.i(hb["j0"].numel() if hb else 0),
            L.ptr(hb.get("p0")), L.ptr(hb.get("p1")), L.i(hb["p0"].numel() if hb else 0),
            L.ptr(sb.get("views")), L.i(nvs), L.ptr(sb.get("i0")), L.ptr(sb.get("i1")), L.i(sb["i0"].numel() if sb else 0),
            L.i(B), L.i(self.center_idx), L.f(self.inp_res[0]), L.f(self.inp_res[1]), self.weights, symp,
            L.ptr(o["joints_3d_abs"]), L.ptr(o["corners_3d_abs"]), L.ptr(o["box_rot_rotmat"]), L.ptr(o["uvd2d"]),
            L.ptr(o["sample_part"]), L.ptr(o["losses"]), L.ptr(o["g_kp3d"] if backward else None),
            L.ptr(o["g_box6d"] if backward else None), L.stream()), "ab_pose_loss_sym")
        return o

    LOSS_KEYS = ("joints_3d_loss", "corners_3d_loss", "joint_ord_loss", "part_ord_loss", "scene_ord_loss", "final_loss")

    def losses_dict(self):
        """Host view of the loss scalars of the last call (synchronises)."""
        v = self.out["losses"].cpu()
        d = {k: v[i] for i, k in enumerate(self.LOSS_KEYS)}
        if self.sym is not None:
            d["sym_corners_3d_loss"] = self.out["sym_loss"].cpu()[0]
        return d

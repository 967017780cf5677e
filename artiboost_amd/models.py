"""Model plugin classes with the reference's names and call contract (anakin/models/arch.py:11-72,
anakin/models/hybridbaseline.py:18-129) on top of the HIP executor (hybridnet.HybridNet).

`Arch(cfg, model_list)(batch) -> {TYPE: preds}`; `preds` carries the 7 keys of hybridbaseline.py:86-96.
`final_loss.backward()` on anything computed from `preds` runs the hand-written backward kernels and leaves the
gradient in `model.flat_param.grad` (one flat tensor -> clip_grad_norm_ / Adam see one parameter)."""
import os
from collections import OrderedDict
from typing import Dict

import torch
import torch.nn as nn

from .hybridnet import HybridNet, ParamStore
from .registry import MODEL, RUNTIME, Queries, enable_lower_param, image_plane_of


def ortho6d_to_rotmat(poses):
    """compute_rotation_matrix_from_ortho6d (anakin/utils/transform.py:578-618)."""
    def nrm(v):
        mag = torch.clamp(torch.sqrt(v.pow(2).sum(1)), min=1e-8)
        return v / mag[:, None]

    x = nrm(poses[:, 0:3])
    z = nrm(torch.cross(x, poses[:, 3:6], dim=1))
    y = torch.cross(z, x, dim=1)
    return torch.stack([x, y, z], dim=2)


def batch_uvd2xyz(uvd, root_joint, intr, inp_res, depth_range=0.4):
    """anakin/utils/transform.py:512-546 (ref_bone_len == 1)."""
    uv = torch.stack([uvd[:, :, 0] * float(inp_res[0]), uvd[:, :, 1] * float(inp_res[1])], dim=2)
    z = (uvd[:, :, 2] - 0.5) * depth_range + root_joint[:, 2:3]
    f = torch.stack([intr[:, 0, 0], intr[:, 1, 1]], 1)[:, None, :]
    c = torch.stack([intr[:, 0, 2], intr[:, 1, 2]], 1)[:, None, :]
    xy = (uv - c) / f * z[..., None]
    return torch.cat([xy, z[..., None]], -1)


class _NetSegment:
    """The network half of an EAGER step as two replayed hipGraphs (forward + soft-argmax | backward) for one input shape.

    The reference-shaped loop (train/train_artiboost.py:66-96) calls `arch_model(batch)`, the criterion, `backward()` and the
    optimizer one after the other from Python; issued kernel by kernel that is ~600 launches (18-25 ms of host time) per step
    for 11 ms of device work.  The first call with a given shape runs eagerly (lazy state, allocator); the second captures
    the same call sequence into a graph whose inputs / outputs / saved activations live at fixed addresses, and every later
    call is one input copy + one replay.  The criterion, the evaluator and the optimizer stay ordinary eager code.
    AB_SEGMENT_GRAPHS=0 disables it."""
    MAX_SHAPES = 4

    def __init__(self):
        self.calls = 0
        self.pending = False          # a grad-mode forward whose backward has not run yet owns the saved activations
        self.fwd = self.bwd = None
        self.x = self.out = self.saved = self.last = self.gk = self.gb = None

    def forward(self, net, image, xpad, grad):
        """-> (logits, kp3d, conf, stat, box6d) at fixed addresses, or None (run eagerly this time)."""
        src = xpad if xpad is not None else image
        self.calls += 1
        if self.pending:
            return None
        if self.fwd is None:
            if self.calls < 2 or torch.cuda.is_current_stream_capturing():
                return None
            self.x = src.detach().clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                logits, box6d = net.forward(image=None if xpad is not None else self.x, xpad=self.x if xpad is not None else None)
                kp3d, conf, stat = net.head_fwd(logits)
                box = box6d.contiguous()
            self.fwd, self.out, self.saved, self.last = g, (logits, kp3d, conf, stat, box), net.saved, net.last
        elif torch.cuda.is_current_stream_capturing():
            return None
        else:
            self.x.copy_(src, non_blocking=True)
        self.fwd.replay()
        if net.training:
            net.p.num_batches_tracked += 1
        net.saved, net.last = self.saved, self.last
        self.pending = bool(grad)
        return self.out

    def backward(self, net, g_kp3d, g_box6d):
        logits, kp3d, conf, stat, _ = self.out
        net.saved, net.last = self.saved, self.last
        self.pending = False
        if torch.cuda.is_current_stream_capturing():
            net.backward(net.head_bwd(logits, kp3d, conf, stat, g_kp3d), g_box6d)
            return
        if self.bwd is None:
            self.gk, self.gb = g_kp3d.detach().clone(), g_box6d.detach().clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self.fwd.pool(), capture_error_mode="thread_local"):
                net.backward(net.head_bwd(logits, kp3d, conf, stat, self.gk), self.gb)
            self.bwd = g
        else:
            self.gk.copy_(g_kp3d, non_blocking=True)
            self.gb.copy_(g_box6d, non_blocking=True)
        self.bwd.replay()


class _NetBridge(torch.autograd.Function):
    """Autograd boundary: forward = HIP forward + fused soft-argmax; backward = HIP backward into the flat grad."""

    @staticmethod
    def forward(ctx, flat_param, owner, image, xpad, seg):
        net = owner.net
        res = seg.forward(net, image, xpad, True) if seg is not None else None
        if res is not None:
            logits, kp3d, conf, stat, box = res
            ctx.seg = seg
            ctx.pack = (logits, kp3d, conf, stat)
            kp3d, conf, box = kp3d.clone(), conf.clone(), box.clone()      # the caller's tensors outlive the next replay
        else:
            logits, box6d = net.forward(image=image, xpad=xpad)
            kp3d, conf, stat = net.head_fwd(logits)
            ctx.seg = None
            ctx.pack = (logits, kp3d, conf, stat)
            box = box6d.contiguous()
        ctx.owner = owner
        ctx.mark_non_differentiable(conf)
        return kp3d, conf, box

    @staticmethod
    def backward(ctx, g_kp3d, g_conf, g_box6d):
        owner = ctx.owner
        net = owner.net
        logits, kp3d, conf, stat = ctx.pack
        if g_kp3d is None:
            g_kp3d = torch.zeros_like(kp3d)
        if g_box6d is None:
            g_box6d = torch.zeros((kp3d.shape[0], 6), dtype=torch.float32, device=kp3d.device)
        g_kp3d, g_box6d = g_kp3d.contiguous().float(), g_box6d.contiguous().float()
        if ctx.seg is not None:
            ctx.seg.backward(net, g_kp3d, g_box6d)
        else:
            dlogits = net.head_bwd(logits, kp3d, conf, stat, g_kp3d)
            net.backward(dlogits, g_box6d)
        owner.flat_param.grad = owner.store.grad      # the kernels wrote it; no copy, no accumulation
        return None, None, None, None, None


class _AssembleFn(torch.autograd.Function):
    """hybridbaseline.py:49-96 in grad mode as ONE launch (ab_pose_assemble) instead of ~35 small torch kernels and their autograd nodes.
    With the fused criterion (Criterion.compute_losses on the raw kp3d / box6d through the `_ab_fuse` link) nothing differentiates through
    these outputs and the backward below never runs; a criterion that does (registry losses + autograd, FUSED_CRITERION off) gets the
    gradient by re-running the torch-op assembly under autograd -- same values to fp32 rounding, paid only on that route."""
    KEYS = ("joints_3d_abs", "corners_3d_abs", "joints_3d", "corners_3d", "uvd2d", "boxroot_3d_abs", "box_rot_rotmat")

    @staticmethod
    def forward(ctx, kp3d, box6d, root_in, intr, corners_can, owner):
        from . import kernels as K
        o = K.pose_assemble(kp3d.contiguous(), box6d if box6d.stride(-1) == 1 else box6d.contiguous(), root_in, intr, corners_can,
                            owner.center_idx, owner.inp_res)
        ctx.owner = owner
        ctx.save_for_backward(kp3d, box6d, root_in, intr, corners_can)
        return tuple(o[k] for k in _AssembleFn.KEYS)

    @staticmethod
    def backward(ctx, *grads):
        kp3d, box6d, root_in, intr, corners_can = ctx.saved_tensors
        owner = ctx.owner
        with torch.enable_grad():
            k_, b_ = kp3d.detach().requires_grad_(True), box6d.detach().requires_grad_(True)
            out = owner._assemble_torch(k_, b_, root_in, intr, corners_can, float(owner.inp_res[1]), float(owner.inp_res[0]))
            outs = [out[k if k != "uvd2d" else "2d_uvd"] for k in _AssembleFn.KEYS]
            pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o.requires_grad]
            gk, gb = torch.autograd.grad([o for o, _ in pairs], [k_, b_], [g for _, g in pairs], allow_unused=True)
        return gk, gb, None, None, None, None


# resnet.py:232-276: block type and stage counts of the five registered backbones
BACKBONES = {"ResNet18": ("basic", (2, 2, 2, 2)), "ResNet34": ("basic", (3, 4, 6, 3)), "ResNet50": ("bottleneck", (3, 4, 6, 3)),
             "ResNet101": ("bottleneck", (3, 4, 23, 3)), "ResNet152": ("bottleneck", (3, 8, 36, 3))}


@MODEL.register_module
class HybridBaseline(nn.Module):
    HEAD_KEY, HEAD_PREFIX, HAS_BOX_HEAD = "HYBRID_HEAD", "hybrid_head", True      # config block / attribute name of the heat-map head

    @enable_lower_param
    def __init__(self, **cfg):
        super().__init__()
        preset = cfg["DATA_PRESET"]
        self.center_idx = preset.get("CENTER_IDX", 9)
        self.inp_res = preset["IMAGE_SIZE"]
        head = cfg[self.HEAD_KEY]
        self.nclasses = head["NCLASSES"]
        self.depth_res = head["DEPTH_RESOLUTION"]
        if cfg["BACKBONE"]["TYPE"] not in BACKBONES:
            raise NotImplementedError(f"backbone {cfg['BACKBONE']['TYPE']}: one of {sorted(BACKBONES)} (resnet.py:232-276)")
        block, layers = BACKBONES[cfg["BACKBONE"]["TYPE"]]
        feat_ch = 512 * (4 if block == "bottleneck" else 1)
        if (head.get("NUM_DECONV_LAYERS", 2), list(head.get("NUM_DECONV_FILTERS", [256, 256])), list(head.get("NUM_DECONV_KERNELS", [4, 4])),
                head.get("INPUT_CHANNEL", feat_ch), bool(head.get("DECONV_WITH_BIAS", False))) != (2, [256, 256], [4, 4], feat_ch, False):
            raise NotImplementedError(f"IntegralDeconvHead: 2 x (ConvTranspose2d 4x4/s2, 256 filters, no bias) on the backbone's {feat_ch} channels only")
        box_dims = (feat_ch, 256, 128)
        if self.HAS_BOX_HEAD:
            bh = cfg.get("BOX_HEAD", {})
            box_dims = tuple(bh.get("LAYERS_N", [feat_ch, 256, 128]))
            if len(box_dims) != 3 or box_dims[0] != feat_ch or bh.get("OUT_CHANNEL", 6) != 6:
                raise NotImplementedError(f"MLP_O: LAYERS_N [{feat_ch}, h1, h2] and OUT_CHANNEL 6 only")
        from .head import norm_code
        norm = norm_code(head.get("NORM_TYPE", "softmax"))          # softmax / sigmoid (simplebaseline.py:16-40); divide_sum raises
        if head.get("FINAL_CONV_KERNEL", 1) != 1:
            raise NotImplementedError("IntegralDeconvHead: 1x1 final conv only")
        if cfg["BACKBONE"].get("PRETRAINED") is True:
            # resnet.py:249-262 fetches torchvision's ImageNet weights (a download); here the backbone keeps its seeded
            # initialisation unless ARCH.PRETRAINED names a checkpoint in the reference's state-dict layout
            import warnings
            warnings.warn("BACKBONE.PRETRAINED: true -- ImageNet weights are a torchvision download and are not fetched; "
                          "pass a converted checkpoint through ARCH.PRETRAINED")
        dev = cfg.get("DEVICE", "cuda")
        cd = cfg.get("COMPUTE_DTYPE", "bf16x3")     # the reference's precision (fp32-grade); "bf16" / "f32" opt in
        self.store = ParamStore(self.nclasses, self.depth_res, device=dev, layers=layers, head_prefix=self.HEAD_PREFIX,
                                box_head=self.HAS_BOX_HEAD, block=block, box_dims=box_dims)
        self.store.init_reference_like(seed=int(cfg.get("INIT_SEED", 1)))
        self.net = HybridNet(self.store, image_size=self.inp_res,
                             compute_dtype=(torch.bfloat16 if cd in ("bf16", torch.bfloat16) else
                                            "bf16x3" if cd in ("bf16x3", "x3") else torch.float32))
        # BACKBONE.FREEZE_BATCHNORM (resnet.py:146-149: bn_layer = FrozenBatchNorm2d): the backbone's BatchNorms are fixed affine maps in
        # both modes, their weight / bias receive no gradient (zero gradient -> Adam leaves them) and carry no num_batches_tracked
        self.net.frozen_bn = self.store.frozen_bn = bool(cfg["BACKBONE"].get("FREEZE_BATCHNORM", False))
        self.net.norm = norm
        # the padded image this model consumes natively: a loader built afterwards with the reference's keywords follows it (registry.RUNTIME)
        RUNTIME["loader_compute_dtype"] = ("u8n" if self.net.x3 and os.environ.get("AB_IMAGE_PLANE", "u8n") == "u8n" else self.net.dtype)
        self.flat_param = nn.Parameter(self.store.flat, requires_grad=True)   # shares storage with the store
        self.flat_param._ab_owner = self                                      # netutils.build_optimizer recognises it
        self.segment_graphs = bool(cfg.get("SEGMENT_GRAPHS", os.environ.get("AB_SEGMENT_GRAPHS", "1") != "0"))
        self._segments = {}
        pretrained = cfg.get("PRETRAINED", "")
        if pretrained:
            self.load_pretrained(pretrained)

    # --- reference-compatible checkpoint I/O (hybridbaseline.py:98-129, utils/io_utils.py:19-51)
    def load_pretrained(self, path):
        ckpt = torch.load(path, map_location="cpu")
        sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
        self.store.load_reference_state_dict(sd, strict=False)
        self.net._packed = False

    def state_dict(self, *a, **k):
        return OrderedDict((k_, v.cpu()) for k_, v in self.store.reference_state_dict().items())

    def load_state_dict(self, sd, strict=True):
        self.store.load_reference_state_dict(sd, strict=strict)
        self.net._packed = False

    def train(self, mode=True):
        super().train(mode)
        self.net.training = mode
        return self

    fused_assembly = os.environ.get("AB_FUSED_ASSEMBLY", "1") != "0"      # eval / no_grad forwards: pose assembly as one kernel

    def _segment(self, image, xpad):
        """The graph pair for this call's (mode, input shape), or None when segment graphs do not apply."""
        src = xpad if xpad is not None else image
        if not self.segment_graphs or src is None or not src.is_cuda:
            return None
        key = (self.training, torch.is_grad_enabled(), xpad is not None, tuple(src.shape), src.dtype)
        seg = self._segments.get(key)
        if seg is None:
            if len(self._segments) >= _NetSegment.MAX_SHAPES:
                return None
            seg = self._segments[key] = _NetSegment()
        return seg

    def _replicate_for_data_parallel(self):
        """nn.DataParallel over several GPUs (train_artiboost.py:131 with more than one visible device) would clone this module
        per device; its parameters, activations and hipGraphs live in ONE device's flat buffers.  Multi-GPU training here is
        one process per GPU (bench.py --gpus N, tools/train_artiboost.py under torch.distributed.run: RCCL all-reduce)."""
        raise RuntimeError("HybridBaseline (HIP) cannot be replicated by nn.DataParallel: run one process per GPU "
                           "(python -m torch.distributed.run --nproc-per-node N tools/train_artiboost.py ...), or restrict "
                           "DataParallel to one device (--gpu_id 0 / CUDA_VISIBLE_DEVICES=0)")

    def params_updated(self):
        """Call after the optimizer changed flat_param (refreshes the compute-precision weight copies)."""
        self.net._packed = False

    def forward(self, inputs: Dict):
        dev = self.store.device
        xpad = inputs.get("image_nhwc4_padded")
        image = None
        if xpad is None:
            image = inputs.get(Queries.IMAGE).to(dev, non_blocking=True)
            H, W = image.shape[2], image.shape[3]
            self.net.image_plane = "f32"
        else:           # (the batch's lazily made CHW `image` is not touched)
            H, W = xpad.shape[1] - 6, xpad.shape[2] - 8
            self.net.image_plane = self._plane_of(inputs, xpad)
        if not self.net._packed or self.flat_param._version != getattr(self, "_seen_version", -1):
            self.net.pack_weights()      # torch-side update (e.g. torch.optim.Adam); the fused optimizer repacks itself
            self._seen_version = self.flat_param._version
        if self.training and torch.is_grad_enabled():
            kp3d, conf, box6d = _NetBridge.apply(self.flat_param, self, image, xpad, self._segment(image, xpad))
        else:
            with torch.no_grad():
                seg = self._segment(image, xpad)
                res = seg.forward(self.net, image, xpad, False) if seg is not None else None
                if res is not None:
                    kp3d, conf, box6d = res[1].clone(), res[2].clone(), res[4].clone()
                else:
                    logits, box6d = self.net.forward(image=image, xpad=xpad)
                    kp3d, conf, _ = self.net.head_fwd(logits)
        return self._assemble(inputs, kp3d, conf, box6d, H, W)

    def _plane_of(self, inputs, xpad):
        """The image plane of a padded NHWC4 input, from the tag its loader wrote (registry.tag_image_plane / IMAGE_PLANE_KEY) -- never
        guessed: an untagged bfloat16 image handed to a bf16x3 model is a TypeError (it could be x / 255 - 0.5 rounded to bf16 or the
        integer plane 2 v - 255; reading one as the other trains on inputs off by 510x)."""
        plane = image_plane_of(inputs, xpad)
        if xpad.dim() == 4 and xpad.dtype == torch.bfloat16 and self.net.x3:
            if plane != "u8n":
                raise TypeError("a bfloat16 padded image for a bf16x3 model must be the loaders' integer plane (compute_dtype=\"u8n\", tagged "
                                f"image_plane 'u8n'); got tag {plane!r} -- build the loader with compute_dtype=\"u8n\" or torch.float32")
            return "u8n"
        if plane == "u8n":
            raise TypeError(f"image_plane 'u8n' (bf16 integers 2 v - 255) needs a bf16x3 model and a bfloat16 [N,H+6,W+8,4] tensor; this model "
                            f"computes in {'bf16x3' if self.net.x3 else self.net.dtype} and got {xpad.dtype} {tuple(xpad.shape)}")
        return "f32"

    def _assemble(self, inputs, kp3d, conf, box6d, H, W):
        """hybridbaseline.py:49-96: uvd -> xyz, 6-D -> R, canonical corners -> camera frame, 2-D re-projection."""
        dev = self.store.device
        root_in = inputs[Queries.ROOT_JOINT].to(dev)
        intr = inputs[Queries.CAM_INTR].to(dev)
        if not kp3d.requires_grad and self.fused_assembly and (W, H) == tuple(self.inp_res):
            # no autograd graph to build (eval / no_grad): the whole pose assembly below is ONE kernel (ab_pose_assemble)
            from . import kernels as K
            f32c = lambda t: t.to(dev, torch.float32).contiguous()      # noqa: E731
            o = K.pose_assemble(kp3d.contiguous(), box6d if box6d.stride(-1) == 1 else box6d.contiguous(), f32c(root_in), f32c(intr),
                                f32c(inputs[Queries.CORNERS_CAN]), self.center_idx, self.inp_res)
            return {"joints_3d_abs": o["joints_3d_abs"], "corners_3d_abs": o["corners_3d_abs"], "joints_3d": o["joints_3d"],
                    "corners_3d": o["corners_3d"], "2d_uvd": o["uvd2d"], "boxroot_3d_abs": o["boxroot_3d_abs"],
                    "box_rot_rotmat": o["box_rot_rotmat"], "kp3d": kp3d, "kp3d_confd": conf}
        if self.fused_assembly and kp3d.is_cuda and (W, H) == tuple(self.inp_res):
            f32c = lambda t: t.to(dev, torch.float32).contiguous()      # noqa: E731
            ja, ca, j, c, uvd, br, rot = _AssembleFn.apply(kp3d, box6d, f32c(root_in), f32c(intr), f32c(inputs[Queries.CORNERS_CAN]), self)
            ja._ab_fuse = dict(kp3d=kp3d, box6d=box6d, inp_res=self.inp_res, center_idx=self.center_idx)
            return {"joints_3d_abs": ja, "corners_3d_abs": ca, "joints_3d": j, "corners_3d": c, "2d_uvd": uvd, "boxroot_3d_abs": br,
                    "box_rot_rotmat": rot, "kp3d": kp3d, "kp3d_confd": conf}
        out = self._assemble_torch(kp3d, box6d, root_in, intr, inputs[Queries.CORNERS_CAN].to(dev), H, W)
        if kp3d.requires_grad:      # lets Criterion.compute_losses run the fused pose/loss kernel on the raw outputs
            out["joints_3d_abs"]._ab_fuse = dict(kp3d=kp3d, box6d=box6d, inp_res=self.inp_res, center_idx=self.center_idx)
        out.update(kp3d=kp3d, kp3d_confd=conf)
        return out

    def _assemble_torch(self, kp3d, box6d, root_in, intr, corners_can_3d, H, W):
        """The assembly as differentiable torch ops (CPU tensors, other image sizes, and _AssembleFn's backward)."""
        pose_3d_abs = batch_uvd2xyz(kp3d, root_in, intr, self.inp_res)
        joints_3d_abs = pose_3d_abs[:, 0:21, :]
        boxroot_3d_abs = pose_3d_abs[:, 21:22, :]
        box_rot_rotmat = ortho6d_to_rotmat(box6d)
        corners_3d_abs = torch.matmul(box_rot_rotmat, corners_can_3d.permute(0, 2, 1)).permute(0, 2, 1) + boxroot_3d_abs
        root_joint = joints_3d_abs[:, self.center_idx, :]
        corners_2d = torch.matmul(intr, corners_3d_abs.permute(0, 2, 1)).permute(0, 2, 1)
        corners_2d = corners_2d[:, :, 0:2] / corners_2d[:, :, 2:3]
        corners_2d = torch.stack([corners_2d[:, :, 0] / W, corners_2d[:, :, 1] / H], dim=2)
        corners_2d_uvd = torch.cat((corners_2d, torch.zeros_like(corners_2d[:, :, 0:1])), dim=2)
        final_2d_uvd = torch.cat((kp3d[:, 0:21, :], corners_2d_uvd, kp3d[:, 21:22, :]), dim=1)
        return {
            "joints_3d_abs": joints_3d_abs,
            "corners_3d_abs": corners_3d_abs,
            "joints_3d": joints_3d_abs - root_joint.unsqueeze(1),
            "corners_3d": corners_3d_abs - root_joint.unsqueeze(1),
            "2d_uvd": final_2d_uvd,
            "boxroot_3d_abs": boxroot_3d_abs,
            "box_rot_rotmat": box_rot_rotmat,
        }


@MODEL.register_module
class SimpleBaseline(HybridBaseline):
    """anakin/models/simplebaseline.py:194-241: backbone + IntegralDeconvHead with ONE heat map per key point (29 = 21 joints + 8 corners,
    no MLP_O box head); the same HIP executor with `pose_head.*` parameter names, no box head and a zero-weight padding class (the final
    layer's channel count must be a multiple of 64 for the weight-gradient kernels).  Losses run through the registry classes + autograd
    (the fused pose/loss kernel is HybridBaseline's assembly)."""
    HEAD_KEY, HEAD_PREFIX, HAS_BOX_HEAD = "HEAD", "pose_head", False

    def _assemble(self, inputs, kp3d, conf, box6d, H, W):
        dev = self.store.device
        kp3d_abs = batch_uvd2xyz(kp3d, inputs[Queries.ROOT_JOINT].to(dev), inputs[Queries.CAM_INTR].to(dev), self.inp_res)
        nj = 21                                           # CONST.NUM_JOINTS (misc.py)
        joints_3d_abs, corners_3d_abs = kp3d_abs[:, :nj], kp3d_abs[:, nj:]
        root_joint = joints_3d_abs[:, self.center_idx, :]
        return {"joints_3d_abs": joints_3d_abs, "corners_3d_abs": corners_3d_abs, "joints_3d": joints_3d_abs - root_joint.unsqueeze(1),
                "corners_3d": corners_3d_abs - root_joint.unsqueeze(1), "2d_uvd": kp3d}


class Arch(nn.Module):
    """anakin/models/arch.py:11-72: DAG of models keyed by TYPE with PREVIOUS edges."""

    def __init__(self, cfg: Dict, model_list):
        super().__init__()
        self._model_list = nn.ModuleList(model_list)
        self._cfg = cfg
        items = cfg["ARCH"]
        if isinstance(items, dict):
            items = [items]
        self.models = {it["TYPE"]: {"id": i, "previous": it["PREVIOUS"]} for i, it in enumerate(items)}
        outdeg = [0] * len(items)
        for v in self.models.values():
            for p in v["previous"]:
                outdeg[self.models[p]["id"]] += 1
        if outdeg.count(0) != 1:
            raise Exception("Arch has multiple roots, a circle or other illegal input.!")
        self.root = items[outdeg.index(0)]["TYPE"]

    @property
    def model_list(self):
        return self._model_list

    @property
    def models_params(self):
        return [{"params": filter(lambda p: p.requires_grad, m.parameters())} for m in self._model_list]

    def forward(self, input: Dict):
        self.outputs = {}
        self._forward(self.root, input)
        return self.outputs

    def _forward(self, mtype, input):
        inputs = dict(input)
        for p in self.models[mtype]["previous"]:
            if p not in self.outputs:
                self._forward(p, input)
            inputs.update(self.outputs[p])
        self.outputs[mtype] = self._model_list[self.models[mtype]["id"]](inputs)

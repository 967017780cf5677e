"""HybridBaseline (ResNet-34 + IntegralDeconvHead(22 x D) + MLP_O) executed on the hand-written HIP kernels.

Reference: anakin/models/hybridbaseline.py:18-96, anakin/models/resnet.py:72-101,142-230,243-248,
anakin/models/simplebaseline.py:75-190, anakin/models/mlp.py:11-25.

Design (MI355X-first, not a translation of the nn.Module tree):
  * all learnable parameters live in ONE flat fp32 buffer in *kernel layout* (conv weights OHWI / K-contiguous, the
    7x7 stem as [64][7][8][4] rows over a zero-bordered NHWC4 image, ConvTranspose weights as the OHWI weights of the
    mirrored conv, the heat-map depth padded 28 -> 32 so every class is one aligned channel run).  Gradients live in
    a second flat buffer of the same shape, so the global-norm clip + Adam are two passes over contiguous memory and
    the DDP all-reduce is a handful of large buckets.
  * activations are NHWC in the compute dtype; forward and backward are explicit kernel sequences (no autograd graph);
    training-mode BatchNorm statistics come for free from the conv epilogue.  Three precisions:
      "bf16x3" (default, the reference's precision): fp32 activations / gradients in HBM, every convolution on the bf16
               MFMA with split operands hi + lo (3 passes, fp32 accumulate: 2^-17 operand precision) -- see csrc/conv_x3.hip;
      "f32"    exact-f32 MFMA everywhere (1/16 of the bf16 rate): the cross-check of bf16x3;
      "bf16"   bf16 operands and activations (fastest; misses the 1e-3 parity bound of the north star).
  * state_dict()/load_state_dict() speak the reference's key names and tensor layouts (torchvision-compatible
    `backbone.*`, `hybrid_head.deconv_layers.*`, `hybrid_head.final_layer.*`, `box_head.layers.*`).
"""
from collections import OrderedDict

import os

import torch

from . import kernels as K
from .head import softargmax3d_fwd, softargmax3d_bwd, softargmax3d_bwd_x3, softargmax3d_stage2

RESNET34_LAYERS = [3, 4, 6, 3]
DEPTH_PITCH = 32
BOX_OUT_PAD = 64


class _Entry:
    __slots__ = ("name", "kind", "ref_shape", "kshape", "offset", "numel", "bn")

    def __init__(self, name, kind, ref_shape, kshape):
        self.name, self.kind, self.ref_shape, self.kshape = name, kind, tuple(ref_shape), tuple(kshape)
        self.numel = 1
        for d in kshape:
            self.numel *= d
        self.offset = 0


def _round_up(x, m):
    return (x + m - 1) // m * m


class ParamStore:
    """Flat parameter / gradient / running-stat storage with reference <-> kernel layout conversion."""

    def __init__(self, nclasses=22, depth=28, device="cuda", layers=(3, 4, 6, 3), head_prefix="hybrid_head", box_head=True,
                 block="basic", box_dims=(512, 256, 128)):
        """layers: BasicBlock counts per stage ((3,4,6,3) = ResNet-34, (2,2,2,2) = ResNet-18: resnet.py:236-248); head_prefix: the
        IntegralDeconvHead's attribute name in the reference module ("hybrid_head" in HybridBaseline, "pose_head" in SimpleBaseline:
        hybridbaseline.py:31, simplebaseline.py:207); box_head: MLP_O present (HybridBaseline only).
        The final layer is laid out for `nclasses_pad` classes (even, so that its channel count is a multiple of 64 for the weight-gradient
        kernels): a padding class has zero weights, zero bias and receives zero gradient."""
        self.nclasses, self.depth = nclasses, depth
        self.nclasses_pad = nclasses + (nclasses & 1)
        self.layers, self.hp, self.box_head = tuple(layers), head_prefix, bool(box_head)
        # block: "basic" (ResNet-18/34, resnet.py:72-101) or "bottleneck" (ResNet-50/101/152, resnet.py:104-141: 1x1 -> 3x3 (stride) -> 1x1 x4)
        self.block, self.expansion = block, (4 if block == "bottleneck" else 1)
        self.feat_ch = 512 * self.expansion                # res_layer4 channels = the head's INPUT_CHANNEL
        self.box_dims = tuple(box_dims)                    # MLP_O LAYERS_N (mlp.py:11-25)
        self.entries = OrderedDict()
        self.buffers = OrderedDict()   # running stats: name -> (offset, C)
        self._build_table()
        off = 0
        for e in self.entries.values():
            e.offset = off
            off += _round_up(e.numel, 64)          # keep every tensor 256-byte aligned
        self.total = off
        boff = 0
        for k in list(self.buffers):
            self.buffers[k] = (boff, self.buffers[k][1])
            boff += _round_up(self.buffers[k][1], 64)
        self.device = torch.device(device)
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.stats = torch.zeros(boff, dtype=torch.float32, device=self.device)
        self.num_batches_tracked = 0

    # ------------------------------------------------------------------ table
    def _add(self, name, kind, ref_shape, kshape):
        self.entries[name] = _Entry(name, kind, ref_shape, kshape)

    def _add_bn(self, prefix, c):
        self._add(prefix + ".weight", "vec", (c,), (c,))
        self._add(prefix + ".bias", "vec", (c,), (c,))
        self.buffers[prefix + ".running_mean"] = (0, c)
        self.buffers[prefix + ".running_var"] = (0, c)

    def _build_table(self):
        C, D, CP, hp = self.nclasses, self.depth, self.nclasses_pad, self.hp
        self._add("backbone.conv1.weight", "stem", (64, 3, 7, 7), (64, 7, 8, 4))
        self._add_bn("backbone.bn1", 64)
        inpl = 64
        for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], self.layers), start=1):
            for b in range(nblk):
                stride = 2 if (b == 0 and li > 1) else 1
                p = f"backbone.layer{li}.{b}"
                if self.block == "bottleneck":
                    outp = planes * 4
                    self._add(p + ".conv1.weight", "conv", (planes, inpl, 1, 1), (planes, 1, 1, inpl))
                    self._add_bn(p + ".bn1", planes)
                    self._add(p + ".conv2.weight", "conv", (planes, planes, 3, 3), (planes, 3, 3, planes))
                    self._add_bn(p + ".bn2", planes)
                    self._add(p + ".conv3.weight", "conv", (outp, planes, 1, 1), (outp, 1, 1, planes))
                    self._add_bn(p + ".bn3", outp)
                    if stride != 1 or inpl != outp:
                        self._add(p + ".downsample.0.weight", "conv", (outp, inpl, 1, 1), (outp, 1, 1, inpl))
                        self._add_bn(p + ".downsample.1", outp)
                    inpl = outp
                    continue
                self._add(p + ".conv1.weight", "conv", (planes, inpl, 3, 3), (planes, 3, 3, inpl))
                self._add_bn(p + ".bn1", planes)
                self._add(p + ".conv2.weight", "conv", (planes, planes, 3, 3), (planes, 3, 3, planes))
                self._add_bn(p + ".bn2", planes)
                if stride != 1 or inpl != planes:
                    self._add(p + ".downsample.0.weight", "conv", (planes, inpl, 1, 1), (planes, 1, 1, inpl))
                    self._add_bn(p + ".downsample.1", planes)
                inpl = planes
        # backbone.fc exists in the reference state_dict (resnet.py:164) but never receives a gradient
        self._add("backbone.fc.weight", "frozen", (1000, self.feat_ch), (1000, self.feat_ch))
        self._add("backbone.fc.bias", "frozen", (1000,), (1000,))
        self._add(hp + ".deconv_layers.0.weight", "deconv", (self.feat_ch, 256, 4, 4), (self.feat_ch, 4, 4, 256))
        self._add_bn(hp + ".deconv_layers.1", 256)
        self._add(hp + ".deconv_layers.3.weight", "deconv", (256, 256, 4, 4), (256, 4, 4, 256))
        self._add_bn(hp + ".deconv_layers.4", 256)
        self._add(hp + ".final_layer.weight", "final_w", (C * D, 256, 1, 1), (CP * DEPTH_PITCH, 1, 1, 256))
        self._add(hp + ".final_layer.bias", "final_b", (C * D,), (CP * DEPTH_PITCH,))
        if not self.box_head:
            return
        d0, d1, d2 = self.box_dims
        self._add("box_head.layers.0.weight", "linear", (d1, d0), (d1, 1, 1, d0))
        self._add("box_head.layers.0.bias", "vec", (d1,), (d1,))
        self._add("box_head.layers.2.weight", "linear", (d2, d1), (d2, 1, 1, d1))
        self._add("box_head.layers.2.bias", "vec", (d2,), (d2,))
        self._add("box_head.layers.4.weight", "linear_pad", (6, d2), (BOX_OUT_PAD, 1, 1, d2))
        self._add("box_head.layers.4.bias", "vec_pad", (6,), (BOX_OUT_PAD,))

    # ------------------------------------------------------------------ views
    def view(self, name, buf=None):
        e = self.entries[name]
        return (self.flat if buf is None else buf)[e.offset:e.offset + e.numel].view(e.kshape)

    def gview(self, name):
        return self.view(name, self.grad)

    def stat(self, name):
        off, c = self.buffers[name]
        return self.stats[off:off + c]

    def trainable_numel(self):
        return sum(e.numel for e in self.entries.values() if e.kind != "frozen")

    # ------------------------------------------------------------------ layout conversion
    def _to_kernel(self, e, t):
        t = t.to(torch.float32)
        C, D = self.nclasses, self.depth
        if e.kind == "stem":
            k = torch.zeros(e.kshape, dtype=torch.float32, device=t.device)
            k[:, :, :7, :3] = t.permute(0, 2, 3, 1)
            return k
        if e.kind == "conv":
            return t.permute(0, 2, 3, 1).contiguous()
        if e.kind == "deconv":       # ConvT weight [Cin_t, Cout_t, kh, kw] -> OHWI of the mirrored conv [Cin_t, kh, kw, Cout_t]
            return t.permute(0, 2, 3, 1).contiguous()
        if e.kind == "final_w":
            k = torch.zeros((self.nclasses_pad, DEPTH_PITCH, 256), dtype=torch.float32, device=t.device)
            k[:C, :D] = t.reshape(C, D, 256)
            return k.reshape(e.kshape)
        if e.kind == "final_b":
            k = torch.zeros((self.nclasses_pad, DEPTH_PITCH), dtype=torch.float32, device=t.device)
            k[:C, :D] = t.reshape(C, D)
            return k.reshape(e.kshape)
        if e.kind == "linear":
            return t.reshape(e.kshape)
        if e.kind == "linear_pad":
            k = torch.zeros(e.kshape, dtype=torch.float32, device=t.device)
            k[:t.shape[0], 0, 0] = t
            return k
        if e.kind == "vec_pad":
            k = torch.zeros(e.kshape, dtype=torch.float32, device=t.device)
            k[:t.shape[0]] = t
            return k
        return t.reshape(e.kshape)

    def _to_reference(self, e, k):
        C, D = self.nclasses, self.depth
        if e.kind == "stem":
            return k[:, :, :7, :3].permute(0, 3, 1, 2).contiguous()
        if e.kind in ("conv", "deconv"):
            return k.permute(0, 3, 1, 2).contiguous()
        if e.kind == "final_w":
            return k.reshape(self.nclasses_pad, DEPTH_PITCH, 256)[:C, :D].reshape(e.ref_shape).contiguous()
        if e.kind == "final_b":
            return k.reshape(self.nclasses_pad, DEPTH_PITCH)[:C, :D].reshape(e.ref_shape).contiguous()
        if e.kind == "linear":
            return k.reshape(e.ref_shape).contiguous()
        if e.kind == "linear_pad":
            return k[:e.ref_shape[0], 0, 0].contiguous()
        if e.kind == "vec_pad":
            return k[:e.ref_shape[0]].contiguous()
        return k.reshape(e.ref_shape).contiguous()

    def load_reference_state_dict(self, sd, strict=True):
        """sd: reference-format tensors keyed like the reference (an optional '_model_list.0.' / 'module.' prefix is
        stripped, hybridbaseline.py:111-126)."""
        clean = {}
        for k, v in sd.items():
            for pre in ("module.", "_model_list.0."):
                if k.startswith(pre):
                    k = k[len(pre):]
            clean[k] = v
        missing = []
        for name, e in self.entries.items():
            if name not in clean:
                missing.append(name)
                continue
            t = clean[name]
            if tuple(t.shape) != e.ref_shape:
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {e.ref_shape}")
            self.view(name).copy_(self._to_kernel(e, t.to(self.device)))
        for name in self.buffers:
            if name in clean:
                self.stat(name).copy_(clean[name].to(self.device).float())
            else:
                missing.append(name)
        # every BatchNorm of the network counts the same training forwards: one host counter stands for all <bn>.num_batches_tracked
        nbt = [int(v) for k, v in clean.items() if k.endswith(".num_batches_tracked")]
        if nbt:
            self.num_batches_tracked = max(nbt)
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:5]}... ({len(missing)})")
        return missing

    def reference_state_dict(self, grads=False):
        """Reference-layout tensors under the reference's keys, in the order of its module tree (a BatchNorm contributes weight,
        bias, running_mean, running_var, num_batches_tracked): what the reference's load_arch(strict=True) expects."""
        out = OrderedDict()
        for name, e in self.entries.items():
            out[name] = self._to_reference(e, self.view(name, self.grad if grads else None))
            pre = name[:-len(".bias")] if name.endswith(".bias") else None
            if not grads and pre is not None and pre + ".running_mean" in self.buffers:
                out[pre + ".running_mean"] = self.stat(pre + ".running_mean").clone()
                out[pre + ".running_var"] = self.stat(pre + ".running_var").clone()
                if not (getattr(self, "frozen_bn", False) and pre.startswith("backbone.")):     # FrozenBatchNorm2d keeps no counter (resnet.py:45-55)
                    out[pre + ".num_batches_tracked"] = torch.tensor(int(self.num_batches_tracked), dtype=torch.int64)
        return out

    def init_reference_like(self, seed=1):
        """kaiming_normal_(fan_out, relu) for convs / deconvs / final conv, BN weight 1 bias 0 (resnet.py:170-176,
        simplebaseline.py:104-118), torch default Linear init for the box head and backbone.fc."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        import math
        for name, e in self.entries.items():
            shp = e.ref_shape
            if e.kind in ("stem", "conv", "final_w"):
                fan_out = shp[0] * shp[2] * shp[3]
                sd[name] = math.sqrt(2.0 / fan_out) * torch.randn(shp, generator=g)
            elif e.kind == "deconv":
                fan_out = shp[0] * shp[2] * shp[3]     # torch's fan_out of a ConvTranspose weight tensor
                sd[name] = math.sqrt(2.0 / fan_out) * torch.randn(shp, generator=g)
            elif e.kind in ("linear", "linear_pad", "frozen") and len(shp) == 2:
                bound = 1.0 / math.sqrt(shp[1])
                sd[name] = (torch.rand(shp, generator=g) * 2 - 1) * bound
            elif name.endswith(".weight") and e.kind == "vec":
                sd[name] = torch.ones(shp)
            elif "box_head" in name or "fc.bias" in name:
                fan_in = self.entries[name.replace(".bias", ".weight")].ref_shape[1]
                sd[name] = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan_in)
            else:
                sd[name] = torch.zeros(shp)
        for name, (_, c) in self.buffers.items():
            sd[name] = torch.ones(c) if name.endswith("running_var") else torch.zeros(c)
        self.load_reference_state_dict(sd)


class HybridNet:
    """Explicit forward / backward of HybridBaseline on the HIP kernels."""

    def __init__(self, store: ParamStore, image_size=(256, 256), compute_dtype=torch.bfloat16):
        self.p = store
        self.W, self.H = int(image_size[0]), int(image_size[1])
        self.x3 = compute_dtype in ("bf16x3", "x3")           # split-bf16 convolutions on fp32 tensors
        self.dtype = torch.float32 if self.x3 else compute_dtype
        self.training = True
        self.image_plane = "f32" # "u8n": forward(xpad=...) receives the loaders' integer plane 2 v - 255 (bf16; AB_DT_U8N) -- bf16x3 only
        self.norm = 0            # HYBRID_HEAD.NORM_TYPE code (head.NORM_CODE): 0 softmax, 1 sigmoid
        self.frozen_bn = False   # BACKBONE.FREEZE_BATCHNORM: backbone BatchNorms are fixed affine maps (resnet.py:33-69, 146-149)
        self.lp = None           # low-precision copy of the flat params (bf16 mode)
        self.tr = {}             # IHWO (data-gradient) copies of conv weights in the compute dtype
        self._packed = False
        self.saved = None

    # ------------------------------------------------------------------ weights in compute precision
    def _dgrad_names(self):
        names = []
        for name, e in self.p.entries.items():
            if e.kind in ("conv", "deconv", "final_w"):
                names.append(name)
        return names

    def refresh_after_update(self, lp_fresh=False):
        """Called by the fused optimizer: the bf16 copy was refreshed in the Adam pass; redo only the transposes."""
        self.pack_weights(skip_cast=lp_fresh)

    def pack_weights(self, skip_cast=False):
        """Refresh compute-precision copies after an optimizer step: one cast pass + IHWO transposes."""
        p = self.p
        if self.x3:
            if self.lp is None or tuple(self.lp.shape) != (2, p.total):
                self.lp = torch.empty((2, p.total), dtype=torch.bfloat16, device=p.device)
                skip_cast = False
            if not skip_cast:
                K.split(p.flat, out=self.lp)                  # (hi, lo) planes of every weight, kernel layout
        elif self.dtype == torch.bfloat16:
            if self.lp is None or self.lp.dtype != torch.bfloat16:
                self.lp = torch.empty(p.total, dtype=torch.bfloat16, device=p.device)
                skip_cast = False
            if not skip_cast:
                K.cast_bf16(p.flat, self.lp)
        else:
            self.lp = p.flat
        if getattr(self, "_tr_plan", None) is None:
            pairs = []
            if self.x3:      # IHWO copies as split planes [2][tot]: written by the transpose launch itself
                tot = sum(_round_up(p.entries[n].numel, 64) for n in self._dgrad_names())
                self._tr_planes = torch.zeros((2, tot), dtype=torch.bfloat16, device=p.device)
                self._tr_lo_off = tot
                off = 0
            for name in self._dgrad_names():
                O, kh, kw, I = p.entries[name].kshape
                if self.x3:
                    n = p.entries[name].numel
                    dst = self._tr_planes[0, off:off + n].view(I, kh, kw, O)         # hi plane; lo at + tot elements
                    self.tr[name] = self._tr_planes[:, off:off + n].view(2, I, kh, kw, O)
                    off += _round_up(n, 64)
                else:
                    dst = self.tr[name] = torch.empty((I, kh, kw, O), dtype=self.dtype, device=p.device)
                pairs.append((p.view(name).reshape(O, kh * kw, I), dst))
            self._tr_pairs = pairs                                   # the flat buffer and the copies are persistent
            self._tr_plan = K.transpose_plan(pairs) or False
            # fp32 [in][out] copies of the box-head weights (its data gradients run as NT products too)
            self.box_t, bpairs = {}, []
            for name in (("box_head.layers.0.weight", "box_head.layers.2.weight", "box_head.layers.4.weight") if p.box_head else ()):
                O, _, _, I = p.entries[name].kshape
                dst = self.box_t[name] = torch.empty((I, O), dtype=torch.float32, device=p.device)
                bpairs.append((p.view(name).reshape(O, 1, I), dst.view(I, 1, O)))
            self._box_pairs = bpairs
            self._box_plan = (K.transpose_plan(bpairs) or False) if bpairs else False
        if self._box_plan:
            K.transpose_oki_batch(self._box_plan)
        else:
            for src, dst in self._box_pairs:
                dst.copy_(src.permute(2, 1, 0))
        if self.x3:
            if not self._tr_plan:
                raise RuntimeError("bf16x3 needs the batched transpose plan (channel counts multiples of 8 / 4)")
            K.transpose_oki_batch_x3(self._tr_plan, self._tr_lo_off)
        elif self._tr_plan:
            K.transpose_oki_batch(self._tr_plan)                     # one launch for all IHWO dgrad copies
        else:
            for src, dst in self._tr_pairs:
                K.transpose_oki(src, dst)
        self._packed = True

    def w(self, name):
        e = self.p.entries[name]
        if self.x3:
            return self.lp[:, e.offset:e.offset + e.numel].view((2,) + e.kshape)
        return self.lp[e.offset:e.offset + e.numel].view(e.kshape)

    # ------------------------------------------------------------------ convolutions in the configured precision
    def _conv_fwd(self, x, name, stride, pad, **kw):
        if kw.get("want_stats") and not self.training:
            # eval mode reads the running statistics: no partial sums are needed, and a forward WITHOUT statistics stays on the kernels whose
            # eval-fold form is bit-identical to conv + ab_bn_apply_x3 (conv3x3r.hip takes the training launches of layer 1 only)
            kw = dict(kw, want_stats=False)
            fn = K.conv2d_fwd_x3 if self.x3 else K.conv2d_fwd
            return fn(x, self.w(name), stride, pad, **kw), None
        if self.x3:
            return K.conv2d_fwd_x3(x, self.w(name), stride, pad, **kw)
        return K.conv2d_fwd(x, self.w(name), stride, pad, **kw)

    def _conv_dgrad(self, dy, name, in_hw, stride, pad, addend=None, bn=None, want_stats=False):
        if self.x3:
            return K.conv2d_dgrad_x3(dy, self.tr[name], in_hw, stride, pad, addend=addend, want_stats=want_stats, bn=bn)
        return K.conv2d_dgrad(dy, self.tr[name], in_hw, stride, pad, addend=addend, bn=bn, want_stats=want_stats)

    # bf16x3, AB_WGRAD_1PASS=1 (a precision / speed STUDY, never the default -- DESIGN 12.1): weight gradients from the hi planes only
    # (one MFMA pass, bf16 operands, fp32 accumulate).  Their rounding does not propagate (nothing consumes a weight gradient but Adam).
    wgrad_1pass = os.environ.get("AB_WGRAD_1PASS", "0") == "1"

    def _conv_wgrad(self, x, dy, kh, kw, stride, pad, out=None, **kws):
        if self.x3 and self.wgrad_1pass:
            return K.conv2d_wgrad(K._planes(x)[0], K._planes(dy)[0], kh, kw, stride, pad, out=out, **kws)
        if self.x3:
            return K.conv2d_wgrad_x3(x, dy, kh, kw, stride, pad, out=out, **kws)
        return K.conv2d_wgrad(x, dy, kh, kw, stride, pad, out=out, **kws)

    # ------------------------------------------------------------------ BN helper
    def _frozen(self, prefix):
        return self.frozen_bn and prefix.startswith("backbone.")

    def _zero_part(self, C):
        """BatchNorm-backward partial sums of a FROZEN BatchNorm: zero, so that ab_bn_bwd* yields dy = scale * dz and zero parameter
        gradients (FrozenBatchNorm2d has no learnable state: its weight / bias are buffers)."""
        z = getattr(self, "_zparts", None)
        if z is None:
            z = self._zparts = {}
        if C not in z:
            z[C] = torch.zeros((1, C, 2), dtype=torch.float32, device=self.p.device)
        return z[C]

    def _bn_params(self, prefix, stats_part, count):
        p = self.p
        if self.training and self._frozen(prefix):
            return K.bn_eval_params(p.view(prefix + ".weight"), p.view(prefix + ".bias"),
                                    p.stat(prefix + ".running_mean"), p.stat(prefix + ".running_var"))
        if self.training:
            return K.bn_finalize(stats_part, count, p.view(prefix + ".weight"), p.view(prefix + ".bias"),
                                 p.stat(prefix + ".running_mean"), p.stat(prefix + ".running_var"))
        ev = getattr(self, "_eval_bnp", None)
        if ev is not None:
            return ev[prefix]                      # this forward's batched launch (forward() -> _eval_params_all)
        return K.bn_eval_params(p.view(prefix + ".weight"), p.view(prefix + ".bias"),
                                p.stat(prefix + ".running_mean"), p.stat(prefix + ".running_var"))

    def _eval_params_all(self):
        """Eval mode: (scale, shift, mean, invstd) of all 38 BatchNorms in ONE launch per forward (they are re-derived at every
        forward: weights and running statistics move between evaluations) instead of one tiny launch each."""
        p = self.p
        if getattr(self, "_eval_desc", None) is None:
            import numpy as np
            rows, off, views = [], 0, {}
            prefixes = [k[:-len(".running_mean")] for k in p.buffers if k.endswith(".running_mean")]
            for pre in prefixes:
                c = p.entries[pre + ".weight"].numel
                rows.append((p.entries[pre + ".weight"].offset, p.entries[pre + ".bias"].offset, p.buffers[pre + ".running_mean"][0],
                             p.buffers[pre + ".running_var"][0], off, c))
                views[pre] = (off, c)
                off += 4 * _round_up(c, 64)
            self._eval_desc = torch.from_numpy(np.asarray(rows, np.int32)).to(p.device)
            self._eval_out = torch.empty(off, dtype=torch.float32, device=p.device)
            self._eval_views = {pre: self._eval_out[o:o + 4 * c].view(4, c) for pre, (o, c) in views.items()}
            self._eval_maxc = max(r[5] for r in rows)
        K.bn_eval_params_batch(p.flat, p.stats, self._eval_desc, self._eval_desc.shape[0], self._eval_maxc, self._eval_out)
        return self._eval_views

    def _bn(self, prefix, y, stats_part, count, res=None, relu=True, feeds_conv=True, keep_f32=False, res_bnp=None):
        """feeds_conv / keep_f32 (bf16x3 only): the activation feeds a convolution (it is written as split planes by this
        pass) / is also needed in fp32 (residual input, ReLU mask of the backward, pooling)."""
        if (self.x3 and feeds_conv and self.training and not self._frozen(prefix) and K.bn_fin_apply_x3_ok(stats_part, y.shape[-1])):
            p = self.p      # few partial rows (layers 3 - 4): the apply pass finalizes the statistics itself, one launch less
            return K.bn_fin_apply_x3(y, stats_part, count, p.view(prefix + ".weight"), p.view(prefix + ".bias"), p.stat(prefix + ".running_mean"),
                                     p.stat(prefix + ".running_var"), res=res, relu=relu, want_f32=keep_f32, res_bnp=res_bnp)
        bnp = self._bn_params(prefix, stats_part, count)
        if self.x3 and feeds_conv:
            return K.bn_apply_x3(y, bnp, res=res, relu=relu, want_f32=keep_f32, res_bnp=res_bnp), bnp
        assert res_bnp is None
        out = K.bn_apply(y, bnp, res=res, relu=relu)
        return out, bnp

    def _bn_bwd(self, *a, **kw):
        """BatchNorm backward; the gradient wrt the conv output goes to convolutions only: split planes under bf16x3.
        bf16x3 with `part`: the data gradient that produced the input already masked and reduced it (conv3x3.hip, X3 = 2)."""
        if kw.pop("frozen", False):
            # frozen BatchNorm: no batch statistics took part in the forward, so dy = scale * (masked gradient) -- the regular passes with
            # all-zero partial sums (and dgamma = dbeta = 0).  A fused reduction that arrived with the gradient is simply not used.
            C = a[2].shape[-1]
            premasked = kw.get("part") is not None
            kw["part"] = self._zero_part(C)
            if self.x3:
                return K.bn_bwd_x3(*a, premasked=premasked, **kw)
            return K.bn_bwd(*a, **kw)
        if self.x3:
            return K.bn_bwd_x3(*a, premasked=kw.get("part") is not None, **kw)
        return K.bn_bwd(*a, **kw)

    # ------------------------------------------------------------------ forward
    def forward(self, image=None, xpad=None):
        """image: float32 NCHW [N,3,H,W] in [-0.5,0.5]   or   xpad: NHWC4 zero-bordered [N,H+6,W+8,4] compute dtype
        (what the renderer writes directly).  Returns kp3d [N,22,3] f32, conf [N,22] f32, box6d [N,6] f32."""
        if not self._packed:
            self.pack_weights()
        p, tr, dt = self.p, self.training, self.dtype
        if xpad is None:
            xpad = K.image_pad_nhwc4(image.contiguous().float(), dt)
        # image_plane == "u8n": a bf16 [N, H+6, W+8, 4] tensor is the loaders' integer plane 2 v - 255 (AB_DT_U8N): the stem's forward and weight
        # gradient run two MFMA passes on it and there is no split pass over the image (fp32 images keep the three-pass path)
        u8n = self.x3 and self.image_plane == "u8n" and xpad.dtype == torch.bfloat16 and xpad.dim() == 4
        if u8n and self.wgrad_1pass:
            raise NotImplementedError("AB_WGRAD_1PASS (a precision study) with the integer image plane")
        if not u8n and xpad.dtype != dt:
            # the stem kernels take ONE dtype code for image and weights: a mismatch would read the weights as the image's type
            raise TypeError(f"HybridNet({'bf16x3' if self.x3 else dt}) needs the padded image in {dt}, got {xpad.dtype} "
                            f"(build the loader with compute_dtype=net.dtype)")
        self._eval_bnp = None if tr else self._eval_params_all()
        if tr and not torch.cuda.is_current_stream_capturing():
            p.num_batches_tracked += 1      # nn.BatchNorm2d's counter (momentum is fixed, so only checkpoints read it); graph replays count themselves
        N = xpad.shape[0]
        H, W = xpad.shape[1] - 6, xpad.shape[2] - 8
        S = {"xpad": xpad, "N": N, "HW": (H, W), "blocks": []}
        if self.x3:
            if not u8n:
                xpad = K.split(xpad)      # planes of THIS step's image (forward and weight gradient of the stem read them)
            S["xpad"] = xpad
            y0, st = K.conv2d_stem_fwd_x3(xpad, self.w("backbone.conv1.weight"), H, W, want_stats=True)
        else:
            y0, st = K.conv2d_stem_fwd(xpad, self.w("backbone.conv1.weight"), H, W, want_stats=True)
        if self.fuse_stem:
            bnp0 = self._bn_params("backbone.bn1", st, N * (H // 2) * (W // 2))
            # BN + ReLU + 3x3/2 max-pool: the 128x128x64 activation is never stored (bf16x3: the pooled planes come from the same pass)
            if self.x3:      # training: + the raw conv output at the winners, all the backward's reduction needs of y0
                res = K.bn_relu_maxpool_fwd_x3(y0, bnp0, want_win=tr and self.pool_win, want_f32=not self.res_planes)
                x, pool_idx = res[0], res[1]
                S["pool_ywin"] = res[2] if len(res) > 2 else None
            else:
                x, pool_idx = K.bn_relu_maxpool_fwd(y0, bnp0)
        else:
            a0, bnp0 = self._bn("backbone.bn1", y0, st, N * (H // 2) * (W // 2), feeds_conv=False)
            x, pool_idx = K.maxpool_fwd(a0)
        if self.x3 and x.dtype != torch.bfloat16 and getattr(x, "_ab_split", None) is None:
            x._ab_split = K.split(x)      # layer1.0 reads the pooled tensor three times (conv1, residual, conv1's weight gradient)
        S.update(y0=y0, bnp0=bnp0, pool_idx=pool_idx)
        inpl = 64
        hp = p.hp
        for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], p.layers), start=1):
            for b in range(nblk):
                stride = 2 if (b == 0 and li > 1) else 1
                pre = f"backbone.layer{li}.{b}"
                if p.block == "bottleneck":
                    x, rec = self._bottleneck_fwd(x, pre, planes, stride, stride != 1 or inpl != planes * 4, last=(li == 4 and b == nblk - 1))
                    S["blocks"].append(rec if tr else dict(pre=pre))
                    inpl = planes * 4
                    continue
                if not tr and self.x3 and self.eval_fold:
                    x = self._eval_block(x, pre, stride, inpl != planes, last=(li == 4 and b == nblk - 1))
                    S["blocks"].append(dict(pre=pre))
                    inpl = planes
                    continue
                y1, st1 = self._conv_fwd(x, pre + ".conv1.weight", stride, 1, want_stats=True)
                cnt = y1.shape[0] * y1.shape[1] * y1.shape[2]
                a1, bnp1 = self._bn(pre + ".bn1", y1, st1, cnt)
                y2, st2 = self._conv_fwd(a1, pre + ".conv2.weight", 1, 1, want_stats=True)
                rec = dict(pre=pre, stride=stride, x=x, y1=y1, a1=a1, bnp1=bnp1, y2=y2, ds=False)
                if stride != 1 or inpl != planes:
                    yd, std_ = self._conv_fwd(x, pre + ".downsample.0.weight", stride, 0, want_stats=True)
                    if self.x3 and self.fuse_ds_bn:      # identity = bn_ds(yd) is applied inside bn2's pass, never stored
                        r, bnpd = yd, self._bn_params(pre + ".downsample.1", std_, cnt)
                    else:
                        r, bnpd = self._bn(pre + ".downsample.1", yd, std_, cnt, relu=False, feeds_conv=False)
                    rec.update(ds=True, yd=yd, bnpd=bnpd)
                else:
                    r, bnpd = x, None
                # bf16x3 + AB_RES_PLANES=1: block outputs exist only as their planes (the next block adds hi + lo as its residual);
                # the last block's output is also kept in fp32 (global average pool, transposed conv head)
                last = li == 4 and b == nblk - 1
                out, bnp2 = self._bn(pre + ".bn2", y2, st2, cnt, res=r, relu=True, keep_f32=(not self.res_planes) or last or not self.x3,
                                     res_bnp=bnpd if (self.x3 and self.fuse_ds_bn) else None)
                rec.update(bnp2=bnp2, out=out)
                if not tr:
                    rec = dict(pre=pre)
                S["blocks"].append(rec)
                x = out
                inpl = planes
        feat = x                                         # res_layer4 [N,h,w,512]
        fmean = K.avgpool_fwd(feat)                      # res_layer4_mean [N,512] f32 (resnet.py:219)
        h4, w4 = feat.shape[-3], feat.shape[-2]
        # ---- IntegralDeconvHead: ConvT == data-gradient of the mirrored stride-2 conv
        # transposed convs: the data-gradient kernel of the mirrored conv, BatchNorm partials from its epilogue
        # (AB_DECONV_STATS=0: separate col_stats passes)
        fused = os.environ.get("AB_DECONV_STATS", "1") != "0"

        def deconv(x, name, hw):
            if tr and fused:
                return self._conv_dgrad(x, name, hw, 2, 1, want_stats=True)
            d = self._conv_dgrad(x, name, hw, 2, 1)
            return d, (K.col_stats(d) if tr else None)

        if not tr and self.x3 and self.eval_fold:      # eval: BatchNorm + ReLU of the two transposed convolutions in their epilogues
            d1 = d2 = bnpd1 = bnpd2 = None
            e1 = K.conv2d_dgrad_x3_affine(feat, self.tr[hp + ".deconv_layers.0.weight"], (2 * h4, 2 * w4), 2, 1,
                                          self._bn_params(hp + ".deconv_layers.1", None, 0))
            e2 = K.conv2d_dgrad_x3_affine(e1, self.tr[hp + ".deconv_layers.3.weight"], (4 * h4, 4 * w4), 2, 1,
                                          self._bn_params(hp + ".deconv_layers.4", None, 0))
        else:
            d1, st1 = deconv(feat, hp + ".deconv_layers.0.weight", (2 * h4, 2 * w4))
            e1, bnpd1 = self._bn(hp + ".deconv_layers.1", d1, st1, N * 4 * h4 * w4)
            d2, st2 = deconv(e1, hp + ".deconv_layers.3.weight", (4 * h4, 4 * w4))
            e2, bnpd2 = self._bn(hp + ".deconv_layers.4", d2, st2, N * 16 * h4 * w4)
        # bf16x3, softmax head: the final layer as the register-resident GEMM with the soft-argmax's first stage in its epilogue
        # (ab_conv1x1_sam_fwd_x3); head_fwd() then only merges the per-tile rows
        wf = self.w(hp + ".final_layer.weight") if self.x3 else None
        sam_part = None          # (travels in self.last with the logits it describes: segment graphs swap `last`, not attributes of the net)
        if (self.x3 and self.fuse_sam and self.norm == 0
                and K.conv1x1_sam_fwd_x3_ok(e2, wf, p.nclasses_pad, p.depth, DEPTH_PITCH)):
            logits, sam_part = K.conv1x1_sam_fwd_x3(e2, wf, p.view(hp + ".final_layer.bias"), p.nclasses_pad, p.depth)
        else:
            logits = self._conv_fwd(e2, hp + ".final_layer.weight", 1, 0, bias=p.view(hp + ".final_layer.bias"))
        # ---- MLP_O box head, always f32 (tiny; keeps the 6-D rotation at full precision)
        m0 = fmean.view(N, p.feat_ch)
        if p.box_head:
            lw = lambda n: p.view(n).view(p.entries[n].kshape[0], -1)      # noqa: E731  ([out][in] rows of the 1x1 layout)
            b1 = K.linear_fwd(m0, lw("box_head.layers.0.weight"), p.view("box_head.layers.0.bias"), relu=True)
            b2 = K.linear_fwd(b1, lw("box_head.layers.2.weight"), p.view("box_head.layers.2.bias"), relu=True)
            b3 = K.linear_fwd(b2, lw("box_head.layers.4.weight"), p.view("box_head.layers.4.bias"))
        else:                     # SimpleBaseline: no MLP_O (simplebaseline.py:205-208); a zero placeholder keeps the call contract
            b1 = b2 = None
            b3 = torch.zeros((N, BOX_OUT_PAD), dtype=torch.float32, device=p.device)
        box6d = b3.view(N, BOX_OUT_PAD)[:, :6]
        S.update(feat=feat, d1=d1, e1=e1, bnpd1=bnpd1, d2=d2, e2=e2, bnpd2=bnpd2, logits=logits, m0=m0, b1=b1, b2=b2)
        self.saved = S if tr else None
        self.last = dict(feat=feat, fmean=fmean, logits=logits, box_raw=b3.view(N, BOX_OUT_PAD), sam_part=sam_part)
        return logits, box6d

    eval_fold = os.environ.get("AB_EVAL_FOLD", "1") != "0"       # bf16x3 eval: BatchNorm folded into the 3x3 conv epilogues

    def _bottleneck_fwd(self, x, pre, planes, stride, has_ds, last):
        """Bottleneck.forward (resnet.py:104-141): 1x1 -> bn -> relu -> 3x3 (stride) -> bn -> relu -> 1x1 (x4) -> bn, + identity /
        downsample(x), relu.  The same kernels as the BasicBlock path (the 1x1 convolutions run on the generic implicit GEMM)."""
        y1, st1 = self._conv_fwd(x, pre + ".conv1.weight", 1, 0, want_stats=True)
        a1, bnp1 = self._bn(pre + ".bn1", y1, st1, y1.shape[0] * y1.shape[1] * y1.shape[2])
        y2, st2 = self._conv_fwd(a1, pre + ".conv2.weight", stride, 1, want_stats=True)
        cnt = y2.shape[0] * y2.shape[1] * y2.shape[2]
        a2, bnp2 = self._bn(pre + ".bn2", y2, st2, cnt)
        y3, st3 = self._conv_fwd(a2, pre + ".conv3.weight", 1, 0, want_stats=True)
        rec = dict(kind="bottleneck", pre=pre, stride=stride, x=x, y1=y1, a1=a1, bnp1=bnp1, y2=y2, a2=a2, bnp2=bnp2, y3=y3, ds=False)
        if has_ds:
            yd, std_ = self._conv_fwd(x, pre + ".downsample.0.weight", stride, 0, want_stats=True)
            if self.x3 and self.fuse_ds_bn:
                r, bnpd = yd, self._bn_params(pre + ".downsample.1", std_, cnt)
            else:
                r, bnpd = self._bn(pre + ".downsample.1", yd, std_, cnt, relu=False, feeds_conv=False)
            rec.update(ds=True, yd=yd, bnpd=bnpd)
        else:
            r, bnpd = x, None
        out, bnp3 = self._bn(pre + ".bn3", y3, st3, cnt, res=r, relu=True, keep_f32=(not self.res_planes) or last or not self.x3,
                             res_bnp=bnpd if (self.x3 and self.fuse_ds_bn and has_ds) else None)
        rec.update(bnp3=bnp3, out=out, y2_last=y3, bnp_last=bnp3)
        return out, rec

    def _bottleneck_bwd(self, dout, rec, dout_part, nxt):
        """Backward of one Bottleneck block; returns (gradient wrt the block input, its fused BatchNorm partials or None)."""
        gv = self.p.gview
        pre, stride, x = rec["pre"], rec["stride"], rec["x"]
        fz = self._frozen(pre)
        dy3, dz = self._bn_bwd(dout, rec["out"], rec["y3"], rec["bnp3"], gv(pre + ".bn3.weight"), gv(pre + ".bn3.bias"),
                           relu=True, want_dz=True, part=dout_part, frozen=fz)
        a2, a1 = rec["a2"], rec["a1"]
        self._wgrad_side(self._conv_wgrad, a2, dy3, 1, 1, 1, 0, out=gv(pre + ".conv3.weight"))
        da2, part2 = self._conv_dgrad(dy3, pre + ".conv3.weight", (dy3.shape[-3], dy3.shape[-2]), 1, 0, bn=(rec["y2"], None, rec["bnp2"]))
        dy2 = self._bn_bwd(da2, a2, rec["y2"], rec["bnp2"], gv(pre + ".bn2.weight"), gv(pre + ".bn2.bias"), relu="recompute", part=part2, frozen=fz)
        self._wgrad_side(self._conv_wgrad, a1, dy2, 3, 3, stride, 1, out=gv(pre + ".conv2.weight"))
        h1, w1 = (a1.shape[-3], a1.shape[-2])
        da1, part1 = self._conv_dgrad(dy2, pre + ".conv2.weight", (h1, w1), stride, 1, bn=(rec["y1"], None, rec["bnp1"]))
        dy1 = self._bn_bwd(da1, a1, rec["y1"], rec["bnp1"], gv(pre + ".bn1.weight"), gv(pre + ".bn1.bias"), relu="recompute", part=part1, frozen=fz)
        self._wgrad_side(self._conv_wgrad, x, dy1, 1, 1, 1, 0, out=gv(pre + ".conv1.weight"))
        hw = (x.shape[-3], x.shape[-2])
        if rec["ds"]:
            dyd = self._bn_bwd(dz, None, rec["yd"], rec["bnpd"], gv(pre + ".downsample.1.weight"), gv(pre + ".downsample.1.bias"), relu=False, frozen=fz)
            self._wgrad_side(self._conv_wgrad, x, dyd, 1, 1, stride, 0, out=gv(pre + ".downsample.0.weight"))
            dx = self._conv_dgrad(dy1, pre + ".conv1.weight", hw, 1, 0)
            return self._conv_dgrad(dyd, pre + ".downsample.0.weight", hw, stride, 0, addend=dx), None
        return self._conv_dgrad(dy1, pre + ".conv1.weight", hw, 1, 0, addend=dz), None

    def _eval_block(self, x, pre, stride, has_ds, last):
        """One BasicBlock in eval mode (resnet.py:85-101 with running statistics): where the 3x3 kernel takes the shape the
        BatchNorm after a convolution (+ residual + ReLU) rides in its epilogue and the fp32 conv output is never stored;
        bit-identical to conv + ab_bn_apply_x3 (AB_EVAL_FOLD=0)."""
        def conv_bn(inp, cname, bname, s, res, relu, want_f32=False):
            w = self.w(cname)
            bnp = self._bn_params(bname, None, 0)
            if s == 1 and K.conv2d_fwd_x3_evalbn_ok(inp, w):
                return K.conv2d_fwd_x3_evalbn(inp, w, bnp, res=res, relu=relu, want_f32=want_f32)
            if res is None and not want_f32:          # strided 3x3: affine + ReLU in the generic kernel's epilogue
                return K.conv2d_fwd_x3_affine(inp, w, bnp, s, 1, relu=relu, planes=True)
            y = K.conv2d_fwd_x3(inp, w, s, 1)
            return K.bn_apply_x3(y, bnp, res=res, relu=relu, want_f32=want_f32)

        a1 = conv_bn(x, pre + ".conv1.weight", pre + ".bn1", stride, None, True)
        if stride != 1 or has_ds:
            r = K.conv2d_fwd_x3_affine(x, self.w(pre + ".downsample.0.weight"), self._bn_params(pre + ".downsample.1", None, 0), stride, 0,
                                       relu=False, planes=False)
        else:
            r = x
        return conv_bn(a1, pre + ".conv2.weight", pre + ".bn2", 1, r, True, want_f32=last or not self.res_planes)

    def head_fwd(self, logits):
        """-> kp3d [N,22,3], conf [N,22], stat (kept for head_bwd)."""
        last = getattr(self, "last", None)
        part = last.get("sam_part") if last is not None and logits is last.get("logits") else None
        if part is not None and part.shape[0] == logits.shape[0]:      # statistics from the GEMM epilogue of THIS forward
            kp3d, conf, stat = softargmax3d_stage2(part, self.p.nclasses_pad)
        else:
            kp3d, conf, stat = softargmax3d_fwd(logits, self.p.nclasses_pad, self.p.depth, DEPTH_PITCH, self.norm)
        if self.p.nclasses_pad != self.p.nclasses:        # the padding class: dropped from what the model sees, kept for the backward
            self._head_full = (kp3d, conf)
            return kp3d[:, :self.p.nclasses].contiguous(), conf[:, :self.p.nclasses].contiguous(), stat
        return kp3d, conf, stat

    def head_bwd(self, logits, kp3d, conf, stat, g_kp3d, g_conf=None):
        """dlogits, written in place over the logits buffer (they are not needed again); bf16x3: as split planes."""
        hp, CP = self.p.hp, self.p.nclasses_pad
        if CP != self.p.nclasses:             # zero gradient for the padding class; forward outputs of all CP classes from head_fwd
            kp3d, conf = self._head_full
            gk = torch.zeros((g_kp3d.shape[0], CP, 3), dtype=torch.float32, device=g_kp3d.device)
            gk[:, :self.p.nclasses].copy_(g_kp3d)
            g_kp3d = gk
            if g_conf is not None:
                gc = torch.zeros((g_conf.shape[0], CP), dtype=torch.float32, device=g_conf.device)
                gc[:, :self.p.nclasses].copy_(g_conf)
                g_conf = gc
        if self.x3:
            # (the final layer's bias gradient = column sums of dlogits comes out of the same pass; backward() sees the tag)
            dbias = self.p.gview(hp + ".final_layer.bias") if (self.sam_bias and self.saved is not None) else None
            return softargmax3d_bwd_x3(logits, CP, self.p.depth, DEPTH_PITCH, kp3d, conf, stat, g_kp3d, g_conf, dbias=dbias,
                                       norm=self.norm)
        return softargmax3d_bwd(logits, CP, self.p.depth, DEPTH_PITCH, kp3d, conf, stat, g_kp3d, g_conf,
                                inplace=True, norm=self.norm)

    # ------------------------------------------------------------------ backward
    # Weight gradients are off the critical path (nothing consumes them before the optimizer), so they CAN be issued on
    # a side stream to co-run with the HBM-bound BatchNorm-backward passes of the layers below.  Measured on MI355X
    # (B=64, 256x256, graph replay): 7366 samples/s with the side stream vs 7715 without -- co-scheduled workgroups evict
    # each other's L2 / LDS residency and the single-queue order is faster.  Round 3, bf16x3 (tools/ab_env.sh AB_WGRAD_OVERLAP 0 1 2):
    # 9.68 ms/step in one queue, 9.84 / 10.01 with the side stream.  Kept as an opt-in (AB_WGRAD_OVERLAP=1).
    overlap_wgrad = os.environ.get("AB_WGRAD_OVERLAP", "0") == "1"
    fuse_stem = os.environ.get("AB_STEM_FUSE", "1") != "0"       # stem BN+ReLU+max-pool as one pass (forward)
    fuse_stem_bwd = os.environ.get("AB_STEM_FUSE_BWD", "0") == "1"   # ... and the gather-based fused backward
    stem_pool_reduce = os.environ.get("AB_STEM_POOL_REDUCE", "1") != "0"   # bf16x3: see _backward_trunk
    res_planes = os.environ.get("AB_RES_PLANES", "1") != "0"     # bf16x3: block outputs only as (hi, lo) planes, no fp32 copy
    pool_win = os.environ.get("AB_POOL_WIN", "1") != "0"          # bf16x3: stem BatchNorm-backward reduction over the pooled elements
    fuse_sam = os.environ.get("AB_FUSE_SAM", "1") != "0"          # bf16x3: soft-argmax stage 1 in the final layer's GEMM epilogue
    sam_bias = os.environ.get("AB_SAM_BIAS", "1") != "0"          # bf16x3: final-layer bias gradient out of the soft-argmax backward
    fuse_ds_bn = os.environ.get("AB_FUSE_DS_BN", "1") != "0"      # bf16x3: the downsample BatchNorm inside bn2's apply pass
    pair_dgrad = os.environ.get("AB_PAIR_DGRAD", "1") != "0"      # bf16x3: conv1 + downsample data gradients of a block in one launch
    pair_dgrad_bn = os.environ.get("AB_PAIR_DGRAD_BN", "1") != "0"      # ... with the BatchNorm-backward reduction of the stage below in its epilogue

    # AB_WGRAD_BATCH=1: the fixed-order slab reductions of a backward stage's weight gradients run as ONE launch at the end
    # of the stage instead of one per layer right behind its slab kernel.  Bit-identical, 38 graph nodes fewer -- and 2 %
    # slower (6.65 vs 6.51 ms/step): a layer's slabs (<= 50 MB) are still in the 256 MB Infinity Cache when reduced at
    # once, a stage's 0.3-0.7 GB are not.  Kept as an opt-in.
    batch_wgrad_reduce = os.environ.get("AB_WGRAD_BATCH", "0") == "1"

    # bf16x3, AB_WGRAD_GROUP=g > 1: the slab reductions of every g consecutive weight gradients run as ONE launch (bit-identical;
    # each deferred gradient keeps its own slab workspace until then).  Measured at B = 64: 10.42 ms/step ungrouped, 10.43 / 10.45 /
    # 10.46 / 10.50 for g = 2 / 3 / 6 / 10 -- the launches saved do not pay for the larger live slab footprint.  Default: off.
    wgrad_group = int(os.environ.get("AB_WGRAD_GROUP", "1"))

    # bf16x3, AB_WGRAD_FUSE=g (default 8, the kernel's maximum -- and never more problems than one round of workgroups holds: 4 on layer 4;
    # 1: off): up to g consecutive SAME-SHAPE 3x3 / stride-1 weight gradients of the backward (a stage's
    # blocks: layer 1 has 6 of one shape, layers 2 - 4 have 7 / 11 / 5) run as ONE slab launch + ONE reduction (ab_conv2d_wgrad_x3_group).
    # At one workgroup per CU every launch writes 256 partial tiles of 147 KB (37.7 MB) and the reduction reads them back: per LAYER before,
    # per GROUP now, and a workgroup's band pipeline ramps up once per group.  The deferred layers' operand planes are kept until the group
    # is launched (a shape change, a full group, or the end of a backward stage: _wgrad_join); gradients are complete only after that.
    # tools/bench_conv_x3.py "probe grp": 72 -> 56 -> 53 us per layer on layer 2 for groups of 1 / 2 / 4, 66 -> 53 -> 48 on layer 3.
    # The step, same box, alternating processes: 8.68 ms ungrouped -> 8.47 (g = 2); 8.91 (2) -> 8.78 (4) on another; 8.69 (4) -> 8.63 (8).
    wgrad_fuse = int(os.environ.get("AB_WGRAD_FUSE", "8"))

    def _wgrad_group_flush(self):
        items, self._wgrp = getattr(self, "_wgrp", None), None
        if not items:
            return
        if len(items) == 1:
            x, dy, out = items[0]
            K.conv2d_wgrad_x3(x, dy, 3, 3, 1, 1, out=out)
        else:
            K.conv2d_wgrad_x3_group(items)

    def _wgrad_side(self, fn, *args, **kw):
        if (self.x3 and self.wgrad_fuse > 1 and not self.overlap_wgrad and not self.wgrad_1pass and fn == self._conv_wgrad
                and len(args) == 6 and tuple(args[2:6]) == (3, 3, 1, 1) and set(kw) == {"out"} and kw["out"] is not None):
            xp, dp = K._planes(args[0]), K._planes(args[1])          # (split now: the planes, not the fp32 tensors, are what is kept)
            key = (tuple(xp[0].shape), tuple(dp[0].shape))
            ok = self.__dict__.setdefault("_wgrp_ok", {})
            if key not in ok:          # the largest group of this shape the kernel takes in one round of workgroups (0: none)
                ok[key] = next((G for G in range(min(self.wgrad_fuse, K.WGRAD_GROUP_MAX), 1, -1) if K.conv2d_wgrad_x3_group_ok(xp, dp, G)), 0)
            if ok[key]:
                pend = getattr(self, "_wgrp", None)
                if pend and (tuple(pend[0][0][0].shape), tuple(pend[0][1][0].shape)) != key:
                    self._wgrad_group_flush()
                    pend = None
                if not pend:
                    pend = self._wgrp = []
                pend.append((xp, dp, kw["out"]))
                if len(pend) >= ok[key]:
                    self._wgrad_group_flush()
                return kw["out"]
        if not self.overlap_wgrad:
            grouped = self.x3 and self.wgrad_group > 1
            if (self.batch_wgrad_reduce and not self.x3) or grouped:
                if getattr(self, "_pending", None) is None:
                    self._pending = K.PendingReductions()
                kw["defer"] = self._pending
                r = fn(*args, **kw)
                if grouped and len(self._pending.descs) >= self.wgrad_group:
                    self._pending.flush()
                return r
            return fn(*args, **kw)
        if getattr(self, "_wg_stream", None) is None:
            self._wg_stream = torch.cuda.Stream(device=self.p.device)
            self._wg_keep = []
        main = torch.cuda.current_stream(self.p.device)
        self._wg_stream.wait_stream(main)
        with torch.cuda.stream(self._wg_stream):
            fn(*args, **kw)
        self._wg_keep.append(args)

    def _wgrad_join(self):
        self._wgrad_group_flush()
        if getattr(self, "_pending", None) is not None:
            self._pending.flush()
        if getattr(self, "_wg_stream", None) is not None:
            torch.cuda.current_stream(self.p.device).wait_stream(self._wg_stream)
            self._wg_keep.clear()

    BWD_STAGES = 3

    def grad_stage_ranges(self):
        """[(lo, hi)] element ranges of the flat gradient completed by backward(stage=0), (stage=1), (stage=2): heads +
        layer4 (68 % of the bytes, produced first), layer3 (27 %), layer2 .. stem (5 %) -- the DDP overlap schedule of
        train.TrainStep all-reduces each range while the next stage computes."""
        o4 = self.p.entries["backbone.layer4.0.conv1.weight"].offset
        o3 = self.p.entries["backbone.layer3.0.conv1.weight"].offset
        return [(o4, self.p.total), (o3, o4), (0, o3)]

    def backward(self, dlogits=None, g_box6d=None, stage=None):
        """dlogits: gradient wrt the logits [N,h,w,22*32] (compute dtype); g_box6d [N,6] f32.
        Fills self.p.grad (overwrites).  Returns nothing (no gradient to the image).
        stage=None runs the whole backward; stage=0 runs box head, heat-map head and layer4 and parks the activation
        gradient, stage=1 continues with layer3, stage=2 with layer2 .. stem (see grad_stage_ranges)."""
        S, p, dt = self.saved, self.p, self.dtype
        if S is None:
            raise RuntimeError("backward() without a training-mode forward()")
        N = S["N"]
        gv = p.gview
        if stage == 1:
            dout, blocks, part = S.pop("_dout"), S.pop("_blocks_left"), S.pop("_dout_part")
            n3 = sum(1 for r in blocks if r["pre"].startswith("backbone.layer3."))
            dout, part = self._backward_blocks(dout, blocks[:n3], part, below=blocks[n3] if n3 < len(blocks) else None)
            S["_dout"], S["_blocks_left"], S["_dout_part"] = dout, blocks[n3:], part
            self._wgrad_join()
            return
        if stage == 2:
            dout, blocks, part = S.pop("_dout"), S.pop("_blocks_left"), S.pop("_dout_part")
            return self._backward_trunk(S, dout, blocks, part)
        hp = p.hp
        # ---- box head (f32)
        g_mean = None
        if p.box_head:
            g3 = getattr(self, "_g3", None)      # padded copy of g_box6d: columns 6.. are zeroed once, only the six live ones are rewritten
            if g3 is None or g3.shape[0] != N or g3.device != p.device:
                g3 = self._g3 = torch.zeros((N, BOX_OUT_PAD), dtype=torch.float32, device=p.device)
            g3[:, :6].copy_(g_box6d)
            lw = lambda n: p.view(n).view(p.entries[n].kshape[0], -1)      # noqa: E731
            lg = lambda n: gv(n).view(p.entries[n].kshape[0], -1)          # noqa: E731
            K.linear_wgrad(g3, S["b2"], lg("box_head.layers.4.weight"), gv("box_head.layers.4.bias"))
            gb2 = K.linear_dgrad(g3, self.box_t["box_head.layers.4.weight"], act_out=S["b2"])
            K.linear_wgrad(gb2, S["b1"], lg("box_head.layers.2.weight"), gv("box_head.layers.2.bias"))
            gb1 = K.linear_dgrad(gb2, self.box_t["box_head.layers.2.weight"], act_out=S["b1"])
            K.linear_wgrad(gb1, S["m0"], lg("box_head.layers.0.weight"), gv("box_head.layers.0.bias"))
            g_mean = K.linear_dgrad(gb1, self.box_t["box_head.layers.0.weight"])
        # ---- head
        e2, e1, feat = S["e2"], S["e1"], S["feat"]
        if self.x3 and dlogits.dtype == torch.bfloat16:       # planes straight from the soft-argmax backward
            if not getattr(dlogits, "_ab_bias_done", False):
                K.col_sum_x3(dlogits, gv(hp + ".final_layer.bias"))
        else:
            K.col_sum(dlogits, gv(hp + ".final_layer.bias"))
            if self.x3:
                dlogits = K.split(dlogits)        # one split serves the weight and the data gradient
        self._wgrad_side(self._conv_wgrad, e2, dlogits, 1, 1, 1, 0, out=gv(hp + ".final_layer.weight"))
        # (bf16x3: the ReLU mask + BatchNorm-backward reduction of the deconvolution below ride in the data gradient's epilogue)
        de2, part2 = self._conv_dgrad(dlogits, hp + ".final_layer.weight", (e2.shape[-3], e2.shape[-2]), 1, 0,
                                      bn=(S["d2"], None, S["bnpd2"])) if self.x3 else (
            self._conv_dgrad(dlogits, hp + ".final_layer.weight", (e2.shape[-3], e2.shape[-2]), 1, 0), None)
        dd2 = self._bn_bwd(de2, e2, S["d2"], S["bnpd2"], gv(hp + ".deconv_layers.4.weight"),
                       gv(hp + ".deconv_layers.4.bias"), relu="recompute", part=part2)
        self._wgrad_side(self._conv_wgrad, dd2, e1, 4, 4, 2, 1, out=gv(hp + ".deconv_layers.3.weight"))
        de1 = self._conv_fwd(dd2, hp + ".deconv_layers.3.weight", 2, 1)
        dd1 = self._bn_bwd(de1, e1, S["d1"], S["bnpd1"], gv(hp + ".deconv_layers.1.weight"),
                       gv(hp + ".deconv_layers.1.bias"), relu="recompute")
        self._wgrad_side(self._conv_wgrad, dd1, feat, 4, 4, 2, 1, out=gv(hp + ".deconv_layers.0.weight"))
        dout = self._conv_fwd(dd1, hp + ".deconv_layers.0.weight", 2, 1)
        if g_mean is not None:
            K.avgpool_bwd(g_mean, dout, accumulate=True)
        # ---- backbone, last block first
        blocks = list(reversed(S["blocks"]))
        if stage == 0:
            n4 = sum(1 for r in blocks if r["pre"].startswith("backbone.layer4."))
            dout, part = self._backward_blocks(dout, blocks[:n4], below=blocks[n4] if n4 < len(blocks) else None)
            S["_dout"], S["_blocks_left"], S["_dout_part"] = dout, blocks[n4:], part
            self._wgrad_join()
            return
        self._backward_trunk(S, dout, blocks)

    def _backward_blocks(self, dout, blocks, dout_part=None, below=None):
        """Backward through `blocks` (last block of the network first).  `dout_part`: BN-backward partial sums for the
        first block's bn2, if the producer of `dout` already reduced them.  `below`: the block that consumes the gradient
        leaving the last entry (its bn2 reduction is fused into that data gradient where the kernel allows).
        Returns (dout, dout_part) for `below`."""
        gv = self.p.gview
        for k, rec in enumerate(blocks):
            pre, stride, x = rec["pre"], rec["stride"], rec["x"]
            nxt = blocks[k + 1] if k + 1 < len(blocks) else below
            if rec.get("kind") == "bottleneck":
                dout, dout_part = self._bottleneck_bwd(dout, rec, dout_part, nxt)
                continue
            fz = self._frozen(pre)
            dy2, dz = self._bn_bwd(dout, rec["out"], rec["y2"], rec["bnp2"], gv(pre + ".bn2.weight"), gv(pre + ".bn2.bias"),
                               relu=True, want_dz=True, part=dout_part, frozen=fz)
            self._wgrad_side(self._conv_wgrad, rec["a1"], dy2, 3, 3, 1, 1, out=gv(pre + ".conv2.weight"))
            # the BN-backward reduction of bn1 rides in the epilogue of the data gradient that produces its input
            da1, part1 = self._conv_dgrad(dy2, pre + ".conv2.weight", (dy2.shape[-3], dy2.shape[-2]), 1, 1,
                                          bn=(rec["y1"], None, rec["bnp1"]))
            dy1 = self._bn_bwd(da1, rec["a1"], rec["y1"], rec["bnp1"], gv(pre + ".bn1.weight"), gv(pre + ".bn1.bias"),
                            relu="recompute", part=part1, frozen=fz)
            self._wgrad_side(self._conv_wgrad, x, dy1, 3, 3, stride, 1, out=gv(pre + ".conv1.weight"))
            bn_below = (nxt["y2"], nxt["out"], nxt["bnp2"]) if nxt is not None else None
            if rec["ds"]:
                dyd = self._bn_bwd(dz, None, rec["yd"], rec["bnpd"], gv(pre + ".downsample.1.weight"),
                               gv(pre + ".downsample.1.bias"), relu=False, frozen=fz)
                self._wgrad_side(self._conv_wgrad, x, dyd, 1, 1, stride, 0, out=gv(pre + ".downsample.0.weight"))
                if self.x3 and stride == 2 and self.pair_dgrad:      # both branches in one launch (the 1x1 as a tap of the 3x3/s2)
                    # ... and the BatchNorm-backward mask + reduction of the stage below in its epilogue (convp.hip)
                    dout = K.conv2d_dgrad_x3_pair(dy1, self.tr[pre + ".conv1.weight"], dyd, self.tr[pre + ".downsample.0.weight"],
                                                  (x.shape[-3], x.shape[-2]), 1, bn=bn_below if self.pair_dgrad_bn else None)
                    dout, dout_part = dout if isinstance(dout, tuple) else (dout, None)
                else:
                    dx = self._conv_dgrad(dy1, pre + ".conv1.weight", (x.shape[-3], x.shape[-2]), stride, 1)
                    dout = self._conv_dgrad(dyd, pre + ".downsample.0.weight", (x.shape[-3], x.shape[-2]), stride, 0, addend=dx)
                    dout_part = None
            elif bn_below is not None:
                dout, dout_part = self._conv_dgrad(dy1, pre + ".conv1.weight", (x.shape[-3], x.shape[-2]), stride, 1,
                                                   addend=dz, bn=bn_below)
            else:
                dout = self._conv_dgrad(dy1, pre + ".conv1.weight", (x.shape[-3], x.shape[-2]), stride, 1, addend=dz)
                dout_part = None
        return dout, dout_part

    def _backward_trunk(self, S, dout, blocks, dout_part=None):
        gv = self.p.gview
        dout, _ = self._backward_blocks(dout, blocks, dout_part)
        # ---- stem
        dy0 = None
        if self.frozen_bn:      # frozen stem BatchNorm: max-pool backward, then the apply pass with zero partial sums (no fused reduction)
            y0 = S["y0"]
            da0 = K.maxpool_bwd(S["pool_idx"], dout, (y0.shape[1], y0.shape[2]))
            dy0 = self._bn_bwd(da0, None, y0, S["bnp0"], gv("backbone.bn1.weight"), gv("backbone.bn1.bias"), relu="recompute", frozen=True)
        elif self.x3 and self.fuse_stem and self.stem_pool_reduce:
            # the max-pool backward pass also masks and reduces for the stem BatchNorm (AB_STEM_POOL_REDUCE=0: separate passes)
            dy0 = K.bn_relu_maxpool_bwd_x3(dout, S["pool_idx"], S["y0"], S["bnp0"], gv("backbone.bn1.weight"), gv("backbone.bn1.bias"),
                                           ywin=S.get("pool_ywin"))
        if dy0 is not None:
            pass
        elif self.fuse_stem_bwd:
            dy0 = K.bn_relu_maxpool_bwd(dout, S["pool_idx"], S["y0"], S["bnp0"], gv("backbone.bn1.weight"), gv("backbone.bn1.bias"))
        else:
            y0 = S["y0"]
            da0 = K.maxpool_bwd(S["pool_idx"], dout, (y0.shape[1], y0.shape[2]))
            dy0 = self._bn_bwd(da0, None, y0, S["bnp0"], gv("backbone.bn1.weight"), gv("backbone.bn1.bias"), relu="recompute")
        H, W = S["HW"]
        if self.x3 and self.wgrad_1pass:
            self._wgrad_side(K.conv2d_stem_wgrad, K._planes(S["xpad"])[0], K._planes(dy0)[0], H, W, out=gv("backbone.conv1.weight"))
        else:
            self._wgrad_side(K.conv2d_stem_wgrad_x3 if self.x3 else K.conv2d_stem_wgrad, S["xpad"], dy0, H, W, out=gv("backbone.conv1.weight"))
        self._wgrad_join()
        self.saved = None

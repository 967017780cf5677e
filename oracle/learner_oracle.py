"""TEST INFRASTRUCTURE ONLY (CPU oracle, torch fp32) -- restatement of the *learner* half of the ArtiBoost
hot path.  Imported only by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; the product path (artiboost_amd/) never imports it.

Pinned against the real reference: tests/golden/learner_*.npz were produced by oracle/gen_golden.py from the
reference's own modules (imported from /root/reference) and tests/test_oracle_golden.py checks this file
against them on CPU.

Each function cites the reference lines it restates (paths relative to /root/reference).
Parameters are a flat {name: tensor} dict using the reference's state_dict key names without the
"_model_list.0." Arch prefix (e.g. "backbone.layer1.0.conv1.weight").
"""
import math
import random
from itertools import combinations, product

import numpy as np
import torch
import torch.nn.functional as F

RESNET34_LAYERS = [3, 4, 6, 3]  # anakin/models/resnet.py:243-248
JOINTS_IDX_PARENTS = [0, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19]  # utils/misc.py:69


# ----------------------------------------------------------------------------- parameters
def param_shapes(nclasses=22, depth=28, layers=None, head_prefix="hybrid_head", box_head=True, bottleneck=False):
    """Ordered (name, shape) list == reference state_dict of Arch(HybridBaseline) minus the Arch prefix
    (resnet.py:142-168, simplebaseline.py:78-101,152-175, mlp.py:11-22).  layers / head_prefix / box_head: the SimpleBaseline
    variant (simplebaseline.py:194-241: `pose_head`, no MLP_O) and the ResNet-18 stage counts (resnet.py:236-241)."""
    layers = list(layers) if layers is not None else RESNET34_LAYERS
    out = []

    def bn(prefix, c):
        out.extend([(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
                    (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", ())])

    out.append(("backbone.conv1.weight", (64, 3, 7, 7)))
    bn("backbone.bn1", 64)
    inpl = 64
    for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], layers), start=1):
        for b in range(nblk):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f"backbone.layer{li}.{b}"
            if bottleneck:                      # Bottleneck (resnet.py:104-141), expansion 4
                outp = planes * 4
                out.append((p + ".conv1.weight", (planes, inpl, 1, 1)))
                bn(p + ".bn1", planes)
                out.append((p + ".conv2.weight", (planes, planes, 3, 3)))
                bn(p + ".bn2", planes)
                out.append((p + ".conv3.weight", (outp, planes, 1, 1)))
                bn(p + ".bn3", outp)
                if stride != 1 or inpl != outp:
                    out.append((p + ".downsample.0.weight", (outp, inpl, 1, 1)))
                    bn(p + ".downsample.1", outp)
                inpl = outp
                continue
            out.append((p + ".conv1.weight", (planes, inpl, 3, 3)))
            bn(p + ".bn1", planes)
            out.append((p + ".conv2.weight", (planes, planes, 3, 3)))
            bn(p + ".bn2", planes)
            if stride != 1 or inpl != planes:
                out.append((p + ".downsample.0.weight", (planes, inpl, 1, 1)))
                bn(p + ".downsample.1", planes)
            inpl = planes
    fch = 2048 if bottleneck else 512
    out.append(("backbone.fc.weight", (1000, fch)))
    out.append(("backbone.fc.bias", (1000,)))
    hp = head_prefix
    out.append((hp + ".deconv_layers.0.weight", (fch, 256, 4, 4)))
    bn(hp + ".deconv_layers.1", 256)
    out.append((hp + ".deconv_layers.3.weight", (256, 256, 4, 4)))
    bn(hp + ".deconv_layers.4", 256)
    out.append((hp + ".final_layer.weight", (nclasses * depth, 256, 1, 1)))
    out.append((hp + ".final_layer.bias", (nclasses * depth,)))
    for i, (a, b) in (zip((0, 2, 4), ((fch, 256), (256, 128), (128, 6))) if box_head else ()):
        out.append((f"box_head.layers.{i}.weight", (b, a)))
        out.append((f"box_head.layers.{i}.bias", (b,)))
    return out


def _key_seed(name, seed):
    h = 1469598103934665603
    for ch in (name + f"#{seed}").encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h & 0x7FFFFFFF


def fill_params(shapes, seed=1, gain=1.0):
    """Deterministic, name-keyed parameter fill (NOT the reference init -- a reproducible stand-in so the same
    weights can be rebuilt on the GPU box without shipping 101 MB).  Conv/linear weights ~ N(0, sqrt(2/fan_out))
    like kaiming_normal_(mode='fan_out') (resnet.py:170-176); BN weight ~ U(0.5,1.5), BN bias ~ N(0,0.1),
    running stats non-trivial so eval-mode parity is meaningful too."""
    params = {}
    for name, shp in shapes:
        g = torch.Generator().manual_seed(_key_seed(name, seed))
        if name.endswith("num_batches_tracked"):
            params[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean"):
            params[name] = 0.1 * torch.randn(shp, generator=g)
        elif name.endswith("running_var"):
            params[name] = 0.5 + torch.rand(shp, generator=g)
        elif len(shp) == 1 and (".bn" in name or "downsample.1" in name or "deconv_layers.1" in name or
                                "deconv_layers.4" in name):
            if name.endswith(".weight"):
                params[name] = 0.5 + torch.rand(shp, generator=g)
            else:
                params[name] = 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            params[name] = 0.05 * torch.randn(shp, generator=g)
        else:
            if "deconv_layers" in name:  # ConvTranspose weight (Cin, Cout, kh, kw): fan_out = Cin*kh*kw per torch
                fan_out = shp[0] * shp[2] * shp[3]
            elif len(shp) == 4:
                fan_out = shp[0] * shp[2] * shp[3]
            else:
                fan_out = shp[0]
            params[name] = gain * math.sqrt(2.0 / fan_out) * torch.randn(shp, generator=g)
    return params


# ----------------------------------------------------------------------------- M1 backbone
def _bn(x, p, prefix, training, eps=1e-5, momentum=0.1, stats=None):
    """nn.BatchNorm2d (resnet.py:147-149).  Training mode normalises with biased batch variance; running stats
    are updated with the unbiased one (torch semantics).  `stats` (dict) collects updated running stats."""
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if stats is not None:
            n = x.numel() / x.shape[1]
            stats[prefix + ".running_mean"] = (1 - momentum) * p[prefix + ".running_mean"] + momentum * mean.detach()
            stats[prefix + ".running_var"] = (1 - momentum) * p[prefix + ".running_var"] + momentum * (
                var.detach() * n / (n - 1))
    else:
        mean, var = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    inv = torch.rsqrt(var + eps)
    return (x - mean[None, :, None, None]) * (inv * w)[None, :, None, None] + b[None, :, None, None]


def resnet34_forward(p, image, training=True, stats=None, feats=None, frozen_bn=False, layers=None, bottleneck=False):
    """ResNet.forward (resnet.py:199-221) with BasicBlock.forward (resnet.py:85-101).
    frozen_bn: BACKBONE.FREEZE_BATCHNORM (resnet.py:146-149 bn_layer = FrozenBatchNorm2d, resnet.py:33-69): every backbone BatchNorm is the
    fixed affine map scale = w * rsqrt(running_var + 1e-5), bias = b - running_mean * scale in BOTH modes (weight / bias are buffers)."""
    if frozen_bn:
        training = False
    layers = list(layers) if layers is not None else RESNET34_LAYERS
    x = F.conv2d(image, p["backbone.conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(x, p, "backbone.bn1", training, stats=stats))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    inpl = 64
    for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], layers), start=1):
        for b in range(nblk):
            stride = 2 if (b == 0 and li > 1) else 1
            pre = f"backbone.layer{li}.{b}"
            res = x
            if bottleneck:                      # Bottleneck.forward (resnet.py:119-141): the stride sits on the 3x3 convolution
                out = F.relu(_bn(F.conv2d(x, p[pre + ".conv1.weight"]), p, pre + ".bn1", training, stats=stats))
                out = F.relu(_bn(F.conv2d(out, p[pre + ".conv2.weight"], stride=stride, padding=1), p, pre + ".bn2", training, stats=stats))
                out = _bn(F.conv2d(out, p[pre + ".conv3.weight"]), p, pre + ".bn3", training, stats=stats)
                if stride != 1 or inpl != planes * 4:
                    res = _bn(F.conv2d(x, p[pre + ".downsample.0.weight"], stride=stride), p, pre + ".downsample.1", training, stats=stats)
                x = F.relu(out + res)
                inpl = planes * 4
                continue
            out = F.conv2d(x, p[pre + ".conv1.weight"], stride=stride, padding=1)
            out = F.relu(_bn(out, p, pre + ".bn1", training, stats=stats))
            out = F.conv2d(out, p[pre + ".conv2.weight"], stride=1, padding=1)
            out = _bn(out, p, pre + ".bn2", training, stats=stats)
            if stride != 1 or inpl != planes:
                res = F.conv2d(x, p[pre + ".downsample.0.weight"], stride=stride)
                res = _bn(res, p, pre + ".downsample.1", training, stats=stats)
            x = F.relu(out + res)
            inpl = planes
        if feats is not None:
            feats[f"res_layer{li}"] = x
    return x, x.mean(3).mean(2)  # res_layer4, res_layer4_mean (resnet.py:217-221)


# ----------------------------------------------------------------------------- M2/M3 head
def softargmax3d(logits, nclasses, depth, height, width, norm_type="softmax"):
    """norm_heatmap(norm_type) + confd + renorm + integral_heatmap3d
    (simplebaseline.py:16-40, 177-190, 43-71).  logits (B, nclasses*depth, H, W) -> uvd (B,C,3), conf (B,C)."""
    B = logits.shape[0]
    x = logits.reshape(B, nclasses, -1)
    if norm_type == "softmax":
        x = F.softmax(x, 2)
    elif norm_type == "sigmoid":
        x = x.sigmoid()
    elif norm_type == "divide_sum":
        x = x / x.sum(dim=2, keepdim=True)
    else:
        raise NotImplementedError(norm_type)
    conf = torch.max(x, dim=-1).values
    x = x / (x.sum(dim=-1, keepdim=True) + 1e-7)
    x = x.reshape(B, nclasses, depth, height, width)
    d_acc = x.sum(dim=[3, 4])
    v_acc = x.sum(dim=[2, 4])
    u_acc = x.sum(dim=[2, 3])
    wd = torch.arange(depth, dtype=x.dtype) / depth
    wv = torch.arange(height, dtype=x.dtype) / height
    wu = torch.arange(width, dtype=x.dtype) / width
    u = (u_acc * wu).sum(-1, keepdim=True)
    v = (v_acc * wv).sum(-1, keepdim=True)
    d = (d_acc * wd).sum(-1, keepdim=True)
    return torch.cat([u, v, d], dim=-1), conf


def head_forward(p, feat, nclasses, depth, training=True, stats=None, keep=None, norm_type="softmax", hp="hybrid_head"):
    """IntegralDeconvHead.forward (simplebaseline.py:177-190); deconv stack (simplebaseline.py:152-175)."""
    x = F.conv_transpose2d(feat, p[hp + ".deconv_layers.0.weight"], stride=2, padding=1)
    x = F.relu(_bn(x, p, hp + ".deconv_layers.1", training, stats=stats))
    x = F.conv_transpose2d(x, p[hp + ".deconv_layers.3.weight"], stride=2, padding=1)
    x = F.relu(_bn(x, p, hp + ".deconv_layers.4", training, stats=stats))
    logits = F.conv2d(x, p[hp + ".final_layer.weight"], p[hp + ".final_layer.bias"])
    if keep is not None:
        keep["logits"] = logits
    H, W = logits.shape[2], logits.shape[3]
    return softargmax3d(logits, nclasses, depth, H, W, norm_type)


# ----------------------------------------------------------------------------- M4 pose assembly
def ortho6d_to_rotmat(poses):
    """compute_rotation_matrix_from_ortho6d (utils/transform.py:578-618)."""
    def nrm(v):
        mag = torch.sqrt(v.pow(2).sum(1))
        mag = torch.max(mag, v.new_tensor([1e-8]))
        return v / mag[:, None]

    x = nrm(poses[:, 0:3])
    z = nrm(torch.cross(x, poses[:, 3:6], dim=1))
    y = torch.cross(z, x, dim=1)
    return torch.stack([x, y, z], dim=2)


def uvd2xyz(uvd, root_joint, intr, inp_res, depth_range=0.4):
    """batch_uvd2xyz (utils/transform.py:512-546) with ref_bone_len == 1."""
    res = torch.tensor([float(inp_res[0]), float(inp_res[1])], dtype=uvd.dtype)
    uv = uvd[:, :, :2] * res
    z = (uvd[:, :, 2] - 0.5) * depth_range + root_joint[:, 2:3]
    fx, fy, cx, cy = intr[:, 0, 0], intr[:, 1, 1], intr[:, 0, 2], intr[:, 1, 2]
    f = torch.stack([fx, fy], 1)[:, None, :]
    c = torch.stack([cx, cy], 1)[:, None, :]
    xy = (uv - c) / f * z[..., None]
    return torch.cat([xy, z[..., None]], -1)


def box_head_forward(p, x):
    """MLP_O.forward (mlp.py:11-25)."""
    x = F.relu(F.linear(x, p["box_head.layers.0.weight"], p["box_head.layers.0.bias"]))
    x = F.relu(F.linear(x, p["box_head.layers.2.weight"], p["box_head.layers.2.bias"]))
    return F.linear(x, p["box_head.layers.4.weight"], p["box_head.layers.4.bias"])


def hybrid_forward(p, batch, inp_res, nclasses=22, depth=28, center_idx=0, training=True, stats=None, keep=None, frozen_bn=False,
                   norm_type="softmax", layers=None, bottleneck=False):
    """HybridBaseline.forward (hybridbaseline.py:37-96)."""
    image = batch["image"]
    H, W = image.shape[2], image.shape[3]
    feat, feat_mean = resnet34_forward(p, image, training, stats=stats, feats=keep, frozen_bn=frozen_bn, layers=layers, bottleneck=bottleneck)
    kp3d, conf = head_forward(p, feat, nclasses, depth, training, stats=stats, keep=keep, norm_type=norm_type)
    box6d = box_head_forward(p, feat_mean)
    pose_abs = uvd2xyz(kp3d, batch["root_joint"], batch["cam_intr"], inp_res)
    joints_abs = pose_abs[:, 0:21]
    boxroot = pose_abs[:, 21:22]
    R = ortho6d_to_rotmat(box6d)
    corners_abs = torch.matmul(R, batch["corners_can"].permute(0, 2, 1)).permute(0, 2, 1) + boxroot
    root = joints_abs[:, center_idx]
    c2d = torch.matmul(batch["cam_intr"], corners_abs.permute(0, 2, 1)).permute(0, 2, 1)
    c2d = c2d[:, :, 0:2] / c2d[:, :, 2:3]
    c2d = torch.stack([c2d[:, :, 0] / W, c2d[:, :, 1] / H], dim=2)
    c2d_uvd = torch.cat([c2d, torch.zeros_like(c2d[:, :, 0:1])], dim=2)
    final_uvd = torch.cat([kp3d[:, 0:21], c2d_uvd, kp3d[:, 21:22]], dim=1)
    return {
        "joints_3d_abs": joints_abs,
        "corners_3d_abs": corners_abs,
        "joints_3d": joints_abs - root[:, None],
        "corners_3d": corners_abs - root[:, None],
        "2d_uvd": final_uvd,
        "boxroot_3d_abs": boxroot,
        "box_rot_rotmat": R,
        "kp3d": kp3d,
        "kp3d_confd": conf,
    }


def simple_forward(p, batch, inp_res, nclasses=29, depth=28, center_idx=0, training=True, stats=None, layers=None, norm_type="softmax",
                   bottleneck=False):
    """SimpleBaseline.forward (simplebaseline.py:211-241): backbone -> pose_head -> uvd2xyz; 21 joints + 8 corners straight from
    the heat maps (no box head)."""
    feat, _ = resnet34_forward(p, batch["image"], training, stats=stats, layers=layers, bottleneck=bottleneck)
    kp3d, conf = head_forward(p, feat, nclasses, depth, training, stats=stats, norm_type=norm_type, hp="pose_head")
    abs_ = uvd2xyz(kp3d, batch["root_joint"], batch["cam_intr"], inp_res)
    joints_abs, corners_abs = abs_[:, :21], abs_[:, 21:]
    root = joints_abs[:, center_idx]
    return {"joints_3d_abs": joints_abs, "corners_3d_abs": corners_abs, "joints_3d": joints_abs - root[:, None],
            "corners_3d": corners_abs - root[:, None], "2d_uvd": kp3d, "kp3d_confd": conf}


# ----------------------------------------------------------------------------- L1-L3 losses
def joints_loss(preds, targs, lambda_j=1.0, lambda_c=0.2):
    """JointsLoss.__call__ (criterions/jointloss.py:25-67): MSE over ALL B*N*3 elements, masked ones are 0."""
    jt = (targs["joints_3d"] + targs["root_joint"][:, None]) * targs["joints_vis"][..., None]
    jp = preds["joints_3d_abs"] * targs["joints_vis"][..., None]
    ct = (targs["corners_3d"] + targs["root_joint"][:, None]) * targs["corners_vis"][..., None]
    cp = preds["corners_3d_abs"] * targs["corners_vis"][..., None]
    lj = F.mse_loss(jp, jt)
    lc = F.mse_loss(cp, ct)
    return lambda_j * lj + lambda_c * lc, {"joints_3d_loss": lj, "corners_3d_loss": lc}


def draw_view_vectors(n_virtual_views):
    """sample_view_vectors (criterions/ordinal.py:59-71) -- consumes torch global RNG: rand(n) then rand(n)."""
    theta = torch.rand(n_virtual_views) * 2.0 * np.pi
    u = torch.rand(n_virtual_views)
    s = torch.sqrt(1.0 - u ** 2)
    nv = torch.stack([s * torch.cos(theta), s * torch.sin(theta), u], dim=1)
    return torch.cat([torch.tensor([[0.0, 0.0, 1.0]]), nv], dim=0)


def draw_pair_subset(npairs):
    """random.shuffle + first third (ordinal.py:165-168) -- consumes python global RNG."""
    idx = list(range(npairs))
    random.shuffle(idx)
    return idx[: npairs // 3]


JOINT_PAIRS = list(combinations(range(21), 2))   # ordinal.py:86-88
PART_PAIRS = list(combinations(range(20), 2))    # ordinal.py:90-92
HO_PAIRS = list(product(range(21), range(8)))    # ordinal.py:243-247


def _ord_joint(a, b, views):
    # jointlevel_ordinal_relation (ordinal.py:39-56): (a-b).n for every view
    return torch.einsum("bpk,vk->bpv", a - b, views)


def hand_ord_loss(preds, targs, views, jsel, psel, lambda_joint=1.0, lambda_part=1.0):
    """HandOrdLoss.__call__ (ordinal.py:144-227) with the RNG draws passed in."""
    vis = targs["joints_vis"][..., None]
    jp = preds["joints_3d_abs"] * vis
    jt = (targs["joints_3d"] + targs["root_joint"][:, None]) * vis
    i0 = [JOINT_PAIRS[i][0] for i in jsel]
    i1 = [JOINT_PAIRS[i][1] for i in jsel]
    gt = _ord_joint(jt[:, i0], jt[:, i1], views)
    pr = _ord_joint(jp[:, i0], jp[:, i1], views)
    lj = torch.log(1.0 + F.relu(-1.0 * torch.sign(gt) * pr)).mean()

    def parts(j):  # joints_2_part_pairs (ordinal.py:98-121)
        return (j - j[:, JOINTS_IDX_PARENTS])[:, 1:]

    pp, pt = parts(jp), parts(jt)
    a0 = [PART_PAIRS[i][0] for i in psel]
    a1 = [PART_PAIRS[i][1] for i in psel]
    gt = torch.einsum("bpk,vk->bpv", torch.cross(pt[:, a0], pt[:, a1], dim=-1), views)  # ordinal.py:17-36
    pr = torch.einsum("bpk,vk->bpv", torch.cross(pp[:, a0], pp[:, a1], dim=-1), views)
    lp = F.relu(-1.0 * torch.sign(gt) * pr).mean()
    return lambda_joint * lj + lambda_part * lp, {"joint_ord_loss": lj, "part_ord_loss": lp}


def scene_ord_loss(preds, targs, views, sel, lambda_scene=1.0):
    """SceneOrdLoss.__call__ (ordinal.py:262-306) with the RNG draws passed in."""
    jv = targs["joints_vis"][..., None]
    cv = targs["corners_vis"][..., None]
    jp = preds["joints_3d_abs"] * jv
    jt = (targs["joints_3d"] + targs["root_joint"][:, None]) * jv
    cp = preds["corners_3d_abs"] * cv
    ct = (targs["corners_3d"] + targs["root_joint"][:, None]) * cv
    i0 = [HO_PAIRS[i][0] for i in sel]
    i1 = [HO_PAIRS[i][1] for i in sel]
    gt = _ord_joint(jt[:, i0], ct[:, i1], views)
    pr = _ord_joint(jp[:, i0], cp[:, i1], views)
    ls = torch.log(1.0 + F.relu(-1.0 * torch.sign(gt) * pr)).mean()
    return lambda_scene * ls, {"scene_ord_loss": ls}


def criterion(preds, targs, lambdas=(0.5, 0.2, 0.1), n_views_hand=20, n_views_scene=40, draws=None):
    """Criterion.compute_losses (criterions/criterion.py:57-67) over [JointsLoss, HandOrdLoss, SceneOrdLoss]
    (config/ho3dv2_clasbased_jlol_artiboost2.yaml:166-172).  RNG order = the reference's: HandOrd draws its
    view vectors, then joint-pair shuffle, then part-pair shuffle; SceneOrd draws views, then pair shuffle."""
    l1, d1 = joints_loss(preds, targs)
    if draws is None:
        hv = draw_view_vectors(n_views_hand)
        jsel = draw_pair_subset(len(JOINT_PAIRS))
        psel = draw_pair_subset(len(PART_PAIRS))
        sv = draw_view_vectors(n_views_scene)
        ssel = draw_pair_subset(len(HO_PAIRS))
        draws = dict(hand_views=hv, joint_sel=jsel, part_sel=psel, scene_views=sv, scene_sel=ssel)
    l2, d2 = hand_ord_loss(preds, targs, draws["hand_views"], draws["joint_sel"], draws["part_sel"])
    l3, d3 = scene_ord_loss(preds, targs, draws["scene_views"], draws["scene_sel"])
    total = lambdas[0] * l1 + lambdas[1] * l2 + lambdas[2] * l3
    out = {}
    out.update(d1); out.update(d2); out.update(d3)
    out["final_loss"] = total
    return total, out, draws


def sym_corner_loss(preds, targs, sym_R, sym_t):
    """SymCornerLoss.__call__ (criterions/symcornerloss.py:49-102, use_ho3d_ycb=False branch): min over the
    object's symmetry set (padded with identities, symcornerloss.py:40-42) of the vis-masked mean squared
    corner error.  sym_R (21,K,3,3), sym_t (21,K,3,1) in metres."""
    obj = targs["obj_idx"].long() - 1
    T = targs["obj_transf"]
    Rs, ts = sym_R[obj], sym_t[obj]                                   # (B,K,3,3), (B,K,3,1)
    can = targs["corners_can"].permute(0, 2, 1)[:, None]              # (B,1,3,8)
    sym_can = torch.matmul(Rs, can) + ts                              # (B,K,3,8)
    gt = torch.matmul(T[:, None, :3, :3], sym_can) + T[:, None, :3, 3:]
    gt = gt.permute(0, 1, 3, 2)                                       # (B,K,8,3)
    vis = targs["corners_vis"]
    pred = (preds["corners_3d_abs"] * vis[..., None])[:, None]
    gt = gt * vis[:, None, :, None]
    return ((gt - pred) ** 2).mean(-1).mean(-1).min(dim=-1)[0].mean()


# ----------------------------------------------------------------------------- V1 metric
def mean_epe_mm(pred_abs, targ_rel, root):
    """Mean3DEPE / ValMetricMean3DEPE2 per-sample value (metrics/meanepe.py:41-58, val_metric.py:84-106)."""
    diff = (pred_abs - (targ_rel + root[:, None])) * 1000.0
    return torch.norm(diff, dim=2).mean(dim=1)


# ----------------------------------------------------------------------------- T1 optimiser step
def clip_and_adam(params, grads, m, v, step, lr=5e-5, max_norm=0.001, b1=0.9, b2=0.999, eps=1e-8):
    """clip_grad_norm_(max_norm) + torch.optim.Adam(lr, wd=0) step (train_artiboost.py:91-96,
    utils/netutils.py:26-33).  `step` is the 1-based step count.  In place; returns the pre-clip total norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for p, g, mi, vi in zip(params, grads, m, v):
        g = g * coef
        mi.mul_(b1).add_(g, alpha=1 - b1)
        vi.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (vi.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(mi, denom, value=-lr / bc1)
    return total


# ----------------------------------------------------------------------------- V1 CCV re-weighting
def ccv_update_method_1(weight_map, val_res, lower, upper):
    """ArtiBoostLoader.update_method_1 (artiboost/artiboost_loader.py:503-524)."""
    ids = list(val_res.keys())
    vals = np.array(list(val_res.values()))
    conf = (vals.max() - vals) / ((vals.max() - vals.min()) + 1e-8)
    upd = 1.0 / (conf + 0.5)
    w = weight_map.clone()
    for i, (o, v, g) in enumerate(ids):
        w[o, v, g] *= float(upd[i])
    return torch.clamp(w, lower, upper)


def ccv_row_col(tidx, n_row, n_col):
    """OVGSet.row_col_calc (artiboost/ovg_set.py:162-170)."""
    return (tidx // (n_row * n_col), (tidx // n_col) % n_row, tidx % n_col)

"""Regression-based baseline of the eval / submit path (BASELINE.json configs[0]:
`submit_reload.py` with a regbased config on CPU, batch size 8, forward only):

  HOPRegNet      anakin/models/hpregnet.py:18-150   ResNet-18 features -> MANO branch (hand) + TransHead (object)
  ManoBranch     anakin/models/mano.py:46-137       MLP -> PCA pose + shape -> MANO layer
  ResNet18       anakin/models/resnet.py:142-236    (the torchvision-keyed backbone, as a plain torch module)

This is NOT the hot path: it is the small forward-only model the reference's own CPU configuration runs, kept as plain torch
modules so that it runs wherever torch does (CPU here, as configs[0] says).  The MANO forward is the torch restatement of the
same arithmetic the HIP kernel `ab_mano_lbs` and `oracle/pose_oracle.mano_lbs` implement (pinned to the reference's in-tree
MANO layer by tests/golden/mano.npz); MANO_RIGHT.pkl is licensed and absent, so the seeded stand-in hand model is used unless
`MANO_ASSETS_ROOT/models/MANO_RIGHT.pkl` exists."""
import os
import pickle
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .registry import BACKBONE, HEAD, MODEL, Queries, build_backbone, build_head, enable_lower_param

MANO_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
MANO_TIPS = [745, 317, 444, 556, 673]
MANO_JOINT_REORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]


# ------------------------------------------------------------------------------------------------ backbone
class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))


class _TorchResNet(nn.Module):
    """resnet.py:142-236 (BasicBlock variants) in plain torch -- the backbone of the regression-based model, which runs where the
    reference runs it (CPU in BASELINE configs[0], any torch device otherwise); `forward(image=...)` returns the res_layer1..4 /
    res_layer4_mean dict.  BACKBONE.PRETRAINED: true (torchvision's ImageNet weights, resnet.py:194-197,249-262) is a download:
    reported and skipped -- the eval configs load the whole model from ARCH.PRETRAINED right after."""
    LAYERS = (2, 2, 2, 2)

    @enable_lower_param
    def __init__(self, **cfg):
        super().__init__()
        if cfg.get("PRETRAINED"):
            import warnings
            warnings.warn(f"{type(self).__name__} PRETRAINED: the ImageNet weights are a torchvision download and are not fetched")
        if cfg.get("FREEZE_BATCHNORM"):
            raise NotImplementedError("FREEZE_BATCHNORM")
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, self.LAYERS[0])
        self.layer2 = self._make_layer(128, self.LAYERS[1], 2)
        self.layer3 = self._make_layer(256, self.LAYERS[2], 2)
        self.layer4 = self._make_layer(512, self.LAYERS[3], 2)
        self.fc = nn.Linear(512, 1000)             # present in the reference's state_dict (resnet.py:164); unused
        self.features = self.output_channel = 512
        for m in self.modules():                   # resnet.py:170-176
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [_BasicBlock(self.inplanes, planes, stride, down)]
        self.inplanes = planes
        layers += [_BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, **kwargs):
        x = self.maxpool(self.relu(self.bn1(self.conv1(kwargs["image"]))))
        f = OrderedDict()
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
            f[f"res_layer{i}"] = x
        f["res_layer4_mean"] = x.mean(3).mean(2).view(x.size(0), -1)
        return f


@BACKBONE.register_module
class ResNet18(_TorchResNet):
    LAYERS = (2, 2, 2, 2)


@BACKBONE.register_module
class ResNet34(_TorchResNet):
    """The backbone config_eval/eval_ho3dv2_regbased_artiboost.yaml names for HOPRegNet (the clasbased HybridBaseline runs its own
    ResNet-34 on the HIP kernels, hybridnet.py)."""
    LAYERS = (3, 4, 6, 3)


# ------------------------------------------------------------------------------------------------ MANO layer (torch)
def _rodrigues(aa):
    """axis-angle [..., 3] -> rotation matrices [..., 3, 3] (exact series near 0)."""
    th = aa.norm(dim=-1, keepdim=True)
    small = th < 1e-6
    ths = torch.where(small, torch.ones_like(th), th)
    k = aa / ths
    K = torch.zeros(aa.shape[:-1] + (3, 3), dtype=aa.dtype, device=aa.device)
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -k[..., 2], k[..., 1], k[..., 2]
    K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -k[..., 0], -k[..., 1], k[..., 0]
    s, c = torch.sin(th)[..., None], torch.cos(th)[..., None]
    eye = torch.eye(3, dtype=aa.dtype, device=aa.device).expand(K.shape)
    R = eye + s * K + (1 - c) * (K @ K)
    return torch.where(small[..., None], eye, R)


class ManoLayerTorch(nn.Module):
    """manotorch.manolayer.ManoLayer(rot_mode="axisang", use_pca, ncomps, center_idx, flat_hand_mean) call contract:
    forward(pose_coeffs [B, 3 + ncomps], betas [B, 10] | None) -> (verts [B,778,3], joints [B,21,3], full_poses [B,48])."""

    def __init__(self, hand_model, ncomps=15, use_pca=True, center_idx=None, flat_hand_mean=False):
        super().__init__()
        self.ncomps, self.use_pca, self.center_idx = ncomps, use_pca, center_idx
        f = lambda k: torch.from_numpy(np.ascontiguousarray(hand_model[k], np.float32))      # noqa: E731
        # MANO constants are assets, not learned state: non-persistent, so this layer adds NO keys to the state_dict.  The
        # reference's checkpoints carry manotorch's own `mano_branch.mano_layer.th_*` buffers (mano.py:24,107 read
        # `_buffers["th_J_regressor"]`, `th_faces`); HOPRegNet drops those keys on load, the constants come from the assets.
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights"):
            self.register_buffer(k, f(k), persistent=False)
        hm = np.zeros(45, np.float32) if flat_hand_mean else np.asarray(hand_model["hands_mean"], np.float32)
        comps = hand_model.get("hands_components")
        comps = np.eye(45, dtype=np.float32) if comps is None else np.asarray(comps, np.float32)
        self.register_buffer("hands_mean", torch.from_numpy(hm), persistent=False)
        self.register_buffer("comps", torch.from_numpy(comps[:ncomps] if use_pca else np.eye(45, dtype=np.float32)), persistent=False)
        self.register_buffer("th_faces", torch.from_numpy(np.asarray(hand_model["faces"], np.int64)), persistent=False)

    def forward(self, pose_coeffs, betas=None):
        B = pose_coeffs.shape[0]
        hand = pose_coeffs[:, 3:] @ self.comps if self.use_pca else pose_coeffs[:, 3:48]
        full = torch.cat([pose_coeffs[:, :3], self.hands_mean[None] + hand], 1)
        if betas is None:
            betas = pose_coeffs.new_zeros((B, 10))
        R = _rodrigues(full.view(B, 16, 3))
        pose_map = (R[:, 1:] - torch.eye(3, dtype=R.dtype, device=R.device)).reshape(B, 135)
        v_shaped = self.v_template[None] + torch.einsum("vkl,bl->bvk", self.shapedirs, betas)
        J = torch.einsum("jv,bvk->bjk", self.J_regressor, v_shaped)
        v_posed = v_shaped + torch.einsum("vkp,bp->bvk", self.posedirs, pose_map)
        G = [None] * 16
        for j in range(16):
            L = torch.zeros((B, 4, 4), dtype=R.dtype, device=R.device)
            L[:, :3, :3], L[:, 3, 3] = R[:, j], 1
            par = MANO_PARENTS[j]
            L[:, :3, 3] = J[:, 0] if par < 0 else J[:, j] - J[:, par]
            G[j] = L if par < 0 else G[par] @ L
        G = torch.stack(G, 1)
        Jh = torch.cat([J, J.new_zeros((B, 16, 1))], 2)
        G2 = G.clone()
        G2[:, :, :, 3] = G2[:, :, :, 3] - torch.einsum("bjmn,bjn->bjm", G, Jh)
        T = torch.einsum("vj,bjmn->bvmn", self.weights, G2)
        verts = torch.einsum("bvmn,bvn->bvm", T, torch.cat([v_posed, v_posed.new_ones((B, 778, 1))], 2))[:, :, :3]
        joints = torch.cat([G[:, :, :3, 3], verts[:, MANO_TIPS]], 1)[:, MANO_JOINT_REORDER]
        if self.center_idx is not None:
            c = joints[:, self.center_idx:self.center_idx + 1]
            verts, joints = verts - c, joints - c
        return verts, joints, full


def load_hand_model(mano_assets_root=None, seed=1):
    """MANO_RIGHT.pkl under `mano_assets_root` when it exists (licensed download), else the seeded MANO-shaped stand-in."""
    p = os.path.join(mano_assets_root or "", "models", "MANO_RIGHT.pkl")
    if mano_assets_root and os.path.isfile(p):
        with open(p, "rb") as f:
            dd = pickle.load(f, encoding="latin1")
        J = dd["J_regressor"]
        return {"v_template": np.asarray(dd["v_template"]), "shapedirs": np.asarray(dd["shapedirs"]), "posedirs": np.asarray(dd["posedirs"]),
                "J_regressor": np.asarray(J.toarray() if hasattr(J, "toarray") else J), "weights": np.asarray(dd["weights"]),
                "hands_mean": np.asarray(dd["hands_mean"]), "hands_components": np.asarray(dd["hands_components"]), "faces": np.asarray(dd["f"])}
    from .assets import make_hand_model
    hm = dict(make_hand_model(seed))
    rng = np.random.default_rng(seed + 11)
    q, _ = np.linalg.qr(rng.standard_normal((45, 45)))
    hm["hands_components"] = q.astype(np.float32)                       # orthonormal PCA basis of the stand-in
    hm["hands_mean"] = (0.1 * rng.standard_normal(45)).astype(np.float32)
    return hm


# ------------------------------------------------------------------------------------------------ heads and the model
@HEAD.register_module
class ManoBranch(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        self.inp_dim, self.ncomps, self.use_pca = cfg["INPUT_DIM"], cfg["NCOMPS"], cfg["USE_PCA"]
        self.center_idx = cfg["CENTER_IDX"]
        self.use_shape = cfg.get("USE_SHAPE", True)
        if not self.use_pca:
            raise NotImplementedError("ManoBranch with USE_PCA: false (16 x 9 rotation-matrix regression, mano.py:76-79,90-96)")
        base = [self.inp_dim, 512, 512]
        layers = []
        for i, o in zip(base[:-1], base[1:]):
            layers += [nn.Linear(i, o), nn.ReLU()]
        self.base_layer = nn.Sequential(*layers)
        self.pose_reg = nn.Linear(base[-1], self.ncomps + 3)
        if self.use_shape:
            self.shape_reg = nn.Sequential(nn.Linear(base[-1], 10))
        self.mano_layer = ManoLayerTorch(load_hand_model(cfg.get("MANO_ASSETS_ROOT")), ncomps=self.ncomps, use_pca=True,
                                         center_idx=self.center_idx, flat_hand_mean=False)
        self.faces = self.mano_layer.th_faces

    def forward(self, feature):
        x = self.base_layer(feature)
        pose = self.pose_reg(x)
        shape = self.shape_reg(x) if self.use_shape else None
        verts, joints, full = self.mano_layer(pose, shape)
        return {"hand_verts_3d": verts, "joints_3d": joints, "mano_shape": shape, "mano_pca_pose": pose, "mano_full_pose": full}


def batch_persp_proj2d(points3d, cam_intr):
    """anakin/utils/transform.py batch_persp_proj2d: K @ p, divide by depth."""
    hom = torch.matmul(cam_intr, points3d.transpose(1, 2)).transpose(1, 2)
    return hom[:, :, :2] / hom[:, :, 2:]


@MODEL.register_module
class HOPRegNet(nn.Module):
    class TransHead(nn.Module):
        def __init__(self, inp_dim, out_dim):
            super().__init__()
            if out_dim not in (3, 9):
                raise ValueError(f"Unrecognized TransHead out dim: {out_dim}")
            self.decoder = nn.Sequential(nn.Linear(inp_dim, inp_dim // 2), nn.ReLU())
            self.final_layer = nn.Linear(inp_dim // 2, out_dim)

        def forward(self, inp):
            return self.final_layer(self.decoder(inp))

    @enable_lower_param
    def __init__(self, **cfg):
        super().__init__()
        self.inp_res = cfg["DATA_PRESET"]["IMAGE_SIZE"]
        self.feature_dim = cfg["HEAD"]["INPUT_DIM"]
        self.center_idx = cfg["DATA_PRESET"]["CENTER_IDX"]
        if cfg.get("MANO_FHB_ADAPTOR", False):
            raise NotImplementedError("MANO_FHB_ADAPTOR (FPHAB skeleton adaptor, hpregnet.py:41-49)")
        self.base_net = build_backbone(cfg["BACKBONE"])
        self.mano_branch = build_head(cfg["HEAD"], default_args=cfg["DATA_PRESET"])
        self.obj_transfhead = HOPRegNet.TransHead(self.feature_dim, 9)
        pretrained = cfg.get("PRETRAINED", "")
        if pretrained:
            if not os.path.isfile(pretrained):
                raise FileNotFoundError(f"=> No {type(self).__name__} checkpoints file found in {pretrained}")
            ck = torch.load(pretrained, map_location="cpu")
            sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
            self.load_state_dict(self.clean_reference_state_dict(sd), strict=True)

    MANO_LAYER_PREFIX = "mano_branch.mano_layer."

    @classmethod
    def clean_reference_state_dict(cls, sd):
        """Reference checkpoint -> the keys this module owns: the DataParallel `module.` prefix goes, and so do the MANO asset
        buffers manotorch's ManoLayer persists (`mano_branch.mano_layer.th_*`; hpregnet.py:59-64 loads strictly over them)."""
        out = {}
        for k, v in sd.items():
            k = k[7:] if k.startswith("module.") else k
            if not k.startswith(cls.MANO_LAYER_PREFIX):
                out[k] = v
        return out

    def recover_mano(self, feature, samples):
        from .models import ortho6d_to_rotmat  # noqa: F401  (same helper module; kept local to avoid an import cycle)
        res = self.mano_branch(feature)
        cam_intr, root = samples[Queries.CAM_INTR].to(feature.device), samples[Queries.ROOT_JOINT].to(feature.device)
        res["joints_3d_abs"] = res["joints_3d"] + root.unsqueeze(1)
        res["hand_verts_3d_abs"] = res["hand_verts_3d"] + root.unsqueeze(1)
        res["joints_2d"] = batch_persp_proj2d(res["joints_3d_abs"], cam_intr)
        res["hand_verts_2d"] = batch_persp_proj2d(res["hand_verts_3d_abs"], cam_intr)
        res["root_joint"] = root
        return res

    def recover_object(self, feature, samples):
        from .models import ortho6d_to_rotmat
        t = self.obj_transfhead(feature)
        rotmat = ortho6d_to_rotmat(t[:, 3:]).view(t.shape[0], 3, 3)
        root, cam_intr = samples[Queries.ROOT_JOINT].to(feature.device), samples[Queries.CAM_INTR].to(feature.device)
        center = root + t[:, :3]
        corners = rotmat.bmm(samples[Queries.CORNERS_CAN].to(feature.device).float().transpose(1, 2)).transpose(1, 2) + center.unsqueeze(1)
        return {"obj_center": center, "corners_3d_abs": corners, "obj_pred_tsl": t[:, :3], "obj_pred_rot": rotmat,
                "corners_2d": batch_persp_proj2d(corners, cam_intr), "box_rot_rotmat": rotmat, "boxroot_3d_abs": center}

    def forward(self, samples):
        image = samples["image"]
        dev = next(self.parameters()).device
        feats = self.base_net(image=image.to(dev))
        mano = self.recover_mano(feats["res_layer4_mean"], samples)
        obj = self.recover_object(feats["res_layer4_mean"], samples)
        obj["corners_3d"] = obj["corners_3d_abs"] - mano["root_joint"].unsqueeze(1)
        return {**mano, **obj}

"""Online synthesis control plane + device data plane: the ArtiBoostLoader of this package.

Reference: anakin/artiboost/artiboost_loader.py:49-340,503-598 (CCV weight map, per-epoch pose generation, mining /
re-weighting), ovg_set.py:104-178 (Categorical sampling of (object, view, grasp) triplets), view_engine.py:17-86,
grasp_engine.py:47-53, preprocessor.py:20-99 + scrambler.py:65-81 (pose generator), rendered_dataset.py:103-274 (GT
assembly + augmentation draws), render_infra.py / utils/renderer.py (render servers).

What changed structurally (MI355X-first):
  * no render-server processes, no multiprocessing queues, no one-pickle-per-sample cache on /dev/shm: `prepare()`
    generates the epoch's poses in batches of 256 on the GPU (HIP LBS kernel) and keeps them as device-resident SoA
    tensors; every training step renders its own batch on the training GPU's stream (render.DeviceRenderer);
  * CCV sampling, view / grasp lookup, GT geometry (bbox crop, jitter, affine, visibility) stay on the host exactly as
    in the reference (float64 numpy, integer truncations included) but vectorised over the epoch;
  * the GrabNet refiner (refiner.py) runs inside the pose generator when the manager config has a
    REFINER block of TYPE "hand_obj" (artiboost_amd/refiner.py); its checkpoint is a download.
"""
import os

import numpy as np
import torch

from . import _lib as L
from .registry import IMAGE_PLANE_KEY, RUNTIME, PlaneTag, Queries, SynthQueries, tag_image_plane
from .render import SAMPLE_DTYPE, DeviceRenderer


# --------------------------------------------------------------------------- rotations (pytorch3d semantics)
def aa_to_rotmat(aa):
    """axis_angle_to_matrix (wrapper anakin/utils/transform.py:42-55)."""
    ang = torch.norm(aa, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    k = torch.where(small, 0.5 - ang * ang / 48.0, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    w = torch.cos(half)[..., 0]
    x, y, z = (aa * k).unbind(-1)
    two_s = 2.0 / (w * w + x * x + y * y + z * z)
    R = torch.stack([1 - two_s * (y * y + z * z), two_s * (x * y - z * w), two_s * (x * z + y * w),
                     two_s * (x * y + z * w), 1 - two_s * (x * x + z * z), two_s * (y * z - x * w),
                     two_s * (x * z - y * w), two_s * (y * z + x * w), 1 - two_s * (x * x + y * y)], dim=-1)
    return R.reshape(aa.shape[:-1] + (3, 3))


def rotmat_to_aa(R):
    """matrix_to_quaternion + quaternion_to_axis_angle (wrapper anakin/utils/transform.py:291-306)."""
    m00, m01, m02 = R[..., 0, 0], R[..., 0, 1], R[..., 0, 2]
    m10, m11, m12 = R[..., 1, 0], R[..., 1, 1], R[..., 1, 2]
    m20, m21, m22 = R[..., 2, 0], R[..., 2, 1], R[..., 2, 2]
    q_abs = torch.sqrt(torch.clamp(torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22,
                                                1 - m00 - m11 + m22], -1), min=0.0))
    cand = torch.stack([torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
                        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
                        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
                        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * torch.clamp(q_abs[..., None], min=0.1))
    best = q_abs.argmax(dim=-1)
    q = torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4)))[..., 0, :]
    norms = torch.norm(q[..., 1:], dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    ang = 2 * half
    small = ang.abs() < 1e-6
    k = torch.where(small, 0.5 - ang * ang / 48.0, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    return q[..., 1:] / k


# --------------------------------------------------------------------------- R1: MANO layer on the HIP kernel
class ManoLayerHIP:
    """ManoLayer(rot_mode='axisang', center_idx=None, flat_hand_mean=True) call contract (grasp_engine.py:90-95):
    __call__(pose [B,48], betas [B,10]) -> verts [B,778,3], joints [B,21,3], transforms_abs [B,16,4,4]."""

    def __init__(self, hand_model, device="cuda"):
        self.dev = torch.device(device)
        f = lambda k: torch.from_numpy(np.ascontiguousarray(hand_model[k], np.float32)).to(self.dev)   # noqa: E731
        self.v_template, self.shapedirs, self.posedirs = f("v_template"), f("shapedirs"), f("posedirs")
        self.J_regressor, self.weights, self.hands_mean = f("J_regressor"), f("weights"), f("hands_mean")

    def __call__(self, pose, betas):
        B = pose.shape[0]
        pose = pose.contiguous().float()
        betas = betas.contiguous().float()
        verts = torch.empty((B, 778, 3), dtype=torch.float32, device=self.dev)
        joints = torch.empty((B, 21, 3), dtype=torch.float32, device=self.dev)
        T = torch.empty((B, 16, 4, 4), dtype=torch.float32, device=self.dev)
        L.check(L.lib().ab_mano_lbs(L.ptr(pose), L.ptr(betas), L.ptr(self.v_template), L.ptr(self.shapedirs),
                                    L.ptr(self.posedirs), L.ptr(self.J_regressor), L.ptr(self.weights),
                                    L.ptr(self.hands_mean), L.i(B), L.ptr(verts), L.ptr(joints), L.ptr(T), L.stream()),
                "ab_mano_lbs")
        return verts, joints, T

    def get_rotation_center(self, betas):
        vs = self.v_template[None] + torch.einsum("vkl,bl->bvk", self.shapedirs, betas)
        return torch.einsum("v,bvk->bk", self.J_regressor[0], vs)


class PoseGenerator:
    """PreProcessorPoseGenerator.forward (preprocessor.py:20-99) with the RandomScrambler (scrambler.py:65-81);
    batched device tensors in, (obj_pose [B,4,4], hand_verts [B,778,3], joints [B,21,3]) out."""

    def __init__(self, mano: ManoLayerHIP, tsl_sigma=0.01, pose_sigma=0.1, refiner=None):
        self.mano, self.tsl_sigma, self.pose_sigma, self.refiner = mano, tsl_sigma, pose_sigma, refiner

    @staticmethod
    def scramble(hand_pose, hand_tsl, rand_pose_angle=None, rand_tsl=None):
        """RandomScrambler.forward (scrambler.py:65-81) with its two Normal draws passed in: every joint's rotation keeps its axis
        and gets `rand_pose_angle` [B,16] added to its angle (which may go negative), the translation gets `rand_tsl` [B,3]."""
        if rand_pose_angle is not None:
            hp = hand_pose.reshape(-1, 16, 3)
            nrm = torch.norm(hp, dim=-1, keepdim=True)
            axis = hp / torch.clamp(nrm, min=1e-7)
            hand_pose = (axis * (nrm[..., 0] + rand_pose_angle)[..., None]).reshape(-1, 48)
        if rand_tsl is not None:
            hand_tsl = hand_tsl + rand_tsl
        return hand_pose, hand_tsl

    def __call__(self, hand_pose, hand_shape, hand_tsl, persp_rotmat, camera_free_transf, z_offset, rand_pose_angle=None,
                 rand_tsl=None, obj_idx=None):
        B = hand_pose.shape[0]
        verts, joints, T = self.mano(hand_pose, hand_shape)
        joints = joints + hand_tsl[:, None]
        Rinv = persp_rotmat.transpose(1, 2)
        op_offset = torch.bmm(Rinv, joints[:, 9, :, None])[..., 0] / 2.0
        cam_sys_offset = z_offset - op_offset
        obj_pose = torch.eye(4, device=verts.device).repeat(B, 1, 1)
        obj_pose[:, :3, :3] = Rinv
        obj_pose[:, :3, 3] = cam_sys_offset
        obj_pose = torch.bmm(camera_free_transf, obj_pose)
        new_glob = rotmat_to_aa(torch.bmm(Rinv, T[:, 0, :3, :3]))
        new_pose = torch.cat([new_glob, hand_pose[:, 3:]], dim=1)
        center = self.mano.get_rotation_center(hand_shape)
        off0 = center - torch.bmm(aa_to_rotmat(hand_pose[:, :3]), center[..., None])[..., 0]
        off1 = center - torch.bmm(aa_to_rotmat(new_pose[:, :3]), center[..., None])[..., 0]
        new_tsl = torch.bmm(Rinv, (off0 + hand_tsl)[..., None])[..., 0] - off1
        new_pose, new_tsl = self.scramble(new_pose, new_tsl, rand_pose_angle, rand_tsl)
        if self.refiner is not None:        # preprocessor.py:73-80: the refiner decodes (and refines) the scrambled grasp
            res = self.refiner({"hand_pose": new_pose, "hand_tsl": new_tsl, "obj_rot": obj_pose[:, :3, :3]}, obj_idx)
            v2, j2, off = res["hand_verts"], res["joints"], cam_sys_offset[:, None]
        else:
            v2, j2, _ = self.mano(new_pose, hand_shape)
            off = (new_tsl + cam_sys_offset)[:, None]
        Rf = camera_free_transf[:, :3, :3]
        final_verts = torch.bmm(v2 + off, Rf.transpose(1, 2))
        final_joints = torch.bmm(j2 + off, Rf.transpose(1, 2))
        return obj_pose, final_verts.contiguous(), final_joints.contiguous()


# --------------------------------------------------------------------------- host geometry (reference semantics)
def align_mat(vec):
    """ViewEngine.caculate_align_mat (view_engine.py:61-86)."""
    vec = vec / np.linalg.norm(vec)
    z = np.array([0.0, 0.0, 1.0])
    zc = np.cross(z, vec)
    K = np.array([[0, -zc[2], zc[1]], [zc[2], 0, -zc[0]], [-zc[1], zc[0], 0]])
    d = float(np.dot(z, vec))
    if d == -1:
        return -np.eye(3)
    if d == 1:
        return np.eye(3)
    return np.eye(3) + K + K.dot(K) / (1 + d)


def perspective_rotmat(persp_id, u_off, th_off, u_bins, theta_bins):
    """ViewEngine.get_perspective_from_id (view_engine.py:35-57)."""
    u_id, th_id = persp_id // theta_bins, persp_id % theta_bins
    u_unit, th_unit = 2 / u_bins, (2 * np.pi) / theta_bins
    u = np.clip((-1 + u_unit / 2) + u_id * u_unit + u_off * u_unit, -1, 1)
    th = np.clip(th_unit / 2 + th_id * th_unit + th_off * th_unit, 0, 2 * np.pi)
    s = np.sqrt(1 - u * u)
    return align_mat(np.array([s * np.cos(th), s * np.sin(th), u]))


def perspective_rotmats(persp_id, u_off, th_off, u_bins, theta_bins):
    """perspective_rotmat for arrays of views: [N] ids / offsets -> [N,3,3] (same arithmetic, vectorised)."""
    persp_id = np.asarray(persp_id)
    u_id, th_id = persp_id // theta_bins, persp_id % theta_bins
    u_unit, th_unit = 2 / u_bins, (2 * np.pi) / theta_bins
    u = np.clip((-1 + u_unit / 2) + u_id * u_unit + np.asarray(u_off) * u_unit, -1, 1)
    th = np.clip(th_unit / 2 + th_id * th_unit + np.asarray(th_off) * th_unit, 0, 2 * np.pi)
    s = np.sqrt(1 - u * u)
    vec = np.stack([s * np.cos(th), s * np.sin(th), u], -1)
    vec = vec / np.linalg.norm(vec, axis=-1, keepdims=True)
    K = np.zeros(vec.shape[:-1] + (3, 3))
    K[..., 0, 2], K[..., 2, 0] = vec[..., 0], -vec[..., 0]              # skew(z x vec), z x vec = (-vy, vx, 0)
    K[..., 1, 2], K[..., 2, 1] = vec[..., 1], -vec[..., 1]
    d = vec[..., 2]
    safe = np.where(d == -1, 1.0, 1 + d)
    R = np.eye(3) + K + (K @ K) / safe[..., None, None]
    R = np.where((d == -1)[..., None, None], -np.eye(3), R)
    return np.where((d == 1)[..., None, None], np.eye(3), R)


def construct_blacklist_map(grasp_pose, u_bins, theta_bins, rng, filter_back_flag=True):
    """ArtiBoostLoader._construct_blacklist_map (artiboost_loader.py:415-500): a CCV triplet is blacklisted when the back
    of the hand faces the camera, th_sgn = (R_persp^T R_wrist [1, 0.2, 0]/|.|) . z < -0.8, with one jittered view drawn per
    triplet (view_engine.get_view -> get_perspective_from_id).  grasp_pose [n_obj, n_grasp, 48] axis-angle (wrist first).
    -> torch.bool [n_obj, n_persp, n_grasp].  (The reference's 57 600-iteration Python loop, here one vectorised pass.)"""
    n_obj, n_grasp = grasp_pose.shape[:2]
    n_persp = u_bins * theta_bins
    out = torch.zeros((n_obj, n_persp, n_grasp), dtype=torch.bool)
    if not filter_back_flag:
        return out
    u_off, th_off = rng.uniform(-0.5, 0.5, (n_obj, n_persp, n_grasp)), rng.uniform(-0.5, 0.5, (n_obj, n_persp, n_grasp))
    ids = np.broadcast_to(np.arange(n_persp)[None, :, None], u_off.shape)
    return back_facing(grasp_pose, perspective_rotmats(ids, u_off, th_off, u_bins, theta_bins))


def back_facing(grasp_pose, persp_rotmat):
    """The test of artiboost_loader.py:478-489 for every triplet: grasp_pose [n_obj, n_grasp, >=3], persp_rotmat
    [n_obj, n_persp, n_grasp, 3, 3] -> torch.bool [n_obj, n_persp, n_grasp]."""
    back = np.array([1.0, 0.2, 0.0])
    back = back / np.linalg.norm(back)
    wrist = aa_to_rotmat(torch.from_numpy(np.asarray(grasp_pose[..., :3], np.float64))).numpy() @ back        # [n_obj, n_grasp, 3]
    # (R_persp^T R_wrist back) . z = (third column of R_persp) . (R_wrist back)
    return torch.from_numpy(np.einsum("ovgk,ogk->ovg", np.asarray(persp_rotmat, np.float64)[..., :, 2], wrist) < -0.8)


def blacklist_cache_path(cfg, n_obj, n_grasp, u_bins, theta_bins, filter_back_flag, root="common/cache/CCV_blacklist",
                         obj_engine_type=None, grasp_engine_type=None):
    """The reference's md5-keyed cache file of the blacklist (artiboost_loader.py:427-448): same identifier fields, same
    JSON / md5 recipe, so a cache written by either side is found by the other."""
    import hashlib
    import json
    origin = cfg.get("OBJ_ENGINE", {}).get("OBJ_ORIGIN_DATASET", "HO3D")
    ident = {"obj_engine_type": obj_engine_type or f"{origin}ObjEngine", "sample_n_obj": n_obj,
             "obj_names": sorted(cfg.get("OBJ_ENGINE", {}).get("OBJ", [])),
             "grasp_engine_type": grasp_engine_type or f"{cfg.get('GRASP_ENGINE', {}).get('GRASP_ORIGIN_DATASET', origin)}GraspEngine",
             "sample_n_grasp": n_grasp,
             "view_engine_u_bins": u_bins, "view_engine_theta_bins": theta_bins, "filter_back_flag": filter_back_flag}
    key = hashlib.md5(json.dumps(ident, sort_keys=True).encode("ascii")).hexdigest()
    return os.path.join(root, f"{key}.pkl")


def _affine_no_rot(center, scale, res):
    a = np.zeros((3, 3))
    ratio = float(res[0]) / float(res[1])
    a[0, 0] = float(res[0]) / scale
    a[1, 1] = float(res[1]) / scale * ratio
    a[0, 2] = res[0] * (-float(center[0]) / scale + 0.5)
    a[1, 2] = res[1] * (-float(center[1]) / scale * ratio + 0.5)
    a[2, 2] = 1
    return a


def get_affine_transform(center, scale, optical_center, out_res, rot=0.0):
    """anakin/utils/transform.py:434-482."""
    rm = np.zeros((3, 3))
    sn, cs = np.sin(rot), np.cos(rot)
    rm[0, :2] = [cs, -sn]
    rm[1, :2] = [sn, cs]
    rm[2, 2] = 1
    ch = np.array([center[0], center[1], 1.0])
    t = np.eye(3)
    t[0, 2], t[1, 2] = -optical_center[0], -optical_center[1]
    ti = t.copy()
    ti[:2, 2] *= -1
    tc = ti.dot(rm).dot(t).dot(ch)
    total = _affine_no_rot(rm.dot(ch)[:2], scale, out_res).dot(rm)
    return total.astype(np.float32), _affine_no_rot(tc[:2], scale, out_res).astype(np.float32)


def assemble_gt(K, joints, obj_pose, corners_can, image_size, raw_size, center_jit_draw, scale_draw, rot, center_idx=0,
                bbox_expand=1.2, center_jit=0.1, scale_jit=0.1):
    """RenderedDataset.__getitem__ geometry for CROP_MODEL root_obj (rendered_dataset.py:127-133,155-254,276-316)."""
    j2d = (K @ joints.T).T
    j2d = j2d[:, :2] / (j2d[:, 2:3] + 1e-8)
    c3d = (obj_pose[:3, :3] @ corners_can.T).T + obj_pose[:3, 3]
    c2d = (K @ c3d.T).T
    c2d = c2d[:, :2] / (c2d[:, 2:3] + 1e-8)
    all2d = np.concatenate([j2d[[0]], c2d], 0)
    mn, mx = all2d.min(0), all2d.max(0)
    center = np.asarray([int((mx[0] + mn[0]) / 2), int((mx[1] + mn[1]) / 2)])      # hodata.py:178-186
    scale = max(mx[0] - mn[0], mx[1] - mn[1]) * bbox_expand
    center = center + (center_jit * scale * np.asarray(center_jit_draw)).astype(int)
    scale = scale * np.clip(scale_draw + 1.0, 1 - scale_jit, 1 + scale_jit)
    rm = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1]]).astype(np.float32)
    aff, post = get_affine_transform(center, scale, [K[0, 2], K[1, 2]], image_size, rot)
    out = {"affine": aff, Queries.CAM_INTR: post.dot(K).astype(np.float32)}
    j3 = rm.dot(joints.astype(np.float32).T).T
    root = j3[center_idx]
    out[Queries.ROOT_JOINT] = root
    out[Queries.JOINTS_3D] = j3 - root
    hom = lambda p: aff.dot(np.concatenate([p, np.ones((p.shape[0], 1))], 1).T).T[:, :2]   # noqa: E731
    j2a = hom(j2d.astype(np.float32)).astype(np.float32)
    out[Queries.JOINTS_2D] = j2a

    def vis(raw2d, aug2d, n):
        v = (raw2d[:, 0] >= 0) & (raw2d[:, 0] < raw_size[0]) & (raw2d[:, 1] >= 0) & (raw2d[:, 1] < raw_size[1])
        if v.sum() < n * 0.4:
            return np.zeros(n, np.float32)
        va = ((aug2d[:, 0] >= 0) & (aug2d[:, 0] < image_size[0]) & (aug2d[:, 1] >= 0) &
              (aug2d[:, 1] < image_size[1])).astype(np.float32)
        return np.zeros(n, np.float32) if va.sum() < n * 0.4 else va

    out[Queries.JOINTS_VIS] = vis(j2d, j2a, 21)
    c3 = rm.dot(c3d.astype(np.float32).T).T
    out[Queries.CORNERS_3D] = c3 - root
    c2a = hom(c2d.astype(np.float32))
    out[Queries.CORNERS_2D] = c2a.astype(np.float32)
    out[Queries.CORNERS_VIS] = vis(c2d, c2a, 8)
    out[Queries.CORNERS_CAN] = corners_can.astype(np.float32)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = rm @ obj_pose[:3, :3].astype(np.float32)
    T[:3, 3] = rm.dot(obj_pose[:3, 3].astype(np.float32))
    out[Queries.OBJ_TRANSF] = T
    return out


def assemble_gt_batch(K, joints, obj_pose, corners_can, image_size, raw_size, center_jit_draw, scale_draw, rot, center_idx=0,
                      bbox_expand=1.2, center_jit=0.1, scale_jit=0.1):
    """assemble_gt for a whole epoch at once (same arithmetic, numpy-vectorised over the S samples; the per-sample Python
    loop costs 75 us per sample, 3 s for a 40 k-sample epoch -- as long as training on it).  joints [S,21,3], obj_pose
    [S,4,4], corners_can [S,8,3] float64; draws [S,2], [S], [S].  Returns the stacked sample dict + "affine" [S,3,3]."""
    K = np.asarray(K, np.float64)

    def project(p):
        h = np.einsum("ij,snj->sni", K, p)
        return h[..., :2] / (h[..., 2:3] + 1e-8)

    j2d = project(joints)
    c3d = np.einsum("sij,snj->sni", obj_pose[:, :3, :3], corners_can) + obj_pose[:, None, :3, 3]
    c2d = project(c3d)
    all2d = np.concatenate([j2d[:, :1], c2d], 1)
    mn, mx = all2d.min(1), all2d.max(1)
    center = ((mx + mn) / 2).astype(np.int64)                                  # int() truncation, hodata.py:178-186
    scale = np.maximum(mx[:, 0] - mn[:, 0], mx[:, 1] - mn[:, 1]) * bbox_expand
    center = center + (center_jit * scale[:, None] * np.asarray(center_jit_draw)).astype(np.int64)
    scale = scale * np.clip(np.asarray(scale_draw) + 1.0, 1 - scale_jit, 1 + scale_jit)
    return gt_core_batch(K, joints, j2d, c3d, c2d, corners_can, obj_pose, center, scale, rot, image_size, raw_size, center_idx)


def gt_core_batch(K, j3d, j2d, c3d, c2d, corners_can, obj_transf, center, scale, rot, image_size, raw_size, center_idx=0,
                  raw_j2d=None, raw_c2d=None, train_split=True):
    """The part of RenderedDataset.__getitem__ / HOdata.__getitem__ after the crop centre and scale are fixed
    (rendered_dataset.py:192-254, hodata.py:361-433), for S samples: affine, rotated 3-D ground truth, warped 2-D ground
    truth, visibility masks, object transform.  K [3,3] or [S,3,3]; raw_* = the un-flipped 2-D annotations the raw-image
    visibility test reads (default: j2d / c2d)."""
    S = j3d.shape[0]
    K = np.asarray(K, np.float64)
    Ks = np.broadcast_to(K, (S, 3, 3))
    rot = np.broadcast_to(np.asarray(rot, np.float64), (S,))
    scale = np.asarray(scale, np.float64)
    cs, sn = np.cos(rot), np.sin(rot)
    rm = np.zeros((S, 3, 3), np.float32)
    rm[:, 0, 0], rm[:, 0, 1], rm[:, 1, 0], rm[:, 1, 1], rm[:, 2, 2] = cs, -sn, sn, cs, 1.0
    # get_affine_transform (anakin/utils/transform.py:434-482), component-wise
    ox, oy = Ks[:, 0, 2], Ks[:, 1, 2]
    cx, cy = np.asarray(center)[:, 0].astype(np.float64), np.asarray(center)[:, 1].astype(np.float64)
    res0, res1 = float(image_size[0]), float(image_size[1])
    ratio = res0 / res1

    def no_rot(c0, c1):
        a = np.zeros((S, 3, 3))
        a[:, 0, 0] = res0 / scale
        a[:, 1, 1] = res1 / scale * ratio
        a[:, 0, 2] = res0 * (-c0 / scale + 0.5)
        a[:, 1, 2] = res1 * (-c1 / scale * ratio + 0.5)
        a[:, 2, 2] = 1
        return a

    rm64 = np.zeros((S, 3, 3))
    rm64[:, 0, 0], rm64[:, 0, 1], rm64[:, 1, 0], rm64[:, 1, 1], rm64[:, 2, 2] = cs, -sn, sn, cs, 1.0
    aff = (no_rot(cs * cx - sn * cy, sn * cx + cs * cy) @ rm64).astype(np.float32)
    dx, dy = cx - ox, cy - oy
    post = no_rot(cs * dx - sn * dy + ox, sn * dx + cs * dy + oy).astype(np.float32)
    out = {"affine": aff, Queries.CAM_INTR: np.einsum("sij,sjk->sik", post, Ks).astype(np.float32)}
    j3 = np.einsum("sij,snj->sni", rm, j3d.astype(np.float32))
    root = j3[:, center_idx]
    out[Queries.ROOT_JOINT] = root
    out[Queries.JOINTS_3D] = j3 - root[:, None]

    def hom(p):                                                                  # transform_coords, float32 points in a float64 product
        ph = np.concatenate([p.astype(np.float32).astype(np.float64), np.ones(p.shape[:-1] + (1,))], -1)
        return np.einsum("sij,snj->sni", aff.astype(np.float64), ph)[..., :2]

    def vis(raw2d, aug2d, n):
        if not train_split:
            return np.ones((S, n), np.float32)
        v = (raw2d[..., 0] >= 0) & (raw2d[..., 0] < raw_size[0]) & (raw2d[..., 1] >= 0) & (raw2d[..., 1] < raw_size[1])
        va = ((aug2d[..., 0] >= 0) & (aug2d[..., 0] < image_size[0]) & (aug2d[..., 1] >= 0) & (aug2d[..., 1] < image_size[1])).astype(np.float32)
        dead = (v.sum(1) < n * 0.4) | (va.sum(1) < n * 0.4)
        va[dead] = 0.0
        return va

    j2a = hom(j2d).astype(np.float32)
    out[Queries.JOINTS_2D] = j2a
    out[Queries.JOINTS_VIS] = vis(j2d if raw_j2d is None else raw_j2d, j2a, j2d.shape[1])
    c3 = np.einsum("sij,snj->sni", rm, c3d.astype(np.float32))
    out[Queries.CORNERS_3D] = c3 - root[:, None]
    c2a = hom(c2d)
    out[Queries.CORNERS_2D] = c2a.astype(np.float32)
    out[Queries.CORNERS_VIS] = vis(c2d if raw_c2d is None else raw_c2d, c2a, c2d.shape[1])
    out[Queries.CORNERS_CAN] = np.asarray(corners_can).astype(np.float32)
    T = np.tile(np.eye(4, dtype=np.float32), (S, 1, 1))
    T[:, :3, :3] = rm @ obj_transf[:, :3, :3].astype(np.float32)
    T[:, :3, 3] = np.einsum("sij,sj->si", rm, obj_transf[:, :3, 3].astype(np.float32))
    out[Queries.OBJ_TRANSF] = T
    return out


# --------------------------------------------------------------------------- the loader
class LazyImageBatch(dict):
    """A batch dict whose `image` entry (the reference's collated float CHW image, rendered_dataset.py:267-271) is produced on first
    access from the padded NHWC4 plane the HIP model consumes: the reference-shaped loop never reads it (the model takes
    `image_nhwc4_padded`), so the 50 MB CHW store per batch happens only for a consumer that asks (`batch["image"]`, `.get`, `in`)."""

    def __init__(self, items, make_image=None):
        super().__init__(items)
        self._make_image = make_image

    def _materialise(self):
        make, self._make_image = self._make_image, None
        if make is not None:
            dict.__setitem__(self, Queries.IMAGE, make())

    def __missing__(self, key):
        if key == Queries.IMAGE and self._make_image is not None:
            self._materialise()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def get(self, key, default=None):
        if key == Queries.IMAGE and self._make_image is not None:
            self._materialise()
        return dict.get(self, key, default)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (key == Queries.IMAGE and self._make_image is not None)


def chw_from_padded(xpad, plane):
    """float32 CHW image [B, 3, H, W] = v / 255 - 0.5 from the zero-bordered NHWC4 tensor, bit-identical to the renderer's own CHW
    store (render.hip warp_jitter_kernel: o = v / 255.0f - 0.5f) for the "u8n" plane (2 v - 255, exact in bf16) and the fp32 image."""
    x = xpad[:, 3:-3, 3:-5, :3].permute(0, 3, 1, 2)
    if plane == "u8n":      # (a division by a DEVICE tensor: torch turns a division by a python scalar into a multiplication by 1 / 255)
        return ((x.float() + 255.0) * 0.5) / torch.full((), 255.0, dtype=torch.float32, device=x.device) - 0.5
    return x.float().contiguous()


class ArtiBoostLoader:
    """The reference's ArtiBoostLoader (artiboost_loader.py:49-340) on the on-GPU renderer: same constructor keywords, same
    public surface (prepare / __iter__ / __len__ / step_eval / sample_weight_map / occurence_map / use_synth /
    update_method_*).  The OVG DataLoader, the pose cache on /dev/shm, the render-server processes and the DataLoader workers
    behind the reference's constructor do not exist here (one in-process batched render per step), so `shuffle`,
    `num_workers`, `pin_memory`, `collate_fn`, `time_f` and the `arg_extra` fields are accepted and unused."""

    def __init__(self, real_train_set=None, arg=None, arg_extra=None, cfg=None, cfg_dataset=None, cfg_preset=None, time_f=None,
                 batch_size=1, shuffle=False, num_workers=0, pin_memory=False, drop_last=False, collate_fn=None, random_seed=1,
                 **kwargs):
        """Reference call site: train/train_artiboost.py:175-190.  cfg = cfg["MANAGER"], cfg_dataset = cfg["DATASET"],
        cfg_preset = cfg["DATA_PRESET"]; arg.device / arg.batch_size as parsed by anakin/opt.py.

        real_train_set: the real training set.  synth_len = int(SYNTH_FACTOR * len(real_train_set)) as in the reference; an
        empty / None set (HO3D and DexYCB are downloads) gives a synthetic-only epoch of cfg["SYNTH_LEN"] samples.
        Iteration over this object yields the synthetic batches; the frames of a non-empty real set (a
        `realdata.HOdataSource`) are mixed in by `realdata.MixedLoader(RealBatcher(real_train_set, ...), self_with_the_synthetic
        share of the batch, batch_size)` -- the MixedDataset of the reference with a static per-batch split.
        Extensions (keyword only): assets (SceneAssets; default: seeded stand-ins for cfg OBJ_ENGINE.OBJ_ORIGIN_DATASET),
        synth_len, device, rank, world_size, grasps, compute_dtype -- the padded image the model consumes: torch.float32 / torch.bfloat16,
        or "u8n" (one bf16 plane of the integers 2 v - 255: the bf16x3 model's native input).  Default (the reference's call, which has
        no such keyword): whatever the most recently built model asked for (registry.RUNTIME; bf16x3 -> "u8n"), resolved when the first
        batch is staged; torch.float32 when no model exists yet."""
        if cfg is None or cfg_preset is None:
            raise TypeError("ArtiBoostLoader needs cfg (the MANAGER block) and cfg_preset (the DATA_PRESET block)")
        from .assets import SceneAssets
        assets = kwargs.pop("assets", None)
        if assets is None:
            assets = SceneAssets(cfg.get("OBJ_ENGINE", {}).get("OBJ_ORIGIN_DATASET", "HO3D"), seed=1)
        real_len = len(real_train_set) if real_train_set is not None else 0
        synth_len = kwargs.pop("synth_len", None)
        if synth_len is None:
            synth_len = int(cfg.get("SYNTH_FACTOR", 0.0) * real_len) if real_len else int(cfg.get("SYNTH_LEN", 0))
        device = kwargs.pop("device", None) or (getattr(arg, "device", None) if arg is not None else None) or "cuda"
        if batch_size is None and arg is not None and getattr(arg, "batch_size", None):
            batch_size = arg.batch_size          # (an explicit batch_size -- 1 included: a rank's share under --batch_size N over N GPUs -- wins)
        self.real_train_set, self.real_len = real_train_set, real_len
        self.shuffle, self.num_workers, self.pin_memory, self.drop_last, self.collat_fn = shuffle, num_workers, pin_memory, drop_last, collate_fn
        self.cfg_dataset = cfg_dataset
        self._setup(assets, cfg, cfg_preset, int(batch_size), synth_len, device=device,
                    compute_dtype=kwargs.pop("compute_dtype", None), random_seed=random_seed,
                    rank=kwargs.pop("rank", 0), world_size=kwargs.pop("world_size", 1), grasps=kwargs.pop("grasps", None))
        self.epoch_len_total = self.real_len + self.synth_len        # the reference's epoch_len (real + synthetic samples)

    @classmethod
    def from_assets(cls, assets, cfg, cfg_preset, batch_size, synth_len, device="cuda", compute_dtype=torch.bfloat16,
                    random_seed=1, rank=0, world_size=1, grasps=None):
        """Direct construction from scene assets and an explicit epoch length (tests, bench.py)."""
        self = cls.__new__(cls)
        self.real_train_set, self.real_len, self.cfg_dataset = None, 0, None
        self._setup(assets, cfg, cfg_preset, batch_size, synth_len, device=device, compute_dtype=compute_dtype,
                    random_seed=random_seed, rank=rank, world_size=world_size, grasps=grasps)
        self.epoch_len_total = self.synth_len
        return self

    def _setup(self, assets, cfg, cfg_preset, batch_size, synth_len, device="cuda", compute_dtype=torch.bfloat16,
               random_seed=1, rank=0, world_size=1, grasps=None):
        self.assets, self.cfg, self.preset = assets, cfg, cfg_preset
        self.dev = torch.device(device)
        self.batch_size, self.synth_len = batch_size, int(synth_len)
        self.rank, self.world = rank, world_size
        # compute_dtype "u8n": the padded image leaves the renderer as ONE bf16 plane of the odd integers 2 v - 255 (AB_DT_U8N) -- what the
        # bf16x3 stem consumes directly (two MFMA passes, no split pass over the image); the network input is that plane / 510
        self._auto_dtype = compute_dtype is None
        self._set_compute_dtype(torch.float32 if compute_dtype is None else compute_dtype)
        ve = cfg["VIEW_ENGINE"]
        self.u_bins, self.theta_bins = ve["PERSP_U_BINS"], ve["PERSP_THETA_BINS"]
        self.z_range = ve["CAMERA_Z_RANGE"]
        self.n_obj = assets.n_obj
        self.n_persp = self.u_bins * self.theta_bins
        self.n_grasp = cfg["GRASP_ENGINE"]["GRASP_NUM"]
        cam = cfg["RENDERER"]["CAM_PARAM"]
        self.render_size = cfg["RENDERER"]["RENDER_SIZE"]
        self.K = np.array([[cam["FX"], 0, cam["CX"]], [0, cam["FY"], cam["CY"]], [0, 0, 1.0]])
        self.image_size = list(cfg_preset["IMAGE_SIZE"])
        self.center_idx = cfg_preset.get("CENTER_IDX", 0)
        self.bbox_expand = float(cfg_preset.get("BBOX_EXPAND_RATIO", 1.2))
        self.blur_radius = 0.1        # RenderedDataset(aug=True): rendered_dataset.py:67 (hard-coded there; 0 when aug is off)
        self.sample_weight_map = torch.ones((self.n_obj, self.n_persp, self.n_grasp), dtype=torch.float32)
        self.occurence_map = torch.zeros((self.n_obj, self.n_persp, self.n_grasp), dtype=torch.bool)
        wu = cfg.get("WEIGHT_UPDATE", {"LOWER": 0.1, "UPPER": 10.0})
        self.sample_weight_lower_bound, self.sample_weight_upper_bound = wu["LOWER"], wu["UPPER"]
        dt = cfg.get("DIST_THRESHOLD", {"LOWER": 8.0, "UPPER": 16.0})
        self.dist_lower_threshold, self.dist_upper_threshold = dt["LOWER"], dt["UPPER"]
        self.update_method_key = cfg.get("UPDATE_METHOD", "method_1")
        self.n_epochs = cfg.get("EPOCH", 100)
        self.use_synth = self.synth_len > 0
        self._seed = random_seed
        self.rng = np.random.default_rng(random_seed)
        self.torch_gen = torch.Generator().manual_seed(random_seed)
        from .assets import make_grasps
        self.grasps = grasps if grasps is not None else make_grasps(self.n_obj, self.n_grasp, seed=random_seed + 5)
        # blacklist of back-of-hand views (artiboost_loader.py:82,124-130): FILTER.BACK defaults to true in the reference
        self.filter_back_flag = cfg["FILTER"].get("BACK", True) if "FILTER" in cfg else True
        self.blacklist_map = self._blacklist(cfg.get("BLACKLIST_CACHE_ROOT"))
        self.sample_weight_map[self.blacklist_map] = 0.0
        sc = cfg.get("SCRAMBLER", {"HAND_TSL_SIGMA": 0.01, "HAND_POSE_SIGMA": 0.1})
        if self.dev.type == "cuda":
            self.mano = ManoLayerHIP(assets.hand, device)
            self.renderer = DeviceRenderer(assets, self.K, self.render_size[0], self.render_size[1], device)
        else:       # host-only instance (epoch planning / CCV bookkeeping); prepare() needs the GPU
            self.mano = self.renderer = None
        self.refiner = None
        rcfg = cfg.get("REFINER")
        if rcfg and rcfg.get("TYPE", "null") != "null" and self.mano is not None:      # preprocessor.py:73-80, refiner.py:151-224
            from .assets import resample_objects
            from .refiner import Refiner
            self.refiner = Refiner.build(rcfg["TYPE"], rcfg, self.mano, device)
            self.refiner.setup(resample_objects(assets, int(rcfg.get("N_SAMPLE_VERTS", 10000)), seed=random_seed + 6))
        self.pose_generator = PoseGenerator(self.mano, sc["HAND_TSL_SIGMA"], sc["HAND_POSE_SIGMA"], refiner=self.refiner)
        self.epoch = None
        self.cursor = 0
        self._resolve_compute_dtype()

    def _set_compute_dtype(self, compute_dtype):
        self.image_plane = "u8n" if (isinstance(compute_dtype, str) and compute_dtype == "u8n") else "f32"
        self.dtype = torch.bfloat16 if self.image_plane == "u8n" else compute_dtype

    def _resolve_compute_dtype(self):
        """A loader built with the reference's keywords follows the model built before it (see __init__)."""
        if self._auto_dtype and RUNTIME.get("loader_compute_dtype") is not None:
            self._set_compute_dtype(RUNTIME["loader_compute_dtype"])

    # ------------------------------------------------------------------ CCV sampling (ovg_set.py:104-132,162-178)
    def _sample_ccv(self, is_train=True):
        # train: == torch.distributions.Categorical(w).sample((n,)) (ovg_set.py:113-114: multinomial with replacement over the
        # normalised weights) but from the loader's own seeded generator, so every DDP rank draws the same epoch.
        # val (OVGSet.val(), ovg_set.py:108-118): uniform over the non-blacklisted triplets WITHOUT replacement.
        if is_train:
            w = self.sample_weight_map.reshape(-1)
            idx = torch.multinomial(w / w.sum(), self.synth_len, replacement=True, generator=self.torch_gen)
        else:
            w = torch.ones_like(self.sample_weight_map)
            w[self.blacklist_map] = 0.0
            if self.synth_len > int(w.sum()):
                raise ValueError(f"val epoch of {self.synth_len} samples exceeds the {int(w.sum())} admissible CCV triplets")
            idx = torch.multinomial(w.reshape(-1), self.synth_len, replacement=False, generator=self.torch_gen)
        o = torch.div(idx, self.n_persp * self.n_grasp, rounding_mode="floor")
        v = torch.div(idx, self.n_grasp, rounding_mode="floor") % self.n_persp
        g = idx % self.n_grasp
        occ = torch.zeros_like(self.occurence_map)
        occ[o, v, g] = True
        self.occurence_map |= occ
        return o.numpy(), v.numpy(), g.numpy()

    def _blacklist(self, cache_root=None):
        """_construct_blacklist_map incl. its pickle cache (only when BLACKLIST_CACHE_ROOT is configured: the reference
        writes under ./common/cache unconditionally)."""
        import pickle
        path = None
        if self.grasps[0] is None:          # planning-only instance without a grasp table
            return torch.zeros((self.n_obj, self.n_persp, self.n_grasp), dtype=torch.bool)
        if cache_root and self.filter_back_flag:
            path = blacklist_cache_path(self.cfg, self.n_obj, self.n_grasp, self.u_bins, self.theta_bins, self.filter_back_flag, cache_root)
            if os.path.exists(path):
                with open(path, "rb") as f:
                    cached = pickle.load(f)
                if list(cached.shape) == [self.n_obj, self.n_persp, self.n_grasp]:
                    return torch.as_tensor(cached, dtype=torch.bool)
        bl = construct_blacklist_map(self.grasps[0], self.u_bins, self.theta_bins, np.random.default_rng(self._seed + 11), self.filter_back_flag)
        if path:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "wb") as f:
                pickle.dump(bl, f)
        return bl

    # ------------------------------------------------------------------ epoch plan (host only; testable without a GPU)
    def plan_epoch(self, is_train=True):
        """Draw the epoch's CCV triplets and every per-sample random draw from the shared seed, then keep this rank's
        slice idx[rank::world]: all ranks consume identical RNG streams, so the union over ranks
        is exactly the single-process epoch and the slices are disjoint."""
        o, v, g = self._sample_ccv(is_train)
        rng = self.rng
        S_all = self.synth_len
        # every rank must run the same number of steps (a rank with one batch more would wait forever in the gradient
        # all-reduce): under DDP the epoch is cut to a multiple of world * batch_size before slicing (drop_last)
        S_use = S_all if self.world == 1 else (S_all // (self.world * self.batch_size)) * self.world * self.batch_size
        if S_use == 0:
            raise ValueError(f"synth_len {S_all} < world_size * batch_size = {self.world * self.batch_size}: no rank would get a batch")
        sl = slice(self.rank, S_use, self.world)
        plan = dict(u_off=rng.uniform(-0.5, 0.5, S_all), th_off=rng.uniform(-0.5, 0.5, S_all),
                    free=rng.uniform(0, 2 * np.pi, S_all), zoff=rng.uniform(self.z_range[0], self.z_range[1], S_all),
                    d_pose=self.pose_generator.pose_sigma * rng.standard_normal((S_all, 16)),
                    d_tsl=self.pose_generator.tsl_sigma * rng.standard_normal((S_all, 3)))
        aug = dict(center=rng.uniform(-1, 1, (S_all, 2)), scale=rng.normal(0, 0.1 / 3.0, S_all),
                   rot=rng.uniform(-0.2 * np.pi, 0.2 * np.pi, S_all), hid=rng.integers(0, self.assets.hand_tex.shape[0], S_all),
                   light=rng.uniform(1.0, 5.0, S_all), bid=rng.integers(0, self.assets.backgrounds.shape[0], S_all),
                   bcrop=rng.integers(self.render_size[0], self.assets.backgrounds.shape[1] + 1, S_all),
                   bx=rng.uniform(0, 1, S_all), by=rng.uniform(0, 1, S_all),
                   order=np.stack([rng.permutation(4) for _ in range(S_all)]),
                   bright=rng.uniform(0.9, 1.1, S_all), contrast=rng.uniform(0.9, 1.1, S_all),
                   sat=rng.uniform(0.9, 1.1, S_all), hue=rng.uniform(-0.075, 0.075, S_all),
                   blur=self.blur_radius * rng.uniform(0, 1, S_all))
        plan = {k: val[sl] for k, val in plan.items()}
        plan["aug"] = {k: val[sl] for k, val in aug.items()}
        plan.update(o=o[sl], v=v[sl], g=g[sl], global_index=np.arange(S_all)[sl])
        return plan

    # ------------------------------------------------------------------ prepare(): per-epoch pose generation
    def generate_render_cache(self, is_train=True):
        """artiboost_loader.py:352-400 under its own name: is_train=False draws the validation-mode epoch (OVGSet.val)."""
        self.prepare(is_train=is_train)

    def prepare(self, cache=None, is_train=True):
        """artiboost_loader.py:279-291,352-400: sample CCV triplets, generate poses (GPU, batches of 256), assemble GT
        (host), upload the epoch as device SoA tensors.  Under DDP every rank draws the same epoch (same seed) and
        keeps its own slice idx[rank::world].
        cache: an epoch of poses read back from the reference's on-disk format (ccv_cache.load_cache: one pickle per
        sample as written by CacheRecorder, cache_recorder.py:22-45) -- used instead of generating poses."""
        if not self.use_synth:
            self.epoch = None
            return
        if self.mano is None:
            raise RuntimeError("ArtiBoostLoader.prepare() renders on the GPU: construct the loader with a cuda device")
        plan = self.plan_epoch(is_train)
        if cache is not None:
            S = len(cache["obj_id"])
            if S > len(plan["o"]):
                raise ValueError(f"cache holds {S} samples, the epoch plan {len(plan['o'])}")
            plan["aug"] = {k: val[:S] for k, val in plan["aug"].items()}
            plan.update(o=np.asarray(cache["obj_id"], np.int64), v=np.asarray(cache["persp_id"], np.int64),
                        g=np.asarray(cache["grasp_id"], np.int64), global_index=plan["global_index"][:S])
            t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.dev)   # noqa: E731
            self._assemble_epoch(plan, t32(cache["obj_pose"]), t32(cache["hand_verts"]), t32(cache["hand_joints"]))
        else:
            self._assemble_epoch(plan, *self._generate_poses(plan))

    def _generate_poses(self, plan):
        """Pose generator over the epoch (preprocessor.py:20-99, batches of 256 as artiboost_loader.py:352-400)."""
        o, v, g = plan["o"], plan["v"], plan["g"]
        u_off, th_off, free, zoff, d_pose, d_tsl = (plan[k] for k in ("u_off", "th_off", "free", "zoff", "d_pose", "d_tsl"))
        pick = lambda a: a   # noqa: E731  (plan_epoch already sliced to this rank)
        S = len(o)
        Rp = perspective_rotmats(v, pick(u_off), pick(th_off), self.u_bins, self.theta_bins)
        fr = pick(free)
        Tf = np.tile(np.eye(4), (S, 1, 1))
        Tf[:, 0, 0], Tf[:, 0, 1], Tf[:, 1, 0], Tf[:, 1, 1] = np.cos(fr), -np.sin(fr), np.sin(fr), np.cos(fr)
        z3 = np.zeros((S, 3))
        z3[:, 2] = pick(zoff)
        gp, gs, gt_ = self.grasps
        dev = self.dev
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)   # noqa: E731
        obj_pose, verts, joints = [], [], []
        for s0 in range(0, S, 256):
            s1 = min(S, s0 + 256)
            op, hv, jt = self.pose_generator(t(gp[o[s0:s1], g[s0:s1]]), t(gs[o[s0:s1], g[s0:s1]]), t(gt_[o[s0:s1], g[s0:s1]]),
                                             t(Rp[s0:s1]), t(Tf[s0:s1]), t(z3[s0:s1]), t(pick(d_pose)[s0:s1]), t(pick(d_tsl)[s0:s1]),
                                             obj_idx=torch.from_numpy(np.ascontiguousarray(o[s0:s1], np.int64)).to(dev))
            obj_pose.append(op); verts.append(hv); joints.append(jt)
        return torch.cat(obj_pose), torch.cat(verts), torch.cat(joints)

    def _assemble_epoch(self, plan, obj_pose_d, verts_d, joints_d):
        """RenderedDataset.__getitem__'s geometry for every sample (rendered_dataset.py:127-254) + the render records."""
        o, v, g, a = plan["o"], plan["v"], plan["g"], plan["aug"]
        S, dev = len(o), self.dev
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev)   # noqa: E731
        obj_pose_h, joints_h = obj_pose_d.cpu().numpy().astype(np.float64), joints_d.cpu().numpy().astype(np.float64)
        # what CacheRecorder would have pickled for this epoch (ccv_cache.export_epoch writes it in the reference's format)
        self.epoch_poses = dict(index=plan["global_index"], obj_id=o, persp_id=v, grasp_id=g, obj_pose=obj_pose_d,
                                hand_verts=verts_d, hand_joints=joints_d)
        # last global sample index at which this rank sees each triplet (only whole batches are trained on): gather_ccv_results
        n_used = (S // self.batch_size) * self.batch_size
        self._last_seen = {(int(a_), int(b_), int(c_)): int(gi) for a_, b_, c_, gi in
                           zip(o[:n_used], v[:n_used], g[:n_used], plan["global_index"][:n_used])}
        # ---- host: GT assembly + render descriptors
        samples = np.zeros(S, SAMPLE_DTYPE)
        samples["obj_id"], samples["hand_tex_id"], samples["bg_id"] = o, a["hid"], a["bid"]
        bgs = self.assets.backgrounds.shape[1]
        samples["bg_w"] = samples["bg_h"] = a["bcrop"]
        samples["bg_x0"] = np.floor(a["bx"] * (bgs - a["bcrop"] + 1)).astype(np.int32)
        samples["bg_y0"] = np.floor(a["by"] * (bgs - a["bcrop"] + 1)).astype(np.int32)
        samples["light"] = a["light"]
        samples["obj_pose"] = obj_pose_h.reshape(S, 16).astype(np.float32)
        fac_of = {0: a["bright"], 1: a["sat"], 2: a["hue"], 3: a["contrast"]}
        order = a["order"].astype(np.int32)
        factor = np.stack([np.choose(order[:, k], [fac_of[0], fac_of[1], fac_of[2], fac_of[3]]) for k in range(4)], 1).astype(np.float32)
        keys = (Queries.CAM_INTR, Queries.ROOT_JOINT, Queries.JOINTS_3D, Queries.JOINTS_2D, Queries.JOINTS_VIS,
                Queries.CORNERS_3D, Queries.CORNERS_2D, Queries.CORNERS_VIS, Queries.CORNERS_CAN, Queries.OBJ_TRANSF)
        r = assemble_gt_batch(self.K, joints_h, obj_pose_h, self.assets.corners_can[o].astype(np.float64), self.image_size,
                              self.render_size, a["center"], a["scale"], a["rot"], self.center_idx, self.bbox_expand)
        gt = {k: r[k] for k in keys}
        full = np.tile(np.eye(3), (S, 1, 1))
        full[:, :2] = r["affine"][:, :2].astype(np.float64)
        inv = np.linalg.inv(full)[:, :2].reshape(S, 6).astype(np.float32)
        ep = {k: t(val) for k, val in gt.items()}
        ep[Queries.OBJ_IDX] = torch.from_numpy(self.assets.obj_idx[o]).to(dev)
        ep[SynthQueries.OBJ_ID] = torch.from_numpy(o.astype(np.int64)).to(dev)
        ep[SynthQueries.PERSP_ID] = torch.from_numpy(v.astype(np.int64)).to(dev)
        ep[SynthQueries.GRASP_ID] = torch.from_numpy(g.astype(np.int64)).to(dev)
        ep[SynthQueries.IS_SYNTH] = torch.ones(S, dtype=torch.bool, device=dev)
        ep[Queries.SAMPLE_IDX] = torch.arange(S, device=dev)
        ep["_samples"] = torch.from_numpy(samples.view(np.uint8).reshape(S, -1)).to(dev)
        ep["_hand_verts"] = verts_d
        ep["_order"] = torch.from_numpy(order).to(dev)
        ep["_factor"] = torch.from_numpy(factor).to(dev)
        ep["_inv_affine"] = torch.from_numpy(inv).to(dev)
        ep["_blur"] = torch.from_numpy(a["blur"].astype(np.float32)).to(dev)
        self.epoch, self.epoch_len, self.cursor = ep, S, 0
        self._pack_batches()

    def _pack_batches(self):
        """Batch-major packed copy of the epoch: one contiguous byte row per (batch, group) so that staging a batch is
        one device copy per group ("gt" / "render") instead of one per tensor (25 launches per step)."""
        B, nb = self.batch_size, self.epoch_len // self.batch_size
        self._layout, self._packed = {}, {}
        for group in ("gt", "render"):
            off, lay = 0, []
            for k, v in self.epoch.items():
                if k.startswith("_") != (group == "render"):
                    continue
                nbytes = B * int(np.prod(v.shape[1:], dtype=np.int64)) * v.element_size()
                lay.append((k, off, nbytes, v.dtype, tuple(v.shape[1:])))
                off += (nbytes + 255) // 256 * 256
            self._layout[group] = (lay, off)
            packed = torch.zeros((max(nb, 1), off), dtype=torch.uint8, device=self.dev)
            for k, o, nbytes, dt, shp in lay:
                if nb:
                    src = self.epoch[k][:nb * B].contiguous().reshape(nb, -1)
                    packed[:, o:o + nbytes] = src.view(torch.uint8) if dt != torch.bool else src.to(torch.uint8)
            self._packed[group] = packed

    # ------------------------------------------------------------------ iteration
    def __len__(self):
        return (self.epoch_len // self.batch_size) if self.epoch is not None else 0

    def new_static_batch(self):
        """Static device buffers one batch wide (inputs of a captured hipGraph)."""
        self._resolve_compute_dtype()
        B, (W, H) = self.batch_size, self.image_size
        ep = self.epoch
        st = {}
        for group in ("gt", "render"):            # typed views into one flat staging buffer per group (see _pack_batches)
            lay, total = self._layout[group]
            flat = st["__flat_" + group] = torch.zeros(total, dtype=torch.uint8, device=self.dev)
            for k, o, nbytes, dt, shp in lay:
                st[k] = flat[o:o + nbytes].view(dt).view((B,) + shp)
        assert set(ep) <= set(st)
        st["image_nhwc4_padded"] = tag_image_plane(torch.zeros((B, H + 6, W + 8, 4), dtype=self.dtype, device=self.dev), self.image_plane)
        return st

    def load_batch(self, static, batch_idx, which="all"):
        """Gather batch `batch_idx` of the epoch into the static buffers (small async device copies).
        which: "all", "gt" (ground-truth / non-underscore keys) or "render" (the `_`-prefixed render inputs)."""
        groups = ("gt", "render") if which == "all" else (which,)
        if all(("__flat_" + g) in static for g in groups):
            for g in groups:
                static["__flat_" + g].copy_(self._packed[g][batch_idx], non_blocking=True)
            return
        s0 = batch_idx * self.batch_size
        for k, v in self.epoch.items():
            if which != "all" and (k.startswith("_") != (which == "render")):
                continue
            static[k].copy_(v[s0:s0 + self.batch_size], non_blocking=True)

    def render_into(self, static, want_chw=False, out_pad=None, out_chw=None):
        """Enqueue the batched render of the samples currently in `static` (hipGraph-capturable).  out_pad / out_chw: rows of a larger
        batch's tensors to render into instead of static's own (realdata.MixedLoader: the synthetic share of a mixed batch)."""
        W, H = self.image_size
        chw = out_chw
        if want_chw and chw is None:
            chw = static.get(Queries.IMAGE)
            if chw is None:
                chw = static[Queries.IMAGE] = torch.empty((self.batch_size, 3, H, W), dtype=torch.float32, device=self.dev)
        self.renderer.render(static["_samples"], static["_hand_verts"], static["_order"], static["_factor"],
                             static["_inv_affine"], W, H, out_pad=static["image_nhwc4_padded"] if out_pad is None else out_pad, out_chw=chw,
                             blur=static["_blur"], pad_code=2 if self.image_plane == "u8n" else None)

    def __iter__(self):
        """Reference-shaped iteration: yields batch dicts of device tensors -- the NHWC4 tensor the HIP model consumes directly (tagged
        with its plane, also under IMAGE_PLANE_KEY) and `image`, the reference's float CHW image, made only when something reads it
        (LazyImageBatch; a bf16 image plane cannot reproduce the fp32 CHW values, so there the renderer writes both as before).
        The tensors are the loader's static staging buffers: a batch is valid until the next one is asked for, as with the reference's
        pinned DataLoader batches once the loop has moved on."""
        if self.epoch is None:
            return
        static = self.new_static_batch()
        pad, plane = static["image_nhwc4_padded"], self.image_plane
        lazy = plane == "u8n" or pad.dtype == torch.float32
        for bi in range(len(self)):
            self.load_batch(static, bi)
            self.render_into(static, want_chw=not lazy)
            items = {k: v for k, v in static.items() if not k.startswith("_")}
            items[IMAGE_PLANE_KEY] = PlaneTag(plane)
            yield LazyImageBatch(items, (lambda: chw_from_padded(pad, plane)) if lazy else None)

    # ------------------------------------------------------------------ mining (artiboost_loader.py:292-340,503-598)
    def get_evaluator_result(self, evaluator):
        """artiboost_loader.py:301-327: average of the per-(object, view, grasp) measures of every validation metric."""
        from .metrics import ValMetricAR2, ValMetricMean3DEPE2
        res = [m.get_measures_averaged() for m in evaluator.metrics_list if isinstance(m, (ValMetricMean3DEPE2, ValMetricAR2))]
        if not res:
            raise ValueError("No validation metric have been found")
        if not all(set(r) == set(res[0]) for r in res):
            raise ValueError("some ccv space idx lost!")
        merged = {k: sum(r[k] for r in res) / len(res) for k in res[0]}
        return self.gather_ccv_results(merged)

    def gather_ccv_results(self, local):
        """under data parallelism every rank has measured only its slice idx[rank::world] of the epoch.
        All ranks exchange their {(o, v, g): value} dicts (one all_gather_object per epoch, <= synth_len entries) and resolve a
        triplet seen by several ranks as the single-process run would: the occurrence with the highest global sample index
        wins ("last write wins", val_metric.py:50-51).  Every rank then applies the identical mining update."""
        if self.world <= 1 or not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return local
        last = getattr(self, "_last_seen", {})
        payload = {k: (float(v), int(last.get(k, -1))) for k, v in local.items()}
        gathered = [None] * self.world
        torch.distributed.all_gather_object(gathered, payload)
        out, best = {}, {}
        for part in gathered:                       # rank order; ties cannot happen (global indices are unique)
            for k, (v, gi) in part.items():
                if k not in best or gi > best[k]:
                    best[k], out[k] = gi, v
        return out

    def step_eval(self, epoch_idx, evaluator):
        self.sample_reweight(self.get_evaluator_result(evaluator), epoch_idx)

    def sample_reweight(self, eval_res, epoch_idx):
        fn = {"method_1": self.update_method_1, "method_2": self.update_method_2, "method_3": self.update_method_3,
              "method_4": self.update_method_4}[self.update_method_key]
        out = fn(self.sample_weight_map, eval_res, self.sample_weight_lower_bound, self.sample_weight_upper_bound,
                 dist_lower_threshold=self.dist_lower_threshold, dist_upper_threshold=self.dist_upper_threshold,
                 epoch_idx=epoch_idx, n_epochs=self.n_epochs)
        self.sample_weight_map = out["sample_weight_map"]

    @staticmethod
    def _confidence(val_res):
        vals = np.array(list(val_res.values()))
        return list(val_res.keys()), vals, (vals.max() - vals) / ((vals.max() - vals.min()) + 1e-8)

    @staticmethod
    def update_method_1(sample_weight_map, val_res, lower, upper, **kw):
        ids, _, conf = ArtiBoostLoader._confidence(val_res)
        upd = (1.0 / (conf + 0.5)).tolist()
        for i, ovg in enumerate(ids):
            sample_weight_map[ovg[0], ovg[1], ovg[2]] *= upd[i]
        return {"sample_weight_map": torch.clamp(sample_weight_map, lower, upper)}

    @staticmethod
    def update_method_2(sample_weight_map, val_res, lower, upper, **kw):
        ids, _, conf = ArtiBoostLoader._confidence(val_res)
        for i, ovg in enumerate(ids):
            sample_weight_map[ovg[0], ovg[1], ovg[2]] += (-0.1 if conf[i] > 0.5 else 0.1)
        return {"sample_weight_map": torch.clamp(sample_weight_map, lower, upper)}

    @staticmethod
    def update_method_3(sample_weight_map, val_res, lower, upper, **kw):
        ids, vals, _ = ArtiBoostLoader._confidence(val_res)
        lo_m, hi_m = vals < kw["dist_lower_threshold"], vals > kw["dist_upper_threshold"]
        for i, ovg in enumerate(ids):
            if lo_m[i]:
                sample_weight_map[ovg[0], ovg[1], ovg[2]] = 0.0
            elif hi_m[i]:
                sample_weight_map[ovg[0], ovg[1], ovg[2]] = 1.0
            else:
                sample_weight_map[ovg[0], ovg[1], ovg[2]] *= 0.5
        return {"sample_weight_map": sample_weight_map, "dist_lower_ratio": lo_m.sum() / len(lo_m)}

    @staticmethod
    def update_method_4(sample_weight_map, val_res, lower, upper, **kw):
        if float(kw["epoch_idx"]) / kw["n_epochs"] < 0.75:
            out = ArtiBoostLoader.update_method_1(sample_weight_map, val_res, lower, upper)
            out["dist_lower_ratio"] = -1.0
            return out
        return ArtiBoostLoader.update_method_3(sample_weight_map, val_res, lower, upper, **kw)

    def synth_shutdown(self):
        self.use_synth = False
        self.epoch = None

"""TEST INFRASTRUCTURE ONLY (CPU oracle, numpy float64/float32) -- restatement of the *pose / geometry* half of
the ArtiBoost synthesis path.

Pinned where the reference is importable (get_affine_transform, transform_coords, get_annot_center/scale,
caculate_align_mat: tests/golden/misc.npz).  NOT pinned ("parity unpinned") where the arithmetic lives in an
absent third-party package: MANO LBS (manotorch, un-pinned git dependency, requirements.txt:178 -- restated from
the in-tree JAX layer anakin/postprocess/iknet/manolayer.py:182-276) and axis-angle<->matrix (pytorch3d@d049cd2e,
requirements.txt:177 -- cross-checked against scipy.spatial.transform.Rotation in tests).
"""
import numpy as np

MANO_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
MANO_TIPS = [745, 317, 444, 556, 673]                     # manolayer.py:263
MANO_JOINT_REORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]  # manolayer.py:267


# --------------------------------------------------------------------------- helpers pinned by misc.npz
def get_affine_trans_no_rot(center, scale, res):
    """utils/transform.py:473-482."""
    a = np.zeros((3, 3))
    ratio = float(res[0]) / float(res[1])
    a[0, 0] = float(res[0]) / scale
    a[1, 1] = float(res[1]) / scale * ratio
    a[0, 2] = res[0] * (-float(center[0]) / scale + 0.5)
    a[1, 2] = res[1] * (-float(center[1]) / scale * ratio + 0.5)
    a[2, 2] = 1
    return a


def get_affine_transform(center, scale, optical_center, out_res, rot=0.0):
    """utils/transform.py:434-470 -> (total_trans f32, affinetrans_post_rot f32)."""
    center = np.asarray(center, dtype=np.float64)
    rm = np.zeros((3, 3))
    sn, cs = np.sin(rot), np.cos(rot)
    rm[0, :2] = [cs, -sn]
    rm[1, :2] = [sn, cs]
    rm[2, 2] = 1
    ch = np.array([center[0], center[1], 1.0])
    origin_rot_center = rm.dot(ch)[:2]
    t = np.eye(3)
    t[0, 2] = -optical_center[0]
    t[1, 2] = -optical_center[1]
    ti = t.copy()
    ti[:2, 2] *= -1
    tc = ti.dot(rm).dot(t).dot(ch)
    post = get_affine_trans_no_rot(origin_rot_center, scale, out_res)
    total = post.dot(rm)
    post_rot = get_affine_trans_no_rot(tc[:2], scale, out_res)
    return total.astype(np.float32), post_rot.astype(np.float32)


def transform_coords(pts, affine):
    """utils/transform.py:422-431."""
    pts = np.asarray(pts)
    h = np.concatenate([pts, np.ones((pts.shape[0], 1))], 1)
    return affine.dot(h.T).T[:, :2]


def annot_center(pts):
    """HOdata.get_annot_center (datasets/hodata.py:178-186): int-truncated bbox centre."""
    mn, mx = pts.min(0), pts.max(0)
    return np.asarray([int((mx[0] + mn[0]) / 2), int((mx[1] + mn[1]) / 2)])


def annot_scale(pts, scale_factor=1.0):
    """HOdata.get_annot_scale (datasets/hodata.py:162-176)."""
    mn, mx = pts.min(0), pts.max(0)
    return max(mx[0] - mn[0], mx[1] - mn[1]) * scale_factor


def align_mat(vec):
    """ViewEngine.caculate_align_mat (artiboost/view_engine.py:61-86): rotation taking +z onto vec."""
    vec = np.asarray(vec, dtype=np.float64)
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


def perspective_from_id(persp_id, u_off, th_off, u_bins=12, theta_bins=24):
    """ViewEngine.get_perspective_from_id (view_engine.py:35-57); u_off/th_off are the two torch.rand(1)-0.5 draws."""
    u_id = persp_id // theta_bins
    th_id = persp_id % theta_bins
    u_unit = 2 / u_bins
    th_unit = (2 * np.pi) / theta_bins
    u = np.clip((-1 + u_unit / 2) + u_id * u_unit + u_off * u_unit, -1, 1)
    th = np.clip(th_unit / 2 + th_id * th_unit + th_off * th_unit, 0, 2 * np.pi)
    s = np.sqrt(1 - u * u)
    return align_mat(np.array([s * np.cos(th), s * np.sin(th), u]))


# --------------------------------------------------------------------------- rotations (pytorch3d semantics)
def aa_to_rotmat(aa):
    """axis-angle (...,3) -> (...,3,3) via quaternion, pytorch3d.transforms.axis_angle_to_matrix semantics
    (wrapper at utils/transform.py:42-55) [third-party, restated from its published algorithm]."""
    aa = np.asarray(aa, dtype=np.float64)
    ang = np.linalg.norm(aa, axis=-1, keepdims=True)
    half = 0.5 * ang
    small = np.abs(ang) < 1e-6
    safe = np.where(small, 1.0, ang)
    k = np.where(small, 0.5 - ang * ang / 48.0, np.sin(half) / safe)
    w = np.cos(half)[..., 0]
    x, y, z = (aa * k)[..., 0], (aa * k)[..., 1], (aa * k)[..., 2]
    two_s = 2.0 / (w * w + x * x + y * y + z * z)
    R = np.stack([
        1 - two_s * (y * y + z * z), two_s * (x * y - z * w), two_s * (x * z + y * w),
        two_s * (x * y + z * w), 1 - two_s * (x * x + z * z), two_s * (y * z - x * w),
        two_s * (x * z - y * w), two_s * (y * z + x * w), 1 - two_s * (x * x + y * y)], axis=-1)
    return R.reshape(aa.shape[:-1] + (3, 3))


def rotmat_to_aa(R):
    """(...,3,3) -> axis-angle (...,3); matrix_to_quaternion + quaternion_to_axis_angle semantics
    (wrapper at utils/transform.py:291-306) [third-party, restated]."""
    R = np.asarray(R, dtype=np.float64)
    m00, m01, m02 = R[..., 0, 0], R[..., 0, 1], R[..., 0, 2]
    m10, m11, m12 = R[..., 1, 0], R[..., 1, 1], R[..., 1, 2]
    m20, m21, m22 = R[..., 2, 0], R[..., 2, 1], R[..., 2, 2]
    q_abs = np.sqrt(np.maximum(0.0, np.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22,
                                               1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)))
    cand = np.stack([
        np.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        np.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
        np.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
        np.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * np.maximum(q_abs[..., None], 0.1))
    best = np.argmax(q_abs, axis=-1)
    q = np.take_along_axis(cand, best[..., None, None], axis=-2)[..., 0, :]
    norms = np.linalg.norm(q[..., 1:], axis=-1, keepdims=True)
    half = np.arctan2(norms, q[..., :1])
    ang = 2 * half
    small = np.abs(ang) < 1e-6
    safe = np.where(small, 1.0, ang)
    k = np.where(small, 0.5 - ang * ang / 48.0, np.sin(half) / safe)
    return q[..., 1:] / k


# --------------------------------------------------------------------------- R1: MANO LBS
def make_synthetic_mano(seed=1):
    """MANO-shaped synthetic hand model: the licensed MANO_RIGHT.pkl is absent, so a seeded
    stand-in with the real topology sizes (778 verts, 1538 faces, 16 joints, real parents / tip ids / reorder).
    Returns dict of float32 arrays."""
    rng = np.random.default_rng(seed)
    # 778 points on a hand-sized blob: palm ellipsoid + 5 finger capsules
    n_palm = 778 - 5 * 100
    pts = []
    u = rng.uniform(-1, 1, (n_palm, 3))
    u /= np.maximum(np.linalg.norm(u, axis=1, keepdims=True), 1e-6)
    pts.append(u * np.array([0.045, 0.05, 0.012]) + np.array([0.0, 0.0, 0.0]))
    finger_base = np.array([[-0.04, 0.02, 0], [-0.02, 0.05, 0], [0.0, 0.055, 0], [0.02, 0.05, 0], [0.04, 0.04, 0]])
    finger_dir = np.array([[-0.7, 0.7, 0], [-0.1, 1, 0], [0, 1, 0], [0.1, 1, 0], [0.25, 0.95, 0]])
    finger_dir /= np.linalg.norm(finger_dir, axis=1, keepdims=True)
    finger_len = np.array([0.06, 0.075, 0.08, 0.075, 0.06])
    for f in range(5):
        t = rng.uniform(0, 1, (100, 1))
        ang = rng.uniform(0, 2 * np.pi, (100, 1))
        radial = 0.008 * np.concatenate([np.cos(ang), np.zeros_like(ang), np.sin(ang)], 1)
        pts.append(finger_base[f] + finger_dir[f] * t * finger_len[f] + radial)
    v_template = np.concatenate(pts, 0)
    # joints: wrist + 3 per finger (MANO order: index, middle, pinky, ring, thumb)
    order = [1, 2, 4, 3, 0]
    J = [np.array([0.0, -0.03, 0.0])]
    for f in order:
        for k in range(3):
            J.append(finger_base[f] + finger_dir[f] * finger_len[f] * (k / 3.0))
    J = np.stack(J)  # (16,3)
    d = np.linalg.norm(v_template[:, None] - J[None], axis=2)  # (778,16)
    w = np.exp(-d / 0.01)
    w /= w.sum(1, keepdims=True)
    # sparse-ish row-stochastic regressor reproducing J approximately
    Jreg = np.exp(-d.T / 0.006)
    Jreg /= Jreg.sum(1, keepdims=True)
    # faces: 1538 triangles from nearest neighbours (fixed, seeded)
    faces = []
    idx = np.argsort(np.linalg.norm(v_template[:, None] - v_template[None], axis=2), axis=1)[:, 1:4]
    for i in range(778):
        faces.append([i, idx[i, 0], idx[i, 1]])
        if len(faces) < 1538:
            faces.append([i, idx[i, 1], idx[i, 2]])
    faces = np.array(faces[:1538], dtype=np.int32)
    # put tip vertex ids at the finger ends so "tips" make geometric sense
    for f, tid in zip(order, MANO_TIPS):
        v_template[tid] = finger_base[f] + finger_dir[f] * finger_len[f] * 1.02
    return {
        "v_template": v_template.astype(np.float32),
        "shapedirs": (1e-3 * rng.standard_normal((778, 3, 10))).astype(np.float32),
        "posedirs": (1e-4 * rng.standard_normal((778, 3, 135))).astype(np.float32),
        "J_regressor": Jreg.astype(np.float32),
        "weights": w.astype(np.float32),
        "faces": faces,
        "hands_mean": np.zeros(45, dtype=np.float32),
    }


def rodrigues(aa):
    """manolayer.py:162-172 (_batch_rodrigues via quaternion; note the +1e-8 inside the norm)."""
    aa = np.asarray(aa, dtype=np.float64)
    n = np.linalg.norm(aa + 1e-8, axis=-1, keepdims=True)
    ax = aa / n
    h = n * 0.5
    w, s = np.cos(h)[..., 0], np.sin(h)
    x, y, z = (s * ax)[..., 0], (s * ax)[..., 1], (s * ax)[..., 2]
    # manolayer.py:135-160 (_quat2mat): normalise then expand
    nq = np.sqrt(w * w + x * x + y * y + z * z)
    w, x, y, z = w / nq, x / nq, y / nq, z / nq
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    R = np.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                  2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                  2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], -1)
    return R.reshape(aa.shape[:-1] + (3, 3))


def mano_lbs(model, pose, betas, dtype=np.float64):
    """MANO forward (manolayer.py:182-276, center_idx=None as at the artiboost call sites, flat_hand_mean=True,
    axis-angle input).  pose (B,48), betas (B,10) -> verts (B,778,3), joints (B,21,3), T_abs (B,16,4,4)."""
    pose = np.asarray(pose, dtype=dtype)
    betas = np.asarray(betas, dtype=dtype)
    B = pose.shape[0]
    vt = model["v_template"].astype(dtype)
    full = pose.copy()
    full[:, 3:] += model["hands_mean"].astype(dtype)
    R = rodrigues(full.reshape(B, 16, 3)).astype(dtype)                      # (B,16,3,3)
    pose_map = (R[:, 1:] - np.eye(3, dtype=dtype)).reshape(B, 135)          # manolayer.py:120-131
    v_shaped = vt[None] + np.einsum("vkl,bl->bvk", model["shapedirs"].astype(dtype), betas)
    J = np.einsum("jv,bvk->bjk", model["J_regressor"].astype(dtype), v_shaped)
    v_posed = v_shaped + np.einsum("vkp,bp->bvk", model["posedirs"].astype(dtype), pose_map)
    G = np.zeros((B, 16, 4, 4), dtype=dtype)
    for j in range(16):
        L = np.zeros((B, 4, 4), dtype=dtype)
        L[:, :3, :3] = R[:, j]
        L[:, 3, 3] = 1
        par = MANO_PARENTS[j]
        if par < 0:
            L[:, :3, 3] = J[:, 0]
            G[:, j] = L
        else:
            L[:, :3, 3] = J[:, j] - J[:, par]
            G[:, j] = G[:, par] @ L
    T_abs = G.copy()
    Jh = np.concatenate([J, np.zeros((B, 16, 1), dtype=dtype)], 2)
    corr = np.einsum("bjmn,bjn->bjm", G, Jh)                                # manolayer.py:252-253
    G2 = G.copy()
    G2[:, :, :, 3] -= corr
    T = np.einsum("vj,bjmn->bvmn", model["weights"].astype(dtype), G2)      # (B,778,4,4)
    vh = np.concatenate([v_posed, np.ones((B, 778, 1), dtype=dtype)], 2)
    verts = np.einsum("bvmn,bvn->bvm", T, vh)[:, :, :3]
    jtr = np.concatenate([T_abs[:, :, :3, 3], verts[:, MANO_TIPS]], 1)[:, MANO_JOINT_REORDER]
    return verts, jtr, T_abs


# --------------------------------------------------------------------------- R2: pose generator glue
def scramble(hand_pose, hand_tsl, rand_pose_angle=None, rand_tsl=None):
    """RandomScrambler.forward (artiboost/scrambler.py:65-81), draws passed in.  PINNED: tests/golden/scrambler.npz is the
    reference class run in the build container (oracle/gen_scrambler_golden.py)."""
    if rand_pose_angle is not None:
        B = hand_pose.shape[0]
        hp = hand_pose.reshape(B, 16, 3)
        nrm = np.linalg.norm(hp, axis=-1, keepdims=True)
        axis = hp / np.clip(nrm, 1e-7, None)
        hand_pose = (axis * (nrm[..., 0] + rand_pose_angle)[..., None]).reshape(B, 48)
    if rand_tsl is not None:
        hand_tsl = hand_tsl + rand_tsl
    return hand_pose, hand_tsl


def pose_generator(model, hand_pose, hand_shape, hand_tsl, persp_rotmat, camera_free_transf, z_offset,
                   rand_pose_angle=None, rand_tsl=None):
    """PreProcessorPoseGenerator.forward (artiboost/preprocessor.py:20-99) + RandomScrambler
    (artiboost/scrambler.py:65-81), WITHOUT the GrabNet refiner ('next' row; weights absent), i.e.
    refiner == identity re-decode of the scrambled pose.  All inputs batched numpy; draws passed in."""
    B = hand_pose.shape[0]
    verts, joints, T = mano_lbs(model, hand_pose, hand_shape)
    hand_verts = verts + hand_tsl[:, None]
    joints = joints + hand_tsl[:, None]
    Rinv = np.transpose(persp_rotmat, (0, 2, 1))
    op_offset = np.einsum("bij,bj->bi", Rinv, joints[:, 9]) / 2.0
    cam_sys_offset = z_offset - op_offset
    obj_pose = np.tile(np.eye(4), (B, 1, 1))
    obj_pose[:, :3, :3] = Rinv
    obj_pose[:, :3, 3] = cam_sys_offset
    obj_pose = camera_free_transf @ obj_pose
    glob = Rinv @ T[:, 0, :3, :3]
    new_glob = rotmat_to_aa(glob)
    new_pose = np.concatenate([new_glob, hand_pose[:, 3:]], 1)
    # rotation centre = root joint of the shaped template (manotorch get_rotation_center) [third-party, restated]
    vt = model["v_template"].astype(np.float64)[None] + np.einsum("vkl,bl->bvk", model["shapedirs"].astype(np.float64), hand_shape)
    center = np.einsum("v,bvk->bk", model["J_regressor"][0].astype(np.float64), vt)
    root_rot = aa_to_rotmat(hand_pose[:, :3])
    off0 = center - np.einsum("bij,bj->bi", root_rot, center)
    new_root = aa_to_rotmat(new_pose[:, :3])
    off1 = center - np.einsum("bij,bj->bi", new_root, center)
    new_tsl = np.einsum("bij,bj->bi", Rinv, off0 + hand_tsl) - off1
    new_pose, new_tsl = scramble(new_pose, new_tsl, rand_pose_angle, rand_tsl)
    v2, j2, _ = mano_lbs(model, new_pose, hand_shape)
    v2 = v2 + new_tsl[:, None] + cam_sys_offset[:, None]
    j2 = j2 + new_tsl[:, None] + cam_sys_offset[:, None]
    Rf = camera_free_transf[:, :3, :3]
    final_verts = np.einsum("bij,bvj->bvi", Rf, v2)
    final_joints = np.einsum("bij,bvj->bvi", Rf, j2)
    return obj_pose, final_verts, final_joints


# --------------------------------------------------------------------------- R4: GT assembly (no pixels)
def assemble_sample_gt(K, joints, obj_pose, corners_can, image_size, draws, center_idx=0, bbox_expand=1.2,
                       center_jit=0.1, scale_jit=0.1, raw_size=None):
    """RenderedDataset.__getitem__ geometry (artiboost/rendered_dataset.py:127-133,155-254), CROP_MODEL root_obj.
    draws = dict(center=(2,) in [-1,1], scale=N(0, scale_jit/3) sample, rot=radians)."""
    raw_size = raw_size or image_size
    j2d = (K @ joints.T).T
    j2d = j2d[:, :2] / (j2d[:, 2:3] + 1e-8)
    c3d = (obj_pose[:3, :3] @ corners_can.T).T + obj_pose[:3, 3]
    c2d = (K @ c3d.T).T
    c2d = c2d[:, :2] / (c2d[:, 2:3] + 1e-8)
    all2d = np.concatenate([j2d[[0]], c2d], 0)
    center = annot_center(all2d)
    scale = annot_scale(all2d) * bbox_expand
    center = center + (center_jit * scale * np.asarray(draws["center"])).astype(int)
    scale = scale * np.clip(draws["scale"] + 1.0, 1 - scale_jit, 1 + scale_jit)
    rot = draws["rot"]
    rm = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1]]).astype(np.float32)
    aff, post = get_affine_transform(center, scale, [K[0, 2], K[1, 2]], image_size, rot)
    out = {"affine": aff, "cam_intr": post.dot(K).astype(np.float32)}
    j3 = rm.dot(joints.astype(np.float32).T).T
    root = j3[center_idx]
    out["root_joint"] = root
    out["joints_3d"] = j3 - root
    j2a = transform_coords(j2d.astype(np.float32), aff).astype(np.float32)
    out["joints_2d"] = j2a

    def vis(raw2d, aug2d, n):
        v = ((raw2d[:, 0] >= 0) & (raw2d[:, 0] < raw_size[0]) & (raw2d[:, 1] >= 0) & (raw2d[:, 1] < raw_size[1]))
        if v.sum() < n * 0.4:
            return np.zeros(n, np.float32)
        va = ((aug2d[:, 0] >= 0) & (aug2d[:, 0] < image_size[0]) & (aug2d[:, 1] >= 0) &
              (aug2d[:, 1] < image_size[1])).astype(np.float32)
        return np.zeros(n, np.float32) if va.sum() < n * 0.4 else va

    out["joints_vis"] = vis(j2d, j2a, 21)
    c3 = rm.dot(c3d.astype(np.float32).T).T
    out["corners_3d"] = c3 - root
    c2a = transform_coords(c2d.astype(np.float32), aff)
    out["corners_2d"] = c2a
    out["corners_vis"] = vis(c2d, c2a, 8)
    out["corners_can"] = corners_can.astype(np.float32)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = rm @ obj_pose[:3, :3].astype(np.float32)
    T[:3, 3] = rm.dot(obj_pose[:3, 3].astype(np.float32))
    out["obj_transf"] = T
    return out

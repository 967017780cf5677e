"""Seeded synthetic stand-ins for the assets the reference downloads (docs/Installation.md:71-180) and that cannot
travel to the GPU box: MANO_RIGHT.pkl, the 51 HTML hand textures, YCB object meshes (HO3D `ds_textured`, DexYCB
`textured_simple`), grasp pickles and background images.  Shapes / counts follow SURVEY.md section 8d; everything is
generated from numpy Generators so the build container and the GPU box produce identical bytes."""
import numpy as np

MANO_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
MANO_TIPS = [745, 317, 444, 556, 673]
MANO_JOINT_REORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]


def _grid_faces(nu, nv, wrap_u=True):
    f = []
    for j in range(nv - 1):
        for i in range(nu if wrap_u else nu - 1):
            a = j * nu + i
            b = j * nu + (i + 1) % nu
            c = (j + 1) * nu + i
            d = (j + 1) * nu + (i + 1) % nu
            f.append((a, c, b))
            f.append((b, c, d))
    return np.asarray(f, dtype=np.int32)


def _vertex_normals(v, f):
    n = np.zeros_like(v)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    for k in range(3):
        np.add.at(n, f[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return n / np.maximum(ln, 1e-12)


def _smooth_noise(rng, h, w, c=3, octaves=4):
    img = np.zeros((h, w, c))
    for o in range(octaves):
        s = 2 ** (o + 2)
        g = rng.uniform(0, 1, (s + 1, s + 1, c))
        ys = np.linspace(0, s, h, endpoint=False)
        xs = np.linspace(0, s, w, endpoint=False)
        y0, x0 = ys.astype(int), xs.astype(int)
        fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
        a = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
        b = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
        img += (a * (1 - fy) + b * fy) / (2 ** o)
    img /= img.max()
    return img


def make_hand_model(seed=1):
    """MANO-shaped hand: 778 verts, 1538 faces, 16 joints with the real parents / fingertip ids / joint reorder.
    Geometry: 5 finger tubes (5 rings x 8... ) + palm grid stitched as one closed-ish surface, hand sized."""
    rng = np.random.default_rng(seed)
    # vertices: palm = 13 x 22 grid on a flattened ellipsoid (286), fingers = 5 x (12 rings x 8) = 480, + 12 spare = 778
    verts, faces = [], []
    nu, nv = 22, 13
    th = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    ph = np.linspace(0.15 * np.pi, 0.85 * np.pi, nv)
    pv = np.stack([0.045 * np.outer(np.sin(ph), np.cos(th)), 0.05 * np.outer(np.cos(ph), np.ones(nu)) * -1.0,
                   0.014 * np.outer(np.sin(ph), np.sin(th))], -1).reshape(-1, 3)
    verts.append(pv)
    faces.append(_grid_faces(nu, nv))
    base = np.array([[-0.035, 0.03, 0], [-0.017, 0.048, 0], [0.0, 0.052, 0], [0.017, 0.048, 0], [0.036, 0.02, 0]])
    direc = np.array([[-0.2, 1, 0], [-0.05, 1, 0], [0, 1, 0], [0.06, 1, 0], [0.75, 0.66, 0]])
    direc /= np.linalg.norm(direc, axis=1, keepdims=True)
    length = np.array([0.07, 0.08, 0.085, 0.075, 0.06])
    off = len(pv)
    tips = []
    for f in range(5):
        ru, rv = 8, 12
        t = np.linspace(0, 1, rv)
        ang = np.linspace(0, 2 * np.pi, ru, endpoint=False)
        side = np.cross(direc[f], [0, 0, 1.0])
        rad = 0.009 * (1 - 0.35 * t)
        ring = (base[f][None, None] + direc[f][None, None] * (t * length[f])[:, None, None]
                + rad[:, None, None] * (np.cos(ang)[None, :, None] * side[None, None] + np.sin(ang)[None, :, None] * np.array([0, 0, 1.0])[None, None]))
        verts.append(ring.reshape(-1, 3))
        faces.append(_grid_faces(ru, rv) + off)
        tips.append(off + (rv - 1) * ru)
        off += ru * rv
    spare = 778 - off
    verts.append(np.array([0.0, -0.04, 0.0])[None] + 0.004 * rng.standard_normal((spare, 3)))
    v_template = np.concatenate(verts, 0)
    fcs = np.concatenate(faces, 0)
    # exactly 1538 faces: pad with degenerate fan triangles over the spare vertices / trim
    extra = []
    k = 0
    while len(fcs) + len(extra) < 1538:
        a = off + (k % spare)
        extra.append((a, off + ((k + 1) % spare), off + ((k + 2) % spare)))
        k += 1
    fcs = np.concatenate([fcs, np.asarray(extra, dtype=np.int32).reshape(-1, 3)], 0)[:1538]
    # joints: wrist + (index, middle, pinky, ring, thumb) x 3   [MANO order]
    order = [1, 2, 4, 3, 0]
    J = [np.array([0.0, -0.045, 0.0])]
    for f in order:
        for kk in range(3):
            J.append(base[f] + direc[f] * length[f] * (kk / 3.0))
    J = np.stack(J)
    # move the designated MANO tip ids onto the finger ends (swap vertex positions so tips are meaningful)
    perm = np.arange(778)
    # geometric fingers f=0..4 are (index-ish left to right); map: f=4 thumb, f=0 index, f=1 middle, f=2 ring, f=3 pinky
    geo2tip = {4: MANO_TIPS[0], 0: MANO_TIPS[1], 1: MANO_TIPS[2], 2: MANO_TIPS[3], 3: MANO_TIPS[4]}
    for f, cur in enumerate(tips):
        want = geo2tip[f]
        i, j = int(np.where(perm == cur)[0][0]), int(np.where(perm == want)[0][0])
        perm[i], perm[j] = perm[j], perm[i]
    inv = np.empty(778, dtype=np.int64)
    inv[perm] = np.arange(778)
    v_template = v_template[inv.argsort()]
    remap = inv.argsort().argsort()
    fcs = remap[fcs].astype(np.int32)
    d = np.linalg.norm(v_template[:, None] - J[None], axis=2)
    w = np.exp(-d / 0.012)
    w[w < 1e-3 * w.max(1, keepdims=True)] = 0
    w /= w.sum(1, keepdims=True)
    Jreg = np.exp(-d.T / 0.006)
    Jreg[Jreg < 1e-2 * Jreg.max(1, keepdims=True)] = 0
    Jreg /= Jreg.sum(1, keepdims=True)
    uv = np.stack([(np.arctan2(v_template[:, 2], v_template[:, 0]) / (2 * np.pi)) % 1.0,
                   np.clip((v_template[:, 1] + 0.06) / 0.2, 0, 0.999)], 1)
    # UV seam: a triangle that straddles the u = 0/1 wrap would interpolate across the whole texture.  As in a textured
    # mesh loaded through trimesh (FMesh.from_trimesh; renderer.py:17-28 get_mapping), the vertices on the low side of such
    # triangles are DUPLICATED with u + 1: render vertices V_dup = 778 + duplicates, `map` takes a render vertex to the
    # MANO vertex whose position it shares; uv / normals are per render vertex, faces index render vertices.
    render_faces = fcs.copy()
    vmap, dup_of = list(range(778)), {}
    uv_list, nrm = [tuple(x) for x in uv], _vertex_normals(v_template, fcs)
    for fi, face in enumerate(fcs):
        us = uv[face, 0]
        if us.max() - us.min() > 0.5:
            for k, vtx in enumerate(face):
                if uv[vtx, 0] < 0.5:
                    if vtx not in dup_of:
                        dup_of[vtx] = len(vmap)
                        vmap.append(int(vtx))
                        uv_list.append((uv[vtx, 0] + 1.0, uv[vtx, 1]))
                    render_faces[fi, k] = dup_of[vtx]
    vmap = np.asarray(vmap, np.int32)
    uv = np.asarray(uv_list)
    fcs_render = render_faces.astype(np.int32)
    return {
        "v_template": v_template.astype(np.float32),
        "shapedirs": (1e-3 * rng.standard_normal((778, 3, 10))).astype(np.float32),
        "posedirs": (1e-4 * rng.standard_normal((778, 3, 135))).astype(np.float32),
        "J_regressor": Jreg.astype(np.float32),
        "weights": w.astype(np.float32),
        "faces": fcs_render,                                               # [1538,3] indices into the V_dup render vertices
        "map": vmap,                                                        # [V_dup] render vertex -> MANO vertex (hand_mapping)
        "hands_mean": np.zeros(45, dtype=np.float32),
        "uv": uv.astype(np.float32),                                        # [V_dup,2]
        "normals": nrm[vmap].astype(np.float32),    # [V_dup,3] rest-pose normals (frender_utils.py:139)
    }


def make_hand_textures(n=51, size=128, seed=2):
    rng = np.random.default_rng(seed)
    out = np.zeros((n, size, size, 3), dtype=np.uint8)
    for i in range(n):
        tone = np.array([0.55 + 0.4 * rng.uniform(), 0.4 + 0.35 * rng.uniform(), 0.3 + 0.3 * rng.uniform()])
        out[i] = np.clip(255 * (tone[None, None] * (0.7 + 0.3 * _smooth_noise(rng, size, size, 1))), 0, 255).astype(np.uint8)
    return out


def make_object(idx, nverts_target, seed=3):
    """Surface of revolution with a per-object profile; ~nverts_target vertices, consistent outward winding,
    bbox-centred (object_engine.py:52,81); returns dict(verts, faces, normals, uv, corners (8,3))."""
    rng = np.random.default_rng(seed * 1000 + idx)
    nu = 128 if nverts_target >= 8192 else 64
    nv = nverts_target // nu
    t = np.linspace(0, 1, nv)
    h = 0.08 + 0.17 * rng.uniform()
    r0 = 0.025 + 0.03 * rng.uniform()
    prof = r0 * (1 + 0.35 * np.sin(2 * np.pi * (t * (1 + idx % 3) + rng.uniform())) * rng.uniform(0.2, 1.0))
    prof *= np.clip(np.minimum(t, 1 - t) * 12, 0.05, 1.0)           # close the ends
    sq = 0.6 * rng.uniform() if idx % 2 else 0.0                    # boxy cross-section for some
    th = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    rr = 1.0 / np.maximum(np.abs(np.cos(th)) ** (1 + 4 * sq) + np.abs(np.sin(th)) ** (1 + 4 * sq), 1e-6) ** (1.0 / (1 + 4 * sq))
    x = prof[:, None] * rr[None] * np.cos(th)[None]
    z = prof[:, None] * rr[None] * np.sin(th)[None] * (0.6 + 0.4 * rng.uniform())
    y = (t[:, None] - 0.5) * h * np.ones((1, nu))
    v = np.stack([x, y, z], -1).reshape(-1, 3)
    f = _grid_faces(nu, nv)
    f = f[:, [0, 2, 1]]                                             # outward orientation for this parametrisation
    v = v - (v.min(0) + v.max(0)) / 2
    mn, mx = v.min(0), v.max(0)
    corners = np.array([[sx, sy, sz] for sx in (mn[0], mx[0]) for sy in (mn[1], mx[1]) for sz in (mn[2], mx[2])])
    uv = np.stack([np.tile(np.arange(nu) / nu, nv), np.repeat(t * 0.999, nu)], 1)
    return {"verts": v.astype(np.float32), "faces": f.astype(np.int32), "normals": _vertex_normals(v, f).astype(np.float32),
            "uv": uv.astype(np.float32), "corners": corners.astype(np.float32)}


def make_object_texture(idx, size=256, seed=4):
    rng = np.random.default_rng(seed * 1000 + idx)
    base = rng.uniform(0.2, 0.9, 3)
    img = base[None, None] * (0.5 + 0.5 * _smooth_noise(rng, size, size, 3))
    stripes = (np.sin(np.linspace(0, 2 * np.pi * (3 + idx % 5), size)) > 0.3).astype(float)
    img = img * (0.75 + 0.25 * stripes[None, :, None])
    return np.clip(255 * img, 0, 255).astype(np.uint8)


def make_backgrounds(n=16, size=768, seed=5):
    rng = np.random.default_rng(seed)
    out = np.zeros((n, size, size, 3), dtype=np.uint8)
    for i in range(n):
        small = _smooth_noise(rng, 192, 192, 3, octaves=5)
        out[i] = np.clip(255 * np.kron(small, np.ones((4, 4, 1))), 0, 255).astype(np.uint8)
    return out


def make_grasps(n_obj, n_grasp=50, seed=6):
    """(pose48 ~ N(0,0.3) clipped +-1.2, shape 0, tsl ~ U(-0.05,0.05)^3) per (object, grasp) -- SURVEY.md 8d."""
    rng = np.random.default_rng(seed)
    pose = np.clip(0.3 * rng.standard_normal((n_obj, n_grasp, 48)), -1.2, 1.2)
    tsl = rng.uniform(-0.05, 0.05, (n_obj, n_grasp, 3))
    return pose.astype(np.float32), np.zeros((n_obj, n_grasp, 10), np.float32), tsl.astype(np.float32)


class SceneAssets:
    """Everything the renderer needs, as packed numpy arrays (one global vertex / face table for all meshes)."""

    def __init__(self, dataset="HO3D", seed=1, tex_size=256):
        self.hand = make_hand_model(seed)
        self.hand_tex = make_hand_textures(51, 128, seed + 1)
        n_obj, nv = (4, 2048) if dataset == "HO3D" else (21, 8192)
        self.objects = [make_object(i, nv, seed + 2) for i in range(n_obj)]
        self.obj_tex = np.stack([make_object_texture(i, tex_size, seed + 3) for i in range(n_obj)])
        self.backgrounds = make_backgrounds(16, 768, seed + 4)
        self.n_obj = n_obj
        # HO3D cfg objects -> YCB class ids (utils/misc.py:97-119): potted_meat_can 9, bleach 12, mustard 5, pitcher 11
        self.obj_idx = np.array([9, 12, 5, 11] if dataset == "HO3D" else list(range(1, 22)), dtype=np.int64)
        self.corners_can = np.stack([o["corners"] for o in self.objects])
        # packed object tables
        vo, fo = [0], [0]
        for o in self.objects:
            vo.append(vo[-1] + len(o["verts"]))
            fo.append(fo[-1] + len(o["faces"]))
        self.obj_vert_off = np.asarray(vo, dtype=np.int32)
        self.obj_face_off = np.asarray(fo, dtype=np.int32)
        self.obj_verts = np.concatenate([o["verts"] for o in self.objects])
        self.obj_normals = np.concatenate([o["normals"] for o in self.objects])
        self.obj_uv = np.concatenate([o["uv"] for o in self.objects])
        self.obj_faces = np.concatenate([o["faces"] for o in self.objects])     # indices local to the object
        self.max_obj_faces = max(len(o["faces"]) for o in self.objects)
        self.max_obj_verts = max(len(o["verts"]) for o in self.objects)


def _subdivide(verts, faces):
    """One midpoint subdivision (each triangle -> 4, one new vertex per unique edge), as trimesh's Trimesh.subdivide()."""
    e = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]), axis=1)
    uniq, inv = np.unique(e, axis=0, return_inverse=True)
    mid = 0.5 * (verts[uniq[:, 0]] + verts[uniq[:, 1]])
    nf = len(faces)
    m01, m12, m20 = (len(verts) + inv.reshape(-1)[k * nf:(k + 1) * nf] for k in range(3))
    f = np.concatenate([np.stack([faces[:, 0], m01, m20], 1), np.stack([m01, faces[:, 1], m12], 1),
                        np.stack([m20, m12, faces[:, 2]], 1), np.stack([m01, m12, m20], 1)])
    return np.concatenate([verts, mid]), f


def resample_objects(assets, n_points=10000, seed=7):
    """HORefiner.resample_obj (anakin/artiboost/refiner.py:169-179) for every object: subdivide until the mesh has at
    least n_points vertices, then keep n_points of them drawn without replacement.  -> float32 [n_obj, n_points, 3]."""
    rng = np.random.default_rng(seed)
    out = []
    for o in assets.objects:
        v, f = np.asarray(o["verts"], np.float64), np.asarray(o["faces"], np.int64)
        while len(v) < n_points:
            v, f = _subdivide(v, f)
        out.append(v[rng.choice(len(v), n_points, replace=False)])
    return np.stack(out).astype(np.float32)

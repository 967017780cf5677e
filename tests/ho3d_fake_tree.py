"""TEST INFRASTRUCTURE -- writes a miniature directory tree with the layout of the HO3D v2 download (README.md:73-91 of the reference;
anakin/datasets/ho3d.py:55-190): HO3D/{train,evaluation}.txt, HO3D/<split>/<seq>/{rgb/NNNN.png, meta/NNNN.pkl}, YCB_models_supp/<obj>/
textured_simple_ds.obj, YCB_models_process/<obj>/ds_textured.obj.  Seeded: oracle/gen_ho3d_reader_golden.py (the reference's reader on this
tree) and tests/test_ho3d_reader.py (this build's reader on the same tree) both call build()."""
import os
import pickle

import numpy as np

OBJS = ["006_mustard_bottle", "010_potted_meat_can", "021_bleach_cleanser"]
TRAIN = [("ABF10", 4, OBJS[2]), ("GPMF12", 3, OBJS[1]), ("SM2", 2, OBJS[0])]
TEST = [("SM1", 3, OBJS[0]), ("MPM10", 2, OBJS[1])]


def _rodrigues_rot(r):
    th = np.linalg.norm(r)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _obj_verts(rng, n=200):
    ext = rng.uniform(0.03, 0.09, 3)
    return (rng.uniform(-1, 1, (n, 3)) * ext + rng.uniform(-0.01, 0.01, 3)).astype(np.float64)


def _write_obj(path, verts, faces):
    with open(path, "w") as f:
        f.write("# stand-in mesh\nmtllib x.mtl\n")
        for v in verts:
            f.write("v %.8f %.8f %.8f\n" % tuple(v))
        for v in verts[:5]:
            f.write("vt %.4f %.4f\n" % (abs(v[0]) % 1, abs(v[1]) % 1))
        for a, b, c in faces:
            f.write("f %d/1 %d/2 %d/3\n" % (a + 1, b + 1, c + 1))


def build(root, seed=7, size=(640, 480), version=2):
    """-> dict(train=[(seq, frame)], test=[...]) of what was written under `root` (a fresh directory).  version=3: the layout of the v3
    download (DATA_ROOT/HO3D_v3, frames as .jpg: ho3d.py:573-596)."""
    name = "HO3D" if version == 2 else "HO3D_v3"
    from PIL import Image
    rng = np.random.default_rng(seed)
    W, H = size
    verts = {}
    for o in OBJS:
        verts[o] = _obj_verts(rng)
        faces = rng.integers(0, len(verts[o]), (50, 3))
        for sub, fn in (("YCB_models_supp", "textured_simple_ds.obj"), ("YCB_models_process", "ds_textured.obj")):
            os.makedirs(os.path.join(root, sub, o), exist_ok=True)
            _write_obj(os.path.join(root, sub, o, fn), verts[o], faces)
    out = {}
    yy, xx = np.mgrid[0:H, 0:W]
    for split, sub, seqs, listing in (("train", "train", TRAIN, "train.txt"), ("test", "evaluation", TEST, "evaluation.txt")):
        lines = []
        for seq, nfr, obj in seqs:
            rgb, meta = os.path.join(root, name, sub, seq, "rgb"), os.path.join(root, name, sub, seq, "meta")
            os.makedirs(rgb, exist_ok=True)
            os.makedirs(meta, exist_ok=True)
            v = verts[obj]
            corners = np.array([[sx, sy, sz] for sx in (v[:, 0].min(), v[:, 0].max()) for sy in (v[:, 1].min(), v[:, 1].max())
                                for sz in (v[:, 2].min(), v[:, 2].max())], np.float32)
            for fi in range(nfr):
                frame = f"{fi:04d}"
                img = np.stack([(xx * (2 + c) + yy * 3 + 50 * np.sin(xx / (7.0 + fi)) + 20 * rng.standard_normal((H, W))) % 256 for c in range(3)], -1)
                if version == 2:
                    Image.fromarray(img.astype(np.uint8)).save(os.path.join(rgb, frame + ".png"), compress_level=int(rng.integers(1, 7)))
                else:
                    Image.fromarray(img.astype(np.uint8)).save(os.path.join(rgb, frame + ".jpg"), quality=int(rng.integers(80, 96)), subsampling=2)
                K = np.array([[614.6 + fi, 0, 320.3], [0, 614.2, 239.7 - fi], [0, 0, 1.0]], np.float32)
                root_j = np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), -rng.uniform(0.4, 0.6)], np.float32)   # HO3D's OpenGL frame: -z forward
                ann = {"camMat": K, "objRot": rng.normal(0, 1.0, (3, 1)).astype(np.float32), "objName": obj,
                       "objTrans": (root_j + rng.uniform(-0.03, 0.03, 3)).astype(np.float32), "objCorners3DRest": corners,
                       "objCorners3D": ((_rodrigues_rot(np.ones(3)) @ corners.T).T).astype(np.float32), "objLabel": 1}
                if split == "train":
                    ann.update(handJoints3D=(root_j + rng.uniform(-0.07, 0.07, (21, 3))).astype(np.float32),
                               handPose=rng.normal(0, 0.3, 48).astype(np.float32), handTrans=root_j.copy(),
                               handBeta=rng.normal(0, 1, 10).astype(np.float32))
                else:
                    ann.update(handJoints3D=root_j.copy(), handBoundingBox=[int(v) for v in rng.integers(100, 400, 4)])
                with open(os.path.join(meta, frame + ".pkl"), "wb") as f:
                    pickle.dump(ann, f, protocol=2)
                lines.append(f"{seq}/{frame}")
        with open(os.path.join(root, name, listing), "w") as f:
            f.write("\n".join(lines) + "\n")
        out[split] = [tuple(l.split("/")) for l in lines]
    return out

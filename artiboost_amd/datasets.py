"""DATASET registry entries and the collate function the reference's train script names
(anakin/datasets/{hodata,ho3d,dexycb}.py; builder.build_dataset, train/train_artiboost.py:114-125,172).

HO3D / DexYCB are downloads (README.md:73-91 of the reference) and are not in this build's environment: the entries below
resolve the reference's YAML (`TYPE: HO3D`, `DATA_ROOT: ./data`, ...) to an EMPTY real set when the data root is absent --
ArtiBoostLoader then trains on the synthetic share alone.  When `DATA_ROOT/HO3D` exists, `HO3D` reads it (round 4: the v2 download's
own layout, ho3d.py:55-190) and serves the getters of `realdata.HOdataSource`, the .png frames as FILE BYTES for the device decode path;
HO3D v3 is the same reader on .jpg frames; DexYCB still raises if its root is given (its reader is not part of this build), so a misconfigured path is not
silently ignored."""
import hashlib
import json
import os
import pickle

import numpy as np
import torch

from .realdata import HOdataSource, annot_center_scale
from .registry import CONST, DATASET


class _DownloadedSet(HOdataSource):
    """Common shell of the real datasets: name, split, augmentation block, and the annotation index when it exists."""
    name = "hodata"
    subdir = ""

    def __init__(self, **cfg):
        self.cfg = cfg
        self.data_split = cfg.get("DATA_SPLIT", "train")
        self.data_root = cfg.get("DATA_ROOT", "./data")
        self.aug = bool(cfg.get("AUG", False))
        self.aug_param = cfg.get("AUG_PARAM") or None
        self.preset = cfg.get("DATA_PRESET", {})
        root = os.path.join(self.data_root, self.subdir)
        self.available = os.path.isdir(root)
        if self.available:
            self._load(root)

    def _load(self, root):
        raise NotImplementedError(f"{self.name}: found {root}, but the annotation reader of this dataset is not part of this "
                                  f"build (SURVEY.md section 8f-3: provide a realdata.HOdataSource over it)")

    def __len__(self):
        return 0


def _rodrigues(r):
    """cv2.Rodrigues(rvec)[0] (ho3d.py:398,414): axis-angle -> rotation matrix, float64."""
    r = np.asarray(r, np.float64).reshape(3)
    th = float(np.linalg.norm(r))
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * K


def _obj_vertices(path):
    """The `v x y z` lines of a Wavefront .obj in file order (trimesh.load(path, process=False).vertices, ho3dutils.py:21-32)."""
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                out.append([float(t) for t in line.split()[1:4]])
    return np.asarray(out, np.float32)


@DATASET.register_module
class HO3D(_DownloadedSet):
    """HO3D v2 from its download (anakin/datasets/ho3d.py:28-560, SPLIT_MODE "paper": train.txt / evaluation.txt): per frame the
    meta/NNNN.pkl annotation (camMat, handJoints3D, objRot, objTrans, objName, objCorners3DRest[, handBoundingBox]) and rgb/NNNN.png.
    Geometry getters restate ho3d.py in its arithmetic (OpenGL -> OpenCV camera flip `cam_extr`, joint re-ordering, the object transform
    w.r.t. the bbox-centred canonical mesh: :405-438, canonical corners :476-485); `get_annots` packs what HOdata.__getitem__ reads.
    The annotation index is cached next to the reference's own cache (common/cache/HO3D/<md5 of the same identifier>.ab.pkl, own format:
    arrays only).  Not built: SPLIT_MODE v1 / v2 (hard-coded sequence lists of ho3dutils), FILTER_NO_CONTACT (needs MANO_RIGHT.pkl)."""
    name, subdir, ext = "HO3D", "HO3D", ".png"
    raw_size = (640, 480)
    REORDER = np.array([0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20])      # ho3d.py:41
    CAM_EXTR = np.diag([1.0, -1.0, -1.0]).astype(np.float32)                                            # ho3d.py:44-49 (rotation part)

    def _load(self, root):
        cfg, preset = self.cfg, self.preset
        self.root = root
        self.split_mode = cfg.get("SPLIT_MODE", "paper")
        if self.split_mode != "paper":
            raise NotImplementedError(f"HO3D SPLIT_MODE {self.split_mode!r}: only the official paper split (train.txt / evaluation.txt) is read")
        if bool(preset.get("FILTER_NO_CONTACT", False)):
            raise NotImplementedError("HO3D FILTER_NO_CONTACT: needs the MANO hand model (ho3dutils.py:116-150); not built")
        self.crop_model = preset.get("CROP_MODEL", "hand_obj")
        self.full_image = bool(preset.get("FULL_IMAGE", False))
        self.mini_factor = float(cfg.get("MINI_FACTOR", 1.0))
        if self.data_split in ("train", "trainval", "val"):
            listing, self.subfolder = "train.txt", "train"
        elif self.data_split == "test":
            listing, self.subfolder = "evaluation.txt", "evaluation"
        else:
            raise ValueError(f"HO3D DATA_SPLIT {self.data_split!r}")
        with open(os.path.join(root, listing)) as f:
            seq_frames = [ln.strip().split("/") for ln in f if ln.strip()]
        if self.data_split == "trainval":                              # ho3d.py:142-145
            seq_frames = seq_frames[:6000]
        elif self.data_split == "val":
            seq_frames = seq_frames[6000:]
        supp = os.path.join(self.data_root, "YCB_models_supp")
        self.obj_verts = {o: _obj_vertices(os.path.join(supp, o, "textured_simple_ds.obj")) for o in sorted(os.listdir(supp))
                          if ".tgz" not in o and os.path.isdir(os.path.join(supp, o))}
        ident = json.dumps({"filter_thresh": float(preset.get("FILTER_THRESH", 0.0)), "data_split": self.data_split, "split_mode": self.split_mode,
                            "fliter_no_contact": False}, sort_keys=True)                                # the reference's cache identifier (:62-69)
        cache = os.path.join("common", "cache", self.name, hashlib.md5(ident.encode("ascii")).hexdigest() + ".ab.pkl")
        ann = None
        if bool(preset.get("USE_CACHE", True)) and os.path.exists(cache):
            with open(cache, "rb") as f:
                ann = pickle.load(f)
            if ann.get("frames") != [tuple(sf) for sf in seq_frames]:
                ann = None
        if ann is None:
            ann = self._read_annotations(seq_frames)
            if bool(preset.get("USE_CACHE", True)):
                # every rank may get here: write to a private temporary and rename, so that a rank starting later never unpickles a
                # half-written file (os.replace is atomic; the ranks write identical bytes)
                os.makedirs(os.path.dirname(cache), exist_ok=True)
                tmp = f"{cache}.{os.getpid()}.tmp"
                with open(tmp, "wb") as f:
                    pickle.dump(ann, f, protocol=4)
                os.replace(tmp, cache)
        self.ann = ann
        self.sample_idxs = list(range(len(ann["frames"])))
        if self.mini_factor != 1.0:                                     # ho3d.py:112-114
            import random
            random.Random(1).shuffle(self.sample_idxs)
            self.sample_idxs = self.sample_idxs[:int(self.mini_factor * len(self.sample_idxs))]
        self.name2id = {v: k for k, v in CONST.YCB_IDX2CLASSES.items()}
        self._can = {}

    def _read_annotations(self, seq_frames):
        n = len(seq_frames)
        out = dict(frames=[tuple(sf) for sf in seq_frames], cam=np.zeros((n, 3, 3), np.float32), joints=np.zeros((n, 21, 3), np.float32),
                   obj_rot=np.zeros((n, 3), np.float32), obj_tsl=np.zeros((n, 3), np.float32), corners_rest=np.zeros((n, 8, 3), np.float32),
                   hand_bbox=np.zeros((n, 4), np.float32), obj_name=[])
        for i, (seq, frame) in enumerate(seq_frames):
            with open(os.path.join(self.root, self.subfolder, seq, "meta", f"{frame}.pkl"), "rb") as f:
                a = pickle.load(f, encoding="latin1")
            j = np.asarray(a["handJoints3D"], np.float32)
            out["joints"][i] = j[None].repeat(21, 0) if j.size == 3 else j            # evaluation frames carry the root only (:172-176)
            out["cam"][i], out["obj_rot"][i], out["obj_tsl"][i] = a["camMat"], np.asarray(a["objRot"]).reshape(3), a["objTrans"]
            out["corners_rest"][i] = a["objCorners3DRest"]
            if "handBoundingBox" in a:
                out["hand_bbox"][i] = a["handBoundingBox"]
            out["obj_name"].append(a["objName"])
        return out

    # ---- HOdataSource
    def __len__(self):
        return len(self.sample_idxs) if self.available else 0

    def get_sample_idxs(self):
        """Dataset indices behind positions 0 .. len-1 (ho3d.py:112-114, hodata.py: __getitem__ reports get_sample_idxs()[idx] as SAMPLE_IDX;
        differs from the position only with MINI_FACTOR != 1)."""
        return self.sample_idxs

    def get_image_path(self, idx):
        seq, frame = self.ann["frames"][self.sample_idxs[idx]]
        return os.path.join(self.root, self.subfolder, seq, "rgb", frame + self.ext)

    def get_image(self, idx):
        from PIL import Image
        return np.asarray(Image.open(self.get_image_path(idx)).convert("RGB"))

    def get_image_bytes(self, idx):
        with open(self.get_image_path(idx), "rb") as f:
            return f.read()

    def _canonical(self, obj):
        """get_obj_verts_can (:385-394): the mesh in the OpenCV frame, centred on its bounding box -> (bbox centre v_0)."""
        if obj not in self._can:
            v = self.CAM_EXTR.dot(self.obj_verts[obj].transpose()).transpose()
            self._can[obj] = (v.min(0) + v.max(0)) / 2
        return self._can[obj]

    def get_annots(self, idx):
        i = self.sample_idxs[idx]
        a, E = self.ann, self.CAM_EXTR
        K = a["cam"][i]
        proj = lambda p: (lambda h: (h / (h[:, 2:] + 1e-6))[:, :2].astype(np.float32))(np.array(K).dot(p.transpose()).transpose())   # noqa: E731  persp_project
        j3 = E.dot(a["joints"][i].transpose()).transpose()[self.REORDER].astype(np.float32)             # get_joints_3d (:262-268)
        obj = a["obj_name"][i]
        v0 = self._canonical(obj)
        rot, tsl = _rodrigues(a["obj_rot"][i]), a["obj_tsl"][i]
        Einv = np.linalg.inv(E)
        rot_cam = E @ (rot @ Einv)                                                                       # get_obj_transf_wrt_cam (:411-438)
        tsl_cam = (E @ (rot @ Einv)).dot(v0) + E.dot(tsl)
        T = np.concatenate([np.concatenate([rot_cam, tsl_cam[:, None]], 1), np.array([[0.0, 0.0, 0.0, 1.0]])], 0).astype(np.float32)
        can = ((E.dot(a["corners_rest"][i].transpose()).transpose() - v0) / 1).astype(np.float32)        # get_corners_can (:476-485)
        c3 = (T[:3, :3].dot(can.transpose()) + T[:3, 3:]).transpose().astype(np.float32)                 # get_corners_3d (:456-462)
        j2, c2 = proj(j3), proj(c3)
        if self.full_image:                                                                              # get_center_scale_wrt_bbox (:284-358)
            center, scale = np.array((self.raw_size[0] / 2, self.raw_size[1] / 2)), self.raw_size[0]
        else:
            test = self.data_split == "test"
            hb = a["hand_bbox"][i]
            hand2d = np.array([[hb[0], hb[1]], [hb[2], hb[3]]], np.float32) if test else j2
            if self.crop_model == "hand":
                pts = hand2d
            elif self.crop_model == "root_obj":
                pts = np.concatenate([j2[[0]], c2], 0)
            elif self.crop_model == "hand_obj":
                pts = np.concatenate([hand2d, c2], 0)
            else:
                raise NotImplementedError(f"CROP_MODEL {self.crop_model!r}")
            center, scale = annot_center_scale(pts)
        return dict(cam_intr=K, joints_3d=j3, joints_2d=j2, corners_3d=c3, corners_2d=c2, corners_can=can, obj_transf=T,
                    obj_idx=self.name2id[obj], side="right", bbox_center=np.asarray(center), bbox_scale=float(scale))


@DATASET.register_module
class HO3DV3(HO3D):
    """HO3D v3 (ho3d.py:573-606): the v2 reader on DATA_ROOT/HO3D_v3 with its frames as baseline .jpg files (device JPEG decode path);
    the annotation cache goes to common/cache/HO3D_v3."""
    name, subdir, ext = "HO3D_v3", "HO3D_v3", ".jpg"


@DATASET.register_module
class DexYCB(_DownloadedSet):
    name, subdir = "DexYCB", "DexYCB"


@DATASET.register_module
class SynthOnly(_DownloadedSet):
    """Explicitly empty real set (`DATASET: {TRAIN: {TYPE: SynthOnly}}`): synthetic-only training."""
    name, subdir = "SynthOnly", "__none__"


def ho_collate(batch):
    """hodata.py:17-62 for the fixed-size queries of the hot path (every sample dict carries same-shaped arrays): stack per
    key.  (The variable-length OBJ_VERTS_* padding of the reference applies to mesh queries no model of this path reads.)"""
    out = {}
    for k in batch[0]:
        v = [b[k] for b in batch]
        if torch.is_tensor(v[0]):
            out[k] = torch.stack(v)
        elif isinstance(v[0], np.ndarray):
            out[k] = torch.from_numpy(np.stack(v))
        elif isinstance(v[0], (int, float, bool, np.integer, np.floating, np.bool_)):
            out[k] = torch.as_tensor(np.asarray(v))
        else:
            out[k] = v
    return out

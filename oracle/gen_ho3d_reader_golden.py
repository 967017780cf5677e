"""TEST INFRASTRUCTURE -- writes tests/golden/ho3d_reader.npz: what the REAL reference reader (anakin/datasets/ho3d.py: class HO3D, SPLIT_MODE
"paper") returns, getter by getter, for every frame of the miniature HO3D v2 tree of tests/ho3d_fake_tree.py (seed 7) -- train and test
split, CROP_MODEL root_obj and hand_obj.  Third-party code the reference calls and this container lacks enters through stand-ins at exactly
two points: cv2.Rodrigues -> scipy.spatial.transform.Rotation.from_rotvec(...).as_matrix(), trimesh.load(path, process=False) -> the `v` / `f`
lines of the file in order (+ manotorch / pyrender / torchvision as the empty stubs of ref_import.load_control_plane: not called here).
Data only: arrays.  Run in the build container: python oracle/gen_ho3d_reader_golden.py"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import ho3d_fake_tree as T   # noqa: E402
import ref_import   # noqa: E402


def main():
    from scipy.spatial.transform import Rotation
    ref_import.load_control_plane()
    cv2 = sys.modules["cv2"]
    cv2.Rodrigues = lambda r: (Rotation.from_rotvec(np.asarray(r, np.float64).reshape(3)).as_matrix(), None)
    trimesh = sys.modules["trimesh"]

    def load(path, process=False):
        v, f = [], []
        for line in open(path):
            if line.startswith("v "):
                v.append([float(t) for t in line.split()[1:4]])
            elif line.startswith("f "):
                f.append([int(t.split("/")[0]) - 1 for t in line.split()[1:4]])
        return types.SimpleNamespace(vertices=np.asarray(v), faces=np.asarray(f), bounding_box_oriented=types.SimpleNamespace(vertices=np.zeros((8, 3))))
    trimesh.load = load
    dep = types.ModuleType("deprecated"); sph = types.ModuleType("deprecated.sphinx")
    sph.deprecated = lambda **k: (lambda fn: fn)
    dep.sphinx = sph
    sys.modules.setdefault("deprecated", dep); sys.modules.setdefault("deprecated.sphinx", sph)
    if not hasattr(np, "asfarray"):                                      # removed in NumPy 2 (ho3dutils.py:28, ho3d.py:393)
        np.asfarray = lambda a, dtype=np.float64: np.asarray(a, dtype=dtype)
    import anakin.datasets.hodata  # noqa
    from anakin.datasets.ho3d import HO3D, HO3DV3
    root = tempfile.mkdtemp(prefix="ho3d_fake_")
    T.build(root, seed=7)
    root3 = tempfile.mkdtemp(prefix="ho3dv3_fake_")
    T.build(root3, seed=9, version=3)
    os.chdir(tempfile.mkdtemp(prefix="ho3d_cache_"))                       # the reference writes common/cache/... relative to the cwd
    out = {}
    ds3 = HO3DV3(DATA_ROOT=root3, DATA_SPLIT="train", SPLIT_MODE="paper", AUG=False, AUG_PARAM=None, MINI_FACTOR=1.0,
                 DATA_PRESET={"USE_CACHE": False, "FILTER_NO_CONTACT": False, "FILTER_THRESH": 0.0, "BBOX_EXPAND_RATIO": 1.2, "FULL_IMAGE": False,
                              "IMAGE_SIZE": [224, 224], "CENTER_IDX": 0, "CROP_MODEL": "root_obj"})
    out["v3.n"] = np.int64(len(ds3))
    for i in range(len(ds3)):                                               # v3: the same getters on the other root, frames as .jpg
        out[f"v3.{i}.joints_3d"], out[f"v3.{i}.obj_transf"] = ds3.get_joints_3d(i), ds3.get_obj_transf(i)
        out[f"v3.{i}.path"] = np.frombuffer(os.path.relpath(ds3.get_image_path(i), root3).encode(), np.uint8)
    for split in ("train", "test"):
        for crop in ("root_obj", "hand_obj"):
            ds = HO3D(DATA_ROOT=root, DATA_SPLIT=split, SPLIT_MODE="paper", AUG=False, AUG_PARAM=None, MINI_FACTOR=1.0,
                      DATA_PRESET={"USE_CACHE": False, "FILTER_NO_CONTACT": False, "FILTER_THRESH": 0.0, "BBOX_EXPAND_RATIO": 1.2, "FULL_IMAGE": False,
                                   "IMAGE_SIZE": [224, 224], "CENTER_IDX": 0, "CROP_MODEL": crop})
            n = len(ds)
            out[f"{split}.n"] = np.int64(n)
            for i in range(n):
                c, s = ds.get_center_scale_wrt_bbox(i)
                out[f"{split}.{crop}.{i}.center"], out[f"{split}.{crop}.{i}.scale"] = np.asarray(c), np.float64(s)
                if crop == "root_obj":
                    pre = f"{split}.{i}."
                    out[pre + "cam_intr"], out[pre + "joints_3d"], out[pre + "joints_2d"] = ds.get_cam_intr(i), ds.get_joints_3d(i), ds.get_joints_2d(i)
                    out[pre + "corners_3d"], out[pre + "corners_2d"], out[pre + "corners_can"] = ds.get_corners_3d(i), ds.get_corners_2d(i), ds.get_corners_can(i)
                    out[pre + "obj_transf"], out[pre + "obj_idx"] = ds.get_obj_transf(i), np.int64(ds.get_obj_idx(i))
                    out[pre + "path"] = np.frombuffer(os.path.relpath(ds.get_image_path(i), root).encode(), np.uint8)
                    assert ds.get_sides(i) == "right"
    np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "ho3d_reader.npz"), **out)
    print(len(out), "arrays")


if __name__ == "__main__":
    main()

"""TEST INFRASTRUCTURE ONLY -- pins R1 (MANO linear-blend skinning) against code the reference itself holds.

The artiboost call sites use manotorch (un-vendored, absent); the reference tree carries the same MANO forward as a JAX layer,
`anakin/postprocess/iknet/manolayer.py:182-276`, whose `__call__` uses nothing but the numpy API.  This script loads THAT
FILE from /root/reference in the build container under a `jax.numpy -> numpy`, `jax.jit -> identity` shim, feeds it a
MANO_RIGHT.pkl written from the build's seeded stand-in hand model (`assets.make_hand_model`; the licensed MANO file is a
download) and dumps inputs + outputs to tests/golden/mano.npz.  `tests/test_oracle_golden.py` checks `pose_oracle.mano_lbs`
against it, `tests/test_gpu_synth.py` checks `ab_mano_lbs`.

The layer is run with center_idx = 0 (it indexes `jtr[:, self.center_idx]`, which has no meaning for None): it returns
verts and joints relative to the wrist joint; the golden therefore pins LBS up to that translation, and the tests add the
remaining identity joints[:, 0] == (J_regressor @ v_shaped)[0] explicitly.

Run:  python oracle/gen_mano_golden.py          (needs /root/reference; the output is committed)"""
import importlib.util
import os
import pickle
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/anakin/postprocess/iknet/manolayer.py"


def _rodrigues(v):
    """cv2.Rodrigues(rvec)[0] (used by the layer's loader on the all-zero rest pose only)."""
    from scipy.spatial.transform import Rotation
    return (Rotation.from_rotvec(np.asarray(v, dtype=np.float64).reshape(3)).as_matrix(),)


def load_reference_layer():
    jax = types.ModuleType("jax")
    jax.jit = lambda f, *a, **k: f
    jnp = types.ModuleType("jax.numpy")
    jnp.__dict__.update({k: getattr(np, k) for k in dir(np) if not k.startswith("__")})
    jax.numpy = jnp
    cv2 = types.ModuleType("cv2")
    cv2.Rodrigues = _rodrigues
    saved = {k: sys.modules.get(k) for k in ("jax", "jax.numpy", "cv2")}
    sys.modules.update({"jax": jax, "jax.numpy": jnp, "cv2": cv2})
    try:
        spec = importlib.util.spec_from_file_location("ref_manolayer", REF)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def write_mano_pkl(hm, root):
    """A MANO_RIGHT.pkl with the fields manolayer.py:55-99 reads, from the stand-in hand model."""
    import scipy.sparse as sp
    os.makedirs(os.path.join(root, "models"))
    parents = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
    dd = {"hands_components": np.eye(45), "hands_mean": np.asarray(hm["hands_mean"], np.float64),
          "shapedirs": np.asarray(hm["shapedirs"], np.float64), "posedirs": np.asarray(hm["posedirs"], np.float64),
          "v_template": np.asarray(hm["v_template"], np.float64), "J_regressor": sp.csc_matrix(np.asarray(hm["J_regressor"], np.float64)),
          "weights": np.asarray(hm["weights"], np.float64), "f": np.asarray(hm["faces"], np.int64),
          "kintree_table": np.stack([np.asarray(parents, np.int64) % (2 ** 32), np.arange(16)]), "bs_type": "lrotmin"}
    with open(os.path.join(root, "models", "MANO_RIGHT.pkl"), "wb") as f:
        pickle.dump(dd, f, protocol=2)


def main():
    from artiboost_amd.assets import make_hand_model
    seed = 1
    hm = make_hand_model(seed)
    mod = load_reference_layer()
    tmp = tempfile.mkdtemp(prefix="ab_mano_")
    write_mano_pkl(hm, tmp)
    layer = mod.ManoLayer(center_idx=0, flat_hand_mean=True, ncomps=45, side="right", mano_root=tmp, use_pca=False)
    rng = np.random.default_rng(7)
    B = 24
    pose = np.clip(0.45 * rng.standard_normal((B, 48)), -1.6, 1.6)
    pose[0] = 0.0
    pose[1, 3:] = 0.0                      # global rotation only
    betas = 0.6 * rng.standard_normal((B, 10))
    betas[0] = 0.0
    verts, jtr, full_pose = layer(pose, betas)
    out = os.path.join(ROOT, "tests", "golden", "mano.npz")
    np.savez_compressed(out, hand_model_seed=np.int64(seed), pose=pose, betas=betas, verts_rel_wrist=np.asarray(verts),
                        joints_rel_wrist=np.asarray(jtr), full_pose=np.asarray(full_pose))
    print("wrote", out, verts.shape, jtr.shape, float(np.abs(verts).max()))


if __name__ == "__main__":
    main()

"""On-disk formats of the reference's ArtiBoost state, so that caches and resume directories
written by either implementation are read by the other:

  * the per-epoch pose cache: one pickle per synthetic sample, `{index:04d}.pkl` =
    {obj_name, obj_id, persp_id, grasp_id, obj_pose (4,4), hand_verts (778,3), hand_joints (21,3)}
    (anakin/artiboost/cache_recorder.py:22-45, read back by rendered_dataset.py:103-116).  This build keeps the epoch as
    device-resident SoA tensors instead; `export_epoch` / `load_cache` convert between the two.
  * the mining state: `<dump>/artiboost/sample_weight/{epoch:03d}_train.pkl`, `.../occurence_map/{epoch:03d}.pkl`
    (pickled numpy arrays) and the empty `.../shutdown` marker (anakin/utils/recorder.py:177-226)."""
import os
import pickle

import numpy as np
import torch


class CacheRecorder:
    """cache_recorder.py:11-51 without the signal handlers: __call__(batch) pickles one file per sample."""

    def __init__(self, cache_root):
        self.cache_root = cache_root
        os.makedirs(cache_root, exist_ok=True)

    def __call__(self, batch):
        n = len(batch["index"])
        host = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in batch.items() if k != "obj_name"}
        for i in range(n):
            rec = {"obj_name": batch["obj_name"][i], "obj_id": int(host["obj_id"][i]), "persp_id": int(host["persp_id"][i]),
                   "grasp_id": int(host["grasp_id"][i]), "obj_pose": host["final_obj_pose"][i],
                   "hand_verts": host["final_hand_verts"][i], "hand_joints": host["final_joints"][i]}
            with open(os.path.join(self.cache_root, f"{int(host['index'][i]):0>4}.pkl"), "wb") as f:
                pickle.dump(rec, f)

    def clear(self):
        import shutil
        if os.path.exists(self.cache_root):
            shutil.rmtree(self.cache_root)
        os.makedirs(self.cache_root, exist_ok=True)


def export_epoch(loader, cache_root, obj_names=None):
    """Write the loader's current epoch in the reference's cache format (one device->host copy, then S small files)."""
    p = loader.epoch_poses
    names = obj_names or loader.cfg.get("OBJ_ENGINE", {}).get("OBJ") or [f"obj{i}" for i in range(loader.n_obj)]
    rec = CacheRecorder(cache_root)
    rec({"index": np.arange(len(p["obj_id"])), "obj_id": p["obj_id"], "persp_id": p["persp_id"], "grasp_id": p["grasp_id"],
         "obj_name": [names[int(o)] for o in p["obj_id"]], "final_obj_pose": p["obj_pose"], "final_hand_verts": p["hand_verts"],
         "final_joints": p["hand_joints"]})
    return len(p["obj_id"])


def load_cache(cache_root):
    """Read a cache directory (ours or the reference's) into stacked arrays, ordered by file index."""
    files = sorted(f for f in os.listdir(cache_root) if f.endswith(".pkl"))
    if not files:
        raise FileNotFoundError(f"no cached samples under {cache_root}")
    out = {k: [] for k in ("obj_name", "obj_id", "persp_id", "grasp_id", "obj_pose", "hand_verts", "hand_joints")}
    for f in files:
        with open(os.path.join(cache_root, f), "rb") as fh:
            rec = pickle.load(fh)
        for k in out:
            out[k].append(rec[k])
    return {k: (v if k == "obj_name" else np.asarray(v)) for k, v in out.items()}


def record_artiboost_loader(loader, epoch, dump_path):
    """Recorder.record_artiboost_loader (anakin/utils/recorder.py:177-202)."""
    for sub, name, arr in (("sample_weight", f"{epoch:0>3}_train.pkl", loader.sample_weight_map),
                           ("occurence_map", f"{epoch:0>3}.pkl", loader.occurence_map)):
        d = os.path.join(dump_path, "artiboost", sub)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "wb") as f:
            pickle.dump(arr.detach().cpu().numpy() if torch.is_tensor(arr) else np.array(arr), f)
    if not loader.use_synth:
        open(os.path.join(dump_path, "artiboost", "shutdown"), "w").close()


def resume_artiboost_loader(loader, resume_epoch, resume_path):
    """Recorder.resume_artiboost_loader (anakin/utils/recorder.py:204-226): state of epoch resume_epoch - 1."""
    epoch = resume_epoch - 1
    with open(os.path.join(resume_path, "artiboost", "sample_weight", f"{epoch:0>3}_train.pkl"), "rb") as f:
        loader.sample_weight_map[:] = torch.from_numpy(pickle.load(f))
    with open(os.path.join(resume_path, "artiboost", "occurence_map", f"{epoch:0>3}.pkl"), "rb") as f:
        loader.occurence_map[:] = torch.from_numpy(pickle.load(f))
    if os.path.exists(os.path.join(resume_path, "artiboost", "shutdown")):
        loader.synth_shutdown()

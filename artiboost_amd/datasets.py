"""DATASET registry entries and the collate function the reference's train script names
(anakin/datasets/{hodata,ho3d,dexycb}.py; builder.build_dataset, train/train_artiboost.py:114-125,172).

HO3D / DexYCB are downloads (README.md:73-91 of the reference) and are not in this build's environment: the entries below
resolve the reference's YAML (`TYPE: HO3D`, `DATA_ROOT: ./data`, ...) to an EMPTY real set when the data root is absent --
ArtiBoostLoader then trains on the synthetic share alone -- and raise if a root IS given but unreadable, so a
misconfigured path is not silently ignored.  Decoding real frames goes through `realdata.HOdataSource`."""
import os

import numpy as np
import torch

from .realdata import HOdataSource
from .registry import DATASET


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
            raise NotImplementedError(f"{self.name}: found {root}, but the annotation reader of this dataset is not part of this "
                                      f"build (SURVEY.md section 8f-3: provide a realdata.HOdataSource over it)")

    def __len__(self):
        return 0


@DATASET.register_module
class HO3D(_DownloadedSet):
    name, subdir = "HO3D", "HO3D"


@DATASET.register_module
class HO3DV3(_DownloadedSet):
    name, subdir = "HO3DV3", "HO3D_v3"


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

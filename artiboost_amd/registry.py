"""Registry / build_from_cfg / builder functions -- the reference's plugin surface
(anakin/utils/registry.py:4-70, anakin/utils/builder.py:5-100, anakin/utils/misc.py:30-38,57-119), re-stated so
the reference's YAML configs (config/*.yaml) build this package's classes unchanged."""
import functools
import inspect
import math
import re
from enum import Enum

import yaml


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._module_dict = dict()

    def __repr__(self):
        return f"{self.__class__.__name__}(name={self._name}, items={list(self._module_dict.keys())})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        if cls.__name__ in self._module_dict:
            raise KeyError(f"{cls.__name__} is already registered in {self.name}")
        self._module_dict[cls.__name__] = cls
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    assert isinstance(cfg, dict) and "TYPE" in cfg
    assert isinstance(default_args, dict) or default_args is None
    args = cfg.copy()
    obj_type = args.pop("TYPE")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f"{obj_type} is not in the {registry.name} registry")
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(obj_type)}")
    if default_args is not None:
        for name, value in default_args.items():
            args.setdefault(name, value)
    return obj_cls(**args)


MODEL = Registry("model")
BACKBONE = Registry("backbone")
NECK = Registry("neck")
HEAD = Registry("head")
LOSS = Registry("loss")
DATASET = Registry("dataset")
METRIC = Registry("metric")


def build(cfg, registry, default_args=None):
    return build_from_cfg(cfg, registry, default_args)


def _build_list(cfg, preset_cfg, registry, **kwargs):
    default_args = {"DATA_PRESET": preset_cfg}
    default_args.update(kwargs)
    if isinstance(cfg, list):
        return [build(c, registry, default_args) for c in cfg]
    return [build(cfg, registry, default_args)]


def build_arch_model_list(cfg, preset_cfg, **kwargs):
    return _build_list(cfg, preset_cfg, MODEL, **kwargs)


def build_evaluator_metric_list(cfg, preset_cfg, **kwargs):
    return _build_list(cfg, preset_cfg, METRIC, **kwargs)


def build_criterion_loss_list(cfg, preset_cfg, **kwargs):
    return _build_list(cfg, preset_cfg, LOSS, **kwargs)


def build_dataset(cfg, preset_cfg, **kwargs):
    default_args = {"DATA_PRESET": preset_cfg}
    default_args.update(kwargs)
    return build(cfg, DATASET, default_args=default_args)


def build_model(cfg, default_args=None):
    return build(cfg, MODEL, default_args=default_args)


def build_head(cfg, default_args=None):
    return build(cfg, HEAD, default_args=default_args)


def build_backbone(cfg, default_args=None):
    return build(cfg, BACKBONE, default_args=default_args)


def build_loss(cfg, default_args=None):
    return build(cfg, LOSS, default_args=default_args)


def build_metric(cfg, default_args=None):
    return build(cfg, METRIC, default_args=default_args)


def enable_lower_param(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        return func(*args, **{k.upper(): v for k, v in kwargs.items()})

    return wrapper


# ---- the image plane a padded NHWC4 image tensor carries (never guessed from dtype or value range) ---------------------------------
# "f32": network input x / 255 - 0.5 in the tensor's own dtype;  "u8n": ONE bf16 plane of the odd integers 2 v - 255 (AB_DT_U8N of the
# renderer / augmenter; network input = plane / 510), what a bf16x3 model's two-pass stem consumes.  The loaders tag the tensors they
# write (tag_image_plane) and their batch dicts carry IMAGE_PLANE_KEY; the models and metrics read the tag.
IMAGE_PLANE_KEY = "image_plane"
# process-wide hint written by the most recently built model: the compute_dtype its loader should be built with ("u8n" for bf16x3,
# torch.bfloat16 / torch.float32 otherwise).  A loader constructed with the REFERENCE's keywords (no compute_dtype) follows it, so the
# unmodified train_artiboost.py:117-190 order (model first, loader second) lands on the model's native plane.
RUNTIME = {"loader_compute_dtype": None}


class PlaneTag(str):
    """The IMAGE_PLANE_KEY entry of a batch dict: a str that passes through the usual whole-batch idioms
    ({k: v.clone() ...}, v.to(dev), v.cuda()) unchanged, so a tagged batch survives them with its tag."""
    __slots__ = ()

    def _same(self, *a, **k):
        return self

    clone = detach = to = cuda = cpu = contiguous = pin_memory = _same


def tag_image_plane(t, plane):
    assert plane in ("f32", "u8n")
    t._ab_plane = PlaneTag(plane)
    return t


def image_plane_of(batch, xpad=None):
    """The plane tag of a batch dict (IMAGE_PLANE_KEY) or of its padded image tensor; None when neither carries one."""
    plane = batch.get(IMAGE_PLANE_KEY) if isinstance(batch, dict) else None
    if plane is None and xpad is not None:
        plane = getattr(xpad, "_ab_plane", None)
    return None if plane is None else PlaneTag(plane)


class TrainMode(Enum):
    TRAIN = 0
    VAL = 1
    TEST = 2


class CONST:
    PI = math.pi
    INT_MAX = 2 ** 32 - 1
    NUM_JOINTS = 21
    NUM_CORNERS = 8
    SIDE = "right"
    DUMMY = "dummy"
    JOINTS_IDX_PARENTS = [0, 0, 1, 2, 3, 0, 5, 6, 7, 0, 9, 10, 11, 0, 13, 14, 15, 0, 17, 18, 19]
    YCB_IDX2CLASSES = {
        1: "002_master_chef_can", 2: "003_cracker_box", 3: "004_sugar_box", 4: "005_tomato_soup_can",
        5: "006_mustard_bottle", 6: "007_tuna_fish_can", 7: "008_pudding_box", 8: "009_gelatin_box",
        9: "010_potted_meat_can", 10: "011_banana", 11: "019_pitcher_base", 12: "021_bleach_cleanser", 13: "024_bowl",
        14: "025_mug", 15: "035_power_drill", 16: "036_wood_block", 17: "037_scissors", 18: "040_large_marker",
        19: "051_large_clamp", 20: "052_extra_large_clamp", 21: "061_foam_brick",
    }


class Queries:
    SAMPLE_IDX = "sample_idx"
    IMAGE = "image"
    CAM_INTR = "cam_intr"
    CORNERS_CAN = "corners_can"
    CORNERS_2D = "corners_2d"
    CORNERS_3D = "corners_3d"
    JOINTS_2D = "joints_2d"
    JOINTS_3D = "joints_3d"
    ROOT_JOINT = "root_joint"
    CORNERS_VIS = "corners_vis"
    JOINTS_VIS = "joints_vis"
    OBJ_TRANSF = "obj_transf"
    OBJ_IDX = "obj_idx"


class SynthQueries:
    IS_SYNTH = "is_synth"
    OBJ_ID = "obj_id"
    PERSP_ID = "persp_id"
    GRASP_ID = "grasp_id"


_camel = re.compile(r"(?<!^)(?=[A-Z])")


def camel_to_snake(name: str) -> str:
    return _camel.sub("_", name).lower()


def update_config(config_file):
    with open(config_file) as f:
        return yaml.load(f, Loader=yaml.FullLoader)

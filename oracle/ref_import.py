"""TEST INFRASTRUCTURE ONLY -- imports the *real* reference (lixiny/ArtiBoost, mounted read-only at
/root/reference) inside the build container so golden vectors can be generated from it.

Nothing here travels to the GPU box in any useful form: /root/reference does not exist there, and the
`-m gpu` tests, smoke() and bench.py never import this module.  It is used by `oracle/gen_golden.py`
(run once, output committed under tests/golden/) and by the optional `-m "not gpu"` cross-checks that
skip themselves when /root/reference is absent.

Recipe follows SURVEY.md Appendix A: the reference's package __init__s eagerly import trimesh / cv2 /
manotorch / pyrender (anakin/datasets/__init__.py:1-3, anakin/models/__init__.py:1-8), none of which is
installed, so the leaf modules are loaded under empty parent packages instead.
"""
import os
import sys
import types
import tempfile

REF_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "anakin"))


_loaded = False


def load():
    """Make `anakin.*` leaf modules of the reference importable. Idempotent."""
    global _loaded
    if _loaded:
        return
    if not available():
        raise RuntimeError("reference tree not present; golden vectors must be used instead")
    import numpy as np

    shim = tempfile.mkdtemp(prefix="ab_shim_")
    with open(os.path.join(shim, "termcolor.py"), "w") as f:
        f.write("def colored(s, *a, **k):\n    return s\n")
    os.makedirs(os.path.join(shim, "pytorch3d"))
    open(os.path.join(shim, "pytorch3d", "__init__.py"), "w").close()
    names = ("axis_angle_to_matrix axis_angle_to_quaternion euler_angles_to_matrix matrix_to_euler_angles "
             "matrix_to_quaternion matrix_to_rotation_6d quaternion_to_axis_angle quaternion_to_matrix "
             "rotation_6d_to_matrix").split()
    with open(os.path.join(shim, "pytorch3d", "transforms.py"), "w") as f:
        f.write("def _absent(*a, **k):\n    raise RuntimeError('pytorch3d is not installed')\n")
        for n in names:
            f.write(f"{n} = _absent\n")
    sys.path[:0] = [shim, REF_ROOT]
    sys.argv = ["oracle"]  # anakin/opt.py:54 parses argv at import
    for alias, ty in (("float", float), ("int", int), ("long", int), ("bool", bool)):
        if not hasattr(np, alias):
            setattr(np, alias, ty)  # view_engine.py:29, hodata.py:56 use NumPy<1.24 aliases
    import anakin  # noqa

    for pkg in ("datasets", "models", "criterions", "metrics", "artiboost"):
        m = types.ModuleType(f"anakin.{pkg}")
        m.__path__ = [f"{REF_ROOT}/anakin/{pkg}"]
        sys.modules[f"anakin.{pkg}"] = m
        setattr(anakin, pkg, m)
    import anakin.models.resnet  # noqa  registers ResNet34
    import anakin.models.simplebaseline  # noqa  registers IntegralDeconvHead
    import anakin.models.mlp as mlp
    import anakin.models.hybridbaseline as hb

    anakin.models.MLP_O = mlp.MLP_O
    anakin.models.HybridBaseline = hb.HybridBaseline  # builder.py:82 exec("from ..models import X")
    import anakin.criterions.jointloss  # noqa
    import anakin.criterions.ordinal  # noqa
    import anakin.criterions.symcornerloss  # noqa
    _loaded = True


def load_control_plane():
    """Additionally stub the heavy third-party imports so anakin.artiboost control-plane *pure functions*
    (update_method_1..4, row_col_calc, caculate_align_mat, ...) can be imported."""
    load()
    from unittest.mock import MagicMock

    def stub(name, **attrs):
        m = MagicMock()
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _B(object):
        def __init__(self, *a, **k):
            pass

    for n in ("trimesh", "trimesh.base", "cv2", "chamfer_distance", "torchvision", "torchvision.transforms",
              "torchvision.transforms.functional", "manotorch", "manotorch.axislayer", "manotorch.utils",
              "manotorch.utils.rodrigues", "manotorch.utils.quatutils", "pyrender.constants", "pyrender.light",
              "pyrender.material", "pyrender.platforms"):
        if n not in sys.modules:
            stub(n)
    if "pyrender" not in sys.modules or isinstance(sys.modules["pyrender"], MagicMock):
        stub("pyrender", Mesh=type("Mesh", (_B,), {}), OffscreenRenderer=type("OffscreenRenderer", (_B,), {}),
             Primitive=type("Primitive", (_B,), {}), Renderer=type("Renderer", (_B,), {}),
             Scene=type("Scene", (_B,), {}))
    stub("manotorch.manolayer", ManoLayer=type("ManoLayer", (_B,), {}), MANOOutput=type("MANOOutput", (_B,), {}))


def load_refiner(hand_model):
    """anakin.artiboost.refiner of the reference, importable: its three absent third-party dependencies are bound to
    stand-ins (documented in oracle/refiner_oracle.py; parity is unpinned at exactly these boundaries):
      chamfer_distance.ChamferDistance -> brute-force nearest neighbour (fp32, first minimum),
      manotorch.manolayer.ManoLayer    -> pose_oracle.mano_lbs on `hand_model` (flat_hand_mean, center_idx None, zero betas),
      pytorch3d rotation conversions   -> pose_oracle.aa_to_rotmat / rotmat_to_aa.
    Everything else that runs (HORefiner.forward, _RefineNet, ResBlock, CRot2rotmat, parms_decode, point2point_signed) is
    the reference's own code."""
    load_control_plane()
    import numpy as np
    import torch
    import pose_oracle as po
    import refiner_oracle as rfo

    class ChamferDistance:
        def __call__(self, x, y):
            d_xy, i_xy = rfo.nearest_dist(x.numpy(), y.numpy())
            d_yx, i_yx = rfo.nearest_dist(y.numpy(), x.numpy())
            t = torch.from_numpy
            return t(d_xy) ** 2, t(d_yx) ** 2, t(i_xy), t(i_yx)

    class MANOOutput:
        def __init__(self, verts, joints):
            self.verts, self.joints = verts, joints

    class ManoLayer(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, pose_coeffs, betas=None):
            b = np.zeros((pose_coeffs.shape[0], 10)) if betas is None else betas.double().numpy()
            v, j, _ = po.mano_lbs(hand_model, pose_coeffs.detach().double().numpy(), b)
            return MANOOutput(torch.from_numpy(v).float(), torch.from_numpy(j).float())

    cd = types.ModuleType("chamfer_distance")
    cd.ChamferDistance = ChamferDistance
    sys.modules["chamfer_distance"] = cd
    ml = types.ModuleType("manotorch.manolayer")
    ml.ManoLayer, ml.MANOOutput = ManoLayer, MANOOutput
    sys.modules["manotorch.manolayer"] = ml
    import anakin.utils.transform as T
    f64 = lambda fn: (lambda x: torch.from_numpy(fn(x.detach().double().numpy())).float())   # noqa: E731
    T.axis_angle_to_matrix = f64(po.aa_to_rotmat)
    T.matrix_to_quaternion = lambda m: m                       # rotmat_to_aa = Compose([matrix_to_quaternion, quaternion_to_axis_angle])
    T.quaternion_to_axis_angle = f64(po.rotmat_to_aa)
    sys.modules.pop("anakin.artiboost.refiner", None)
    import anakin.artiboost.refiner as R
    return R


def build_reference_model_and_criterion(image_size=224, heatmap=28, depth=28, center_idx=0, seed=1):
    """Reference Arch(HybridBaseline) + Criterion from the reference's own training YAML
    (config/ho3dv2_clasbased_jlol_artiboost2.yaml), random init from `seed`."""
    load()
    import yaml
    import torch
    from anakin.utils import builder
    from anakin.models.arch import Arch
    from anakin.criterions.criterion import Criterion

    cfg = yaml.safe_load(open(f"{REF_ROOT}/config/ho3dv2_clasbased_jlol_artiboost2.yaml"))
    cfg["ARCH"]["BACKBONE"]["PRETRAINED"] = False  # model_zoo download otherwise (resnet.py:194-197)
    cfg["DATA_PRESET"]["IMAGE_SIZE"] = [image_size, image_size]
    cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [heatmap, heatmap]
    cfg["DATA_PRESET"]["CENTER_IDX"] = center_idx
    cfg["ARCH"]["HYBRID_HEAD"]["DEPTH_RESOLUTION"] = depth
    torch.manual_seed(seed)
    model = Arch(cfg, builder.build_arch_model_list(cfg["ARCH"], preset_cfg=cfg["DATA_PRESET"]))
    crit = Criterion(cfg, builder.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"],
                                                           LAMBDAS=cfg["LAMBDAS"]))
    return cfg, model, crit

"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the REAL reference
(/root/reference, imported via oracle/ref_import.py) on seeded inputs.  Run in the build container:

    python oracle/gen_golden.py

The outputs are data (inputs + expected outputs), committed under tests/golden/.  No reference source
travels.  Weights are NOT stored: they are regenerated anywhere by learner_oracle.fill_params (name-keyed
deterministic fill) and loaded into the reference modules here.
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
import learner_oracle as lo  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")


sys.path.insert(0, os.path.join(HERE, "..", "tests"))
from gen_batch import make_batch  # noqa: E402


def load_params_into_reference(model, params):
    sd = model.state_dict()
    for k in sd:
        sd[k] = params[k.replace("_model_list.0.", "")].clone()
    model.load_state_dict(sd)


def gen_learner(tag, size, heat, depth, B, seed):
    cfg, model, crit = ref_import.build_reference_model_and_criterion(size, heat, depth, center_idx=0, seed=seed)
    params = lo.fill_params(lo.param_shapes(22, depth), seed=seed)
    load_params_into_reference(model, params)
    batch = make_batch(B, size, seed + 100)
    out = {}
    for mode in ("train", "eval"):
        model.train(mode == "train")
        feats = {}
        hb = model._model_list[0]
        hooks = [hb.backbone.register_forward_hook(lambda m, i, o: feats.update(o)),
                 hb.hybrid_head.final_layer.register_forward_hook(lambda m, i, o: feats.update(logits=o))]
        preds = model(batch)["HybridBaseline"]
        for h in hooks:
            h.remove()
        for k, v in preds.items():
            out[f"{mode}.pred.{k}"] = v.detach().numpy().copy()
        for k in ("res_layer1", "res_layer2", "res_layer3", "res_layer4"):
            f = feats[k].detach()
            out[f"{mode}.feat.{k}.mean_c"] = f.mean(dim=(0, 2, 3)).numpy().copy()
            out[f"{mode}.feat.{k}.sample"] = f[:, ::16, ::3, ::3].numpy().copy()
        out[f"{mode}.feat.res_layer4_mean"] = feats["res_layer4_mean"].detach().numpy().copy()
        lg = feats["logits"].detach()
        out[f"{mode}.logits.sample"] = lg[:, ::37, ::5, ::5].numpy().copy()
        out[f"{mode}.logits.absmean"] = lg.abs().mean().numpy().copy()
        if mode == "train":
            # losses with pinned RNG (ordinal.py draws from torch + python global RNGs)
            random.seed(seed + 7)
            torch.manual_seed(seed + 7)
            model.zero_grad()
            total, losses = crit.compute_losses(preds, batch)
            total.backward()
            for k, v in losses.items():
                out[f"loss.{k}"] = v.detach().numpy().copy()
            gn = {}
            for n, p in model.named_parameters():
                if p.grad is None:
                    continue
                k = n.replace("_model_list.0.", "")
                gn[k] = float(p.grad.norm())
            out["grad.names"] = np.array(list(gn.keys()))
            out["grad.norms"] = np.array(list(gn.values()), dtype=np.float64)
            named = dict(model.named_parameters())
            out["grad.conv1.sample"] = named["_model_list.0.backbone.conv1.weight"].grad[::8, :, ::3, ::3].numpy().copy()
            out["grad.final_bias"] = named["_model_list.0.hybrid_head.final_layer.bias"].grad.numpy().copy()
            out["grad.box4.weight"] = named["_model_list.0.box_head.layers.4.weight"].grad.numpy().copy()
            out["grad.l3.0.ds.sample"] = named["_model_list.0.backbone.layer3.0.downsample.0.weight"].grad[::16, ::16, 0, 0].numpy().copy()
            out["grad.deconv3.sample"] = named["_model_list.0.hybrid_head.deconv_layers.3.weight"].grad[::32, ::32].numpy().copy()
            # running stats after one training forward
            sd = model.state_dict()
            out["stat.bn1.running_mean"] = sd["_model_list.0.backbone.bn1.running_mean"].numpy().copy()
            out["stat.bn1.running_var"] = sd["_model_list.0.backbone.bn1.running_var"].numpy().copy()
            out["stat.l4.2.bn2.running_var"] = sd["_model_list.0.backbone.layer4.2.bn2.running_var"].numpy().copy()
            # one optimiser step exactly as train_artiboost.py:91-96
            opt = torch.optim.Adam(model.models_params, lr=5e-5, weight_decay=0.0)
            tn = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.001)
            opt.step()
            out["opt.total_norm"] = tn.numpy().copy()
            named = dict(model.named_parameters())
            out["opt.conv1.delta.sample"] = (named["_model_list.0.backbone.conv1.weight"].detach()
                                             - params["backbone.conv1.weight"])[::8, :, ::3, ::3].numpy().copy()
            out["opt.final_bias.delta"] = (named["_model_list.0.hybrid_head.final_layer.bias"].detach()
                                           - params["hybrid_head.final_layer.bias"]).numpy().copy()
            load_params_into_reference(model, params)  # restore for the eval pass
    out["meta"] = np.array([size, heat, depth, B, seed])
    np.savez_compressed(os.path.join(OUT, f"learner_{tag}.npz"), **out)
    print("wrote", tag, sum(v.nbytes for v in out.values()) / 1e3, "KB")


def gen_head_only(seed=3):
    """softmax+integral head alone (simplebaseline.py:16-71,177-190) at both geometries, small B."""
    ref_import.load()
    from anakin.models.simplebaseline import norm_heatmap, integral_heatmap3d
    out = {}
    for tag, (C, D, H, W) in {"g224": (22, 28, 28, 28), "g256": (22, 28, 32, 32), "tiny": (3, 4, 5, 6)}.items():
        g = torch.Generator().manual_seed(seed)
        B = 2
        logits = (4.0 * torch.randn(B, C * D, H, W, generator=g)).requires_grad_(True)
        x = logits.reshape(B, C, -1)
        x = norm_heatmap("softmax", x)
        conf = torch.max(x, dim=-1).values
        x = x / (x.sum(dim=-1, keepdim=True) + 1e-7)
        x = x.contiguous().view(B, C, D, H, W)
        uvd = integral_heatmap3d(x)
        gu = torch.randn(uvd.shape, generator=g)
        gc = torch.randn(conf.shape, generator=g)
        ((uvd * gu).sum() + (conf * gc).sum()).backward()
        out[f"{tag}.seed"] = np.array([seed, B, C, D, H, W])
        out[f"{tag}.uvd"] = uvd.detach().numpy().copy()
        out[f"{tag}.conf"] = conf.detach().numpy().copy()
        out[f"{tag}.g_uvd"] = gu.numpy().copy()
        out[f"{tag}.g_conf"] = gc.numpy().copy()
        dl = logits.grad
        out[f"{tag}.dlogits.sample"] = dl.reshape(B, C, -1)[:, :, ::97].numpy().copy()
        out[f"{tag}.dlogits.abs_sum"] = dl.abs().sum().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "head.npz"), **out)
    print("wrote head")


def gen_misc(seed=5):
    """Pure helper functions: affine transform, CCV update/row-col, view alignment, symmetry table, metric."""
    ref_import.load_control_plane()
    from anakin.utils import transform as T
    out = {}
    rng = np.random.default_rng(seed)
    cs, ss, rs, tots, posts = [], [], [], [], []
    for i in range(6):
        c = rng.integers(150, 350, size=2).astype(np.float64)
        s = float(rng.uniform(120, 300))
        r = float(rng.uniform(-0.6, 0.6))
        tot, post = T.get_affine_transform(c, s, [256.0, 256.0], [224, 224] if i % 2 else [256, 256], rot=r)
        cs.append(c); ss.append(s); rs.append(r); tots.append(tot); posts.append(post)
    out["affine.center"] = np.array(cs); out["affine.scale"] = np.array(ss); out["affine.rot"] = np.array(rs)
    out["affine.total"] = np.array(tots); out["affine.post"] = np.array(posts)
    pts = rng.uniform(0, 512, size=(9, 2))
    out["coords.pts"] = pts
    out["coords.out"] = T.transform_coords(pts, tots[0])
    from anakin.datasets.hodata import HOdata
    out["annot.center"] = HOdata.get_annot_center(pts)
    out["annot.scale"] = np.array(HOdata.get_annot_scale(pts))
    # CCV
    from anakin.artiboost.artiboost_loader import ArtiBoostLoader
    from anakin.artiboost.ovg_set import OVGSet
    from anakin.artiboost.view_engine import ViewEngine
    w = torch.ones(4, 288, 50)
    ids = [(int(a), int(b), int(c)) for a, b, c in zip(rng.integers(0, 4, 40), rng.integers(0, 288, 40), rng.integers(0, 50, 40))]
    vals = rng.uniform(5, 60, 40)
    res = dict(zip(ids, vals.tolist()))
    out["ccv.ids"] = np.array(list(res.keys())); out["ccv.vals"] = np.array(list(res.values()))
    for m in (1, 2, 3):
        fn = getattr(ArtiBoostLoader, f"update_method_{m}")
        r = fn(w.clone(), res, 0.1, 10.0, dist_lower_threshold=8.0, dist_upper_threshold=16.0, epoch_idx=3, n_epochs=10)
        out[f"ccv.m{m}"] = np.array([float(r["sample_weight_map"][i]) for i in res.keys()])
    t = torch.tensor(rng.integers(0, 4 * 288 * 50, 64))
    b, r_, c = OVGSet.row_col_calc(t, 288, 50)
    out["ccv.tidx"] = t.numpy().copy(); out["ccv.bidx"] = b.numpy().copy(); out["ccv.ridx"] = r_.numpy().copy(); out["ccv.cidx"] = c.numpy().copy()
    vecs = rng.normal(size=(8, 3)); vecs[0] = [0, 0, 1]; vecs[1] = [0, 0, -1]
    out["view.vecs"] = vecs
    out["view.align"] = np.array([ViewEngine.caculate_align_mat(v.copy()) for v in vecs])
    # symmetry table (bop_misc.py:18-65) on a synthetic models_info
    from anakin.utils.bop_toolkit.bop_misc import get_symmetry_transformations
    info = {"symmetries_discrete": [[-1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]],
            "symmetries_continuous": [{"axis": [0, 0, 1], "offset": [1.0, 2.0, 0.0]}]}
    tr = get_symmetry_transformations(info, 0.25)
    out["sym.R"] = np.array([x["R"] for x in tr]); out["sym.t"] = np.array([x["t"] for x in tr])
    np.savez_compressed(os.path.join(OUT, "misc.npz"), **out)
    print("wrote misc")


def gen_blacklist(seed=9):
    """ArtiBoostLoader._construct_blacklist_map of the real reference (artiboost_loader.py:415-500) on a small CCV space,
    called with a stand-in `self` that carries the real ViewEngine (jittered views from the seeded torch RNG) and a table
    of grasps; the views it drew are recorded so that the restatement can be checked on exactly the same triplets."""
    import types
    ref_import.load_control_plane()
    import pose_oracle as po
    import anakin.utils.transform as T
    T.axis_angle_to_matrix = lambda x: torch.from_numpy(po.aa_to_rotmat(x.detach().double().numpy())).float()   # pytorch3d stand-in
    from anakin.artiboost.artiboost_loader import ArtiBoostLoader
    from anakin.artiboost.view_engine import ViewEngine
    n_obj, u_bins, th_bins, n_grasp = 2, 4, 6, 7
    rng = np.random.default_rng(seed)
    grasps = (1.5 * rng.standard_normal((n_obj, n_grasp, 48))).astype(np.float32)
    ve = ViewEngine.__new__(ViewEngine)
    ve.persp_u_bins, ve.persp_theta_bins = u_bins, th_bins
    views = []

    def get_view(vi):
        R = ViewEngine.get_perspective_from_id(ve, torch.tensor(vi))
        views.append(np.asarray(R, np.float64))
        return R, np.eye(4), np.zeros(3)

    fake = types.SimpleNamespace(
        obj_engine=types.SimpleNamespace(obj_names=[f"obj{i}" for i in range(n_obj)]),
        grasp_engine=types.SimpleNamespace(get_obj_grasp=lambda name, gi: (grasps[int(name[3:]), gi], None, None)),
        view_engine=types.SimpleNamespace(persp_u_bins=u_bins, persp_theta_bins=th_bins, get_view=get_view))
    cwd = os.getcwd()
    import tempfile
    os.chdir(tempfile.mkdtemp())                         # the method writes its pickle cache under ./common/cache
    torch.manual_seed(seed)
    try:
        bl = ArtiBoostLoader._construct_blacklist_map(fake, n_obj, u_bins * th_bins, n_grasp, True)
        cache = [os.path.join(r, f) for r, _, fs in os.walk("common") for f in fs]
        assert len(cache) == 1
        ident = os.path.basename(cache[0])
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(OUT, "blacklist.npz"), grasps=grasps, views=np.array(views).reshape(n_obj, u_bins * th_bins, n_grasp, 3, 3),
                        blacklist=bl.numpy(), u_bins=u_bins, th_bins=th_bins, cache_name=ident)
    print("wrote blacklist", bl.shape, float(bl.float().mean()), ident)


def gen_state_files(seed=12):
    """Files written by the reference's own writers -- CacheRecorder.__call__ (cache_recorder.py:22-45) and
    Recorder.record_sample_weight / record_sample_occurence / record_shutdown (utils/recorder.py:182-202) -- on small seeded
    inputs, committed as fixtures under tests/golden/ref_state/ for the readers in artiboost_amd/ccv_cache.py."""
    import types
    import shutil
    ref_import.load_control_plane()
    from anakin.artiboost.cache_recorder import CacheRecorder
    git = types.ModuleType("git")
    git.Repo = object
    sys.modules.setdefault("git", git)
    from anakin.utils.recorder import Recorder
    Recorder = Recorder.__wrapped__                      # @singleton (utils/misc.py:41-50) keeps the class here
    root = os.path.join(OUT, "ref_state")
    shutil.rmtree(root, ignore_errors=True)
    os.makedirs(os.path.join(root, "cache"))
    g = torch.Generator().manual_seed(seed)
    batch = {"index": torch.tensor([3, 17]), "obj_id": torch.tensor([1, 0]), "persp_id": torch.tensor([200, 5]),
             "grasp_id": torch.tensor([49, 7]), "obj_name": ["021_bleach_cleanser", "010_potted_meat_can"],
             "final_obj_pose": torch.randn(2, 4, 4, generator=g), "final_hand_verts": torch.randn(2, 778, 3, generator=g),
             "final_joints": torch.randn(2, 21, 3, generator=g)}
    CacheRecorder.__call__(types.SimpleNamespace(cache_root=os.path.join(root, "cache")), batch)
    w = torch.rand(2, 6, 5, generator=g) * 3
    occ = torch.rand(2, 6, 5, generator=g) > 0.5
    fake = types.SimpleNamespace(dump_path=os.path.join(root, "dump"))
    os.makedirs(os.path.join(root, "dump", "artiboost"))
    Recorder.record_sample_weight(fake, w, 4)
    Recorder.record_sample_occurence(fake, occ, 4)
    Recorder.record_shutdown(fake, types.SimpleNamespace(use_synth=False))
    np.savez_compressed(os.path.join(root, "expected.npz"), weight=w.numpy(), occ=occ.numpy(),
                        **{k: v.numpy() for k, v in batch.items() if k != "obj_name"})
    print("wrote ref_state", sorted(os.path.relpath(os.path.join(r, f), root) for r, _, fs in os.walk(root) for f in fs))


def gen_real_sample(seed=21):
    """HOdata.__getitem__ of the real reference (anakin/datasets/hodata.py:315-450) on a stand-in subclass that serves
    seeded frames and annotations (the datasets are downloads): the full sample dict incl. the augmented image, plus the
    augmentation draws, recovered by replaying the same RNG calls in the same order.  torchvision (absent) enters through
    stand-ins for to_tensor / normalize only."""
    import random
    import types
    from PIL import Image
    ref_import.load_control_plane()
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvf.to_tensor = lambda im: torch.from_numpy(np.asarray(im, np.uint8).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    tvf.normalize = lambda t, mean, std: (t - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None]
    sys.modules["torchvision.transforms.functional"] = tvf
    sys.modules["torchvision.transforms"].functional = tvf
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]      # `import a.b.c as x` walks attributes
    sys.modules.pop("anakin.datasets.hodata", None)
    from anakin.datasets.hodata import HOdata
    from anakin.utils import img_augment

    class _NP:                                    # NumPy < 2 semantics of np.uint8(negative float) (img_augment.py:187): C cast + wrap
        def __getattr__(self, k):
            return getattr(np, k)

        class uint8(np.uint8):
            def __new__(cls, v=0):
                return np.uint8(int(v) & 0xFF)
    img_augment.np = _NP()
    from torch.distributions.normal import Normal
    from torch.distributions.uniform import Uniform
    W, H, res = 192, 160, 64
    rng = np.random.default_rng(seed)
    n = 3
    yy, xx = np.mgrid[0:H, 0:W]
    frames = np.stack([np.stack([(xx * (3 + i) + yy * 2 + 40 * np.sin(xx / (5.0 + c) + i) + 30 * rng.standard_normal((H, W))) % 256
                                 for c in range(3)], -1) for i in range(n)]).astype(np.uint8)
    K = np.array([[150.0, 0, 96.0], [0, 150.0, 80.0], [0, 0, 1.0]], np.float32)
    ann = []
    for i in range(n):
        j3 = (rng.uniform(-0.06, 0.06, (21, 3)) + [0.01 * i, 0.0, 0.5]).astype(np.float32)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = po_aa(rng.standard_normal(3)); T[:3, 3] = [0.02, -0.01, 0.52]
        can = (rng.uniform(0.03, 0.06, 3) * np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])).astype(np.float32)
        c3 = (T[:3, :3] @ can.T).T + T[:3, 3]
        ann.append(dict(j3=j3, c3=c3.astype(np.float32), can=can, T=T, side="left" if i == 1 else "right", obj_idx=3 + i))

    class Fake(HOdata):
        def __init__(self, **cfg):
            super().__init__(**cfg)
            self.raw_size = (W, H)
        get_sample_idxs = lambda self: list(range(n))                                   # noqa: E731
        get_image = lambda self, idx: Image.fromarray(frames[idx])                      # noqa: E731
        get_cam_intr = lambda self, idx: K.copy()                                       # noqa: E731
        get_joints_3d = lambda self, idx: ann[idx]["j3"].copy()                         # noqa: E731
        get_joints_2d = lambda self, idx: HOdata.persp_project(ann[idx]["j3"], K)       # noqa: E731
        get_corners_3d = lambda self, idx: ann[idx]["c3"].copy()                        # noqa: E731
        get_corners_2d = lambda self, idx: HOdata.persp_project(ann[idx]["c3"], K)      # noqa: E731
        get_corners_can = lambda self, idx: ann[idx]["can"].copy()                      # noqa: E731
        get_obj_transf = lambda self, idx: ann[idx]["T"].copy()                         # noqa: E731
        get_obj_idx = lambda self, idx: ann[idx]["obj_idx"]                             # noqa: E731
        get_sides = lambda self, idx: ann[idx]["side"]                                  # noqa: E731

        def get_center_scale_wrt_bbox(self, idx):                                       # CROP_MODEL root_obj, ho3d.py:318-324
            all2d = np.concatenate([self.get_joints_2d(idx)[[0]], self.get_corners_2d(idx)], 0)
            return HOdata.get_annot_center(all2d), HOdata.get_annot_scale(all2d)
    for name in ("get_image_path get_hand_verts_3d get_hand_verts_2d get_hand_faces get_obj_faces get_obj_verts_transf "
                 "get_obj_verts_2d get_obj_verts_can get_sample_identifier").split():
        setattr(Fake, name, lambda self, idx: None)
    Fake.__abstractmethods__ = frozenset()
    cfg = dict(DATA_ROOT="", DATA_SPLIT="train", AUG=True, AUG_PARAM={"SCALE_JIT": 0.1, "CENTER_JIT": 0.1, "MAX_ROT": 0.2},
               DATA_PRESET={"USE_CACHE": False, "FILTER_NO_CONTACT": False, "FILTER_THRESH": 0.0, "BBOX_EXPAND_RATIO": 1.2,
                            "CROP_MODEL": "root_obj", "FULL_IMAGE": False, "IMAGE_SIZE": [res, res], "CENTER_IDX": 0})
    ds = Fake(**cfg)
    out = {"frames": frames, "K": K, "res": res}
    for i in range(n):
        for k, v in ann[i].items():
            out[f"ann{i}.{k}"] = np.asarray(v)
        out[f"ann{i}.j2"], out[f"ann{i}.c2"] = ds.get_joints_2d(i), ds.get_corners_2d(i)
        c, s = ds.get_center_scale_wrt_bbox(i)
        out[f"ann{i}.bbox_center"], out[f"ann{i}.bbox_scale"] = np.asarray(c), np.asarray(s)
        torch.manual_seed(seed + i); random.seed(seed + i)
        sample = ds[i]
        # replay the draws (hodata.py:350-358,436-441; img_augment.py:6-46) from the same seeds
        torch.manual_seed(seed + i); random.seed(seed + i)
        cj = Uniform(low=-1, high=1).sample((2,)).numpy()
        sj = Normal(0, 0.1 / 3.0).sample().item()
        rot = Uniform(low=-0.2 * np.pi, high=0.2 * np.pi).sample().item()
        blur = Uniform(low=0, high=1).sample().item() * 0.1
        b_, c_, s_, h_ = (random.uniform(0.9, 1.1), random.uniform(0.9, 1.1), random.uniform(0.9, 1.1), random.uniform(-0.075, 0.075))
        ops = [0, 1, 2, 3]                       # apply_jitter's list order: brightness, saturation, hue, contrast
        random.shuffle(ops)
        fac = {0: b_, 1: s_, 2: h_, 3: c_}
        out[f"draw{i}.center"], out[f"draw{i}.scale"], out[f"draw{i}.rot"], out[f"draw{i}.blur"] = cj, np.array(sj), np.array(rot), np.array(blur)
        out[f"draw{i}.order"], out[f"draw{i}.factor"] = np.array(ops, np.int32), np.array([fac[o] for o in ops], np.float32)
        for k, v in sample.items():
            v = v.numpy() if torch.is_tensor(v) else np.asarray(v)
            if k == "image":
                v = np.round((v + 0.5) * 255).astype(np.uint8)          # exact: the tensor is k/255 - 0.5
            out[f"sample{i}.{k}"] = v
    np.savez_compressed(os.path.join(OUT, "real_sample.npz"), **out)
    print("wrote real_sample", os.path.getsize(os.path.join(OUT, "real_sample.npz")), sorted(k for k in out if k.startswith("sample0")))


def po_aa(aa):
    import pose_oracle as po
    return po.aa_to_rotmat(np.asarray(aa, np.float64)).astype(np.float32)


def gen_eval_metrics(seed=31):
    """Reference eval metrics on seeded predictions: PCK family (pckmetric.py), Mean2DEPE (meanepe.py), AR / MSSD (bopAR.py),
    ValMetricAR2 (val_metric.py) and HOSubmitEpochPass.dump_json / get_order_idxs (hodata_submit_epoch_pass.py)."""
    import json
    import tempfile
    ref_import.load_control_plane()
    from anakin.metrics.pckmetric import Hand3DPCKMetric, Obj2DPCKMetric
    from anakin.metrics.meanepe import Mean2DEPE
    from anakin.metrics.bopAR import AR
    from anakin.metrics.val_metric import ValMetricAR2
    g = torch.Generator().manual_seed(seed)
    B = 12
    r = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
    targs = {"joints_3d": 0.05 * r(B, 21, 3), "joints_2d": 100 + 20 * r(B, 21, 2), "corners_3d": 0.05 * r(B, 8, 3),
             "corners_2d": 100 + 20 * r(B, 8, 2), "joints_vis": (torch.rand(B, 21, generator=g) > 0.2).float(),
             "corners_vis": (torch.rand(B, 8, generator=g) > 0.2).float(), "root_joint": 0.5 + 0.05 * r(B, 3),
             "corners_can": 0.05 * r(B, 8, 3), "obj_idx": torch.randint(1, 4, (B,), generator=g),
             "is_synth": torch.rand(B, generator=g) > 0.3, "obj_id": torch.randint(0, 3, (B,), generator=g),
             "persp_id": torch.randint(0, 288, (B,), generator=g), "grasp_id": torch.randint(0, 50, (B,), generator=g)}
    T = torch.eye(4).repeat(B, 1, 1)
    T[:, :3, :3] = torch.from_numpy(po_aa(r(B, 3).numpy()))
    T[:, :3, 3] = 0.5 + 0.05 * r(B, 3)
    targs["obj_transf"] = T
    preds = {"joints_3d": targs["joints_3d"] + 0.015 * r(B, 21, 3), "joints_2d": targs["joints_2d"] + 6 * r(B, 21, 2),
             "corners_3d": targs["corners_3d"] + 0.015 * r(B, 8, 3), "corners_2d": targs["corners_2d"] + 6 * r(B, 8, 2),
             "box_rot_rotmat": torch.from_numpy(po_aa(r(B, 3).numpy())), "boxroot_3d_abs": (T[:, :3, 3] + 0.01 * r(B, 3))[:, None],
             "joints_3d_abs": targs["joints_3d"] + targs["root_joint"][:, None] + 0.01 * r(B, 21, 3)}
    preds["corners_3d_abs"] = (T[:, :3, :3] @ targs["corners_can"].transpose(1, 2)).transpose(1, 2) + T[:, None, :3, 3] + 0.01 * r(B, 8, 3)
    info = {"1": {}, "2": {"symmetries_discrete": [[-1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]]},
            "3": {"symmetries_continuous": [{"axis": [0, 0, 1], "offset": [0, 0, 0]}]}}
    out = {f"t.{k}": v.numpy() for k, v in targs.items()}
    out.update({f"p.{k}": v.numpy() for k, v in preds.items()})
    out["model_info"] = np.array(json.dumps(info))
    pck_cfg = dict(VAL_MIN=0.0, VAL_MAX=0.05, STEPS=20)
    for name, cls, cfg in (("hand3d", Hand3DPCKMetric, pck_cfg), ("obj2d", Obj2DPCKMetric, dict(VAL_MIN=0.0, VAL_MAX=30.0, STEPS=15))):
        m = cls(**cfg)
        m.feed(preds, targs); m.feed(preds, targs)
        for k, v in m.get_measures().items():
            out[f"{name}.{k}"] = np.asarray(v)
        out[f"{name}.pck_all"] = np.asarray(m.get_pck_all(0.02 if name == "hand3d" else 10.0))
    m = Mean2DEPE(VAL_KEYS=["joints_2d", "corners_2d"], MILLIMETERS=True)
    m.feed(preds, targs)
    out["mean2d"] = np.array([m.get_measures()[k] for k in ("joints_2d_mepe", "corners_2d_mepe")])
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(info, f)
    ar_cfg = dict(USE_MSSD=True, MODEL_INFO_PATH=f.name, MAX_SYM_DISC_STEP=0.25, MSSD_USE_CORNERS=True, DATA_PRESET={"CENTER_IDX": 0})
    for tag, extra in (("ar", {}), ("ar_center", {"MSSD_USE_CENTER_IDX": True}), ("ar_ycb", {"USE_HO3D_YCB": True})):
        a = AR(**dict(ar_cfg, **extra))
        a.feed(preds, targs)
        meas = a.get_measures()
        out[f"{tag}.keys"] = np.array(sorted(meas))
        out[f"{tag}.vals"] = np.array([meas[k] for k in sorted(meas)])
    v = ValMetricAR2(**ar_cfg)
    v.feed(preds, targs)
    avg = v.get_measures_averaged()
    out["val_ar2.ids"] = np.array(sorted(avg))
    out["val_ar2.vals"] = np.array([avg[k] for k in sorted(avg)])
    # submission file
    import types
    sys.modules.setdefault("anakin.submit", types.ModuleType("anakin.submit")).__path__ = [f"{ref_import.REF_ROOT}/anakin/submit"]
    try:
        from anakin.submit.hodata_submit_epoch_pass import HOSubmitEpochPass
        with tempfile.TemporaryDirectory() as d:
            xyz = [x.numpy() for x in preds["joints_3d_abs"][:3]]
            vts = [np.zeros((778, 3))] * 3
            HOSubmitEpochPass.dump_json(None, os.path.join(d, "pred.json"), xyz, vts, codalab=False)
            out["submit.json"] = np.array(open(os.path.join(d, "pred.json")).read())
        out["submit.reorder"], out["submit.unorder"] = (np.asarray(x) for x in HOSubmitEpochPass.get_order_idxs())
    except Exception as e:      # heavy import chain (fitting unit, viz): record why and keep the metric goldens
        print("submit pass not importable:", repr(e)[:200])
    np.savez_compressed(os.path.join(OUT, "eval_metrics.npz"), **out)
    print("wrote eval_metrics", sorted(k for k in out if not k.startswith(("t.", "p."))))


def gen_refiner(B=3, seed=4, n_iters=3):
    """HORefiner.forward of the real reference (refiner.py:181-224) on seeded grasps, weights from
    refiner_oracle.fill_params, third-party stand-ins as documented in ref_import.load_refiner."""
    import refiner_oracle as rfo
    sys.path.insert(0, os.path.join(HERE, ".."))
    from artiboost_amd.assets import SceneAssets, resample_objects
    assets = SceneAssets("HO3D", seed=1)
    R = ref_import.load_refiner(assets.hand)
    net = R._RefineNet(n_iters=n_iters)
    params = rfo.fill_params(seed)
    sd = net.state_dict()
    for k in sd:
        sd[k] = params[k].clone()
    net.load_state_dict(sd)
    net.eval()
    ref = R.HORefiner.__new__(R.HORefiner)               # __init__ would torch.load the GrabNet checkpoint (a download)
    torch.nn.Module.__init__(ref)
    ref.refine_net = net
    pts = resample_objects(assets, 10000, seed=7)
    ref.obj_idx = {f"obj{i}": i for i in range(assets.n_obj)}
    ref.register_buffer("resampled_objs_buffer", torch.from_numpy(pts))
    pose, tsl, rot, oi = rfo.make_inputs(assets, B, seed)
    out = {"in.hand_pose": pose, "in.hand_tsl": tsl, "in.obj_rot": rot, "in.obj_idx": oi}
    with torch.no_grad():
        res = ref({"hand_pose": torch.from_numpy(pose), "hand_tsl": torch.from_numpy(tsl), "obj_rot": torch.from_numpy(rot)},
                  [f"obj{i}" for i in oi])
        # pieces: one ResBlock, CRot2rotmat, the first-iteration distances
        x = torch.from_numpy(np.random.default_rng(seed).standard_normal((B, rfo.IN_SIZE)).astype(np.float32))
        out["rb1.in"], out["rb1.out"] = x.numpy(), net.rb1(x).numpy().copy()
        c6 = torch.from_numpy(np.random.default_rng(seed + 1).standard_normal((B * 16, 6)).astype(np.float32))
        out["crot.in"], out["crot.out"] = c6.numpy(), R.CRot2rotmat(c6).numpy().copy()
        verts = net.mano_layer(torch.from_numpy(pose)).verts + torch.from_numpy(tsl)[:, None]
        vo = torch.transpose(torch.bmm(torch.from_numpy(rot), torch.transpose(torch.from_numpy(pts[oi]), -2, -1)), -2, -1)
        out["h2o.first"] = R.point2point_signed(verts, vo).numpy().copy()
    for k, v in res.items():
        out["out." + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "refiner.npz"), **out)
    print("wrote refiner", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if "--refiner" in sys.argv:
        gen_refiner()
        sys.exit(0)
    if "--blacklist" in sys.argv:
        gen_blacklist()
        sys.exit(0)
    if "--state" in sys.argv:
        gen_state_files()
        sys.exit(0)
    if "--real" in sys.argv:
        gen_real_sample()
        sys.exit(0)
    if "--eval" in sys.argv:
        gen_eval_metrics()
        sys.exit(0)
    gen_head_only()
    gen_misc()
    gen_learner("g224", 224, 28, 28, B=2, seed=1)
    gen_learner("g256", 256, 32, 28, B=2, seed=2)
    gen_refiner()
    gen_blacklist()
    gen_state_files()
    gen_real_sample()
    gen_eval_metrics()

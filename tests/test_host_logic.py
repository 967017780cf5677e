"""CPU: host-side logic of the package (registry/config, parameter layout conversion, CCV bookkeeping, GT geometry,
rotations) against the reference's golden vectors and the oracle; the N>1 data-parallel pieces under gloo, world 2."""
import os
import sys

import numpy as np
import pytest
import torch
import yaml

import learner_oracle as lo
import pose_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    return yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))


def test_registry_builds_reference_yaml_schema():
    from artiboost_amd import registry as R
    import artiboost_amd.criterions  # noqa: F401
    import artiboost_amd.metrics  # noqa: F401
    cfg = _cfg()
    losses = R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"])
    assert [type(l).__name__ for l in losses] == ["JointsLoss", "HandOrdLoss", "SceneOrdLoss"]
    metrics = R.build_evaluator_metric_list(cfg["EVALUATOR"], preset_cfg=cfg["DATA_PRESET"])
    assert [type(m).__name__ for m in metrics] == ["LossesMetric", "Mean3DEPE", "ValMetricMean3DEPE2"]
    with pytest.raises(KeyError):
        R.build_from_cfg({"TYPE": "Nope"}, R.LOSS)


def test_criterion_cpu_matches_reference_golden(golden_dir):
    """The plugin-level losses are device-agnostic torch code: check them on CPU against the reference's golden."""
    import random
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from gen_batch import make_batch
    g = np.load(os.path.join(golden_dir, "learner_g224.npz"))
    size, heat, depth, B, seed = [int(x) for x in g["meta"]]
    cfg = _cfg()
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    batch = make_batch(B, size, seed + 100)
    preds = {"joints_3d_abs": torch.from_numpy(g["train.pred.joints_3d_abs"]), "corners_3d_abs": torch.from_numpy(g["train.pred.corners_3d_abs"])}
    random.seed(seed + 7); torch.manual_seed(seed + 7)
    total, losses = crit.compute_losses(preds, batch)
    for k in ("joints_3d_loss", "corners_3d_loss", "joint_ord_loss", "part_ord_loss", "scene_ord_loss", "final_loss"):
        np.testing.assert_allclose(losses[k].numpy().reshape(-1), g[f"loss.{k}"].reshape(-1), rtol=1e-4, atol=1e-8, err_msg=k)


def test_param_store_roundtrip_and_layouts():
    from artiboost_amd.hybridnet import ParamStore
    st = ParamStore(22, 28, device="cpu")
    params = lo.fill_params(lo.param_shapes(22, 28), seed=3)
    st.load_reference_state_dict({"_model_list.0." + k: v for k, v in params.items()})
    back = st.reference_state_dict()
    # the reference's resume is load_arch(strict=True): same keys in the same order, <bn>.num_batches_tracked (int64 scalars) included
    assert list(back) == [k for k, _ in lo.param_shapes(22, 28)]
    nbt = max(int(v) for k, v in params.items() if k.endswith("num_batches_tracked"))
    for k, v in params.items():
        if k.endswith("num_batches_tracked"):
            assert back[k].dtype == torch.int64 and back[k].shape == () and int(back[k]) == nbt == st.num_batches_tracked
            continue
        np.testing.assert_array_equal(back[k].numpy(), v.numpy(), err_msg=k)
    # layout spot checks: OHWI, padded stem, padded depth, padded box head
    w = params["backbone.layer2.0.conv1.weight"]
    np.testing.assert_array_equal(st.view("backbone.layer2.0.conv1.weight")[5, 2, 1, 7].item(), w[5, 7, 2, 1].item())
    stem = st.view("backbone.conv1.weight")
    assert stem.shape == (64, 7, 8, 4) and float(stem[:, :, 7].abs().max()) == 0 and float(stem[..., 3].abs().max()) == 0
    fw = st.view("hybrid_head.final_layer.weight").reshape(22, 32, 256)
    assert float(fw[:, 28:].abs().max()) == 0
    np.testing.assert_array_equal(fw[3, 5].numpy(), params["hybrid_head.final_layer.weight"][3 * 28 + 5, :, 0, 0].numpy())
    assert st.total % 64 == 0 and st.trainable_numel() < st.total
    with pytest.raises(ValueError):
        st.load_reference_state_dict({"backbone.conv1.weight": torch.zeros(64, 3, 3, 3)}, strict=False)


def test_param_store_variants_speak_the_reference_keys():
    """SimpleBaseline on ResNet-18 (simplebaseline.py:194-241, resnet.py:236-241): `pose_head.*` names, no box head, BasicBlock stage counts
    (2, 2, 2, 2) -- the key list (order included) is the one the reference's own class produced when oracle/gen_simplebaseline_golden.py
    ran (asserted there against lo.param_shapes)."""
    from artiboost_amd.hybridnet import ParamStore
    st = ParamStore(29, 28, device="cpu", layers=(2, 2, 2, 2), head_prefix="pose_head", box_head=False)
    want = lo.param_shapes(29, 28, layers=(2, 2, 2, 2), head_prefix="pose_head", box_head=False)
    sd = st.reference_state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, tuple(s)) for k, s in want]
    assert st.nclasses_pad == 30 and st.view("pose_head.final_layer.weight").shape == (960, 1, 1, 256)
    params = lo.fill_params(want, seed=2)
    st.load_reference_state_dict(params)
    back = st.reference_state_dict()
    for k, v in params.items():
        if v.dtype.is_floating_point:
            np.testing.assert_array_equal(back[k].numpy(), v.numpy(), err_msg=k)
    assert float(st.view("pose_head.final_layer.weight").reshape(30, 32, 256)[29].abs().max()) == 0.0      # the padding class


def test_gt_geometry_and_views_match_reference(golden_dir):
    from artiboost_amd import synth
    g = np.load(os.path.join(golden_dir, "misc.npz"))
    for i in range(len(g["affine.scale"])):
        res = [224, 224] if i % 2 else [256, 256]
        tot, post = synth.get_affine_transform(g["affine.center"][i], float(g["affine.scale"][i]), [256.0, 256.0], res, float(g["affine.rot"][i]))
        np.testing.assert_allclose(tot, g["affine.total"][i], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(post, g["affine.post"][i], rtol=1e-6, atol=1e-5)
    for v, m in zip(g["view.vecs"], g["view.align"]):
        np.testing.assert_allclose(synth.align_mat(v.copy()), m, rtol=1e-9, atol=1e-12)
    # assemble_gt (product) == pose_oracle.assemble_sample_gt (restatement of rendered_dataset.py:155-254)
    rng = np.random.default_rng(0)
    K = np.array([[435.0, 0, 256.0], [0, 435.0, 256.0], [0, 0, 1.0]])
    for _ in range(5):
        joints = np.array([0.0, 0.0, 0.5]) + 0.05 * rng.standard_normal((21, 3))
        pose = np.eye(4); pose[:3, :3] = po.aa_to_rotmat(rng.standard_normal(3)); pose[:3, 3] = [0.02, -0.01, 0.5]
        can = np.array([[sx, sy, sz] for sx in (-.04, .04) for sy in (-.06, .06) for sz in (-.03, .03)])
        d = dict(center=rng.uniform(-1, 1, 2), scale=rng.normal(0, 0.03), rot=rng.uniform(-0.6, 0.6))
        ref = po.assemble_sample_gt(K, joints, pose, can, [256, 256], d, raw_size=[512, 512])
        got = synth.assemble_gt(K, joints, pose, can, [256, 256], [512, 512], d["center"], d["scale"], d["rot"])
        for k in ("cam_intr", "root_joint", "joints_3d", "joints_2d", "joints_vis", "corners_3d", "corners_2d", "corners_vis", "obj_transf"):
            np.testing.assert_allclose(got[k], ref[k], rtol=1e-6, atol=1e-6, err_msg=k)


def test_rotations_vs_scipy_and_oracle():
    from scipy.spatial.transform import Rotation
    from artiboost_amd import synth
    rng = np.random.default_rng(1)
    aa = rng.standard_normal((64, 3)) * rng.uniform(0.01, 3.0, (64, 1))
    aa[0] = 0
    R = synth.aa_to_rotmat(torch.from_numpy(aa)).numpy()
    np.testing.assert_allclose(R, Rotation.from_rotvec(aa).as_matrix(), atol=1e-12)
    np.testing.assert_allclose(R, po.aa_to_rotmat(aa), atol=1e-12)
    back = synth.rotmat_to_aa(torch.from_numpy(R)).numpy()
    np.testing.assert_allclose(Rotation.from_rotvec(back).as_matrix(), R, atol=1e-9)
    np.testing.assert_allclose(back, po.rotmat_to_aa(R), atol=1e-9)


def test_ccv_update_methods_match_reference(golden_dir):
    from artiboost_amd.synth import ArtiBoostLoader as AL
    g = np.load(os.path.join(golden_dir, "misc.npz"))
    ids = [tuple(int(x) for x in r) for r in g["ccv.ids"]]
    res = dict(zip(ids, g["ccv.vals"].tolist()))
    for m in (1, 2, 3):
        fn = getattr(AL, f"update_method_{m}")
        out = fn(torch.ones(4, 288, 50), res, 0.1, 10.0, dist_lower_threshold=8.0, dist_upper_threshold=16.0, epoch_idx=3, n_epochs=10)
        np.testing.assert_allclose(np.array([float(out["sample_weight_map"][i]) for i in res]), g[f"ccv.m{m}"], rtol=1e-6)


def test_metrics_epe_and_val_metric():
    from artiboost_amd.metrics import Mean3DEPE, ValMetricMean3DEPE2
    g = torch.Generator().manual_seed(0)
    B = 6
    preds = {"joints_3d_abs": torch.randn(B, 21, 3, generator=g), "corners_3d_abs": torch.randn(B, 8, 3, generator=g)}
    targs = {"joints_3d": torch.randn(B, 21, 3, generator=g), "corners_3d": torch.randn(B, 8, 3, generator=g),
             "root_joint": torch.randn(B, 3, generator=g), "is_synth": torch.tensor([1, 1, 0, 1, 1, 1], dtype=torch.bool),
             "obj_id": torch.tensor([0, 1, -1, 1, 2, 3]), "persp_id": torch.tensor([5, 6, -1, 6, 7, 8]), "grasp_id": torch.tensor([1, 2, -1, 2, 3, 4])}
    m = Mean3DEPE(VAL_KEYS=["corners_3d_abs", "joints_3d_abs"], MILLIMETERS=True)
    m.feed(preds, targs)
    ref = lo.mean_epe_mm(preds["joints_3d_abs"], targs["joints_3d"], targs["root_joint"])
    np.testing.assert_allclose(m.get_measures()["joints_3d_abs_mepe"], float(ref.mean()), rtol=1e-5)
    vm = ValMetricMean3DEPE2(VAL_KEYS=["corners_3d_abs", "joints_3d_abs"], MILLIMETERS=True)
    vm.feed(preds, targs)
    avg = vm.get_measures_averaged()
    assert set(avg) == {(0, 5, 1), (1, 6, 2), (2, 7, 3), (3, 8, 4)}          # real sample dropped, duplicate key: last write wins
    refc = lo.mean_epe_mm(preds["corners_3d_abs"], targs["corners_3d"], targs["root_joint"])
    np.testing.assert_allclose(avg[(1, 6, 2)], 0.5 * (float(ref[3]) + float(refc[3])), rtol=1e-5)


# ------------------------------------------------------------------ N > 1 (gloo, world_size 2, CPU)
def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from artiboost_amd.assets import SceneAssets
    from artiboost_amd.synth import ArtiBoostLoader
    from artiboost_amd.train import allreduce_flat_
    # (1) bucketed gradient averaging == mean of the per-rank gradients
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    allreduce_flat_(g, world, bucket_elems=256)
    ok1 = torch.allclose(g, torch.arange(1000, dtype=torch.float32) * 1.5)
    # (2) epoch sharding: same seed on every rank, disjoint slices, union == single-process epoch
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    assets = SceneAssets.__new__(SceneAssets)      # planning needs only a few fields; skip mesh generation
    assets.n_obj, assets.hand_tex, assets.backgrounds = 4, np.zeros((51, 1, 1, 3), np.uint8), np.zeros((16, 768, 1, 3), np.uint8)
    assets.hand = None
    ld = ArtiBoostLoader.from_assets(assets, cfg["MANAGER"], cfg["DATA_PRESET"], 4, 40, device="cpu", random_seed=7, rank=rank, world_size=world,
                         grasps=(None, None, None))
    plan = ld.plan_epoch()
    ld1 = ArtiBoostLoader.from_assets(assets, cfg["MANAGER"], cfg["DATA_PRESET"], 4, 40, device="cpu", random_seed=7, rank=0, world_size=1,
                          grasps=(None, None, None))
    full = ld1.plan_epoch()
    idx = plan["global_index"]
    ok2 = (np.array_equal(plan["o"], full["o"][idx]) and np.array_equal(plan["aug"]["rot"], full["aug"]["rot"][idx])
           and np.array_equal(idx, np.arange(40)[rank::world]))
    gathered = [None] * world
    dist.all_gather_object(gathered, idx.tolist())
    ok3 = sorted(sum(gathered, [])) == list(range(40))
    # (3) mining results: every rank ends with the single-process dict (highest global index wins on shared triplets)
    o_, v_, g_ = full["o"], full["v"], full["g"]
    o_[7], v_[7], g_[7] = o_[2], v_[2], g_[2]                 # a triplet seen by both ranks (global samples 2 and 7)
    vals = np.arange(40, dtype=np.float64) + 0.5               # "error" of global sample i
    single = {}
    for i in range(40):
        single[(int(o_[i]), int(v_[i]), int(g_[i]))] = vals[i]
    local = {}
    ld._last_seen = {}
    for i in idx:
        t = (int(o_[i]), int(v_[i]), int(g_[i]))
        local[t] = vals[i]
        ld._last_seen[t] = int(i)
    merged = ld.gather_ccv_results(local)
    ok4 = merged == single and ld1.gather_ccv_results(single) is single
    q.put((rank, bool(ok1), bool(ok2), bool(ok3 and ok4)))
    dist.destroy_process_group()


def test_ddp_pieces_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True, True, True), (1, True, True, True)], res


def test_assemble_gt_batch_equals_per_sample_assembly():
    """The epoch-vectorised GT assembly == RenderedDataset.__getitem__'s per-sample arithmetic (assemble_gt, itself checked
    against the reference golden above), including samples pushed out of the crop (visibility masks)."""
    from artiboost_amd import synth
    from artiboost_amd.registry import Queries
    rng = np.random.default_rng(3)
    S = 64
    K = np.array([[435.0, 0, 256.0], [0, 435.0, 256.0], [0, 0, 1.0]])
    joints = rng.uniform(-0.08, 0.08, (S, 21, 3)) + [0.0, 0.0, 0.5]
    joints[::7, :, 0] += 0.5                                   # far off-centre: raw / cropped visibility rules kick in
    pose = np.tile(np.eye(4), (S, 1, 1))
    pose[:, :3, :3] = po.aa_to_rotmat(rng.standard_normal((S, 3)))
    pose[:, :3, 3] = rng.uniform(-0.05, 0.05, (S, 3)) + [0.0, 0.0, 0.5]
    pose[::5, 0, 3] += 0.6
    can = rng.uniform(0.03, 0.08, (S, 1, 3)) * np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])[None]
    cj, sj, rot = rng.uniform(-1, 1, (S, 2)), rng.normal(0, 0.1 / 3, S), rng.uniform(-0.2 * np.pi, 0.2 * np.pi, S)
    got = synth.assemble_gt_batch(K, joints, pose, can, [256, 256], [512, 512], cj, sj, rot)
    nz = 0
    for i in range(S):
        ref = synth.assemble_gt(K, joints[i], pose[i], can[i], [256, 256], [512, 512], cj[i], sj[i], rot[i])
        for k, v in ref.items():
            np.testing.assert_allclose(got[k][i], v, rtol=1e-6, atol=1e-5, err_msg=f"{i}.{k}")
        nz += int(ref[Queries.JOINTS_VIS].sum() == 0) + int(ref[Queries.CORNERS_VIS].sum() == 0)
    assert nz > 0                                              # the all-zero visibility branches were exercised


def test_bench_gpus_flag_launches_the_ranks():
    """`python bench.py --gpus 2` is the whole multi-GPU command: it starts 2 ranks itself (torch.distributed.run, gloo here)
    and the line reports the world size the ranks actually formed -- not the flag."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert lines == [{"dry_launch": True, "n_gpus": 2, "gpus_arg": 2}]


def test_dropin_surface_matches_reference_signatures():
    """The constructor keywords of train/train_artiboost.py:175-190 and the import paths of :9-22 exist (the alias package
    `anakin` resolves them to artiboost_amd)."""
    import inspect
    import sys
    from unittest import mock
    with mock.patch.object(sys, "argv", ["x"]):
        for m in [k for k in sys.modules if k in ("anakin.opt", "anakin.opt_extra")]:
            del sys.modules[m]
        from anakin.artiboost import ArtiBoostLoader
        from anakin.criterions.criterion import Criterion  # noqa: F401
        from anakin.datasets.hodata import ho_collate
        from anakin.metrics.evaluator import Evaluator  # noqa: F401
        from anakin.models.arch import Arch  # noqa: F401
        from anakin.opt import arg, cfg  # noqa: F401
        from anakin.opt_extra import data_generation_manager_parse
        from anakin.utils import builder
        from anakin.utils.etqdm import etqdm  # noqa: F401
        from anakin.utils.logger import logger  # noqa: F401
        from anakin.utils.misc import CONST, TrainMode  # noqa: F401
        from anakin.utils.netutils import build_optimizer, build_scheduler
        from anakin.utils.recorder import Recorder  # noqa: F401
        from anakin.utils.summarizer import Summarizer  # noqa: F401
        assert data_generation_manager_parse().ovg_batch_size == 256
    names = list(inspect.signature(ArtiBoostLoader.__init__).parameters)[1:15]
    assert names == ["real_train_set", "arg", "arg_extra", "cfg", "cfg_dataset", "cfg_preset", "time_f", "batch_size", "shuffle",
                     "num_workers", "pin_memory", "drop_last", "collate_fn", "random_seed"]
    # datasets: the reference YAML's TRAIN entry builds (empty: ./data is a download)
    ds = builder.build_dataset({"TYPE": "HO3D", "DATA_SPLIT": "train", "DATA_ROOT": "/nonexistent", "AUG": True}, preset_cfg={})
    assert len(ds) == 0
    b = ho_collate([{"a": np.zeros((2, 3), np.float32), "i": 1}, {"a": np.ones((2, 3), np.float32), "i": 2}])
    assert b["a"].shape == (2, 2, 3) and b["i"].tolist() == [1, 2]
    # optimizer / scheduler builders with **cfg["TRAIN"]
    p = torch.nn.Parameter(torch.zeros(4))
    tr = dict(OPTIMIZER="adam", LR=5e-5, WEIGHT_DECAY=0, LR_DECAY_STEP=100, LR_DECAY_GAMMA=1.0, BATCH_SIZE=128)
    opt = build_optimizer([{"params": [p]}], **tr)
    assert isinstance(opt, torch.optim.Adam) and opt.param_groups[0]["lr"] == 5e-5
    assert isinstance(build_scheduler(opt, **tr), torch.optim.lr_scheduler.StepLR)
    assert isinstance(build_optimizer([{"params": [p]}], **dict(tr, OPTIMIZER="sgd", MOMENTUM=0.9)), torch.optim.SGD)
    # a host-only loader through the reference keywords (no GPU): epoch length from SYNTH_LEN, mining maps in place
    mcfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config",
                                            "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    class A:  # noqa: E701
        device, batch_size, workers = "cpu", 4, 0
    ld = ArtiBoostLoader(ds, arg=A(), arg_extra=None, cfg=dict(mcfg["MANAGER"], SYNTH_LEN=40, EPOCH=2), cfg_dataset=mcfg["DATASET"],
                         cfg_preset=mcfg["DATA_PRESET"], time_f=0.0, batch_size=4, shuffle=True, num_workers=0, pin_memory=True,
                         drop_last=True, collate_fn=ho_collate, random_seed=1)
    assert ld.synth_len == 40 and ld.use_synth and ld.sample_weight_map.shape == (4, 288, 50) and ld.batch_size == 4


def test_dataparallel_replication_is_refused_with_instructions():
    """nn.DataParallel over > 1 device clones modules per device (torch/nn/parallel/replicate.py calls
    `_replicate_for_data_parallel`); the HIP model owns one device's flat buffers and says what to do instead."""
    from artiboost_amd.models import HybridBaseline
    with pytest.raises(RuntimeError, match="one process per GPU"):
        HybridBaseline._replicate_for_data_parallel(object.__new__(HybridBaseline))


def test_train_script_launches_one_rank_per_listed_gpu():
    """train/train_artiboost.py with the reference's `--gpu_id 0,1` starts two ranks under torch.distributed.run (the reference
    wraps the model in nn.DataParallel instead, train_artiboost.py:131,249-257); --dry-launch reports the world size and exits."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "train", "train_artiboost.py"), "--cfg",
                          os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml"), "--gpu_id", "0,1", "--gpu_render_id", "0,1",
                          "--batch_size", "16", "--dry-launch"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"dry_launch": True, "n_gpus": 2, "gpu_id": ["0", "1"]}


def test_every_type_named_by_the_reference_yamls_is_registered():
    """Every TYPE the reference's shipped YAMLs name (config/*.yaml, config_eval/*.yaml: models, backbones / heads of the
    regression model, losses, metrics, datasets) resolves in this build's registries.  Reads the reference checkout when it is
    there (the build container); skipped elsewhere."""
    import glob
    import yaml
    files = sorted(glob.glob("/root/reference/config/*.yaml") + glob.glob("/root/reference/config_eval/*.yaml"))
    if not files:
        pytest.skip("no reference checkout")
    from artiboost_amd import registry as R
    import artiboost_amd.criterions, artiboost_amd.datasets, artiboost_amd.hpregnet, artiboost_amd.metrics, artiboost_amd.models  # noqa: F401,E401
    missing = []
    for f in files:
        c = yaml.safe_load(open(f))
        archs = c["ARCH"] if isinstance(c["ARCH"], list) else [c["ARCH"]]
        for a in archs:
            if R.MODEL.get(a["TYPE"]) is None:
                missing.append((os.path.basename(f), "MODEL", a["TYPE"]))
            if a["TYPE"] == "HOPRegNet":            # its backbone / head are registry-built torch modules
                if R.BACKBONE.get(a["BACKBONE"]["TYPE"]) is None:
                    missing.append((os.path.basename(f), "BACKBONE", a["BACKBONE"]["TYPE"]))
                if R.HEAD.get(a["HEAD"]["TYPE"]) is None:
                    missing.append((os.path.basename(f), "HEAD", a["HEAD"]["TYPE"]))
        for key, reg in (("CRITERION", R.LOSS), ("EVALUATOR", R.METRIC)):
            for e in c.get(key) or []:
                if reg.get(e["TYPE"]) is None:
                    missing.append((os.path.basename(f), key, e["TYPE"]))
        for split in ("TRAIN", "TEST"):
            d = (c.get("DATASET") or {}).get(split)
            if d and R.DATASET.get(d["TYPE"]) is None:
                missing.append((os.path.basename(f), "DATASET", d["TYPE"]))
    assert not missing, missing


def test_alias_leaf_modules_resolve():
    """The deeper import paths of the reference (`from anakin.criterions.ordinal import HandOrdLoss`, ...) resolve to this build's
    classes for everything that is implemented here."""
    import importlib
    want = {"anakin.artiboost.artiboost_loader": ["ArtiBoostLoader"], "anakin.artiboost.refiner": ["Refiner", "HORefiner"],
            "anakin.criterions.jointloss": ["JointsLoss"], "anakin.criterions.ordinal": ["HandOrdLoss", "SceneOrdLoss"],
            "anakin.criterions.symcornerloss": ["SymCornerLoss"], "anakin.criterions.honetloss": ["ManoLoss"],
            "anakin.datasets.ho3d": ["HO3D", "HO3DV3"], "anakin.datasets.dexycb": ["DexYCB"], "anakin.metrics.metric": ["Metric", "AverageMeter"],
            "anakin.metrics.meanepe": ["Mean3DEPE", "Mean2DEPE"], "anakin.metrics.pckmetric": ["Hand3DPCKMetric", "Obj3DPCKMetric"],
            "anakin.metrics.val_metric": ["ValMetricMean3DEPE2", "ValMetricAR2"], "anakin.metrics.vismetric": ["Vis2DMetric", "VisMetric"],
            "anakin.metrics.lossesmetric": ["LossesMetric"], "anakin.metrics.bopAR": ["AR"], "anakin.models.hybridbaseline": ["HybridBaseline"],
            "anakin.models.hpregnet": ["HOPRegNet"], "anakin.models.mano": ["ManoBranch"], "anakin.models.resnet": ["ResNet18", "ResNet34"],
            "anakin.submit.submit_epoch_pass": ["SubmitEpochPass"], "anakin.submit.hodata_submit_epoch_pass": ["HOSubmitEpochPass"],
            "anakin.utils.transform": ["batch_uvd2xyz", "compute_rotation_matrix_from_ortho6d"]}
    for mod, names in want.items():
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)


def test_frozen_batchnorm_refuses_optimizers_that_would_move_its_buffers():
    """FrozenBatchNorm2d's affine terms are buffers in the reference (resnet.py:33-69); here they sit in the flat parameter, protected by a
    zero gradient that only the decay-free Adam honours: weight decay or SGD must be refused, not silently diverge."""
    import types
    import torch
    from artiboost_amd import netutils
    p = torch.nn.Parameter(torch.zeros(8))
    owner = types.SimpleNamespace(flat_param=p, store=types.SimpleNamespace(frozen_bn=True))
    p._ab_owner = owner
    with pytest.raises(NotImplementedError):
        netutils.build_optimizer([p], OPTIMIZER="Adam", LR=1e-4, WEIGHT_DECAY=1e-4)
    with pytest.raises(NotImplementedError):
        netutils.build_optimizer([p], OPTIMIZER="SGD", LR=1e-4)
    assert isinstance(netutils.build_optimizer([p], OPTIMIZER="Adam", LR=1e-4), torch.optim.Adam)      # CPU parameter: torch's Adam, no decay
    owner.store.frozen_bn = False
    assert isinstance(netutils.build_optimizer([p], OPTIMIZER="SGD", LR=1e-4, WEIGHT_DECAY=1e-4), torch.optim.SGD)


def test_traffic_table_counts_every_kernel_of_the_step():
    """bench.py's `roofline.traffic` comes from a committed reduction of the PMC passes (tools/pmc_traffic.py) that sums per-kernel counters
    into families BY KERNEL NAME: a kernel added to the step and not to those lists is counted nowhere (round 5: four convolution kernels
    were missing and the conv stack read 14.2 GB instead of 16.6).  Every kernel of the committed per-kernel table that moves more than 2 MB
    per step must belong to a family -- torch's own elementwise / copy kernels aside -- and the file's totals must be those of its table."""
    import csv
    import importlib.util
    import json
    import bench
    root = os.path.join(os.path.dirname(__file__), "..")
    spec = importlib.util.spec_from_file_location("pmc_traffic", os.path.join(root, "tools", "pmc_traffic.py"))
    P = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(P)
    known = sum((m for _, m in P.FAMILIES), ())
    name = bench.PMC_FILE["bf16x3"]
    summ = json.load(open(os.path.join(root, "profiles", name)))
    conv = total = 0.0
    for line in open(os.path.join(root, "profiles", name[:-5] + ".csv")):
        parts = next(csv.reader([line])) if line.startswith('"') else []
        if len(parts) != 4:
            continue
        try:
            mb = float(parts[2]) + float(parts[3])
        except ValueError:
            continue
        b = P.base(parts[0])
        if b in known:
            total += mb
            conv += mb if b in P.CONV else 0.0
        else:
            assert mb < 2.0 or b.startswith("at::") or b.startswith("__amd_rocclr"), f"{b}: {mb:.0f} MB per step in no family"
    assert abs(conv * 1e6 - summ["conv_stack_bytes_per_step"]) < 1e-3 * summ["conv_stack_bytes_per_step"]
    assert abs(total * 1e6 - summ["all_kernels_bytes_per_step"]) < 1e-3 * summ["all_kernels_bytes_per_step"]


def test_lazy_image_batch_and_plane_tag():
    """synth.LazyImageBatch makes `image` on first access only; registry.PlaneTag survives the whole-batch idioms."""
    import torch
    from artiboost_amd.registry import IMAGE_PLANE_KEY, PlaneTag, image_plane_of, tag_image_plane
    from artiboost_amd.synth import LazyImageBatch, chw_from_padded
    pad = torch.zeros(2, 8 + 6, 8 + 8, 4, dtype=torch.bfloat16)
    v = torch.randint(0, 256, (2, 8, 8, 3))
    pad[:, 3:-3, 3:-5, :3] = (2 * v - 255).to(torch.bfloat16)
    calls = []
    b = LazyImageBatch({"image_nhwc4_padded": tag_image_plane(pad, "u8n"), IMAGE_PLANE_KEY: PlaneTag("u8n")},
                       lambda: calls.append(1) or chw_from_padded(pad, "u8n"))
    assert "image" in b and "image" not in list(b) and not calls
    moved = {k: x.clone() for k, x in b.items()}                       # the usual whole-batch idioms keep the tag
    assert image_plane_of(moved, moved["image_nhwc4_padded"]) == "u8n" and moved[IMAGE_PLANE_KEY].to("cpu") == "u8n"
    assert image_plane_of({}, pad) == "u8n" and image_plane_of({}, pad.clone()) is None
    img = b["image"]
    assert calls == [1] and b.get("image") is img and "image" in list(b) and calls == [1]
    assert torch.equal(img, v.permute(0, 3, 1, 2).float() / 255.0 - 0.5)
    import pytest
    with pytest.raises(KeyError):
        b["nope"]
    assert LazyImageBatch({"a": 1}).get("image") is None


def test_evaluator_host_feeds_match_the_reference_semantics():
    """Evaluator.feed_all on host tensors (nothing to defer): Mean3DEPE's running mean in mm, the unseen-object filter on the corner
    errors (meanepe.py:62-66), LossesMetric's means with None entries skipped, ValMetricMean3DEPE2's last write per CCV triplet."""
    import types
    import numpy as np
    import torch
    from artiboost_amd.metrics import Evaluator, LossesMetric, Mean3DEPE, ValMetricMean3DEPE2
    ev = Evaluator({}, [LossesMetric(), Mean3DEPE(VAL_KEYS=["joints_3d_abs", "corners_3d_abs"], MILLIMETERS=True,
                                                   arg=types.SimpleNamespace(filter_unseen_obj_idxs=[3])),
                        ValMetricMean3DEPE2(VAL_KEYS=["joints_3d_abs"], MILLIMETERS=True)])
    g = torch.Generator().manual_seed(0)
    js, cs, nkeep = [], [], 0
    for step in range(3):
        B = 4
        tj, tc, root = torch.randn(B, 21, 3, generator=g), torch.randn(B, 8, 3, generator=g), torch.randn(B, 3, generator=g)
        pj, pc = tj + root[:, None] + 0.01 * torch.randn(B, 21, 3, generator=g), tc + root[:, None] + 0.01 * torch.randn(B, 8, 3, generator=g)
        oi = torch.tensor([1, 3, 2, 3])
        targs = {"joints_3d": tj, "corners_3d": tc, "root_joint": root, "obj_idx": oi, "obj_id": torch.tensor([0, 0, 1, 1]),
                 "persp_id": torch.tensor([5, 5, 6, 6]), "grasp_id": torch.tensor([0, 0, 0, step]), "is_synth": torch.tensor([True, True, False, True])}
        ev.feed_all({"joints_3d_abs": pj, "corners_3d_abs": pc}, targs, {"final_loss": torch.tensor(1.0 + step), "aux": None, "k": 2.0})
        js.append((1000 * (pj - tj - root[:, None])).norm(dim=2).mean(1))
        cs.append((1000 * (pc - tc - root[:, None])).norm(dim=2).mean(1)[oi != 3])
        last_j = js[-1]
    m = ev.get_measures_all()
    np.testing.assert_allclose(m["joints_3d_abs_mepe"], torch.cat(js).mean().item(), rtol=1e-6)
    np.testing.assert_allclose(m["corners_3d_abs_mepe"], torch.cat(cs).mean().item(), rtol=1e-6)
    assert m["final_loss"] == 2.0 and m["k"] == 2.0 and "aux" not in m
    table = m["joints_3d_abs"]
    assert set(table) == {(0, 5, 0), (1, 6, 0), (1, 6, 1), (1, 6, 2)}                 # the real sample (is_synth False) never enters
    np.testing.assert_allclose(table[(0, 5, 0)], last_j[1].item(), rtol=1e-6)          # later write of the same triplet wins
    assert "final_loss" in str(ev)

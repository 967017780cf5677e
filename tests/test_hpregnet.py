"""BASELINE.json configs[0]: the regression-based model (HOPRegNet: ResNet-18 + MANO branch + object TransHead) through the
submit pass on CPU, batch size 8, forward only (train/submit_reload.py:26-79, anakin/models/hpregnet.py:18-150).

Pinned by tests/golden/hpregnet.npz (oracle/gen_hpregnet_golden.py: the reference's ResNet18, TransHead and recover_object run
on seeded weights) and, for the MANO arithmetic, by the oracle that tests/golden/mano.npz pins to the reference's MANO layer."""
import json
import os

import numpy as np
import torch
import yaml

import pose_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _seeded_modules(seed):
    from artiboost_amd import hpregnet
    torch.manual_seed(seed)
    net = hpregnet.ResNet18(PRETRAINED=False, FREEZE_BATCHNORM=False)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    head = hpregnet.HOPRegNet.TransHead(512, 9)
    return net.eval(), head.eval()


def test_backbone_and_object_head_match_reference_golden(golden_dir):
    import types
    from artiboost_amd import hpregnet
    g = np.load(os.path.join(golden_dir, "hpregnet.npz"))
    net, head = _seeded_modules(int(g["seed"]))
    samples = {k: torch.from_numpy(g["sample." + k]) for k in ("cam_intr", "root_joint", "corners_can")}
    with torch.no_grad():
        feats = net(image=torch.from_numpy(g["image"]))
        self_like = types.SimpleNamespace(obj_transfhead=head)
        obj = hpregnet.HOPRegNet.recover_object(self_like, feats["res_layer4_mean"], samples)
    np.testing.assert_allclose(feats["res_layer4_mean"].numpy(), g["res_layer4_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(feats["res_layer1"][:, ::8, ::5, ::5].numpy(), g["res_layer1_sample"], rtol=1e-5, atol=1e-6)
    for k in ("obj_center", "corners_3d_abs", "obj_pred_tsl", "obj_pred_rot", "corners_2d", "box_rot_rotmat", "boxroot_3d_abs"):
        np.testing.assert_allclose(obj[k].numpy(), g["obj." + k], rtol=1e-5, atol=1e-5, err_msg=k)


def test_mano_layer_torch_matches_oracle():
    """PCA pose + shape -> MANO forward, centred on CENTER_IDX (mano.py:99-106 contract) == the pinned numpy oracle."""
    from artiboost_amd import hpregnet
    hm = hpregnet.load_hand_model(None)
    layer = hpregnet.ManoLayerTorch(hm, ncomps=15, use_pca=True, center_idx=9, flat_hand_mean=False)
    rng = np.random.default_rng(0)
    pc = (0.5 * rng.standard_normal((6, 18))).astype(np.float32)
    be = (0.5 * rng.standard_normal((6, 10))).astype(np.float32)
    pc[0] = 0
    v, j, full = layer(torch.from_numpy(pc), torch.from_numpy(be))
    full_ref = np.concatenate([pc[:, :3], hm["hands_mean"][None] + pc[:, 3:] @ hm["hands_components"][:15]], 1)
    hm0 = dict(hm, hands_mean=np.zeros(45, np.float32))
    vr, jr, _ = po.mano_lbs(hm0, full_ref, be)
    np.testing.assert_allclose(full.numpy(), full_ref, atol=1e-6)
    np.testing.assert_allclose(v.numpy(), vr - jr[:, 9:10], atol=2e-6)
    np.testing.assert_allclose(j.numpy(), jr - jr[:, 9:10], atol=2e-6)
    assert float(np.abs(j.numpy()[:, 9]).max()) == 0.0


def test_configs0_submit_pass_on_cpu_bs8(tmp_path):
    """The whole configs[0] plumbing: YAML -> registry -> Arch(HOPRegNet) -> HOSubmitEpochPass over batches of 8 on CPU ->
    evaluator measures + the HO3D CodaLab prediction file."""
    from artiboost_amd import registry as R
    from artiboost_amd import hpregnet  # noqa: F401  (registers HOPRegNet / ManoBranch / ResNet18)
    from artiboost_amd.metrics import Evaluator
    from artiboost_amd.models import Arch
    from artiboost_amd.submit import HOSubmitEpochPass
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "eval_ho3dv2_regbased_artiboost_cpu.yaml")))
    torch.manual_seed(cfg["TRAIN"]["MANUAL_SEED"])
    model = Arch(cfg, R.build_arch_model_list(cfg["ARCH"], preset_cfg=cfg["DATA_PRESET"]))
    evaluator = Evaluator(cfg, R.build_evaluator_metric_list(cfg["EVALUATOR"], preset_cfg=cfg["DATA_PRESET"]))
    assert next(model.parameters()).device.type == "cpu"
    g = torch.Generator().manual_seed(3)
    bs, nb = cfg["TRAIN"]["BATCH_SIZE"], 2
    batches = []
    for _ in range(nb):
        root = torch.tensor([0.0, 0.0, 0.6]) + 0.05 * torch.randn((bs, 3), generator=g)
        batches.append({"image": torch.rand((bs, 3, 224, 224), generator=g) - 0.5,
                        "cam_intr": torch.tensor([[617.0, 0, 112.0], [0, 617.0, 112.0], [0, 0, 1.0]]).repeat(bs, 1, 1),
                        "root_joint": root, "corners_can": 0.05 * (torch.rand((bs, 8, 3), generator=g) * 2 - 1),
                        "joints_3d": 0.05 * torch.randn((bs, 21, 3), generator=g), "corners_3d": 0.05 * torch.randn((bs, 8, 3), generator=g),
                        "joints_2d": 224 * torch.rand((bs, 21, 2), generator=g), "corners_2d": 224 * torch.rand((bs, 8, 2), generator=g),
                        "joints_vis": torch.ones(bs, 21), "corners_vis": torch.ones(bs, 8),
                        "is_synth": torch.zeros(bs, dtype=torch.bool), "obj_idx": torch.ones(bs, dtype=torch.long)})
    dump = str(tmp_path / "pred_SUBMIT.json")
    from artiboost_amd.criterions import Criterion
    crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
    assert [type(l).__name__ for l in crit.loss_list] == ["ManoLoss", "JointsLoss", "HandOrdLoss", "SceneOrdLoss"]
    assert type(model.model_list[0].base_net if hasattr(model.model_list[0], "base_net") else model.model_list[0]).__name__
    joints = HOSubmitEpochPass({"DUMP": True})(0, batches, model, criterion=crit, evaluator=evaluator, rank=0, dump_path=dump)
    assert {"mano_shape", "mano_pca_pose", "scene_ord_loss", "final_loss"} <= set(evaluator.get_measures_all_striped()["LossesMetric"])
    assert evaluator.dump_images()["Vis2DMetric"].shape == (6 * 224, 2 * 6 * 224, 3)
    assert len(joints) == bs * nb and joints[0].shape == (21, 3)
    xyz, verts = json.load(open(dump))
    assert len(xyz) == bs * nb and len(xyz[0]) == 21 and len(verts[0]) == 778
    assert os.path.exists(dump.replace(".json", ".zip"))
    meas = evaluator.get_measures_all_striped()
    assert np.isfinite(list(meas["Mean3DEPE"].values())).all()
    # the 7 output keys the metrics / submit pass read, with the reference's shapes
    out = model(batches[0])["HOPRegNet"]
    assert out["joints_3d_abs"].shape == (bs, 21, 3) and out["corners_3d_abs"].shape == (bs, 8, 3)
    assert out["hand_verts_3d"].shape == (bs, 778, 3) and out["box_rot_rotmat"].shape == (bs, 3, 3)
    np.testing.assert_allclose(out["joints_3d"][:, cfg["DATA_PRESET"]["CENTER_IDX"]].detach().numpy(), 0.0, atol=1e-7)


def test_submit_reload_script_with_the_reference_command_line(tmp_path):
    """train/submit_reload.py (the reference's command line, submit_reload.py:26-79) on CPU, batch size 8: BASELINE configs[0] by
    one command -- evaluator record, CodaLab JSON + zip."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "train", "submit_reload.py"), "--cfg",
                          os.path.join(ROOT, "config", "eval_ho3dv2_regbased_artiboost_cpu.yaml"), "--batch_size", "8", "--submit_dump",
                          "--random_frames", "12"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path),
                         env=dict(os.environ, CUDA_VISIBLE_DEVICES=""))
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("submit:")][-1]
    assert "12 frames on cpu" in line and "joints_3d_abs_mepe" in line
    exp = os.path.join(tmp_path, "exp", os.listdir(tmp_path / "exp")[0])
    js = [f for f in os.listdir(exp) if f.endswith("_SUBMIT.json")]
    assert len(js) == 1 and os.path.exists(os.path.join(exp, js[0].replace(".json", ".zip")))
    xyz, verts = json.load(open(os.path.join(exp, js[0])))
    assert len(xyz) == 12 and len(xyz[0]) == 21 and len(verts[0]) == 778
    assert os.path.exists(os.path.join(exp, "evaluations", "test_eval.txt"))


def test_reference_checkpoint_key_set_loads_strictly(tmp_path, golden_dir):
    """The reference's regbased checkpoint is read with strict=True (hpregnet.py:59-64): HOPRegNet must own exactly the reference
    module's keys (tests/golden/hpregnet_keys.json: state_dict names + shapes of the reference's own HOPRegNet, built by
    oracle/gen_hpregnet_keys.py) apart from manotorch's persisted MANO asset buffers `mano_branch.mano_layer.th_*`, which are
    dropped on load -- and a checkpoint carrying them (+ DataParallel's `module.` prefix) must load through ARCH.PRETRAINED."""
    from artiboost_amd import hpregnet
    keys = json.load(open(os.path.join(golden_dir, "hpregnet_keys.json")))
    pre = hpregnet.HOPRegNet.MANO_LAYER_PREFIX
    for bb in ("ResNet18", "ResNet34"):
        cfg = {"TYPE": "HOPRegNet", "PRETRAINED": "", "BACKBONE": {"TYPE": bb, "PRETRAINED": False, "FREEZE_BATCHNORM": False},
               "HEAD": {"TYPE": "ManoBranch", "INPUT_DIM": 512, "NCOMPS": 15, "USE_PCA": True, "USE_SHAPE": True, "MANO_ASSETS_ROOT": "assets/mano_v1_2"},
               "DATA_PRESET": {"IMAGE_SIZE": [224, 224], "CENTER_IDX": 9}}
        torch.manual_seed(0)
        net = hpregnet.HOPRegNet(**cfg)
        ours = {k: list(v.shape) for k, v in net.state_dict().items()}
        ref = {k: s for k, s in keys[bb].items() if not k.startswith(pre)}
        assert any(k.startswith(pre + "th_") for k in keys[bb])           # the reference checkpoint does carry them
        assert ours == ref, (sorted(set(ours) ^ set(ref))[:8])
        # a reference-shaped checkpoint: every reference key (MANO buffers included), `module.` prefix, seeded values
        g = torch.Generator().manual_seed(1)
        sd = {"module." + k: (torch.randn(s, generator=g) if "num_batches" not in k and "th_faces" not in k else torch.zeros(s, dtype=torch.long))
              for k, s in keys[bb].items()}
        path = str(tmp_path / f"{bb}.pth.tar")
        torch.save({"state_dict": sd}, path)
        net2 = hpregnet.HOPRegNet(**dict(cfg, PRETRAINED=path))
        for k, v in net2.state_dict().items():
            assert torch.equal(v, sd["module." + k]), k

"""Real-data half of the training mix (anakin/datasets/hodata.py:315-450, mixed_dataset.py).
Golden: tests/golden/real_sample.npz = the sample dicts produced by the REAL HOdata.__getitem__ on a stand-in subclass
serving seeded frames / annotations (oracle/gen_golden.py gen_real_sample), one of them a left hand (flip path)."""
import os

import numpy as np
import pytest
import torch

import render_oracle as ro

GOLD = os.path.join(os.path.dirname(__file__), "golden", "real_sample.npz")
GT = ("cam_intr", "root_joint", "joints_3d", "joints_2d", "joints_vis", "corners_3d", "corners_2d", "corners_vis", "corners_can", "obj_transf")



def _same_entry(a, b, msg):
    """Batch entries are device tensors -- bit-identical -- or plain tags (the IMAGE_PLANE_KEY string)."""
    if torch.is_tensor(a):
        torch.testing.assert_close(a, b, rtol=0, atol=0, msg=str(msg))
    else:
        assert a == b, msg


class GoldenSource:
    """HOdataSource over the golden frames (index modulo 3)."""
    sides = "right"

    def __init__(self):
        self.g = np.load(GOLD)
        self.raw_size = (self.g["frames"].shape[2], self.g["frames"].shape[1])
        self.n = 12

    def __len__(self):
        return self.n

    def get_image(self, idx):
        return self.g["frames"][idx % 3]

    def get_annots(self, idx):
        g, i = self.g, idx % 3
        return dict(cam_intr=g["K"], joints_3d=g[f"ann{i}.j3"], joints_2d=g[f"ann{i}.j2"], corners_3d=g[f"ann{i}.c3"],
                    corners_2d=g[f"ann{i}.c2"], corners_can=g[f"ann{i}.can"], obj_transf=g[f"ann{i}.T"], obj_idx=int(g[f"ann{i}.obj_idx"]),
                    side=str(g[f"ann{i}.side"]), bbox_center=g[f"ann{i}.bbox_center"], bbox_scale=float(g[f"ann{i}.bbox_scale"]))


def _draws(g, i):
    return dict(center=g[f"draw{i}.center"], scale=float(g[f"draw{i}.scale"]), rot=float(g[f"draw{i}.rot"]))


def _image_close(got, ref_u8):
    """got float [3,h,w] = k/255 - 0.5; ref uint8.  Exact up to nearest-neighbour ties at .0 source coordinates."""
    k = np.round((got + 0.5) * 255).astype(np.int64)
    assert np.abs((k / 255.0 - 0.5) - got).max() < 1e-6
    return (k != ref_u8.astype(np.int64)).any(axis=0).mean()


def test_gt_and_image_match_reference_getitem():
    from artiboost_amd.realdata import assemble_real_gt
    src = GoldenSource()
    g = src.g
    res = int(g["res"])
    assert [str(g[f"ann{i}.side"]) for i in range(3)] == ["right", "left", "right"]
    for i in range(3):
        r = assemble_real_gt(src.get_annots(i), [res, res], src.raw_size, _draws(g, i), center_idx=0)
        assert r["flip"] == (i == 1)
        for k in GT:
            np.testing.assert_allclose(r[k], g[f"sample{i}.{k}"], rtol=1e-5, atol=1e-5, err_msg=f"{i}.{k}")
        assert r["obj_idx"] == int(g[f"sample{i}.obj_idx"])
        inv = np.linalg.inv(np.vstack([r["affine"][:2], [0, 0, 1]]).astype(np.float64))[:2].reshape(-1)
        img = ro.augment(src.get_image(i), g[f"draw{i}.order"], g[f"draw{i}.factor"], inv, float(g[f"draw{i}.blur"]), r["flip"], res, res)
        assert _image_close(img, g[f"sample{i}.image"]) < 4e-3, i
        assert not bool(g[f"sample{i}.is_synth"]) and int(g[f"sample{i}.obj_id"]) == -1


def test_batched_real_gt_equals_per_sample_and_reference():
    from artiboost_amd.realdata import assemble_real_gt, assemble_real_gt_batch
    src = GoldenSource()
    g = src.g
    res = int(g["res"])
    idxs = [0, 1, 2, 1, 0]
    draws = dict(center=np.stack([g[f"draw{i}.center"] for i in idxs]), scale=np.array([float(g[f"draw{i}.scale"]) for i in idxs]),
                 rot=np.array([float(g[f"draw{i}.rot"]) for i in idxs]))
    out = assemble_real_gt_batch([src.get_annots(i) for i in idxs], [res, res], src.raw_size, draws, center_idx=0)
    assert out["flip"].tolist() == [False, True, False, True, False]
    for n, i in enumerate(idxs):
        ref = assemble_real_gt(src.get_annots(i), [res, res], src.raw_size, _draws(g, i), center_idx=0)
        for k in GT + ("affine",):
            np.testing.assert_allclose(out[k][n], ref[k], rtol=1e-6, atol=1e-5, err_msg=f"{n}.{k}")
            if k != "affine":
                np.testing.assert_allclose(out[k][n], g[f"sample{i}.{k}"], rtol=1e-5, atol=1e-5)
    plain = assemble_real_gt_batch([src.get_annots(i) for i in idxs], [res, res], src.raw_size, None, center_idx=0, train_split=False)
    assert (plain["joints_vis"] == 1).all() and (plain["corners_vis"] == 1).all()


@pytest.mark.gpu
def test_real_batch_on_gpu_vs_oracle_and_reference():
    from artiboost_amd.realdata import RealBatcher
    src = GoldenSource()
    g = src.g
    res = int(g["res"])
    rb = RealBatcher(src, {"IMAGE_SIZE": [res, res], "CENTER_IDX": 0, "BBOX_EXPAND_RATIO": 1.2}, compute_dtype=torch.float32)
    draws = dict(center=np.stack([g[f"draw{i}.center"] for i in range(3)]), scale=np.array([float(g[f"draw{i}.scale"]) for i in range(3)]),
                 rot=np.array([float(g[f"draw{i}.rot"]) for i in range(3)]), blur=np.array([float(g[f"draw{i}.blur"]) for i in range(3)], np.float32),
                 order=np.stack([g[f"draw{i}.order"] for i in range(3)]), factor=np.stack([g[f"draw{i}.factor"] for i in range(3)]))
    draws["blur"][2] = 0.0999            # make the blur act on one sample
    pad = torch.zeros((3, res + 6, res + 8, 4), dtype=torch.float32, device="cuda")
    b = rb.batch([0, 1, 2], draws, out_pad=pad)
    host = rb.assemble([0, 1, 2], draws)
    img = b["image"].cpu().numpy()
    for i in range(3):
        ref = ro.augment(src.get_image(i), draws["order"][i], draws["factor"][i], host["inv"][i], float(draws["blur"][i]), bool(host["flip"][i]), res, res)
        np.testing.assert_array_equal(img[i], ref)                       # HIP == CPU oracle, bit for bit (incl. flip and blur)
        if i < 2:
            assert _image_close(img[i], g[f"sample{i}.image"]) < 4e-3    # == the reference's own output
        for k in GT:
            np.testing.assert_allclose(b[k][i].cpu().numpy(), g[f"sample{i}.{k}"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(pad[:, 3:-3, 3:-5, :3].permute(0, 3, 1, 2).cpu().numpy(), img)
    assert not b["is_synth"].any() and (b["obj_id"] == -1).all() and b["obj_idx"].tolist() == [3, 4, 5]
    # no augmentation: identity chain
    rb0 = RealBatcher(src, {"IMAGE_SIZE": [res, res], "CENTER_IDX": 0}, aug=False, compute_dtype=torch.float32)
    b0 = rb0.batch([0])
    h0 = rb0.assemble([0])
    ref0 = ro.augment(src.get_image(0), h0["order"][0], h0["factor"][0], h0["inv"][0], 0.0, False, res, res)
    np.testing.assert_array_equal(b0["image"][0].cpu().numpy(), ref0)


class JpegSource(GoldenSource):
    """The golden frames as .jpg files: get_image decodes them with Pillow (the reference's path, ho3d.py:228-231), get_image_bytes hands the
    file to the device decoder."""

    def __init__(self, quality=92, subsampling=2):
        import io
        from PIL import Image
        super().__init__()
        self.files = []
        for f in self.g["frames"]:
            b = io.BytesIO()
            Image.fromarray(f).save(b, "JPEG", quality=quality, subsampling=subsampling)
            self.files.append(b.getvalue())

    def get_image(self, idx):
        import io
        from PIL import Image
        return np.asarray(Image.open(io.BytesIO(self.files[idx % 3])).convert("RGB"))

    def get_image_bytes(self, idx):
        return self.files[idx % 3]


class PillowOnly(JpegSource):
    get_image_bytes = None


@pytest.mark.gpu
@pytest.mark.parametrize("subsampling", [0, 2])
def test_real_batch_from_jpeg_files_equals_pillow_path(subsampling):
    """A source that serves .jpg FILES (decoded on the device by ab_jpeg_decode_batch) gives bit-identical batches -- augmented images
    and ground truth -- to the same source decoded with Pillow on the host (the reference's Image.open(...).convert("RGB"))."""
    pytest.importorskip("PIL")
    from artiboost_amd.realdata import RealBatcher
    a, b = JpegSource(subsampling=subsampling), PillowOnly(subsampling=subsampling)
    res = int(a.g["res"])
    cfg = {"IMAGE_SIZE": [res, res], "CENTER_IDX": 0, "BBOX_EXPAND_RATIO": 1.2}
    ra, rbb = RealBatcher(a, cfg, compute_dtype=torch.float32, seed=5), RealBatcher(b, cfg, compute_dtype=torch.float32, seed=5)
    for it in range(2):
        idxs = [0, 1, 2, 4, 8][: 5 - it]
        ha, hb = ra.assemble(idxs), rbb.assemble(idxs)           # (both consume one set of augmentation draws)
        assert ha["files"] is not None and ha["frames"] is None and hb["files"] is None
        ba, bb = ra.batch(idxs), rbb.batch(idxs)
        for k in ba:
            _same_entry(ba[k], bb[k], k)
    # a file the device decoder does not cover (progressive): the batch falls back to get_image
    import io
    from PIL import Image
    pb = io.BytesIO()
    Image.fromarray(a.g["frames"][0]).save(pb, "JPEG", progressive=True)
    a.files[0] = b.files[0] = pb.getvalue()
    assert ra.assemble([0, 1])["files"] is None and rbb.assemble([0, 1])["files"] is None
    torch.testing.assert_close(ra.batch([0, 1])["image"], rbb.batch([0, 1])["image"], rtol=0, atol=0)


@pytest.mark.gpu
def test_mixed_loader_batches():
    """MixedDataset semantics with a static split: real rows first (is_synth False, CCV ids -1), synthetic rows after."""
    import copy
    from test_gpu_synth import _loader
    from artiboost_amd.realdata import MixedLoader, RealBatcher
    from artiboost_amd.synth import ArtiBoostLoader
    src = GoldenSource()
    assets, proto = _loader(size=64)
    B = 8
    n_synth = MixedLoader.n_synth_for(B, len(src), proto.synth_len)
    synth = ArtiBoostLoader.from_assets(assets, proto.cfg, proto.preset, n_synth, proto.synth_len, compute_dtype=torch.float32, random_seed=3)
    synth.prepare()
    real = RealBatcher(src, proto.preset, compute_dtype=torch.float32)
    ml = MixedLoader(real, synth, B)
    assert ml.n_real + ml.n_synth == B and ml.n_synth == n_synth and 0 < ml.n_real < B
    seen_real = []
    nb = 0
    for batch in ml:
        nb += 1
        assert batch["image"].shape == (B, 3, 64, 64) and batch["image_nhwc4_padded"].shape == (B, 70, 72, 4)
        assert batch["is_synth"].tolist() == [False] * ml.n_real + [True] * ml.n_synth
        assert (batch["obj_id"][:ml.n_real] == -1).all() and (batch["obj_id"][ml.n_real:] >= 0).all()
        assert batch["joints_3d"].shape == (B, 21, 3) and torch.isfinite(batch["image"]).all()
        np.testing.assert_array_equal(batch["image_nhwc4_padded"][:, 3:-3, 3:-5, :3].permute(0, 3, 1, 2).cpu().numpy(), batch["image"].cpu().numpy())
        seen_real += batch["sample_idx"][:ml.n_real].tolist()
    assert nb == len(ml) and len(set(seen_real)) == len(seen_real)          # a permutation: no real sample twice per epoch
    # data-parallel sharding of the real set: disjoint slices of one shared permutation
    shards = []
    for r in range(2):
        mr = MixedLoader(real, synth, B, seed=7, rank=r, world_size=2)
        shards.append(mr.rng.permutation(mr.real_len)[r::2].tolist())
        assert len(mr) == min(len(shards[-1]) // mr.n_real, len(synth))
    assert not set(shards[0]) & set(shards[1]) and sorted(shards[0] + shards[1]) == list(range(len(src)))
    synth.synth_shutdown()
    ml.update()
    assert ml.n_real == B and ml.n_synth == 0
    assert next(iter(ml))["is_synth"].sum() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16x3", "bf16"])
def test_graph_replayed_training_step_over_mixed_batches(dtype):
    """TrainStep (hipGraph replay, renderer=None) fed with MixedLoader batches: the same step as epoch_pass over the mixed
    loader (train_artiboost.py:66-96); replay is deterministic and the loss is finite on real + synthetic rows."""
    import random
    import yaml
    from test_gpu_synth import _loader
    from artiboost_amd import registry as R
    from artiboost_amd.criterions import Criterion
    from artiboost_amd.models import Arch
    from artiboost_amd.optim import FusedClipAdam
    from artiboost_amd.realdata import MixedLoader, RealBatcher
    from artiboost_amd.synth import ArtiBoostLoader
    from artiboost_amd.train import TrainStep
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = GoldenSource()
    B, size = 8, 64
    cdt = torch.bfloat16 if dtype == "bf16" else torch.float32

    def run(use_graph):
        random.seed(5); torch.manual_seed(5); np.random.seed(5)
        assets, proto = _loader(size=size)
        cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
        cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [size, size], [size // 8, size // 8]
        n_synth = MixedLoader.n_synth_for(B, len(src), proto.synth_len)
        synth = ArtiBoostLoader.from_assets(assets, proto.cfg, cfg["DATA_PRESET"], n_synth, proto.synth_len, compute_dtype=cdt, random_seed=3)
        synth.prepare()
        ml = MixedLoader(RealBatcher(src, cfg["DATA_PRESET"], compute_dtype=cdt, seed=2), synth, B, seed=4)
        arch = dict(cfg["ARCH"], COMPUTE_DTYPE=dtype, INIT_SEED=3)
        model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
        crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
        hb = model.model_list[0]
        opt = FusedClipAdam(model.models_params, lr=1e-3, max_norm=1.0, model=hb)
        model.train()
        batches = list(ml)
        static = {k: v.clone() for k, v in batches[0].items()}
        ts = TrainStep(model, crit, opt, static, use_graph=use_graph, renderer=None)
        vals = []
        for b in batches + batches:
            _, losses, _ = ts(b)
            vals.append(losses.float().cpu().numpy().copy())
        return np.stack(vals), hb.store.flat.detach().cpu().numpy().copy()

    lg, wg = run(True)
    lg2, wg2 = run(True)
    le, we = run(False)
    assert np.isfinite(lg).all() and np.isfinite(le).all() and len(lg) >= 2
    np.testing.assert_array_equal(lg, lg2)            # replay is deterministic
    np.testing.assert_array_equal(wg, wg2)
    np.testing.assert_array_equal(lg, le)             # the capture warm-up leaves no trace: graph == eager
    np.testing.assert_array_equal(wg, we)


@pytest.mark.gpu
def test_stream_prefetcher_yields_the_loaders_batches():
    """StreamPrefetcher (next batch decoded / augmented / rendered on a side stream while the current one is consumed) hands over exactly the
    batches MixedLoader yields on its own, .jpg frames decoded on the device included; consumer work between batches does not disturb them."""
    pytest.importorskip("PIL")
    import yaml
    from test_gpu_synth import _loader
    from artiboost_amd.realdata import MixedLoader, RealBatcher, StreamPrefetcher
    from artiboost_amd.synth import ArtiBoostLoader
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    B, size = 8, 64

    def make(decode_group=1, decode_ahead=False):      # (MixedLoader's defaults: 4, True)
        src = JpegSource()
        assets, proto = _loader(size=size)
        cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
        cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [size, size], [size // 8, size // 8]
        n_synth = MixedLoader.n_synth_for(B, len(src), proto.synth_len)
        synth = ArtiBoostLoader.from_assets(assets, proto.cfg, cfg["DATA_PRESET"], n_synth, proto.synth_len, compute_dtype=torch.float32, random_seed=3)
        synth.prepare()
        return MixedLoader(RealBatcher(src, cfg["DATA_PRESET"], compute_dtype=torch.float32, seed=2), synth, B, seed=4, decode_group=decode_group, decode_ahead=decode_ahead)

    plain = [{k: v.clone() for k, v in b.items()} for b in make()]
    got = []
    junk = torch.randn(2048, 2048, device="cuda")
    for b in StreamPrefetcher(make()):
        junk = junk @ junk * 1e-3                       # the consumer's own work on the main stream while the next batch is produced
        got.append({k: v.clone() for k, v in b.items()})
    torch.cuda.synchronize()
    grouped = [{k: v.clone() for k, v in b.items()} for b in make(decode_group=3)]      # frames of three batches decoded per call
    ahead = []
    for b in make(decode_group=2, decode_ahead=True):      # ... and the next group on a side stream while this one is consumed
        junk = junk @ junk * 1e-3
        ahead.append({k: v.clone() for k, v in b.items()})
    torch.cuda.synchronize()
    # ... and the whole assembly on a worker thread two batches ahead (ThreadedPrefetcher), over the buffer ring it must respect
    from artiboost_amd.realdata import ThreadedPrefetcher
    threaded = []
    ml = make(decode_group=2, decode_ahead=True)
    ml.reuse_buffers, ml.want_chw = 4, True
    for b in ThreadedPrefetcher(ml, depth=2):
        junk = junk @ junk * 1e-3
        threaded.append({k: v.clone() for k, v in b.items()})
    torch.cuda.synchronize()
    with pytest.raises(ValueError, match="reuse_buffers"):
        ThreadedPrefetcher(ml, depth=3)
    early = []
    for b in ThreadedPrefetcher(make(), depth=2):      # a consumer that stops early: the worker is released and joined
        early.append({k: v.clone() for k, v in b.items()})
        break
    assert len(got) == len(plain) == len(grouped) == len(ahead) == len(threaded) >= 2 and len(early) == 1
    for other in (got, grouped, ahead, threaded, early):
        for a, b in zip(plain, other):
            assert a.keys() == b.keys()
            for k in a:
                _same_entry(a[k], b[k], k)


class _OneProgressive(JpegSource):
    """Twelve DISTINCT .jpg files (the golden frames shifted by idx); file 0 is progressive: the device decoder refuses it."""

    def __init__(self):
        import io
        from PIL import Image
        super().__init__()
        self.files = []
        for i in range(self.n):
            b = io.BytesIO()
            Image.fromarray(np.roll(self.g["frames"][i % 3], 7 * i, axis=1)).save(b, "JPEG", quality=92, progressive=(i == 0))
            self.files.append(b.getvalue())

    def get_image(self, idx):
        import io
        from PIL import Image
        return np.asarray(Image.open(io.BytesIO(self.files[idx])).convert("RGB"))

    def get_image_bytes(self, idx):
        return self.files[idx]


class _OneProgressivePillow(_OneProgressive):
    get_image_bytes = None


@pytest.mark.gpu
def test_group_that_cannot_be_predecoded_does_not_share_the_side_stream_decoder():
    """Round-3 advisor finding: with decode_ahead the NEXT group is decoded on a side stream; a group whose own predecode bailed out (one file
    the device decoder does not cover) decodes batch by batch on the MAIN stream at the same time -- with a decoder of its own, not the side
    stream's device blob / workspace.  Every batch must equal the all-Pillow source's, bit for bit, whatever the interleaving."""
    pytest.importorskip("PIL")
    from artiboost_amd.realdata import MixedLoader, RealBatcher
    cfg = {"IMAGE_SIZE": [64, 64], "CENTER_IDX": 0, "BBOX_EXPAND_RATIO": 1.2}
    junk = torch.randn(1024, 1024, device="cuda")
    for seed in (1, 2, 3):
        ref = [{k: v.clone() for k, v in b.items()} for b in
               MixedLoader(RealBatcher(_OneProgressivePillow(), cfg, compute_dtype=torch.float32, seed=5), None, 2, seed=seed, decode_group=1, decode_ahead=False)]
        ml = MixedLoader(RealBatcher(_OneProgressive(), cfg, compute_dtype=torch.float32, seed=5), None, 2, seed=seed, decode_group=2, decode_ahead=True)
        got = []
        for b in ml:
            junk = junk @ junk * 1e-3
            got.append({k: v.clone() for k, v in b.items()})
        torch.cuda.synchronize()
        assert len(got) == len(ref) == 6
        assert ml.real._jpeg is not None and ml.real._jpeg_side is not None and ml.real._jpeg is not ml.real._jpeg_side
        for a, r in zip(got, ref):
            for k in r:
                _same_entry(a[k], r[k], k)


@pytest.mark.gpu
def test_mixed_batches_from_the_buffer_ring_equal_fresh_ones():
    """MixedLoader(want_chw=False, reuse_buffers=n) -- both halves write their frames straight into the batch's image tensor, which comes from a ring
    of n zero-bordered buffers -- yields, batch by batch, the same tensors as the default loader (fresh tensors, float CHW `image` as well); the
    ring hands a buffer out again n batches later with its borders still zero."""
    from test_gpu_synth import _loader
    from artiboost_amd.realdata import MixedLoader, RealBatcher
    from artiboost_amd.synth import ArtiBoostLoader
    B = 4

    def make(**kw):
        src = GoldenSource()
        assets, proto = _loader(size=64)
        n_synth = MixedLoader.n_synth_for(B, len(src), proto.synth_len)
        synth = ArtiBoostLoader.from_assets(assets, proto.cfg, proto.preset, n_synth, proto.synth_len, compute_dtype=torch.float32, random_seed=3)
        synth.prepare()
        return MixedLoader(RealBatcher(src, proto.preset, compute_dtype=torch.float32, seed=2), synth, B, seed=4, **kw)
    ref = [{k: v.clone() for k, v in b.items()} for b in make()]
    seen = []
    for i, b in enumerate(make(want_chw=False, reuse_buffers=3)):
        assert "image" not in b and set(b) == set(ref[i]) - {"image"}
        for k, v in b.items():
            _same_entry(v, ref[i][k], (i, k))
        pad = b["image_nhwc4_padded"]
        seen.append(pad.data_ptr())
        assert float(pad[:, :3].abs().max()) == 0 and float(pad[:, :, :3].abs().max()) == 0 and float(pad[:, -3:].abs().max()) == 0
    assert len(ref) == len(seen) >= 4 and len(set(seen)) == 3 and seen[3] == seen[0]


def test_mixed_loader_epoch_length_is_the_same_on_every_rank():
    """Advisor finding (round 4): len(range(rank, real_len, world)) differs by one between ranks when real_len % world != 0; a rank with one
    more batch would hang in the gradient all-reduce.  The real share is cut to world * (real_len // world) before the rank slices."""
    from types import SimpleNamespace
    from artiboost_amd.realdata import MixedLoader

    class Synth:      # what MixedLoader reads of the ArtiBoostLoader: nominal epoch length, batch size, trimmed step count
        use_synth, epoch = True, object()

        def __init__(self, synth_len, bs, steps):
            self.synth_len, self.batch_size, self._steps = synth_len, bs, steps

        def __len__(self):
            return self._steps

    for real_len, factor, world, B in ((85805, 0.2, 2, 64), (11, 0.0, 2, 3), (1001, 0.6, 8, 16), (66034, 0.6, 8, 128), (37, 0.5, 3, 4)):
        synth_len = int(real_len * factor)
        lens, perms = [], []
        for rank in range(world):
            real = SimpleNamespace(src=range(real_len))
            if synth_len:
                ns = MixedLoader.n_synth_for(B, real_len, synth_len)
                synth = Synth(synth_len, ns, (synth_len // world) // ns if ns else 0)
            else:
                synth = None
            ml = MixedLoader(real, synth, B, seed=5, rank=rank, world_size=world)
            lens.append(len(ml))
            perms.append(ml._epoch_perm())
        assert len(set(lens)) == 1, (real_len, factor, world, B, lens)
        assert len({len(p) for p in perms}) == 1 and lens[0] * ml.n_real <= len(perms[0])
        allp = np.concatenate(perms)
        assert len(set(allp.tolist())) == len(allp)          # disjoint slices of one shared permutation

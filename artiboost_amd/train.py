"""One training step of the coupled hot path (train/train_artiboost.py:66-96 `epoch_pass` body):
   [render next synthetic batch] -> arch_model(batch) -> criterion -> backward -> [grad all-reduce] -> clip -> Adam.

`TrainStep` runs it eagerly (reference-shaped, every call re-issues ~600 launches from Python) or as replayed
hipGraphs (one capture, then ~tens of microseconds of host work per step): everything between the batch being
resident in HBM and the parameters being updated is device work on one stream; per-step host inputs (loss RNG draws,
Adam bias corrections, CCV sample descriptors) go through small pinned -> device copies issued before the replay.
With world_size > 1 the backward is captured as three graphs; the RCCL all-reduce of each finished range of the flat
gradient buffer runs on a side stream while the next stage computes."""
import os

import torch

from .registry import IMAGE_PLANE_KEY, Queries, SynthQueries, image_plane_of, tag_image_plane


# hipGraph captures use thread-local error mode: with a process group alive, RCCL's watchdog / heartbeat threads query
# events concurrently, which under the default global mode can invalidate a capture in progress on this thread
CAPTURE_MODE = "thread_local"


def allreduce_flat_(g, world, group=None, bucket_elems=8 << 20):
    """In-place average of the flat gradient over `world` ranks in fixed-size buckets (32 MiB of fp32 by default:
    large enough to run RCCL at link rate over xGMI, small enough that the first bucket is on the wire while the
    later ones are still being enqueued).  Works for any backend (RCCL on GPU, gloo in the CPU tests).
    SUM + this library's own 1 / world pass, not ReduceOp.AVG: RCCL implements AVG as PreMulSum, whose gfx950 ring kernels
    multiply with `v_pk_mul_f32` -- and these collectives run on the comm stream BESIDE the backward's MFMA kernels, where packed
    fp32 is not safe on this part (librccl.so disassembled: `runRing<float, FuncPreMulSum<float>, ...>` holds
    v_pk_mul / v_pk_fma / v_pk_add_f32, `runRing<float, FuncSum<float>, ...>` none).  world a power of two: bit-identical to AVG."""
    n = g.numel()
    for s in range(0, n, bucket_elems):
        torch.distributed.all_reduce(g[s:s + bucket_elems], op=torch.distributed.ReduceOp.SUM, group=group)
    if g.is_cuda:
        from . import _lib as L
        L.check(L.lib().ab_scale_f32(L.ptr(g), L.l(n), L.f(1.0 / world), L.stream()), "ab_scale_f32")
    else:
        g.mul_(1.0 / world)
    return g


def rccl_env_defaults():
    """Call before init_process_group("nccl").  Ring only: RCCL's tree kernels for a float SUM (`runTreeUpDown<float, FuncSum<float>, ...>`)
    add with `v_pk_add_f32`, the ring kernels do not (see allreduce_flat_); on one xGMI node the ring is what RCCL picks for these
    4 - 32 MiB buckets anyway.  An explicit NCCL_ALGO in the environment wins."""
    os.environ.setdefault("NCCL_ALGO", "Ring")


class TrainStep:
    def __init__(self, arch_model, criterion, optimizer, example_batch, use_graph=True, dist_group=None,
                 renderer=None, fused_criterion=True, pipeline_render=False):
        self.model = arch_model
        self.hb = arch_model.model_list[0]
        self.crit = criterion
        self.opt = optimizer
        self.dev = self.hb.store.device
        self.use_graph = use_graph
        self.renderer = renderer          # object with .render_into(static_batch) enqueuing device work
        self.group = dist_group
        self.world = torch.distributed.get_world_size(dist_group) if dist_group is not None else 1
        # ("__flat_*" are the loader's packed staging buffers; clones would not alias their typed views)
        self.static = {k: (v.to(self.dev).clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()
                       if not k.startswith("__flat_")}
        self.out = None
        self.g_fwd_bwd = None
        self.g_opt = None
        self._fake_comm = os.environ.get("AB_FAKE_COMM") == "1"     # timing probe: the DDP stream choreography without RCCL
        # AB_DDP_SINGLE_RANK=1 with a process group of ONE rank: the whole multi-rank schedule (three backward graphs, bucketed
        # SUM all-reduces + 1 / world on the comm stream, post-all-reduce clip) with the real collective -- what a 1-GPU box can execute
        # of the RCCL path (tests/test_gpu_bench.py::test_rccl_single_rank_schedule)
        self.comm = self.world > 1 or (dist_group is not None and os.environ.get("AB_DDP_SINGLE_RANK") == "1")
        self.comm_stream = torch.cuda.Stream(device=self.dev) if (self.comm or self._fake_comm) else None
        self.steps = 0
        # DDP overlap: the backward is captured as three graphs (heads + layer4 | layer3 | layer2 .. stem).  The first
        # stage produces 68 % of the gradient bytes, the second 27 %; each range is all-reduced on the comm stream while
        # the next stage computes, so only the last 5 MB are exposed.  AB_DDP_SPLIT=1 forces the split on one GPU
        # (tests), =0 disables it.
        env = os.environ.get("AB_DDP_SPLIT", "")
        self.split = bool(use_graph and fused_criterion and (self.comm or env == "1") and env != "0")   # (cleared below
        #                                                                                            if the criterion is not fusable)
        self.g_bwd_rest = []
        # Render/learn pipelining (the reference overlaps them through DataLoader worker processes,
        # artiboost_loader.py:195-260): the batch for step i+1 is rendered on a side stream while step i trains.
        # `rstatic` holds the render inputs of the NEXT batch and its own image buffer; the image is handed over by one
        # device copy at the end of the step, so a single captured graph with fixed addresses serves every step.
        # pipeline_render="opt": batch i+1 is rendered on the side stream while step i's gradient all-reduce and clip+Adam
        # run; it is written straight into the learn buffer, which step i no longer reads by then -- no second image, no
        # copy.  Measured on one MI355X: 9505 vs 9890 samples/s without it -- like the other two-stream variants (see
        # DESIGN.md section 5) concurrency across streams costs more here than it hides; opt-in only.
        self.pipeline_opt = bool(pipeline_render == "opt" and renderer is not None)
        self.g_render = None
        self.pipeline = bool(pipeline_render and not self.pipeline_opt and renderer is not None)
        # the padded image's plane comes from the tag its loader wrote (registry.tag_image_plane / the batch's IMAGE_PLANE_KEY), carried over
        # to the static clone; HybridBaseline._plane_of refuses an untagged bfloat16 image on a bf16x3 model
        pad0 = self.static.get("image_nhwc4_padded") if isinstance(self.static, dict) else None
        net0 = getattr(self.hb, "net", None)
        if net0 is not None and torch.is_tensor(pad0):
            plane = image_plane_of(example_batch, example_batch.get("image_nhwc4_padded"))
            if plane is not None:
                tag_image_plane(pad0, plane)
                self.static[IMAGE_PLANE_KEY] = plane
            net0.image_plane = self._plane = self.hb._plane_of(self.static, pad0)
        self.rstatic = None
        self.render_stream = None
        if self.pipeline_opt:
            self.rstatic = {k: v.clone() for k, v in self.static.items() if torch.is_tensor(v) and k.startswith("_") and not k.startswith("__flat_")}
            self.render_stream = torch.cuda.Stream(device=self.dev)
        if self.pipeline:
            self.rstatic = {k: v.clone() for k, v in self.static.items() if torch.is_tensor(v) and k.startswith("_") and not k.startswith("__flat_")}
            self.rstatic["image_nhwc4_padded"] = torch.zeros_like(self.static["image_nhwc4_padded"])
            self.render_stream = torch.cuda.Stream(device=self.dev)
        self.fused = None
        self.model_key = type(self.hb).__name__                      # the key of this model's outputs in Arch's result dict
        if not getattr(self.hb, "HAS_BOX_HEAD", True):               # SimpleBaseline: the fused pose/loss kernel is HybridBaseline's assembly
            fused_criterion = False
            self.split = False
            self.use_graph = False
        if fused_criterion:
            from .criterions import FusedPoseCriterion
            try:
                self.fused = FusedPoseCriterion(criterion, self.hb.inp_res, self.hb.center_idx)
            except NotImplementedError:       # a loss outside the fused kernel (e.g. SymCornerLoss): autograd criterion
                self.fused = None
                self.split = False
                self.use_graph = False        # the registry losses index with host lists / move constants: not capturable
        if self.use_graph:
            self.opt.use_device_hyper(self.dev)

    # ------------------------------------------------------------------ pieces
    def _fwd_bwd(self):
        if self.pipeline:
            cur = torch.cuda.current_stream(self.dev)
            self.render_stream.wait_stream(cur)
            with torch.cuda.stream(self.render_stream):
                self.renderer.render_into(self.rstatic)
            out = self._learn()
            cur.wait_stream(self.render_stream)
            self.static["image_nhwc4_padded"].copy_(self.rstatic["image_nhwc4_padded"])
            return out
        if self.renderer is not None and not self.pipeline_opt:
            self.renderer.render_into(self.static)
        return self._learn()

    def _launch_render_next(self):
        """pipeline_opt: the learn graphs are enqueued -- once they have run the image buffer is free; start the render of
        the next batch on the side stream (it overlaps the all-reduce and the optimizer graph)."""
        if not self.pipeline_opt:
            return
        self.render_stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.render_stream):
            self.g_render.replay()

    def _render_next(self):
        """pipeline_opt: render the batch staged in `rstatic` into the learn image buffer."""
        self.rstatic["image_nhwc4_padded"] = self.static["image_nhwc4_padded"]
        self.renderer.render_into(self.rstatic)

    def _learn(self):
        if self.fused is not None:
            return self._fwd_bwd_fused()
        preds = self.model(self.static)[self.model_key]
        total, losses = self.crit.compute_losses(preds, self.static)
        self.opt.zero_grad(set_to_none=True)
        total.backward()
        return preds, total, losses

    def _fwd_bwd_fused(self):
        """Autograd-free path: HIP forward -> fused soft-argmax -> fused pose+criterion(+backward) -> HIP backward."""
        hb, net, st = self.hb, self.hb.net, self.static
        net.training = True
        if not net._packed:
            net.pack_weights()
        net.image_plane = getattr(self, "_plane", "f32") if st.get("image_nhwc4_padded") is not None else "f32"
        logits, _ = net.forward(image=st.get(Queries.IMAGE), xpad=st.get("image_nhwc4_padded"))
        kp3d, conf, stat = net.head_fwd(logits)
        o = self.fused(kp3d, net.last["box_raw"], net.last["box_raw"].shape[-1], st)
        dlogits = net.head_bwd(logits, kp3d, conf, stat, o["g_kp3d"])
        net.backward(dlogits, o["g_box6d"], stage=0 if self._capturing_split else None)
        hb.flat_param.grad = hb.store.grad
        preds = dict(o, kp3d=kp3d, kp3d_confd=conf)
        return preds, o["losses"], o

    _capturing_split = False

    def predictions(self):
        """The step's predictions under the NINE keys HybridBaseline.forward returns (hybridbaseline.py:86-96).  The fused path keeps
        only what the criterion kernel writes; the root-relative and box-root entries the evaluator's PCK / visual metrics read are
        derived here (a few small device ops, outside the graphs)."""
        if self.fused is None:
            return self.out[0]
        o, st = self.fused.out, self.static
        ja, ca, R = o["joints_3d_abs"], o["corners_3d_abs"], o["box_rot_rotmat"]
        root = ja[:, self.hb.center_idx:self.hb.center_idx + 1]
        can = st[Queries.CORNERS_CAN].to(ca.dtype)
        boxroot = (ca - torch.matmul(R, can.permute(0, 2, 1)).permute(0, 2, 1)).mean(1, keepdim=True)
        return {"joints_3d_abs": ja, "corners_3d_abs": ca, "joints_3d": ja - root, "corners_3d": ca - root, "2d_uvd": o["uvd2d"],
                "boxroot_3d_abs": boxroot, "box_rot_rotmat": R, "kp3d": self.out[0]["kp3d"], "kp3d_confd": self.out[0]["kp3d_confd"]}

    def _optim(self):
        self.opt.step()

    def _allreduce_range(self, lo, hi):
        """Enqueue the averaging all-reduce of grad[lo:hi] on the comm stream, ordered after everything enqueued so far
        on the compute stream."""
        self.comm_stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self.comm_stream):
            if self._fake_comm:
                self.hb.store.grad[lo:hi].mul_(1.0)          # same bytes touched once, no communication
            else:
                allreduce_flat_(self.hb.store.grad[lo:hi], self.world, self.group)

    def _allreduce(self):
        """Average the flat gradient across ranks on the side stream (RCCL over xGMI), bucketed so the first buckets'
        transfer overlaps the later ones' launch; clip uses the post-all-reduce norm (single-process semantics)."""
        g = self.hb.store.grad
        cur = torch.cuda.current_stream(self.dev)
        self.comm_stream.wait_stream(cur)
        with torch.cuda.stream(self.comm_stream):
            allreduce_flat_(g, self.world, self.group)
        cur.wait_stream(self.comm_stream)

    # ------------------------------------------------------------------ capture
    def _snapshot(self):
        """Everything the capture warm-up step modifies: weights, BatchNorm running statistics, Adam moments and step counts,
        and the host RNG streams the loss draws consume."""
        import random
        import numpy as np
        store = self.hb.store
        snap = dict(flat=store.flat.detach().clone(), stats=store.stats.clone(), nbt=store.num_batches_tracked,
                    graph_steps=self.opt.graph_steps, rng=(random.getstate(), np.random.get_state(), torch.get_rng_state()), opt={})
        for p, st in self.opt.state.items():
            snap["opt"][p] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
        if self.pipeline:       # the pipelined warm-up ends by handing the NEXT batch's image to the learn buffer: undo that too
            snap["image"] = self.static["image_nhwc4_padded"].clone()
        return snap

    def _restore(self, snap):
        import random
        import numpy as np
        store = self.hb.store
        with torch.no_grad():
            store.flat.copy_(snap["flat"])
            store.stats.copy_(snap["stats"])
            if "image" in snap:
                self.static["image_nhwc4_padded"].copy_(snap["image"])
        store.num_batches_tracked = snap["nbt"]
        for p, st in self.opt.state.items():
            old = snap["opt"].get(p)
            for k, v in st.items():
                if torch.is_tensor(v):
                    v.copy_(old[k]) if old is not None else v.zero_()      # moments created by the warm-up start from zero
                else:
                    st[k] = old[k] if old is not None else 0
        self.opt.graph_steps = snap["graph_steps"]
        random.setstate(snap["rng"][0]); np.random.set_state(snap["rng"][1]); torch.set_rng_state(snap["rng"][2])
        self.hb.net.pack_weights()                                         # compute-precision copies of the restored weights
        torch.cuda.synchronize(self.dev)

    def _capture(self):
        # The capture needs one real step first (allocator warm-up, lazily created optimizer state).  Its effects are undone
        # afterwards, so a graph-replayed run is the same sequence of updates as the eager one, bit for bit, and a
        # checkpoint can be loaded before the first step.
        snap = self._snapshot()
        self.crit.draw(self.dev)
        self.opt.advance_hyper()
        self.opt.graph_steps = 0
        torch.cuda.synchronize(self.dev)
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):          # warm-up on a side stream (allocator, lazy state) before capture
            self.crit.freeze_draws(True)
            self._fwd_bwd()
            if self.comm:                      # the warm-up is a real optimizer step: it must see the averaged gradient too
                self._allreduce()
            self._optim()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        self.g_fwd_bwd = torch.cuda.CUDAGraph()
        if self.split:
            self._capturing_split = True
            try:
                with torch.cuda.graph(self.g_fwd_bwd, capture_error_mode=CAPTURE_MODE):
                    self.out = self._fwd_bwd()                       # ... up to and including layer4's backward
                self.g_bwd_rest = []
                for st in range(1, self.hb.net.BWD_STAGES):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=self.g_fwd_bwd.pool(), capture_error_mode=CAPTURE_MODE):
                        self.hb.net.backward(stage=st)
                    self.g_bwd_rest.append(g)
            finally:
                self._capturing_split = False
        else:
            with torch.cuda.graph(self.g_fwd_bwd, capture_error_mode=CAPTURE_MODE):
                self.out = self._fwd_bwd()
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, pool=self.g_fwd_bwd.pool(), capture_error_mode=CAPTURE_MODE):
            self._optim()
        if self.pipeline_opt:
            self.g_render = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_render, pool=self.g_fwd_bwd.pool(), capture_error_mode=CAPTURE_MODE):
                self._render_next()
        self._restore(snap)

    # ------------------------------------------------------------------ public
    def stage(self, loader, batch_idx):
        """Gather the inputs of step `batch_idx` from the loader's planned epoch: ground truth of this batch and, when
        pipelined, the render inputs of the next one (whose image this step produces)."""
        if not (self.pipeline or self.pipeline_opt):
            loader.load_batch(self.static, batch_idx)
            return
        loader.load_batch(self.static, batch_idx, which="gt")
        loader.load_batch(self.rstatic, (batch_idx + 1) % max(len(loader), 1), which="render")

    def prime(self, loader, batch_idx):
        """Pipelined mode: render batch `batch_idx` eagerly so that the first step has an image to learn from."""
        if self.pipeline_opt:
            loader.load_batch(self.rstatic, batch_idx, which="render")
            self._render_next()
            return
        if not self.pipeline:
            return
        loader.load_batch(self.rstatic, batch_idx, which="render")
        self.renderer.render_into(self.rstatic)
        self.static["image_nhwc4_padded"].copy_(self.rstatic["image_nhwc4_padded"])

    def load_batch(self, batch):
        """Copy a host/device batch into the static input buffers (async on the current stream)."""
        for k, v in batch.items():
            if torch.is_tensor(v) and k in self.static:
                self.static[k].copy_(v, non_blocking=True)

    def __call__(self, batch=None):
        if batch is not None:
            self.load_batch(batch)
        if not self.use_graph:
            if self.fused is not None:
                self.crit.draw(self.dev)
            self.out = self._fwd_bwd()
            if self.pipeline_opt:
                cur = torch.cuda.current_stream(self.dev)
                self.render_stream.wait_stream(cur)
                with torch.cuda.stream(self.render_stream):
                    self._render_next()
            if self.comm:
                self._allreduce()
            self._optim()
            if self.pipeline_opt:
                torch.cuda.current_stream(self.dev).wait_stream(self.render_stream)
        else:
            if self.g_fwd_bwd is None:
                self._capture()
            self.crit.draw(self.dev)
            self.opt.advance_hyper()
            self.g_fwd_bwd.replay()
            self.hb.store.num_batches_tracked += 1           # the replayed forward is a training-mode BatchNorm forward
            if self.split:
                ranges = self.hb.net.grad_stage_ranges()
                comm = self.comm or self._fake_comm
                if comm:
                    self._allreduce_range(*ranges[0])
                for g, rng in zip(self.g_bwd_rest, ranges[1:]):
                    g.replay()
                    if comm:
                        self._allreduce_range(*rng)
                self._launch_render_next()
                if comm:
                    torch.cuda.current_stream(self.dev).wait_stream(self.comm_stream)
            else:
                self._launch_render_next()
                if self.comm:
                    self._allreduce()
            self.g_opt.replay()
            if self.pipeline_opt:
                torch.cuda.current_stream(self.dev).wait_stream(self.render_stream)
        self.steps += 1
        return self.out


class DeferredEpochMetrics:
    """Feeds an Evaluator without a device synchronisation per step (the reference's epoch_pass calls
    `evaluator.feed_all(predicts, batch, losses)` after every batch, train_artiboost.py:96-98, which moves tensors to the host
    each time).  The fused pose/loss kernel already leaves every sample's joint / corner EPE in mm and the eight loss scalars
    on the device: `collect()` stacks them (device-side copies on the step's stream), `flush()` makes ONE transfer at the end
    of the epoch and replays the steps into the metrics in their original order, so Mean3DEPE, LossesMetric and
    ValMetricMean3DEPE2 (last write per CCV triplet wins) end up exactly as with per-step feeding."""

    def __init__(self, ts: TrainStep, capacity: int, evaluator=None):
        assert ts.fused is not None, "needs the fused criterion (per-sample EPE comes from ab_pose_loss)"
        self.ts, self.n = ts, 0
        # metrics that need the full predictions (PCK, AR, ...) are still fed after every step
        self.direct = [m for m in (evaluator.metrics_list if evaluator is not None else []) if not self._deferrable(m)]
        B, dev = ts.static[Queries.ROOT_JOINT].shape[0], ts.dev
        self.epe = torch.zeros((capacity, B, 2), dtype=torch.float32, device=dev)         # (joints, corners) mm
        self.losses = torch.zeros((capacity, 8), dtype=torch.float32, device=dev)
        self.ids = torch.zeros((capacity, B, 4), dtype=torch.int64, device=dev)           # obj, persp, grasp, is_synth

    @staticmethod
    def _deferrable(m):
        from .metrics import LossesMetric, Mean3DEPE, ValMetricMean3DEPE2
        if isinstance(m, (ValMetricMean3DEPE2, LossesMetric)):
            return True
        return type(m) is Mean3DEPE and m.to_millimeters and all(k in ("joints_3d_abs", "corners_3d_abs") for k in m.val_keys_list)

    def collect(self):
        o, st, i = self.ts.fused.out, self.ts.static, self.n
        if self.direct:
            full = self.ts.predictions()
            for m in self.direct:
                m.feed(full, st)
        self.epe[i].copy_(o["sample_part"][:, 5:7])
        self.losses[i].copy_(o["losses"])
        self.ids[i].copy_(torch.stack([st[SynthQueries.OBJ_ID], st[SynthQueries.PERSP_ID], st[SynthQueries.GRASP_ID],
                                       st[SynthQueries.IS_SYNTH].to(torch.int64)], 1))
        self.n += 1

    def flush(self, evaluator, summarizer=None):
        """summarizer: also replay the per-step loss scalars into Summarizer.summarize_losses (epoch_pass does that after
        every TRAIN batch, train_artiboost.py:101-103)."""
        from .metrics import LossesMetric, Mean3DEPE, ValMetricMean3DEPE2
        n = self.n
        epe, losses, ids = self.epe[:n].cpu().numpy(), self.losses[:n].cpu().numpy(), self.ids[:n].cpu().numpy()
        col = {"joints_3d_abs": 0, "corners_3d_abs": 1}
        keys = self.ts.fused.LOSS_KEYS
        for m in evaluator.metrics_list:
            if m in self.direct:
                continue
            if isinstance(m, ValMetricMean3DEPE2):
                for key in m.val_keys_list:
                    for s in range(n):
                        for b in range(epe.shape[1]):
                            if ids[s, b, 3]:
                                m.storage[key][tuple(int(x) for x in ids[s, b, :3])] = epe[s, b, col[key]]
            elif isinstance(m, Mean3DEPE):
                for key in m.val_keys_list:
                    for s in range(n):
                        m.avg_meters[key].update(float(epe[s, :, col[key]].sum()), n=epe.shape[1])
            elif isinstance(m, LossesMetric):
                for s in range(n):
                    m.feed(None, None, losses={k: losses[s, i] for i, k in enumerate(keys)})
        if summarizer is not None:
            for s in range(n):
                summarizer.summarize_losses({k: losses[s, i] for i, k in enumerate(keys)})
        self.n = 0

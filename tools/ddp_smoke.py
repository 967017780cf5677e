"""Two-rank smoke of the DDP step on whatever devices are visible (both ranks may share one GPU when the backend is gloo):
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/ddp_smoke.py
Checks: the split-graph + side-stream all-reduce path runs -- by default with the north-star schedule (the render of batch
t+1 on a second stream under the last all-reduce range and the optimizer; AB_DDP_OVERLAP=0 turns it off) in the benchmarked
precision (AB_DDP_DTYPE, default bf16x3) -- losses are finite, and the ranks hold identical weights.
With --nproc-per-node 1 and AB_DDP_SINGLE_RANK=1 the same schedule runs with a ONE-rank RCCL group: the real init_process_group("nccl"),
bucketed SUM all-reduces + 1 / world on the comm stream between the backward graphs -- what a 1-GPU box can execute of the RCCL path."""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import yaml
from artiboost_amd import registry as R
from artiboost_amd.assets import SceneAssets
from artiboost_amd.criterions import Criterion
from artiboost_amd.models import Arch
from artiboost_amd.optim import FusedClipAdam
from artiboost_amd.synth import ArtiBoostLoader
from artiboost_amd.train import TrainStep

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
import random
import numpy as np
random.seed(100 + rank); np.random.seed(100 + rank); torch.manual_seed(100 + rank)      # the ordinal losses draw their pairs from these
ngpu = torch.cuda.device_count()
dev = f"cuda:{rank % ngpu}"
torch.cuda.set_device(dev)
backend = "nccl" if ngpu >= world else "gloo"
if backend == "nccl":
    from artiboost_amd.train import rccl_env_defaults
    rccl_env_defaults()
dist.init_process_group(backend)
root = os.path.join(os.path.dirname(__file__), "..")
cfg = yaml.safe_load(open(os.path.join(root, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
dtype = os.environ.get("AB_DDP_DTYPE", "bf16x3")
overlap = os.environ.get("AB_DDP_OVERLAP", "1") != "0"
arch = dict(cfg["ARCH"], COMPUTE_DTYPE=dtype, DEVICE=dev, INIT_SEED=1)
model = Arch({"ARCH": arch}, R.build_arch_model_list(arch, preset_cfg=cfg["DATA_PRESET"]))
crit = Criterion(cfg, R.build_criterion_loss_list(cfg["CRITERION"], preset_cfg=cfg["DATA_PRESET"], LAMBDAS=cfg["LAMBDAS"]))
hb = model.model_list[0]
opt = FusedClipAdam(model.models_params, lr=1e-3, max_norm=1.0, model=hb)
assets = SceneAssets("HO3D", seed=1)
loader = ArtiBoostLoader.from_assets(assets, dict(cfg["MANAGER"], EPOCH=1), cfg["DATA_PRESET"], 8, 8 * world * 6, device=dev,
                         compute_dtype=hb.net.dtype, random_seed=1, rank=rank, world_size=world)
loader.prepare()
static = loader.new_static_batch()
loader.load_batch(static, 0)
model.train()
ts = TrainStep(model, crit, opt, static, use_graph=True, dist_group=dist.group.WORLD, renderer=loader,
               pipeline_render="opt" if overlap else False)
ts.static = static
assert ts.split, "world_size > 1 must take the split-graph path"
ts.prime(loader, 0)
for i in range(5):
    ts.stage(loader, i % len(loader))
    _, losses, _ = ts()
torch.cuda.synchronize()
assert torch.isfinite(losses).all()
w = hb.store.flat.detach().clone()
ws = [torch.empty_like(w) for _ in range(world)]
dist.all_gather(ws, w)
same = all(torch.equal(ws[0], x) for x in ws)
if rank == 0:
    print(f"backend={backend} world={world} dtype={dtype} render_overlap={overlap} comm={ts.comm} final_loss={float(losses[5]):.9f} "
          f"weight_sum={float(w.double().sum()):.12f} weights_identical_across_ranks={same}")
assert same
dist.destroy_process_group()

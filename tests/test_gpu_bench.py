"""bench.py end to end on the GPU box: the two-rank schedule the driver's SCALE run will execute (one process per rank, split
backward graphs, side-stream gradient averaging, next-batch render overlapped with the last range + optimizer), on shared devices
over gloo when the box has one GPU -- so the first real multi-GPU run exercises a path that has run whole at least once -- and the
BASELINE configs[1] eval leg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-3000:]
    return json.loads(lines[-1])


def test_bench_two_ranks_end_to_end():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--allow-shared-devices", "--steps", "3", "--warmup", "2",
           "--sustain", "0.5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["config"]["render_overlap"] is True and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 128
    assert line["final_loss"] is not None and 0 < line["final_loss"] < 1.0
    assert line["value"] > 0 and line["ms_per_step"] > 0
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and roof["achieved"] and 0 < roof["frac"] < 1
    assert line["sustained"]["sustained_samples_per_s"] > 0


def test_bench_eval_leg_configs1():
    """`bench.py --eval`: BASELINE configs[1] (eval-mode forward, bs 64) as a line of its own with its roofline."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--eval", "--steps", "10", "--warmup", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = _last_json(r.stdout)
    assert "eval" in line["metric"] and line["dtype"] == "bf16x3" and line["n_gpus"] == 1
    assert line["value"] > 1000 and line["ms_per_step"] < 64.0
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and roof["conv_launches"] >= 36


def test_dropin_loop_without_segment_graphs_stays_device_bound():
    """The reference-shaped loop (tools/bench_dropin.py: arch_model(batch) -> compute_losses -> feed_all -> backward -> clip -> step through
    the anakin.* imports) issued kernel by kernel -- AB_SEGMENT_GRAPHS=0, every launch a torch.ops.artiboost_hip.* call -- at the benchmark
    geometry: the host keeps ahead of the device (round-2 review item 8: <= 12 ms per step).  Round 6: the evaluator no longer synchronises
    per step and the loader hands over the integer image plane, so the stated bound holds again on the slow boxes of the pool too (round 5
    measured 10.7 ms, 12.6 on a box whose graph-replayed step ran 9.25 ms instead of 8.7)."""
    env = dict(os.environ, AB_SEGMENT_GRAPHS="0")
    env.pop("AB_BINDING", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_dropin.py"), "--steps", "20"], capture_output=True, text=True,
                       timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = _last_json(r.stdout)
    assert line["segment_graphs"] is False and line["batch"] == 64 and line["size"] == 256
    assert line["ms_per_step"] <= 12.0, line
    assert 0 < line["final_loss"] < 1.0



def test_dropin_epoch_pass_leg_is_within_a_tenth_of_the_headline():
    """bench.py's `dropin_epoch_pass` leg: the reference's own epoch_pass loop through the anakin.* aliases (tools/bench_dropin.py) on the same
    box as the graph-replayed headline step -- the boundary the north star names.  Round-5 review: 1.28 x the headline; required <= 1.10."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--sustain", "0", "--no-eval-leg", "--no-dexycb-leg", "--no-study-leg",
           "--no-jpeg-leg", "--no-mixed-leg", "--no-rccl-leg"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = _last_json(r.stdout)
    d = line["dropin_epoch_pass"]
    assert "error" not in d, d
    assert d["image_plane"] == "u8n" and d["steps"] == 30
    assert d["ratio_to_headline"] <= 1.10, (d, line["ms_per_step"])
    assert d["ms_per_step_no_feed"] <= d["ms_per_step"] * 1.03


def test_rccl_single_rank_schedule():
    """What a 1-GPU box can execute of the RCCL path: tools/ddp_smoke.py under torch.distributed.run with ONE rank and
    AB_DDP_SINGLE_RANK=1 -- init_process_group("nccl") on the device, the three-graph backward with bucketed SUM all-reduces (+ 1 / world) on
    the comm stream, the render of the next batch under the last range, post-all-reduce clip + Adam.  An average over one rank is the
    identity, so five steps must end in bit-identical weights to the same schedule with the collective replaced by a touch of the same
    bytes (AB_FAKE_COMM=1)."""
    import re
    import subprocess
    import sys
    outs = []
    for i, extra in enumerate(({"AB_DDP_SINGLE_RANK": "1"}, {"AB_FAKE_COMM": "1", "AB_DDP_SPLIT": "1"})):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                            "--master-port", str(29541 + i), os.path.join(ROOT, "tools", "ddp_smoke.py")], env=env, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        line = [l for l in r.stdout.splitlines() if l.startswith("backend=")][-1]
        outs.append(line)
    assert "backend=nccl world=1" in outs[0] and "comm=True" in outs[0] and "comm=False" in outs[1], outs
    key = lambda l: re.search(r"final_loss=(\S+) weight_sum=(\S+)", l).groups()      # noqa: E731
    assert key(outs[0]) == key(outs[1]), outs


def test_bench_rccl_single_rank_mode():
    """`bench.py --rccl-single-rank`: init_process_group("nccl", device_id=...) exactly as the N > 1 launch does it, the multi-rank
    schedule with the real collective over one rank, one JSON line that says so."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-single-rank", "--steps", "3", "--warmup", "2", "--sustain", "0"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"].endswith("one_rank_rccl") and d["config"]["render_overlap"] is True
    assert d["value"] > 0 and d["final_loss"] == d["final_loss"] and d["roofline"]["frac"] > 0


def test_roofline_is_timed_inside_the_replayed_step():
    """Round-4 review item 6: the bench line's conv-stack time comes from the graph replay (wall-clock stamps captured around every
    conv-stack call), the eager-event number is stated beside it, and the two describe the same launches: same count, and the replayed
    kernels are not slower than the eager ones by more than the instrument's boundary uncertainty (they run 3 - 5 % FASTER: no launch gaps,
    warmer clocks), never below 0.85 of them."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--sustain", "0", "--no-eval-leg",
           "--no-dexycb-leg", "--no-study-leg", "--no-jpeg-leg", "--no-mixed-leg", "--no-rccl-leg", "--no-dropin-leg"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    roof = _last_json(r.stdout)["roofline"]
    assert "graph_stamp_error" not in roof, roof
    # (113 conv-stack calls per step through round 5; round 6 groups the 29 3x3 weight gradients into 6 launches: 90)
    assert roof["conv_ms_source"].startswith("graph replay") and roof["conv_launches_per_step"] >= 85
    g, e = roof["conv_ms_per_step"], roof["conv_ms_per_step_eager"]
    assert 0.85 * e <= g <= 1.03 * e, roof
    assert 0.5 < roof["stamp_boundary_us"] < 4.0, roof
    line = _last_json(r.stdout)
    assert g < line["ms_per_step"]                        # the dominant family is shorter than the step it is part of


def test_training_beside_the_worker_thread_gives_the_same_losses():
    """The mixed real + synthetic loop with its batch assembly on the training thread and on a worker thread with its own stream
    (realdata.ThreadedPrefetcher): the same batches in the same order, so the same losses to the last bit.  This is the loop in which
    the render of the next batch first ran BESIDE the step's MFMA kernels and came out a few grey levels off in a few dozen pixels
    (packed-fp32 results beside v_mfma_f32_32x32x16_bf16: DESIGN 15.10) -- with packed fp32 in the library the losses of the two
    modes part at a random step."""
    import yaml
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_mixed
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", "ho3dv2_clasbased_artiboost_mi355x.yaml")))
    cfg["DATA_PRESET"]["IMAGE_SIZE"], cfg["DATA_PRESET"]["HEATMAP_SIZE"] = [256, 256], [32, 32]
    group = "frames of 4 batches decoded per call one group ahead on a side stream"
    runs = []
    for mode in ("same stream, " + group, "worker thread two batches ahead, " + group, "worker thread two batches ahead, " + group):
        r = bench_mixed.train_loop(cfg, steps=24, modes=(mode,), quiet=True)
        runs.append((r["first_losses"], r["final_loss"]))
    assert runs[0] == runs[1] == runs[2], runs

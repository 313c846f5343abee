"""bench.py brings up its own ranks: `python bench.py --gpus 2` (no external launcher) on ONE GPU with the gloo
backend as the rehearsal of the RCCL path -- the JSON line must say n_gpus 2, carry the whole-job value of both ranks,
and both ranks must end with identical parameters after an update from the all-reduced gradients."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("config,data,issue", [("S256", "D0", "--graph"), ("S256", "D1", "--graph"), ("S256", "D1", "--eager"),
                                               ("S256", "D1", "--split-graph"), ("MIX", "D0", "--split-graph")])
def test_bench_two_ranks_gloo_rehearsal(config, data, issue):
    """--graph (the default): each rank replays its captured step, one all-reduce after the replay.  --split-graph (round 4):
    the step as TWO captured graphs split at the staged backward's hand-over, the first gradient segment's all-reduce started
    between their replays.  --eager: the staged backward hands the two halves of the flat gradient buffer to GradSync from
    inside loss.backward().  Through gloo the segments are reduced synchronously, same arithmetic.  MIX: the ragged batch runs
    on the layer-major kernels, whose backward is not staged -- the split step degrades to one all-reduce behind the second
    (empty) graph, and the one-launch stack kernels leave 64 CUs to the collective's kernels (hexgnn_stack_reserve_cus)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", config,
           "--data", data, "--steps", "6", "--warmup", "2", "--preheat-ms", "20", "--no-cpu-baseline", "--no-split", issue]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 512
    assert out["replicas_identical"] is True
    assert out["scaling"] == "weak" and out["value"] > 0
    assert abs(out["value"] - 512 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    assert out["config"]["collective"]["backend"].startswith("gloo")
    assert out["config"]["collective"]["overlapped_with_backward"] == (issue != "--graph")
    assert out["config"]["hip_graph"] == (issue != "--eager")


@pytest.mark.timeout(900)
def test_bench_four_ranks_gloo_rehearsal():
    """Four self-spawned ranks on the one GPU (the box allows six GPU processes: 4 ranks + this one): weak scaling
    bookkeeping (global batch 1024, value = 1024 graphs / step time) and identical replicas after the averaged update."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--config", "S256",
           "--data", "D1", "--steps", "4", "--warmup", "1", "--preheat-ms", "10", "--no-cpu-baseline", "--no-split"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["config"]["global_batch"] == 1024 and out["config"]["parallelism"] == "dp4"
    assert out["replicas_identical"] is True and out["scaling"] == "weak"
    assert abs(out["value"] - 1024 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("config", ["S256", "MIX"])
def test_bench_two_ranks_strong_scaling(config):
    """`--scaling strong`: ONE global 256-graph batch per step split over the ranks by edge count (ragged MIX: the two
    parts differ in graph count but not, beyond one graph, in edges), loss weights that make the averaged gradient the
    global mean, value = 256 graphs per step time, the same-set check on the first step, identical replicas."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", config,
           "--data", "D1", "--steps", "4", "--warmup", "1", "--preheat-ms", "10", "--sustain-s", "0", "--no-cpu-baseline",
           "--no-split", "--scaling", "strong"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["global_batch"] == 256
    assert 100 <= out["config"]["graphs_per_gpu"] <= 156
    assert out["replicas_identical"] is True
    assert abs(out["value"] - 256 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]

"""bench.py end to end on a small batch: the default mode (the step replayed from one HIP graph, in a child process), --eager,
and --graph in-process -- the JSON line's contract fields, the roofline / cpu_baseline objects, and that the two modes agree."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=600):
    env = dict(os.environ)
    for k in ("DMP_BENCH_CHILD", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "16", "--steps", "3", "--warmup", "2"] + list(flags),
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def _check_contract(d, steps=3, warmup=2):
    assert d["metric"].startswith("(pattern,graph) pairs/sec") and d["unit"] == "pairs/s"
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    assert d["step_ms_min"] <= d["step_ms_median"] <= d["step_ms_max"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches"] > 0 and r["avg_us"] > 0


def test_default_mode_replays_and_eager_agrees(gpu):
    d = _run("--no-cpu-baseline")
    _check_contract(d)
    assert d["config"]["launch"].startswith("one HIP graph replay per step") and "launch_fallback" not in d["config"]
    assert d["config"]["eager_ms_per_step"] > 0
    e = _run("--eager", "--no-cpu-baseline")
    _check_contract(e)
    assert e["config"]["launch"] == "eager launches" and e["config"]["eager_ms_per_step"] is None
    g = _run("--graph", "--no-cpu-baseline")                # in this process tree: no child
    _check_contract(g)
    assert g["config"]["launch"].startswith("one HIP graph replay per step")


def test_cpu_baseline_object(gpu):
    d = _run("--eager", timeout=900)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "pairs/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c


def test_two_ranks_replay_their_forward_backward(gpu):
    """--gpus 2 --graph: each rank replays forward + backward + gradient pack from one HIP graph, the all-reduce and the
    optimizer update follow eagerly (here: two ranks time-sharing the one GPU over gloo -- the plumbing, not a scaling run)."""
    d = _run("--gpus", "2", "--backend", "gloo", "--single-device", "--graph", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo"
    assert d["config"]["launch"].startswith("forward + backward + gradient pack as one HIP graph replay per rank")
    assert d["value"] > 0 and abs(d["value"] - 2 * 16 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]

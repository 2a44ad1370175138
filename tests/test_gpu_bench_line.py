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
    # `frac` is on SURVEY 8(d)'s byte count, the kernel's own byte count beside it
    assert r["bytes_basis"].startswith("SURVEY") and r["bytes_own"] >= r["bytes_per_launch"] and r["frac_own_bytes"] >= r["frac"]
    if "rows_not_fetched" in r:      # the batch's 0 / 1 gates: rows the launch leaves out, and the bytes it does move
        assert 0 < r["rows_not_fetched"] and 0 < r["bytes_per_launch"] < r["bytes_survey_all_rows"]
        assert (r["frac"] < 1.0 or r.get("served_from") == "infinity cache") and "note" in r
    assert "rate_check" not in d, d.get("rate_check")      # (bench.check_rates found nothing to refuse)
    _no_rate_above_the_peak(d)
    gk = d["gate_kept"]
    assert 0 < gk["edge_rows"] <= gk["of_edge_rows"] and 0 < gk["node_rows"] <= gk["of_node_rows"]


def _no_rate_above_the_peak(obj, path="line"):
    """Nothing in the line may exceed 8000 GB/s or a fraction of 1.0 (bench.check_rates enforces it before printing; here once
    more on what was printed)."""
    if isinstance(obj, dict):
        # (the one marked exception: a launch whose bytes fit in the chip's caches -- 256 MiB Infinity Cache + 32 MiB L2 -- can be fed faster than HBM delivers)
        sizes = [obj[k] for k in ("bytes", "bytes_per_launch", "bytes_own") if obj.get(k)]
        cached = obj.get("served_from") == "infinity cache" and sizes and min(sizes) < ((256 + 32) << 20) * 5 // 4
        for k, v in obj.items():
            if isinstance(v, (dict, list)):
                _no_rate_above_the_peak(v, path + "." + k)
            elif cached:
                continue
            elif isinstance(v, (int, float)) and not isinstance(v, bool):
                if k in ("gbps", "hbm_gbps") or (k == "achieved" and obj.get("unit") == "GB/s"):
                    assert v <= 8000.0, (path, k, v)
                if k.startswith("frac") or k.endswith("_frac"):
                    assert v <= 1.0, (path, k, v)
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            _no_rate_above_the_peak(v, "%s[%d]" % (path, i))


def test_default_mode_replays_and_eager_agrees(gpu):
    d = _run("--no-cpu-baseline")
    _check_contract(d)
    assert d["config"]["launch"].startswith("one HIP graph replay per step") and "launch_fallback" not in d["config"]
    assert d["config"]["eager_ms_per_step"] > 0
    # the two secondary measurements of the default line: 200 further steps, and the step with the gate capacity set
    x = d["steps_extended"]
    assert x["steps"] == 200 and x["step_ms_min"] <= x["step_ms_median"] <= x["step_ms_p90"] <= x["step_ms_max"] and x["value"] > 0
    gc = d["gate_compact"]
    assert gc["ms_per_step"] > 0 and 0 < gc["capacity"] < gc["target_edge_rows"] and gc["unit"] == "pairs/s"
    assert abs(gc["value"] - 16 / (gc["ms_per_step"] * 1e-3)) <= 0.01 * gc["value"]
    gd = d["gate_dense"]                                       # the control: the same step without the filter's gates
    assert gd["ms_per_step"] > 0 and gd["unit"] == "pairs/s" and abs(gd["value"] - 16 / (gd["ms_per_step"] * 1e-3)) <= 0.01 * gd["value"]
    # ... a first-class number: its own scatter-add objects over EVERY row, its own kernel and MFMA tables
    for key in ("roofline", "roofline_bwd"):
        r = gd[key]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["avg_us"] > 0 and "rows_not_fetched" not in r
        assert r["bytes_basis"].startswith("SURVEY") and r["bytes_own"] > 0 and r["bytes_per_launch"] > 0
        assert r["same_size_copy"]["avg_us"] > 0 and 0 < r["same_size_copy"]["frac"] <= 1.0
    assert gd["roofline"]["bytes_own"] >= gd["roofline"]["bytes_per_launch"]      # (the forward kernel writes [N, 2H]: more than SURVEY's count)
    assert d["roofline"]["same_size_copy"]["frac"] > 0 and d["roofline"]["own_rate_over_copy_rate"] > 0
    assert gd["kernels"] and gd["eager_ms_per_step"] > 0 and "kern" not in gd
    # the MFMA table: bf16x6 kernels against the dense bf16 peak (6 piece products per fp32 product), f32-input kernels against theirs
    for tab in (d["mfma_kernels"], gd["mfma_kernels"]):
        assert tab
        for name, m in tab.items():
            x6 = m["arithmetic"].startswith("bf16x6")
            assert m["peak"] == (2500.0 if x6 else 157.3) and abs(m["frac"] - m["pipe_tflops"] / m["peak"]) < 2e-3 and m["frac"] < 1.0
            assert abs(m["pipe_tflops"] - (6.0 if x6 else 1.0) * m["tflops"]) <= 0.6 and 0 < m["hbm_frac"] <= 1.0
    e = _run("--eager", "--no-cpu-baseline")
    _check_contract(e)
    assert e["config"]["launch"] == "eager launches" and e["config"]["eager_ms_per_step"] is None
    # the all-rows step as the headline of its own line (what the gate_dense profiles are taken with)
    n = _run("--graph", "--filter-net", "None", "--no-cpu-baseline", "--no-all-outputs", "--extended-steps", "0")
    assert "NO filter net" in n["config"]["workload"] and n["gate_dense"] is None and n["value"] > 0
    assert "rows_not_fetched" not in n["roofline"] and n["roofline"]["bytes_own"] >= n["roofline"]["bytes_per_launch"]
    g = _run("--graph", "--no-cpu-baseline")                # in this process tree: no child
    _check_contract(g)
    assert g["config"]["launch"].startswith("one HIP graph replay per step")


def test_cpu_baseline_object(gpu):
    d = _run("--eager", timeout=900)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "pairs/s" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    # SURVEY 8(d)'s protocol: 3 warm-up steps, at least 10 timed steps, the median
    assert c["protocol"].startswith("3 warm-up + ") and int(c["protocol"].split("+ ")[1].split(" ")[0]) >= 10
    assert c["step_s_min"] <= c["step_s_median"] <= c["step_s_max"] and abs(c["value"] - 32 / c["step_s_median"]) <= 0.01 * c["value"]   # (B = 32 pairs per CPU step)


def test_two_ranks_replay_their_forward_backward(gpu):
    """--gpus 2 --graph: each rank replays forward + backward + gradient pack from one HIP graph, the all-reduce and the
    optimizer update follow eagerly (here: two ranks time-sharing the one GPU over gloo -- the plumbing, not a scaling run)."""
    d = _run("--gpus", "2", "--backend", "gloo", "--single-device", "--graph", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo"
    assert d["config"]["launch"].startswith("forward + backward + gradient pack as one HIP graph replay per rank")
    assert d["value"] > 0 and abs(d["value"] - 2 * 16 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]


def test_two_ranks_on_the_scaling_workload_default_mode_and_their_gradient(gpu, tmp_path):
    """`bench.py --gpus 2 --workload 4 --batch 64` (BASELINE configs[3]'s shapes, two ranks time-sharing the one GPU over
    gloo): without a mode flag every rank replays forward + backward + gradient pack; both ranks are counted; the line
    carries per-rank host / all-reduce-wait / device times; and the ranks' averaged gradient equals the gradient of ONE
    process on the global batch (rank 0's pairs, then rank 1's)."""
    import torch as th
    two, one = str(tmp_path / "two.pt"), str(tmp_path / "one.pt")
    flags = ["--workload", "4", "--batch", "64", "--no-cpu-baseline", "--steps", "2", "--warmup", "2"]
    d = _run("--gpus", "2", "--backend", "gloo", "--single-device", "--dump-grad", two, *flags, timeout=1200)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo" and d["launch_fallback"] is None
    assert d["launch_mode"] == "hip_graph_replay_front+eager_allreduce_adamw"
    assert [r["rank"] for r in d["per_rank"]] == [0, 1]
    for r in d["per_rank"]:
        assert r["host_ms_per_step"] > 0 and r["allreduce_wait_ms"] is not None and r["allreduce_wait_ms"] >= 0 and r["step_ms_median"] > 0
    assert d["config"]["global_batch"] == 128 and "configs[3]" in d["config"]["workload"]
    e = _run("--emulate-world", "2", "--eager", "--no-all-outputs", "--dump-grad", one, *flags, timeout=1200)
    assert e["n_gpus"] == 1 and e["config"]["global_batch"] == 128
    a, b = th.load(two), th.load(one)
    assert a["world"] == 2 and b["world"] == 1 and b["batch"] == 128
    assert th.equal(a["pred_c"].view(-1), b["pred_c"].view(-1)[:64]) or float((a["pred_c"].view(-1) - b["pred_c"].view(-1)[:64]).abs().max()) <= 1e-4 * max(1.0, float(b["pred_c"].abs().max()))
    scale = float(b["flat"].abs().max())
    assert scale > 0 and float((a["flat"] - b["flat"]).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("mode", ["default", "--eager"])
def test_one_rank_rccl_group_runs_the_multi_rank_path(gpu, mode, tmp_path):
    """`bench.py --gpus 1 --force-collective` forms a ONE-rank nccl (= RCCL) process group on the single
    GPU and runs the code path of the N > 1 step -- default mode: StepGraph(front) replayed next to RCCL's stream, then the
    all-reduce and AdamW launched eagerly; --eager: sync.sync(async_op=True) -> the next batch's prepare_joint ->
    sync.finish (stream-ordered Work.wait) -> AdamW.  One rank: the all-reduce is the identity, so the step must equal the
    collective-free step -- the gradient dump of the forced run equals the plain run's bit for bit.  No scaling claim."""
    import torch as th
    flags = ["--force-collective", "--no-cpu-baseline", "--extended-steps", "0"] + ([] if mode == "default" else [mode])
    forced = str(tmp_path / "forced.pt")
    d = _run(*flags, "--dump-grad", forced)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["backend"] == "nccl (RCCL)" and d["collective_forced"] is True
    assert d["value"] > 0 and abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]
    if mode == "default":
        assert d["launch_mode"] == "hip_graph_replay_front+eager_allreduce_adamw" and d["launch_fallback"] is None
    else:
        assert d["launch_mode"] == "eager"
    pr = d["per_rank"][0]
    assert pr["rank"] == 0 and pr["allreduce_wait_ms"] is not None and pr["allreduce_wait_ms"] >= 0.0    # the wait was timed: the collective ran
    plain = str(tmp_path / "plain.pt")
    _run("--eager", "--no-cpu-baseline", "--extended-steps", "0", "--no-all-outputs", "--no-gate-compact", "--dump-grad", plain)
    a, b = th.load(forced), th.load(plain)
    assert th.equal(a["flat"], b["flat"]) and th.equal(a["pred_c"], b["pred_c"])

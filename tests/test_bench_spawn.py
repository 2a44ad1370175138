"""bench.py starts its own rank processes: ``python bench.py --gpus N`` with no launcher must run N ranks (VERDICT r2:
the flag used to be parsed and ignored, so a scaling run printed an N = 1 number).  ``--spawn-check`` runs everything of
the multi-rank plumbing that needs no GPU: the parent starts N fresh interpreters with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_*, they form the process group (gloo here), count themselves with an all-reduce, rank 0 prints the line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return env


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout          # rank 0's line, once
    return json.loads(lines[0])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n", [2, 3, 8])      # 8: the target node's rank count
def test_gpus_flag_starts_that_many_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--backend", "gloo", "--single-device", "--steps", "1", "--spawn-check"],
                       env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert r.returncode == 0, r.stderr
    line = _line(r.stdout)
    assert line["n_gpus"] == n and line["ranks_seen"] == n and line["backend"] == "gloo"
    assert line["launcher"].startswith("bench.py")


@pytest.mark.timeout(300)
def test_under_an_external_launcher_no_second_spawn():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--spawn-check"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=280)
    assert r.returncode == 0, r.stderr
    assert _line(r.stdout)["n_gpus"] == 1


@pytest.mark.timeout(300)
def test_world_size_must_equal_gpus():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--spawn-check"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=280)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


@pytest.mark.timeout(300)
def test_parent_fails_when_a_rank_fails():
    """Without a GPU the real (non --spawn-check) ranks exit with an error: the parent must report it, not hang."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--single-device", "--steps", "1"], env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert r.returncode != 0
    assert "needs an AMD GPU" in r.stderr and "rank process failed" in r.stderr


def test_default_mode_tries_the_graph_child_then_the_eager_child():
    """``python bench.py`` at N = 1 replays the step from a HIP graph in a child process and falls back to an eager child
    when that one fails (no GPU needed: the children are stubbed by DMP_BENCH_CHILD_SELFTEST)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode, want, fallback in (("graph_ok", "graph", False), ("graph_fails", "eager", True)):
        env = dict(os.environ, DMP_BENCH_CHILD_SELFTEST=mode)
        env.pop("DMP_BENCH_CHILD", None)
        env.pop("WORLD_SIZE", None)
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["config"]["launch"] == want
        assert ("launch_fallback" in d["config"]) == fallback
        assert (b"failed" in p.stderr) == fallback

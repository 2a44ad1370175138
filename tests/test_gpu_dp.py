"""The data-parallel path with the PRODUCT model on a real GPU: two freshly started processes (gloo, both on cuda:0 --
the GPU box has one device; RCCL replaces gloo on a multi-GPU node, the code path is the same ``torch.distributed``
calls) each run their contiguous half of a global batch through the full DMPNN on the fused path; the averaged flat
gradient and the parameters after one AdamW step must equal a single-process run over the whole batch (losses are batch
means, train.py:463-480, so the average of the shard gradients is the global-batch gradient)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("overlap", [False, True])
def test_two_rank_gradient_equals_global_batch(overlap, gpu, tmp_path):
    sys.path.insert(0, HERE)
    import dp_worker
    import bench
    world, out = 2, str(tmp_path / "rank0.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world), DP_TEST_BATCH="64",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), out, "1" if overlap else "0"],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(out)
    # single process, whole batch
    cfg = dict(bench.CFG, batch=64, act="leaky_relu", emb="Equivariant")
    grad, params, loss = dp_worker.one_step(cfg, gpu, 0, 1, False)
    ref_g, ref_p = grad.cpu().numpy(), params.cpu().numpy()
    scale = max(1.0, float(np.abs(ref_g).max()))
    assert got["grad"].shape == ref_g.shape
    assert float(np.abs(got["grad"] - ref_g).max()) <= 2e-5 * scale, float(np.abs(got["grad"] - ref_g).max())
    # after one AdamW step (lr 1e-3): the first Adam update is lr * g / (|g| + eps), i.e. it amplifies differences in
    # gradients near zero -- compared where the gradient is clearly non-zero, bounded by 2 lr everywhere
    clear = np.abs(ref_g) > 1e3 * max(float(np.abs(got["grad"] - ref_g).max()), 1e-30)
    assert clear.sum() > 100
    assert float(np.abs(got["params"] - ref_p)[clear].max()) <= 1e-5
    assert float(np.abs(got["params"] - ref_p).max()) <= 2.1e-3
    assert abs(float(got["losses"].mean()) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))   # mean of shard means = global mean
    assert float(got["losses"][0]) != float(got["losses"][1])    # the ranks did see different shards


def test_micro_batched_step_equals_whole_batch_step(gpu):
    """bench.py's config-4 form of a step (M micro-batches, gradients summed, each slice's mean loss weighted 1 / M)
    against the same step over the whole shard at once: identical gradient up to fp32 summation order."""
    import bench
    flats = []
    for m in (1, 4):
        cfg = dict(bench.CFG, batch=64, act="leaky_relu", emb="Equivariant", micro_batches=m)
        step, _ = bench.build_step(cfg, bench.make_shard(cfg, 0, gpu), gpu)
        assert step.micro_batches == m
        step()
        flats.append(step.sync.flat.clone())
    scale = max(1.0, float(flats[0].abs().max()))
    assert float((flats[0] - flats[1]).abs().max()) <= 2e-5 * scale
    # round 3: rows are addressed by index, so config 4's 8.4 M-row shard is one pass too (4 in round 2: 32-bit byte offsets)
    assert bench.micro_batches_for(dict(bench.CFG)) == 1 and bench.micro_batches_for(dict(bench.CFG4)) == 1

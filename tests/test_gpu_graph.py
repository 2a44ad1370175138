"""HIP-graph replay of a training step (dp.StepGraph, harness.GraphedTrainStep, FlatAdamW(capturable=True)): a replayed
step must compute what the eager step computes -- same losses step by step, same parameters at the end.

Every case runs in a fresh child process (``test_*`` below start ``test_inner_*`` through pytest with
DMP_GRAPH_TESTS_INLINE=1): what goes wrong with a recording on this stack goes wrong as a segfault or a GPU memory fault
of the whole process (see dp.StepGraph), and one such failure must not take the rest of the GPU suite with it."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu
INLINE = os.environ.get("DMP_GRAPH_TESTS_INLINE") == "1"
inner = pytest.mark.skipif(not INLINE, reason="runs in a child process started by its test_* wrapper")
outer = pytest.mark.skipif(INLINE, reason="wrapper: the child process runs the inner case")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _isolated(node):
    """Run ``tests/test_gpu_graph.py::<node>`` in a child process; fail with its output if it fails or dies."""
    env = dict(os.environ, DMP_GRAPH_TESTS_INLINE="1")
    r = subprocess.run([sys.executable, "-X", "faulthandler", "-m", "pytest", "%s::%s" % (os.path.abspath(__file__), node), "-q", "-x",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-25:])
    assert r.returncode == 0 and " passed" in r.stdout, "child exited with %s:\n%s" % (r.returncode, tail)


@outer
def test_capturable_adamw_matches_the_host_step(gpu):
    _isolated("test_inner_capturable_adamw_matches_the_host_step")


@outer
def test_step_graph_records_once_per_signature(gpu):
    _isolated("test_inner_step_graph_records_once_per_signature")


@outer
@pytest.mark.parametrize("with_reg", [False, True])
def test_graphed_training_equals_eager_training(with_reg, gpu):
    _isolated("test_inner_graphed_training_equals_eager_training[%s]" % with_reg)


@outer
def test_fit_with_graph_replay(gpu):
    _isolated("test_inner_fit_with_graph_replay")


@outer
def test_optimizer_state_dict_after_replays(gpu):
    _isolated("test_inner_optimizer_state_dict_after_replays")


@outer
def test_recorded_pack_clears_missing_slices(gpu):
    _isolated("test_inner_recorded_pack_clears_missing_slices")


@outer
def test_fit_with_graph_replay_and_scheduled_regulariser(gpu):
    _isolated("test_inner_fit_with_graph_replay_and_scheduled_regulariser")


@inner
def test_inner_capturable_adamw_matches_the_host_step(gpu):
    """dmp_adamw_step_dev (step count and learning rate in device memory) against dmp_adamw_step_skip (host values)
    over a few steps with a learning-rate change and a parameter range without gradient."""
    from dualmessagepassing_amd.dp import FlatAdamW
    gen = th.Generator().manual_seed(3)
    p0 = th.randn(10007, generator=gen).to(gpu)
    ps = [p0.clone().requires_grad_(True) for _ in range(2)]
    opts = [FlatAdamW([ps[0]], lr=3e-3, weight_decay=1e-2, amsgrad=True),
            FlatAdamW([ps[1]], lr=3e-3, weight_decay=1e-2, amsgrad=True, capturable=True)]
    for t in range(7):
        g = th.randn(10007, generator=gen).to(gpu)
        for p, opt in zip(ps, opts):
            p.grad = g.clone()
            p._dmp_live_runs = [(0, 4000), (6000, 4007)]            # [4000, 6000): no gradient this step
            if t == 4:
                opt.param_groups[0]["lr"] = 1e-3
            opt.step()
        assert float((ps[0] - ps[1]).abs().max()) <= 1e-7 * float(ps[0].abs().max()), t
    assert th.equal(ps[0][4000:6000], p0[4000:6000]) and th.equal(ps[1][4000:6000], p0[4000:6000])
    assert float(opts[1].state[ps[1]]["dev"][0]) == 7.0 and float(opts[1].state[ps[1]]["dev"][1]) == 1e-3


@inner
def test_inner_step_graph_records_once_per_signature(gpu):
    from dualmessagepassing_amd.dp import StepGraph
    acc = th.zeros(4, device=gpu)
    calls = []

    def fn(meta, x):
        calls.append(meta)
        acc.add_(x.sum())
        return x * meta + 1.0, acc

    sg = StepGraph(fn, max_shapes=1)
    xs = [th.full((4,), float(i), device=gpu) for i in range(5)]
    want_acc = 0.0
    for i, x in enumerate(xs):
        y, a = sg(3, x)
        want_acc += 4.0 * i
        assert th.equal(y, x * 3 + 1.0) and float(a[0]) == want_acc, i
    assert sg.eager_calls == 1 and sg.replays == 4 and len(calls) == 2          # eager, recording; then replays only
    y, _ = sg(3, th.ones(6, device=gpu))                                        # a second signature: beyond max_shapes -> eager
    y, _ = sg(3, th.ones(6, device=gpu))
    assert y.shape == (6,) and sg.eager_calls == 3 and sg.replays == 4


@inner
@pytest.mark.parametrize("with_reg", [False, True])
def test_inner_graphed_training_equals_eager_training(with_reg, gpu):
    """Three epochs of count-loss training (batches of two shapes: the last batch of an epoch is smaller), eager vs
    HIP-graph replay from the same initial state: the per-step losses and the final parameters agree to fp32 rounding
    of the optimizer's bias corrections; the learning rate changes between epochs and reaches the replayed steps."""
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
    from dualmessagepassing_amd.harness import GraphedTrainStep, SyntheticPairs, train_epoch
    ds = SyntheticPairs(56, 3, 2, 8, 16, 2, 1, seed=11)
    th.manual_seed(5)
    base = build_model(**ds.model_config(hid_dim=64, layers=2, rep_act_func="leaky_relu", pred_act_func="leaky_relu")).to(gpu)
    runs = []
    for graphed in (False, True):
        model = copy.deepcopy(base)
        sync = FlatGradSync(model)
        master = sync.flatten_parameters()
        start = master.detach().clone()
        opt = FlatAdamW([master], lr=2e-3, weight_decay=1e-5, amsgrad=True, capturable=graphed)
        step = GraphedTrainStep(model, opt, sync, with_rep_reg=with_reg) if graphed else None
        trace = []
        for epoch in range(3):
            opt.param_groups[0]["lr"] = 2e-3 / (1 + epoch)
            train_epoch(model, opt, ds, 16, gpu, sync=sync, neg_slp=0.01, rep_reg_w=0.05 if with_reg else 0.0,
                        order=np.random.default_rng(epoch).permutation(len(ds)), trace=trace, graph=step)
        th.cuda.synchronize()
        runs.append((th.stack([t[0] for t in trace]).cpu(), master.detach().clone(), step))
    (l0, p0, _), (l1, p1, step) = runs
    assert step.steps.replays >= 3 * 4 - 4 and step.steps.eager_calls == 2, (step.steps.replays, step.steps.eager_calls)
    assert l0.shape == l1.shape == (12,)
    assert th.allclose(l0, l1, rtol=1e-5, atol=1e-6), (l0 - l1).abs().max()
    assert float((p0 - p1).abs().max()) <= 1e-5 * float(p0.abs().max()), float((p0 - p1).abs().max())
    assert float((p0 - start).abs().max()) > 1e-3                           # it did train


@inner
def test_inner_fit_with_graph_replay(tmp_path, gpu):
    """harness.fit(graph=True): the run trains through recorded steps (uniform synthetic set: two batch shapes) and leaves
    the same run directory as an eager fit; its dev metric after every epoch equals the eager run's."""
    import os
    from dualmessagepassing_amd import dataio
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
    from dualmessagepassing_amd.harness import SyntheticPairs, fit
    ds = SyntheticPairs(72, 3, 2, 8, 16, 2, 1, seed=9)
    train, dev = ds.subset(range(56)), ds.subset(range(56, 72))
    config = ds.model_config(hid_dim=64, layers=2)
    th.manual_seed(2)
    base = build_model(**config).to(gpu)
    hists = []
    for graphed in (False, True):
        model = copy.deepcopy(base)
        sync = FlatGradSync(model)
        opt = FlatAdamW([sync.flatten_parameters()], lr=2e-3, weight_decay=1e-5, amsgrad=True, capturable=graphed)
        run = os.path.join(tmp_path, "run%d" % graphed)
        hists.append(fit(model, opt, train, dev, 4, 16, gpu, save_dir=run, config=config, sync=sync, neg_slp=0.01, seed=3,
                         graph=graphed))
        assert os.path.exists(dataio.checkpoint_path(run, 3)) and "eval-MAE" in dataio.get_best_epochs(os.path.join(run, "log.txt"))
    for a, b in zip(*hists):
        assert a["dev"]["eval_metric"] == pytest.approx(b["dev"]["eval_metric"], rel=1e-4, abs=1e-5)
        assert a["train"]["bp_loss"] == pytest.approx(b["train"]["bp_loss"], rel=1e-4, abs=1e-5)


@inner
def test_inner_optimizer_state_dict_after_replays(gpu):
    """replays advance the step count on the device only.  ``state_dict()`` must carry the true count, and the
    state must load into a plain (non-capturable) FlatAdamW that then continues exactly like the capturable one."""
    from dualmessagepassing_amd.dp import FlatAdamW, StepGraph
    gen = th.Generator().manual_seed(4)
    p = th.randn(4099, generator=gen).to(gpu).requires_grad_(True)
    g = th.randn(4099, generator=gen).to(gpu)
    p.grad = g.clone()
    opt = FlatAdamW([p], lr=1e-2, weight_decay=1e-2, amsgrad=True, capturable=True)

    def fn(gr):
        p.grad.copy_(gr)
        opt.step()
        return p.detach().sum().view(1)
    run = StepGraph(fn, optimizer=opt)
    with run.on_stream():
        for _ in range(6):                                            # eager, record + replay, 4 replays
            run(g)
    th.cuda.synchronize()
    assert run.replays == 5
    sd = copy.deepcopy(opt.state_dict())                              # load_state_dict keeps same-device tensors by reference
    st = sd["state"][0]
    assert st["step"] == 6 and "dev" not in st and "dev_lr" not in st
    q = p.detach().clone().requires_grad_(True)
    plain = FlatAdamW([q], lr=1e-2, weight_decay=1e-2, amsgrad=True)
    plain.load_state_dict(sd)
    q.grad = g.clone()
    plain.step()                                                      # step 7 on the host-side count
    with run.on_stream():
        run(g)                                                        # step 7 replayed
    th.cuda.synchronize()
    assert float((p - q).abs().max()) <= 1e-7 * float(p.abs().max())
    # and back: a capturable optimizer that loads the state rebuilds its device count from it
    r = q.detach().clone().requires_grad_(True)
    cap = FlatAdamW([r], lr=1e-2, weight_decay=1e-2, amsgrad=True, capturable=True)
    cap.load_state_dict(copy.deepcopy(plain.state_dict()))
    r.grad, q.grad = g.clone(), g.clone()
    cap.step()
    plain.step()
    assert float(cap.state[r]["dev"][0]) == 8.0
    assert float((r - q).abs().max()) <= 1e-7 * float(q.abs().max())


@inner
def test_inner_recorded_pack_clears_missing_slices(gpu):
    """``FlatGradSync.pack`` recorded in a graph must clear the slices of parameters without a gradient at every
    replay -- an eager step with a LARGER live set in between leaves its gradients there otherwise."""
    from dualmessagepassing_amd.dp import FlatGradSync, StepGraph

    class Two(th.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = th.nn.Parameter(th.ones(64))
            self.b = th.nn.Parameter(th.ones(32))
    m = Two().to(gpu)
    sync = FlatGradSync(m)
    x = th.arange(64, dtype=th.float32, device=gpu)

    def only_a(t):
        sync.detach_grads()
        (m.a * t).sum().backward()
        sync.pack()
        return sync.flat.clone()
    run = StepGraph(only_a)
    with run.on_stream():
        run(x)
        out = run(x)                                                  # recorded: b has no gradient
        assert float(out[64:].abs().max()) == 0.0
        sync.detach_grads()                                           # an eager step where BOTH get gradients
        ((m.a * x).sum() + (m.b * 3.0).sum()).backward()
        sync.pack()
        assert float(sync.flat[64:96].min()) == 3.0
        out = run(x)                                                  # replay: b's slice must be cleared again
        th.cuda.synchronize()
        assert run.replays == 2
        assert float(out[64:].abs().max()) == 0.0 and th.equal(out[:64], x)
        sync.detach_grads()                                           # an eager step with the recording's live set clears too
        ((m.a * x).sum() + (m.b * 3.0).sum()).backward()
        sync.pack()
        sync.detach_grads()
        (m.a * x).sum().backward()
        sync.pack()
        assert float(sync.flat[64:].abs().max()) == 0.0


@inner
def test_inner_fit_with_graph_replay_and_scheduled_regulariser(gpu):
    """``fit(graph=True, schedule=RunSchedule(config))`` with a non-zero / annealed ``rep_reg_w`` in the config
    used to raise mid-epoch (the recorded step was built without the regulariser)."""
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
    from dualmessagepassing_amd.harness import RunSchedule, SyntheticPairs, fit
    ds = SyntheticPairs(48, 3, 2, 8, 16, 2, 1, seed=9)
    train, dev = ds.subset(range(32)), ds.subset(range(32, 48))
    config = ds.model_config(hid_dim=64, layers=2)
    th.manual_seed(2)
    base = build_model(**config).to(gpu)
    hists = []
    for graphed in (False, True):
        model = copy.deepcopy(base)
        sync = FlatGradSync(model)
        opt = FlatAdamW([sync.flatten_parameters()], lr=2e-3, weight_decay=1e-5, amsgrad=True, capturable=graphed)
        run_cfg = dict(config, train_batch_size=16, train_epochs=3, lr=2e-3, scheduler="constant", rep_reg_w=0.05, neg_pred_slp=0.01)
        hists.append(fit(model, opt, train, dev, 3, 16, gpu, sync=sync, seed=3, graph=graphed, schedule=RunSchedule(run_cfg, len(train))))
    for a, b in zip(*hists):
        assert a["train"]["bp_loss"] == pytest.approx(b["train"]["bp_loss"], rel=1e-4, abs=1e-5)

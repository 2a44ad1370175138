"""The index builds of a step on the side stream (dualmessagepassing_amd/side.py, dmpnn.prefetch_joint_indexes; round 5).

Everything the joint pass derives from the batch's structure and its two 0 / 1 gates (the union's CSR, coefficients, selectors,
row masks, kept-row lists, the kept edges' tiles / CSR / incidence CSR, the pooling indexes) is issued on a second stream beside
the embedding and first-layer kernels.  Same kernels, same inputs, same order of additions -- only the stream differs: the step
must equal the one-stream step BIT FOR BIT, eagerly and as a replayed HIP graph (where the fork becomes a branch of the graph),
and every stage must be joined when the forward pass returns."""
import os
import sys

import pytest
import torch as th

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(gpu, batch=96, filt="ScalarFilter"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_compact import _model_and_batch, _outputs_and_grads
    sys.path.insert(0, ROOT)
    import bench
    cfg = dict(bench.CFG, batch=batch, filter=filt)
    bench_, shard, step, model = _model_and_batch(cfg, gpu)
    return bench_, cfg, shard, step, model, _outputs_and_grads


@pytest.mark.parametrize("lazy,filt", [(True, "ScalarFilter"), (False, "ScalarFilter"), (True, "None")])
def test_step_with_the_side_stream_equals_the_one_stream_step_bit_for_bit(lazy, filt, gpu):
    """(``filt`` "None": the model without a filter net -- the branch then carries the structure's own indexes only.)"""
    from dualmessagepassing_amd import dmpnn, side
    bench_, cfg, shard, step, model, run = _setup(gpu, filt=filt)
    side.USE_SIDE_STREAM = False
    try:
        ref_out, ref_flat = run(bench_, cfg, shard, step, model, lazy)
    finally:
        side.USE_SIDE_STREAM = True
    forked = []
    real = dmpnn.prefetch_joint_indexes

    def spy(*a, **k):
        real(*a, **k)
        forked.append(len(side.pending()))

    dmpnn.prefetch_joint_indexes = spy
    try:
        import dualmessagepassing_amd.basemodel  # noqa: F401  (the model looks the function up in dmpnn at call time)
        out, flat = run(bench_, cfg, shard, step, model, lazy)
    finally:
        dmpnn.prefetch_joint_indexes = real
    assert forked and max(forked) >= 5, "the prefetch did not fork its stages: %r" % (forked,)
    assert not side.pending(), "stages left unjoined after the pass: %r" % (side.pending(),)
    for k, v in ref_out.items():
        if v is None:
            assert out[k] is None
        else:
            assert th.equal(out[k], v), k
    assert th.equal(flat, ref_flat)


def test_side_stream_stages_are_waited_in_order_and_joined(gpu):
    """``side.fork`` / ``mark`` / ``wait`` / ``join`` on their own: a stage's wait drops the earlier stages, work issued in a fork
    is visible after its wait, a nested fork is a no-op, nothing stays pending after ``join``."""
    from dualmessagepassing_amd import side
    x = th.zeros(1 << 20, device=gpu)
    with side.fork() as f:
        assert f and side.active()
        with side.fork() as inner:
            assert inner is False
        a = x + 1.0
        side.mark("a")
        b = a * 3.0
        side.mark("b")
        c = b - 1.0
        side.mark("c")
    assert side.pending() == ["a", "b", "c", "_end"]
    side.wait("b")
    assert side.pending() == ["c", "_end"]
    assert float(b.sum()) == 3.0 * (1 << 20)
    side.wait("nope")                                   # an unknown stage: nothing happens
    side.join()
    assert not side.pending() and float(c.sum()) == 2.0 * (1 << 20)
    side.USE_SIDE_STREAM = False
    try:
        with side.fork() as f:
            assert f is False and not side.active()
    finally:
        side.USE_SIDE_STREAM = True


def test_replayed_step_with_the_forked_branch_equals_the_one_stream_recording(gpu):
    """``dp.StepGraph`` records bench.py's step (forward + backward + gradient pack) with its forked index branch and replays it:
    predictions and the flat gradient equal those of a recording made with the side stream off, bit for bit."""
    sys.path.insert(0, ROOT)
    import bench
    from dualmessagepassing_amd import side
    from dualmessagepassing_amd.dp import StepGraph
    cfg = dict(bench.CFG, batch=64)
    shard = bench.make_shard(cfg, 0, gpu)
    step, model = bench.build_step(dict(cfg, graph=True), shard, gpu, 1)
    gen = th.Generator().manual_seed(7)
    with th.no_grad():
        for prm in step.sync.params:
            prm.add_((0.05 * th.randn(prm.shape, generator=gen)).to(gpu))
    res = {}
    for on in (False, True):
        side.USE_SIDE_STREAM = on
        try:
            g = StepGraph(lambda: step.front(), optimizer=None, max_shapes=1)
            with g.on_stream():
                g()                                   # eager
                g()                                   # recorded + replayed
                step.sync.flat.fill_(float("nan"))
                g()                                   # replayed
                assert g.replays == 2
                th.cuda.synchronize()
                res[on] = (step.last_pred_c.detach().clone(), step.sync.flat.detach().clone())
        finally:
            side.USE_SIDE_STREAM = True
        assert not side.pending()
    assert bool(th.isfinite(res[True][1]).all())
    assert th.equal(res[True][0], res[False][0]) and th.equal(res[True][1], res[False][1])


def test_failed_fork_leaves_no_stage_behind(gpu):
    """an exception inside ``fork()`` must not leave its stages pending (a later ``wait`` by name would order the
    caller behind products that were never made); the state is per host thread."""
    import threading
    from dualmessagepassing_amd import side
    x = th.ones(1 << 16, device=gpu)
    with side.fork():
        y = x + 1.0
        side.mark("kept")
    with pytest.raises(RuntimeError, match="boom"):
        with side.fork():
            z = x * 2.0
            side.mark("lost")
            raise RuntimeError("boom")
    assert side.pending() == ["kept", "_end"] and not side.active()
    seen = []
    t = threading.Thread(target=lambda: seen.append(side.pending()))     # another thread: its own (empty) list
    t.start(); t.join()
    assert seen == [[]]
    side.join()
    assert not side.pending() and float(y.sum()) == 2.0 * (1 << 16) and float(z.sum()) == 2.0 * (1 << 16)


def test_row_mask_miss_on_the_main_stream_waits_for_the_side_stream(gpu):
    """``fused.gate_row_mask`` called on the caller's stream BEFORE any wait, for a gate that the side stream is
    still writing (a memo miss: the prefetch's and the consumer's conditions drifted apart).  The launch is ordered behind the
    side stream's ``erows`` stage, so the mask is the finished gate's."""
    from dualmessagepassing_amd import fused, side
    R = 1 << 22
    src = (th.arange(R, device=gpu) % 3 == 0).float()
    th.cuda.synchronize()
    with side.fork():
        big = th.zeros(64 << 20, device=gpu)
        for _ in range(6):                   # a few hundred microseconds of side-stream work ahead of the gate's writer
            big.add_(1.0)
        gate = th.empty(R, device=gpu)
        gate.copy_(src)
        side.mark("erows")
    mask = fused.gate_row_mask(gate)         # a miss: nothing memoised this mask
    side.join()
    th.cuda.synchronize()
    bits = ((mask.view(-1, 1).to(th.int64) >> th.arange(32, device=gpu)) & 1).reshape(-1)[:R].float()
    assert th.equal(bits, src)

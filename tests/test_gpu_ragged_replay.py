"""Replay for ragged batches.

The reference trains on bucket-sorted RAGGED batches (SubgraphCountingMatching/utils/sampler.py:10-84, train.py:1283-1290):
every batch has its own node / edge totals, so a step recorded per exact shape never replays.  ``PairDataset.batch_arrays(pad=)``
extends a batch by inert pairs to one of a few capacity levels; pairs are independent in the model, an inert pair carries no loss.
Here: the reference's own ragged model run (tests/golden/fullmodel_ragged.npz) through the padded batch -- predictions and
gradients of the real pairs as the reference has them -- and a padded, replayed training run against the exact-shape eager run."""
import numpy as np
import pytest
import torch as th

from conftest import golden_files, load_golden

pytestmark = pytest.mark.gpu


def _t(a):
    return th.from_numpy(np.asarray(a))


def _close(got, ref, tol, what):
    got, ref = got.detach().double().cpu(), _t(ref).double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(1.0, float(ref.abs().max())) if ref.numel() else 1.0
    err = float((got - ref).abs().max()) if ref.numel() else 0.0
    assert err <= tol * scale, "%s: max err %g (scale %g)" % (what, err, scale)


def _padded_graph(d, t, dev, extra, cap_n, cap_e):
    """The fixture's batched graph ``t`` + ``extra`` inert graphs (harness.PairDataset._inert_graphs) that take its totals to the caps."""
    from dualmessagepassing_amd.graph import BatchedGraph
    from dualmessagepassing_amd.harness import PairDataset
    bnn, bne = np.asarray(d[t + "_bnn"]), np.asarray(d[t + "_bne"])
    inert = PairDataset._inert_graphs((int(bnn.sum()), int(bne.sum())), cap_n, cap_e, extra)
    off = int(bnn.sum()) + np.concatenate([[0], np.cumsum([g["num_nodes"] for g in inert])])[:-1]
    src = np.concatenate([np.asarray(d[t + "_src"])] + [g["src"] + o for g, o in zip(inert, off)])
    dst = np.concatenate([np.asarray(d[t + "_dst"])] + [g["dst"] + o for g, o in zip(inert, off)])
    nn_ = np.concatenate([bnn, [g["num_nodes"] for g in inert]]).astype(np.int64)
    ne_ = np.concatenate([bne, [len(g["src"]) for g in inert]]).astype(np.int64)
    g = BatchedGraph(_t(src).to(dev), _t(dst).to(dev), int(nn_.sum()), _t(nn_).to(dev), _t(ne_).to(dev))
    n_pad, e_pad = int(nn_.sum() - bnn.sum()), int(ne_.sum() - bne.sum())
    for k, v in d.items():
        if k.startswith(t + "_ndata."):
            name, v = k.split(".", 1)[1], np.asarray(v)
            fill = np.concatenate([np.arange(x["num_nodes"]) for x in inert]) if name == "id" else np.zeros(n_pad, v.dtype)
            if name in ("in_deg", "out_deg"):       # cached degrees (dataset.py:1230-1236): the ring's
                deg = [np.bincount(x["dst" if name == "in_deg" else "src"], minlength=x["num_nodes"]) for x in inert]
                fill = np.concatenate(deg)
            g.ndata[name] = _t(np.concatenate([v, fill.astype(v.dtype)])).to(dev)
        if k.startswith(t + "_edata."):
            name, v = k.split(".", 1)[1], np.asarray(v)
            fill = np.concatenate([np.arange(len(x["src"])) for x in inert]) if name == "id" else np.zeros(e_pad, v.dtype)
            g.edata[name] = _t(np.concatenate([v, fill.astype(v.dtype)])).to(dev)
    return g


def test_reference_ragged_run_through_a_padded_batch(gpu):
    """fullmodel_ragged.npz (the reference's own model on 6 ragged pairs) with 6 inert pairs appended and both sides padded to a
    capacity: the real pairs' ``pred_c`` (2e-4) and the gradients of their sum (3e-4) as the reference has them."""
    from dualmessagepassing_amd.basemodel import build_model
    d = load_golden([p for p in golden_files("fullmodel_") if "ragged" in p][0])
    config = {str(k): eval(str(v)) for k, v in zip(d["config_keys"], d["config_vals"])}
    model = build_model(**config)
    model.load_state_dict({k[3:]: _t(v) for k, v in d.items() if k.startswith("sd.")}, strict=True)
    model.to(gpu)
    B = len(d["p_bnn"])
    pattern = _padded_graph(d, "p", gpu, B, 64, 160)
    graph = _padded_graph(d, "g", gpu, B, 160, 640)
    assert pattern.number_of_nodes() == 64 and pattern.number_of_edges() == 160 and graph.number_of_nodes() == 160 and graph.number_of_edges() == 640
    out = model(pattern, graph)
    assert out["pred_c"].shape == (2 * B, 1)
    _close(out["pred_c"][:B], d["out.pred_c"], 2e-4, "pred_c of the real pairs")
    out["pred_c"][:B].sum().backward()
    n = 0
    for k, p in model.named_parameters():
        if "grad." + k in d:
            _close(p.grad, d["grad." + k], 3e-4, "grad " + k)
            n += 1
    assert n > 20


def test_padded_batches_replay_and_train_like_the_exact_shapes(gpu):
    """A ragged dataset (harness.SmallLikePairs) for two epochs: ``GraphedTrainStep`` with padding replays almost every step
    (at most ``max_shapes`` levels, each run eagerly once and recorded once) and ends where the exact-shape eager run ends."""
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
    from dualmessagepassing_amd.harness import GraphedTrainStep, SmallLikePairs, train_epoch
    ds = SmallLikePairs(224, seed=3, threads=4)
    B, epochs = 32, 2
    finals, steps = [], None
    for graphed in (False, True):
        th.manual_seed(0)
        model = build_model(**ds.model_config(hid_dim=64, layers=3, rep_act_func="leaky_relu", pred_act_func="leaky_relu",
                                              emb_net="Equivariant")).to(gpu)
        sync = FlatGradSync(model)
        opt = FlatAdamW([sync.flatten_parameters()], lr=2e-4, weight_decay=1e-5, amsgrad=True, capturable=graphed)
        step = GraphedTrainStep(model, opt, sync, max_shapes=4) if graphed else None
        losses = []
        ctx = step.steps.on_stream() if graphed else None
        if ctx is not None:
            ctx.__enter__()
        try:
            for ep in range(epochs):
                r = train_epoch(model, opt, ds, B, gpu, sync=sync, neg_slp=0.01, order=np.random.default_rng(ep).permutation(len(ds)), graph=step)
                losses.append(r["bp_loss"])
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
        th.cuda.synchronize()
        finals.append((sync.master.detach().clone(), losses))
        if graphed:
            steps = step.steps
    n_steps = epochs * (len(ds) // B)
    assert steps.replays >= n_steps - 2 * 4 and steps.eager_calls <= 4, (steps.replays, steps.eager_calls, n_steps)
    (p0, l0), (p1, l1) = finals
    scale = float(p0.abs().max())
    assert float((p0 - p1).abs().max()) <= 2e-4 * scale, float((p0 - p1).abs().max()) / scale
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (l0, l1)


def test_padded_replay_with_gate_compact_drops_no_step(gpu):
    """``fit(graph=True, gate_compact=...)`` on a RAGGED dataset.  The inert target graphs of a padded batch carry a
    label their inert patterns do not use, so the filter gates wipe every inert row: a capacity calibrated on the real pairs
    holds for the padded batches (nothing dropped) and the run ends where the exact-shape eager run on every edge row ends."""
    from dualmessagepassing_amd import harness
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
    ds = harness.SmallLikePairs(160, seed=4, threads=4)
    train, dev = ds.subset(range(128)), ds.subset(range(128, 160))
    # the inert rows are dead under the gates: a padded batch keeps exactly the real pairs' rows
    pad = train.pad_buckets(32, levels=4)
    meta, tensors = train.batch_arrays(np.arange(32), gpu, pad=pad)
    pattern, graph = train.graphs_from_arrays(meta, tensors)
    th.manual_seed(0)
    probe = build_model(**ds.model_config(hid_dim=128, layers=3)).to(gpu)
    real_p, real_g = train.batchify(np.arange(32), gpu)[:2]
    kept_padded, e_padded = probe.gate_kept_edges(pattern, graph)
    kept_real, e_real = probe.gate_kept_edges(real_p, real_g)
    assert kept_padded == kept_real and e_padded > e_real, (kept_padded, kept_real, e_padded, e_real)
    hist = {}
    for mode in ("plain", "padded+compact"):
        th.manual_seed(0)
        model = build_model(**ds.model_config(hid_dim=128, layers=3)).to(gpu)
        sync = FlatGradSync(model)
        opt = FlatAdamW([sync.flatten_parameters()], lr=5e-4, weight_decay=1e-5, amsgrad=True, capturable=mode != "plain")
        hist[mode] = harness.fit(model, opt, train, dev, 2, 32, gpu, sync=sync, seed=1,
                                 gate_compact=1.25 if mode != "plain" else False, graph=mode != "plain")
        if mode != "plain":
            assert model.gate_capacity, "the gates removed too little for a capacity to be set"
            assert hist[mode][-1]["dropped_steps"] == 0, hist[mode][-1]
    for a, b in zip(hist["plain"], hist["padded+compact"]):
        for part in ("train", "dev"):
            for k, v in a[part].items():
                assert abs(v - b[part][k]) <= 2e-3 * max(1.0, abs(v)), (part, k, v, b[part][k])

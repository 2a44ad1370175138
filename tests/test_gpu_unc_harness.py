"""The UNC training loop (unc_harness: main.py:99-211 of the reference's UNC model) on a small two-community graph:
file formats round-trip, the link-prediction loss falls, linked nodes end up closer than unlinked ones, the output pass
covers the sampled nodes with the reference's blending rule."""
import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu


def _two_communities(rng, n=240, m=1500):
    half = n // 2
    u = rng.integers(0, n, 4 * m)
    v = rng.integers(0, n, 4 * m)
    same = (u < half) == (v < half)
    keep = (u != v) & (same | (rng.random(4 * m) < 0.03))          # mostly intra-community edges
    pairs = np.unique(np.stack([u[keep], v[keep]], 1), axis=0)[:m]
    return np.stack([pairs[:, 0], np.zeros(len(pairs), np.int64), pairs[:, 1]], 1)


@pytest.mark.parametrize("sampler", ["neighbor", "randomwalk"])
def test_unsupervised_training_learns_the_link_structure(sampler, gpu):
    from dualmessagepassing_amd.unc import TrainModel
    from dualmessagepassing_amd.unc_harness import collect_node_embeddings, graph_of, train_unsupervised
    rng = np.random.default_rng(7)
    n = 240
    trip_np = _two_communities(rng, n)
    graph, trip = graph_of(trip_np, n, 1, gpu)
    th.manual_seed(0)
    model = TrainModel(None, n, 64, 1, 0, num_hidden_layers=2, dropout=0.0, reg_param=0.01).to(gpu)
    logs = []
    hist = train_unsupervised(model, graph, trip, n_epochs=6, graph_batch_size=300, lr=5e-3, sampler=sampler, sample_depth=2,
                              sample_width=8, negative_sample=3, rescale_epochs=False, seed=1, log=logs.append)
    assert len(hist) == len(logs) >= 2 and np.isfinite(hist).all()
    assert min(hist) < 0.9 * hist[0], hist                       # the link-prediction loss falls
    if len(hist) < 6:                                            # stopped early: exactly at the first rise
        assert hist[-1] > hist[-2] and all(b <= a for a, b in zip(hist[:-2], hist[1:-1]))
    emb, covered = collect_node_embeddings(model, graph, trip, graph_batch_size=300, sampler=sampler, sample_depth=2,
                                           sample_width=8, negative_sample=3, seed=2)
    assert emb.shape == (n, 64) and bool(th.isfinite(emb).all())
    touched = th.zeros(n, dtype=th.bool, device=gpu)
    touched[trip[:, 0]] = True
    touched[trip[:, 2]] = True
    assert bool((covered | ~touched).all())                      # every endpoint of a training edge was sampled at least once
    # untouched rows keep the table's values
    table = model.model.node_emb.weight.detach()
    if bool((~covered).any()):
        assert th.equal(emb[~covered], table[~covered])
    # the trained score separates true edges from random pairs
    with th.no_grad():
        model.eval()
        fake = th.stack([th.randint(0, n, (trip.size(0),), device=gpu), th.zeros(trip.size(0), dtype=th.int64, device=gpu),
                         th.randint(0, n, (trip.size(0),), device=gpu)], 1)
        pos, neg = model.calc_score(emb, trip), model.calc_score(emb, fake)
    assert float(pos.mean()) > float(neg.mean()) + 0.1, (float(pos.mean()), float(neg.mean()))


def test_training_loop_at_the_reference_default_sample_width(gpu):
    """``train_unsupervised`` / ``collect_node_embeddings`` default to sample_width = 128 (main.py:294); the
    default call must run (a hub with more than 128 in-edges makes the sampler actually select)."""
    from dualmessagepassing_amd.unc import TrainModel
    from dualmessagepassing_amd.unc_harness import collect_node_embeddings, graph_of, train_unsupervised
    rng = np.random.default_rng(3)
    n = 400
    trip_np = _two_communities(rng, n)
    hub = np.stack([np.arange(1, 301), np.zeros(300, np.int64), np.zeros(300, np.int64)], 1)   # 300 edges into node 0
    trip_np = np.unique(np.concatenate([trip_np, hub]), axis=0)
    graph, trip = graph_of(trip_np, n, 1, gpu)
    assert int(graph.in_degrees().max()) > 128
    th.manual_seed(0)
    model = TrainModel(None, n, 32, 1, 0, num_hidden_layers=2, dropout=0.0, reg_param=0.01).to(gpu)
    hist = train_unsupervised(model, graph, trip, n_epochs=2, graph_batch_size=500, lr=5e-3, sampler="neighbor", negative_sample=2,
                              rescale_epochs=False, seed=1)
    assert len(hist) >= 1 and np.isfinite(hist).all()
    emb, covered = collect_node_embeddings(model, graph, trip, graph_batch_size=500, sampler="neighbor", negative_sample=2, seed=2)
    assert emb.shape == (n, 32) and bool(th.isfinite(emb).all())


def test_padded_sampled_step_equals_the_step_on_the_sub_graph_itself(gpu):
    """``unc_harness.pad_sampled`` + ``unc.PaddedRows`` (a sampled sub-graph padded to fixed capacities with inert nodes and
    edges: what lets its step replay): loss and EVERY parameter gradient of the padded step equal the step on the sub-graph
    itself -- BatchNorm statistics, the per-relation means and the regularisers' means run over the real rows only, an inert
    row contributes nothing -- and so do the BatchNorm running statistics afterwards."""
    import copy
    from dualmessagepassing_amd.graph import BatchedGraph
    from dualmessagepassing_amd.unc import PaddedRows, TrainModel
    from dualmessagepassing_amd.unc_harness import graph_of, pad_sampled
    from dualmessagepassing_amd.unc_sampling import generate_sampled_graph_and_labels_unsupervised
    rng = np.random.default_rng(3)
    n = 600
    trip_np = _two_communities(rng, n)
    graph, trip = graph_of(trip_np, n, 1, gpu)
    th.manual_seed(0)
    model = TrainModel(None, n, 64, 1, 0, num_hidden_layers=2, dropout=0.0, reg_param=0.01).to(gpu)
    twin = copy.deepcopy(model)
    gen = th.Generator(device=gpu).manual_seed(5)
    sub, samples, labels = generate_sampled_graph_and_labels_unsupervised(graph, trip[:256], 2, 8, 0.5, 3, generator=gen, sampler="neighbor")
    et = sub.edata["type"]
    emb, _ = model(sub, sub.ndata["_ID"], et, sub.edata["norm"])
    loss = model.get_unsupervised_loss(sub, emb, et, samples, labels)
    loss.backward()
    (ncap, ecap), (src, dst, etp, norm, nid, counts) = pad_sampled(sub, et, 512, 2048)
    assert ncap > sub.number_of_nodes() and ecap >= sub.number_of_edges() and ncap % 512 == 0 and ecap % 2048 == 0
    g = BatchedGraph(src, dst, ncap)
    g._dmp_valid = PaddedRows(counts[0:1], counts[1:2], ncap, ecap)
    emb2, _ = twin(g, nid, etp, norm)
    loss2 = twin.get_unsupervised_loss(g, emb2, etp, samples, labels)
    loss2.backward()
    assert abs(float(loss2) - float(loss)) <= 2e-6 * max(1.0, abs(float(loss)))
    N, E = sub.number_of_nodes(), sub.number_of_edges()
    assert float((emb2[0][:N] - emb[0]).abs().max()) <= 2e-5 * max(1.0, float(emb[0].abs().max()))
    assert float((emb2[1][:E] - emb[1]).abs().max()) <= 2e-5 * max(1.0, float(emb[1].abs().max()))
    assert float((emb2[2] - emb[2]).abs().max()) <= 2e-5 * max(1.0, float(emb[2].abs().max()))
    for (name, p), q in zip(model.named_parameters(), twin.parameters()):
        if p.grad is None:
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, name
            continue
        # (a bias in front of a BatchNorm has gradient zero in exact arithmetic: both sides hold rounding noise ~1e-7 there)
        s = max(1e-6, float(p.grad.abs().max()))
        assert float((q.grad - p.grad).abs().max()) <= 5e-5 * s + 1e-6, (name, float((q.grad - p.grad).abs().max()), s)
    for (name, b), c in zip(model.named_buffers(), twin.buffers()):
        assert th.allclose(b.float(), c.float(), rtol=1e-5, atol=1e-6), name


@pytest.mark.parametrize("sampler", ["neighbor", "randomwalk"])
def test_replayed_sampled_training_follows_the_eager_loss_curve(sampler, gpu):
    """``train_unsupervised(replay=True)`` (``SampledStep``: every step after the sampling is one HIP-graph replay on padded
    capacities) against the eager loop from the same seed -- the same sub-graphs, step by step: the per-epoch mean losses agree
    to 1e-3, most steps were replays, and only a few shapes were recorded."""
    import copy
    from dualmessagepassing_amd import unc_harness
    from dualmessagepassing_amd.unc import TrainModel
    rng = np.random.default_rng(11)
    n = 900
    trip_np = _two_communities(rng, n)
    graph, trip = unc_harness.graph_of(trip_np, n, 1, gpu)
    th.manual_seed(0)
    model = TrainModel(None, n, 64, 1, 0, num_hidden_layers=2, dropout=0.0, reg_param=0.01).to(gpu)
    twin = copy.deepcopy(model)
    kw = dict(n_epochs=3, graph_batch_size=300, lr=5e-3, sampler=sampler, sample_depth=2, sample_width=8, negative_sample=3,
              rescale_epochs=False, seed=1)
    ref = unc_harness.train_unsupervised(model, graph, trip, **kw)
    made = []
    real = unc_harness.SampledStep
    unc_harness.SampledStep = lambda *a, **k: (made.append(real(*a, **k)), made[-1])[1]
    try:
        got = unc_harness.train_unsupervised(twin, graph, trip, replay=True, **kw)
    finally:
        unc_harness.SampledStep = real
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        assert abs(a - b) <= 1e-3 * max(1.0, abs(b)), (got, ref)
    sg = made[0].steps
    assert sg.replays >= 0.5 * (sg.replays + sg.eager_calls), (sg.replays, sg.eager_calls)
    # (the parameters themselves are not compared: Adam moves an entry whose gradient is rounding noise -- a bias in front of
    # a BatchNorm -- by its full rate in the noise's direction; the node embeddings the run is for are)
    e0, e1 = model.model.node_emb.weight.detach(), twin.model.node_emb.weight.detach()
    assert float((e0 - e1).abs().mean()) <= 0.01 * float(e0.abs().mean())      # (single entries: the same amplification)
    assert len(sg._graphs) <= 8

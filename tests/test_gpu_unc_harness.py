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

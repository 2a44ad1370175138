"""UNC mini-batch construction on the device (unc_sampling.py): the integer parts against the
reference's own functions (tests/golden/unc_sampling.npz, utils.py:539-567), the random parts through
the properties that define them (DGL's samplers are not available; their random streams are not pinned)."""
import numpy as np
import pytest
import torch as th

from conftest import golden_files, load_golden

pytestmark = pytest.mark.gpu


def _graph(n, m, gpu, seed=0):
    from dualmessagepassing_amd.graph import BatchedGraph
    rng = np.random.default_rng(seed)
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    g = BatchedGraph(th.from_numpy(src).to(gpu), th.from_numpy(dst).to(gpu), n)
    g.edata["type"] = th.from_numpy(rng.integers(0, 3, m)).to(gpu)
    return g, src, dst


def test_negative_sampling_and_nid_conversion_match_reference(gpu):
    from dualmessagepassing_amd.unc_sampling import convert_subgraph_nids, negative_sampling
    d = load_golden(golden_files("unc_sampling")[0])
    neg = negative_sampling(th.from_numpy(d["pos"]).to(gpu), int(d["num_entity"]), int(d["negative_rate"]),
                            values=th.from_numpy(d["values"]).to(gpu), choices=th.from_numpy(d["choices"]).to(gpu))
    assert np.array_equal(neg.cpu().numpy(), d["neg"])
    mapped = convert_subgraph_nids(th.from_numpy(d["ori"]).to(gpu), th.from_numpy(d["subg_nids"]).to(gpu), 200)
    assert np.array_equal(mapped.cpu().numpy(), d["mapped"])
    # drawn on the device: replaced endpoints differ from the originals, relations and the other endpoint stay
    pos = th.from_numpy(d["pos"]).to(gpu)
    gen = th.Generator(device=gpu).manual_seed(1)
    neg2 = negative_sampling(pos, 40, 4, generator=gen)
    rep = pos.repeat(4, 1)
    changed_s, changed_o = neg2[:, 0] != rep[:, 0], neg2[:, 2] != rep[:, 2]
    assert bool((changed_s ^ changed_o).all()) and th.equal(neg2[:, 1], rep[:, 1])
    assert int(neg2[:, [0, 2]].min()) >= 0 and int(neg2[:, [0, 2]].max()) < 40


def test_sample_in_edges_properties(gpu):
    from dualmessagepassing_amd.unc_sampling import sample_in_edges
    g, src, dst = _graph(300, 6000, gpu)
    nodes = th.arange(0, 300, 3, device=gpu)
    gen = th.Generator(device=gpu).manual_seed(5)
    mask = sample_in_edges(g, nodes, 7, gen).cpu().numpy()
    indeg = np.bincount(dst, minlength=300)
    got = np.bincount(dst[mask], minlength=300)
    want = np.where(np.arange(300) % 3 == 0, np.minimum(indeg, 7), 0)
    assert np.array_equal(got, want)
    gen2 = th.Generator(device=gpu).manual_seed(5)
    assert np.array_equal(mask, sample_in_edges(g, nodes, 7, gen2).cpu().numpy())       # same generator state, same sample
    other = sample_in_edges(g, nodes, 7, th.Generator(device=gpu).manual_seed(6)).cpu().numpy()
    assert (other != mask).any()
    # roughly uniform: over many draws every in-edge of a node with 2*width in-edges is picked about half the time
    v = int(np.argmax(indeg == indeg[indeg >= 14].min())) if (indeg >= 14).any() else 0
    picks = np.zeros(len(dst))
    only_v = th.tensor([v], device=gpu)
    for s in range(200):
        picks += sample_in_edges(g, only_v, indeg[v] // 2, th.Generator(device=gpu).manual_seed(100 + s)).cpu().numpy()
    freq = picks[dst == v] / 200
    assert abs(freq.mean() - (indeg[v] // 2) / indeg[v]) < 1e-9 and freq.min() > 0.25 and freq.max() < 0.75


def test_subgraph_sampling_and_batch_construction(gpu):
    from dualmessagepassing_amd.unc_sampling import generate_sampled_graph_and_labels_unsupervised, sample_subgraph_by_neighbors
    g, src, dst = _graph(500, 4000, gpu, seed=2)
    seeds = th.tensor([3, 77, 78, 410], device=gpu)
    gen = th.Generator(device=gpu).manual_seed(9)
    sub, nid = sample_subgraph_by_neighbors(g, seeds, depth=2, width=5, generator=gen)
    nid_h = nid.cpu().numpy()
    assert np.all(np.diff(nid_h) > 0) and set([3, 77, 78, 410]) <= set(nid_h.tolist())
    u, v = sub.all_edges(form="uv", order="eid")
    eid = sub.edata["_ID"].cpu().numpy()
    assert np.array_equal(nid_h[u.cpu().numpy()], src[eid]) and np.array_equal(nid_h[v.cpu().numpy()], dst[eid])
    assert th.equal(sub.edata["type"], g.edata["type"][sub.edata["_ID"]])
    assert int(sub.in_degrees().max()) <= 5
    deg = (sub.in_degrees() + sub.out_degrees()).cpu().numpy()
    assert all(deg[i] > 0 or nid_h[i] in (3, 77, 78, 410) for i in range(len(nid_h)))
    # whole batch: positives + negatives in subgraph ids, labels, edge norms, dropped edges
    rng = np.random.default_rng(4)
    pick = rng.choice(4000, 32, replace=False)
    edges = th.from_numpy(np.stack([src[pick], rng.integers(0, 3, 32), dst[pick]], 1)).to(gpu)
    sub, samples, labels = generate_sampled_graph_and_labels_unsupervised(g, edges, 2, 6, 0.5, 3, th.Generator(device=gpu).manual_seed(3))
    assert samples.shape == (128, 3) and float(labels.sum()) == 32 and labels[:32].min() == 1
    nid = sub.ndata["_ID"]
    assert th.equal(nid[samples[:32, 0]], edges[:, 0]) and th.equal(nid[samples[:32, 2]], edges[:, 2])
    assert int(samples[:, [0, 2]].min()) >= 0 and int(samples[:, [0, 2]].max()) < sub.number_of_nodes()
    assert sub.edata["norm"].shape == (sub.number_of_edges(), 1)


def test_random_walk_and_in_edge_sampling_kernels_match_oracle(gpu):
    """dmp_random_walks / dmp_sample_in_edges (utils.py:279-313: DGL's random_walk + sample_neighbors there) against
    the oracle restatement with the same counter-based generator: every trace entry and every mask bit exact; plus the
    walk / sampling properties themselves."""
    import graph_oracle as GO
    from dualmessagepassing_amd.graph import BatchedGraph
    from dualmessagepassing_amd import unc_sampling as S
    rng = np.random.default_rng(8)
    n, e = 60, 400
    src = rng.integers(0, n - 5, e).astype(np.int64)          # the last 5 nodes have no out-edges: walks end there
    dst = rng.integers(0, n, e).astype(np.int64)
    g = BatchedGraph(th.from_numpy(src).to(gpu), th.from_numpy(dst).to(gpu), n)
    seeds = np.array([0, 3, 3, 59, 17, 58], np.int64)
    for seed in (1, 2 ** 40 + 12345):
        traces, visited = S.random_walks(g, th.from_numpy(seeds).to(gpu), walks=7, depth=4, seed=seed)
        ref = GO.random_walks(src, dst, n, seeds, 7, 4, seed)
        assert np.array_equal(traces.cpu().numpy(), ref)
        assert np.array_equal(visited.cpu().numpy(), np.isin(np.arange(n), ref[ref >= 0]))
        # every step follows an existing edge; a dead end stays -1
        edges = set(zip(src.tolist(), dst.tolist()))
        for row in ref:
            for a, b in zip(row[:-1], row[1:]):
                assert (a, b) in edges or b == -1
                assert a != -1 or b == -1
        wanted = rng.random(n) < 0.6
        for width in (1, 5, 64, 65, 128, 300):   # > 64: the wave-per-node kernel (the reference's default width is 128)
            mask = S.sample_in_edges_device(g, th.from_numpy(wanted).to(gpu), width, seed=seed)
            want = GO.sample_in_edges(dst, n, wanted, width, seed)
            assert np.array_equal(mask.cpu().numpy(), want)
            kept = np.bincount(dst[want], minlength=n)
            indeg = np.bincount(dst, minlength=n)
            assert np.array_equal(kept, np.where(wanted, np.minimum(indeg, width), 0))
    assert not np.array_equal(GO.random_walks(src, dst, n, seeds, 7, 4, 1), GO.random_walks(src, dst, n, seeds, 7, 4, 2))


def test_randomwalk_subgraph_sampler(gpu):
    """sample_subgraph_by_randomwalks (utils.py:279-313): the subgraph's nodes are the seeds plus endpoints of sampled
    edges, ids compacted in ascending parent order, every kept edge is an in-edge of a walked node, frames copied."""
    from dualmessagepassing_amd.graph import BatchedGraph
    from dualmessagepassing_amd import unc_sampling as S
    rng = np.random.default_rng(5)
    n, e = 300, 2500
    src, dst = rng.integers(0, n, e).astype(np.int64), rng.integers(0, n, e).astype(np.int64)
    g = BatchedGraph(th.from_numpy(src).to(gpu), th.from_numpy(dst).to(gpu), n)
    g.edata["type"] = th.arange(e, device=gpu) % 3
    seeds = th.tensor([1, 50, 299], device=gpu)
    sub, nid = S.sample_subgraph_by_randomwalks(g, seeds, depth=2, width=4, seed=77)
    again, nid2 = S.sample_subgraph_by_randomwalks(g, seeds, depth=2, width=4, seed=77)
    assert th.equal(nid, nid2) and th.equal(sub.edata["_ID"], again.edata["_ID"])        # reproducible from the seed
    nid_h, eid_h = nid.cpu().numpy(), sub.edata["_ID"].cpu().numpy()
    assert np.all(np.diff(nid_h) > 0) and set([1, 50, 299]) <= set(nid_h.tolist())
    u, v = sub.all_edges()
    assert np.array_equal(nid_h[u.cpu().numpy()], src[eid_h]) and np.array_equal(nid_h[v.cpu().numpy()], dst[eid_h])
    assert th.equal(sub.edata["type"], g.edata["type"][sub.edata["_ID"]])
    assert np.bincount(dst[eid_h], minlength=n).max() <= 4
    used = np.zeros(n, bool); used[src[eid_h]] = True; used[dst[eid_h]] = True; used[[1, 50, 299]] = True
    assert np.array_equal(np.nonzero(used)[0], nid_h)
    out = S.generate_sampled_graph_and_labels_unsupervised(g, th.tensor([[1, 0, 50], [299, 2, 7]], device=gpu), 2, 4, 0.8, 2,
                                                           generator=th.Generator(device=gpu).manual_seed(3), sampler="randomwalk")
    assert out[1].shape == (6, 3) and out[2].tolist() == [1, 1, 0, 0, 0, 0] and "norm" in out[0].edata


def test_in_edge_sampler_beyond_64_on_hub_nodes(gpu):
    """widths above 64 (the reference's default sample width is 128, main.py:294) used to be refused.  Hubs with
    in-degrees above the width exercise the wave-per-node selection: exact against the oracle, tie rule included."""
    import graph_oracle as GO
    from dualmessagepassing_amd.graph import BatchedGraph
    from dualmessagepassing_amd import unc_sampling as S
    rng = np.random.default_rng(21)
    n = 300
    hubs = np.array([0, 7, 299])
    dst = np.concatenate([rng.integers(0, n, 2000), np.repeat(hubs, [700, 129, 1025])]).astype(np.int64)
    rng.shuffle(dst)
    src = rng.integers(0, n, len(dst)).astype(np.int64)
    g = BatchedGraph(th.from_numpy(src).to(gpu), th.from_numpy(dst).to(gpu), n)
    indeg = np.bincount(dst, minlength=n)
    for seed in (5, 2 ** 33 + 9):
        for width in (65, 128, 129, 700, 1024):
            for wanted in (np.ones(n, bool), rng.random(n) < 0.5):
                mask = S.sample_in_edges_device(g, th.from_numpy(wanted).to(gpu), width, seed=seed).cpu().numpy()
                want = GO.sample_in_edges(dst, n, wanted, width, seed)
                assert np.array_equal(mask, want), (seed, width)
                assert np.array_equal(np.bincount(dst[mask], minlength=n), np.where(wanted, np.minimum(indeg, width), 0))


def _chi2_bound(df, p=1e-6):
    from scipy.stats import chi2
    return float(chi2.ppf(1.0 - p, df))


def test_random_walks_follow_dgl_random_walk_in_distribution(gpu):
    """(f)4, VERDICT r4 item 10.  ``dgl.sampling.random_walk(g, nodes, length)`` as UNC utils.py:291-293 uses it: every step
    takes one of the current node's OUT-EDGES uniformly at random (parallel edges count with their multiplicity), walks are
    independent, a node without out-edges ends the walk and the rest of the trace is -1.  DGL itself is absent and its random
    stream is third-party, so the kernel is held to these semantics IN DISTRIBUTION: chi-square tests of the one-step and
    two-step visit frequencies and of the independence of consecutive walks (fixed seed: deterministic, p = 1e-6 bounds)."""
    from dualmessagepassing_amd.graph import BatchedGraph
    from dualmessagepassing_amd import unc_sampling as S
    # node 0: out-edges to 1 (x3, parallel), 2 (x1), 3 (x2); 1 -> {4, 5}; 2 -> {5}; 3 -> {0, 4, 4, 6}; 4: dead end; 5 -> {0}; 6 -> {6} (self-loop)
    edges = [(0, 1)] * 3 + [(0, 2)] + [(0, 3)] * 2 + [(1, 4), (1, 5), (2, 5), (3, 0), (3, 4), (3, 4), (3, 6), (5, 0), (6, 6)]
    n = 7
    src = np.array([a for a, _ in edges], np.int64)
    dst = np.array([b for _, b in edges], np.int64)
    g = BatchedGraph(th.from_numpy(src).to(gpu), th.from_numpy(dst).to(gpu), n)
    P = np.zeros((n, n))
    for a, b in edges:
        P[a, b] += 1.0
    outdeg = P.sum(1)
    P = np.divide(P, outdeg[:, None], out=np.zeros_like(P), where=outdeg[:, None] > 0)
    W = 60000
    traces, visited = S.random_walks(g, th.tensor([0, 4], device=gpu), walks=W, depth=3, seed=2026)
    t = traces.cpu().numpy().reshape(2, W, 4)
    a = t[0]
    assert (a[:, 0] == 0).all()
    # one step: multinomial over the out-edges of node 0 (multiplicities 3 : 1 : 2)
    o1 = np.bincount(a[:, 1], minlength=n).astype(float)
    e1 = W * P[0]
    assert (o1[e1 == 0] == 0).all()
    x1 = float((((o1 - e1) ** 2) / np.where(e1 > 0, e1, 1.0))[e1 > 0].sum())
    assert x1 <= _chi2_bound(int((e1 > 0).sum()) - 1), (x1, o1, e1)
    # two steps: the Markov chain's two-step law, the dead end 4 reached after one step yields -1 afterwards
    p2 = P[0] @ P
    dead_after_1 = P[0, outdeg == 0].sum()
    o2 = np.bincount(a[:, 2][a[:, 2] >= 0], minlength=n).astype(float)
    e2 = W * p2
    x2 = float((((o2 - e2) ** 2) / np.where(e2 > 0, e2, 1.0))[e2 > 0].sum())
    assert abs((a[:, 2] == -1).mean() - dead_after_1) < 1e-9          # node 0 has no dead-end successor: no walk has ended yet
    assert x2 <= _chi2_bound(int((e2 > 0).sum()) - 1), (x2, o2, e2)
    # three steps: walks that reached node 4 (a dead end) at step 2 read -1 at step 3 and nothing else does
    ended = a[:, 2] == 4
    assert ended.any() and (a[ended, 3] == -1).all() and (a[~ended, 3] >= 0).all()
    p3 = p2 @ P
    o3 = np.bincount(a[:, 3][a[:, 3] >= 0], minlength=n).astype(float)
    e3 = W * p3
    x3 = float((((o3 - e3) ** 2) / np.where(e3 > 0, e3, 1.0))[e3 > 0].sum())
    assert x3 <= _chi2_bound(int((e3 > 0).sum())), (x3, o3, e3)       # (+ the "ended" cell: the counts no longer sum to W)
    assert abs(ended.mean() - p2[4]) <= 5.0 * np.sqrt(p2[4] * (1 - p2[4]) / W)
    # a seed that is a dead end: the whole trace after the seed is -1; visited = exactly the nodes on some trace
    assert (t[1][:, 0] == 4).all() and (t[1][:, 1:] == -1).all()
    assert np.array_equal(visited.cpu().numpy(), np.isin(np.arange(n), t[t >= 0]))
    # independence of the walks of one seed: the first steps of walk i and walk i + 1 (contingency table, 3 x 3 cells)
    f, s = a[:-1, 1], a[1:, 1]
    tab = np.zeros((n, n))
    np.add.at(tab, (f, s), 1.0)
    keep = P[0] > 0
    tab = tab[np.ix_(keep, keep)]
    exp = np.outer(tab.sum(1), tab.sum(0)) / tab.sum()
    xi = float(((tab - exp) ** 2 / exp).sum())
    assert xi <= _chi2_bound((keep.sum() - 1) ** 2), (xi, tab)
    # another seed of the generator: another sample of the same law
    t2 = S.random_walks(g, th.tensor([0], device=gpu), walks=W, depth=1, seed=7)[0].cpu().numpy()
    assert not np.array_equal(t2[:, 1], a[:, 1])
    o = np.bincount(t2[:, 1], minlength=n).astype(float)
    assert float((((o - e1) ** 2) / np.where(e1 > 0, e1, 1.0))[e1 > 0].sum()) <= _chi2_bound(int((e1 > 0).sum()) - 1)


def test_in_edge_sampler_follows_dgl_sample_neighbors_in_distribution(gpu):
    """``dgl.sampling.sample_neighbors(g, nodes, fanout, edge_dir="in")`` (UNC utils.py:294-296,326-335, replace=False): for every
    requested node ``fanout`` of its in-edges uniformly WITHOUT replacement -- every in-edge with probability fanout / indeg,
    every pair with fanout (fanout - 1) / (indeg (indeg - 1)), parallel edges as distinct edges -- all of them when there are at
    most ``fanout``, none for nodes that were not requested.  Held in distribution over 4000 launches with different seeds."""
    from dualmessagepassing_amd.graph import BatchedGraph
    from dualmessagepassing_amd import unc_sampling as S
    rng = np.random.default_rng(4)
    # node 0: 12 in-edges (two of them parallel copies of the same source); node 1: 3 in-edges; node 2: 9 in-edges, not requested
    src = np.concatenate([np.array([5, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]), np.array([5, 6, 7]), rng.integers(3, 16, 9)]).astype(np.int64)
    dst = np.concatenate([np.zeros(12), np.ones(3), np.full(9, 2)]).astype(np.int64)
    perm = rng.permutation(src.size)                           # edge ids in no particular order
    src, dst = src[perm], dst[perm]
    n, width, R = 16, 5, 4000
    g = BatchedGraph(th.from_numpy(src).to(gpu), th.from_numpy(dst).to(gpu), n)
    wanted = th.tensor([0, 1], device=gpu)
    masks = th.stack([S.sample_in_edges_device(g, wanted, width, seed=1000 + r) for r in range(R)]).cpu().numpy()
    e0, e1, e2 = np.nonzero(dst == 0)[0], np.nonzero(dst == 1)[0], np.nonzero(dst == 2)[0]
    assert (masks[:, e0].sum(1) == width).all() and masks[:, e1].all() and not masks[:, e2].any()
    d = e0.size
    p = width / d
    inc = masks[:, e0].mean(0)
    z = np.abs(inc - p) / np.sqrt(p * (1 - p) / R)
    assert z.max() <= 4.8, (z, inc)                             # 12 edges, two-sided 1e-6 each
    pairs = masks[:, e0].astype(float).T @ masks[:, e0].astype(float) / R
    pp = width * (width - 1) / (d * (d - 1))
    iu = np.triu_indices(d, 1)
    zp = np.abs(pairs[iu] - pp) / np.sqrt(pp * (1 - pp) / R)
    assert zp.max() <= 5.3, (zp.max(), pairs[iu].min(), pairs[iu].max(), pp)    # 66 pairs

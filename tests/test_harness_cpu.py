"""Host-side pieces of the harness: the exact subgraph-isomorphism counter against networkx's VF2
monomorphism enumeration, and the synthetic dataset's layout (reversed edges as train.py:299-327)."""
import networkx as nx
import numpy as np

from dualmessagepassing_amd.harness import SyntheticPairs, count_subisomorphisms


def _nx_count(ps, pd, pvl, pel, gs, gd, gvl, gel):
    P, G = nx.DiGraph(), nx.DiGraph()
    for i, l in enumerate(pvl):
        P.add_node(i, label=int(l))
    for i, l in enumerate(gvl):
        G.add_node(i, label=int(l))
    for u, v, l in zip(ps, pd, pel):
        P.add_edge(int(u), int(v), label=int(l))
    for u, v, l in zip(gs, gd, gel):
        G.add_edge(int(u), int(v), label=int(l))
    gm = nx.algorithms.isomorphism.DiGraphMatcher(
        G, P, node_match=lambda a, b: a["label"] == b["label"], edge_match=lambda a, b: a["label"] == b["label"])
    return sum(1 for _ in gm.subgraph_monomorphisms_iter())


def test_counter_matches_networkx_vf2():
    rng = np.random.default_rng(0)
    seen_positive = 0
    for _ in range(25):
        pn, gn = int(rng.integers(2, 5)), int(rng.integers(5, 9))
        pe, ge = int(rng.integers(1, pn * (pn - 1) + 1)), int(rng.integers(4, gn * (gn - 1) // 2))
        from dualmessagepassing_amd.harness import _er_edges
        ps, pd = _er_edges(pn, pe, rng)
        gs, gd = _er_edges(gn, ge, rng)
        pvl, gvl = rng.integers(0, 2, pn), rng.integers(0, 2, gn)
        pel, gel = rng.integers(0, 2, pe), rng.integers(0, 2, ge)
        c = count_subisomorphisms(ps, pd, pvl, pel, gs, gd, gvl, gel)
        assert c == _nx_count(ps, pd, pvl, pel, gs, gd, gvl, gel)
        seen_positive += c > 0
    assert seen_positive >= 5


def test_synthetic_pairs_layout():
    ds = SyntheticPairs(6, 3, 3, 8, 14, 2, 2, seed=1)
    assert len(ds) == 6
    s = ds.samples[0]["graph"]
    e = 14
    assert len(s["src"]) == 2 * e and np.array_equal(s["src"][e:], s["dst"][:e]) and np.array_equal(s["dst"][e:], s["src"][:e])
    assert np.array_equal(s["rev"], np.r_[np.zeros(e, bool), np.ones(e, bool)])
    assert np.array_equal(s["eid"], np.r_[np.arange(e), e + np.arange(e)]) and np.array_equal(s["elabel"][e:], s["elabel"][:e] + 2)
    cfg = ds.model_config(hid_dim=16, layers=2)
    assert cfg["max_nge"] == 2 * e and cfg["max_ngel"] == 4 and cfg["rep_net"] == "DMPNN"


def test_step_schedules_match_reference_tables():
    """``harness.scheduled_value`` / ``harness.lr_factor`` against tables emitted by the reference's own ``anneal_fn`` /
    ``cyclical_fn`` / scheduler ``lr_lambda``s (tests/golden/schedules.npz, oracle/make_golden.py::gen_schedules)."""
    import numpy as np
    from conftest import golden_files, load_golden
    from dualmessagepassing_amd.harness import lr_factor, scheduled_value
    d = load_golden(golden_files("schedules")[0])
    for i, spec in enumerate(d["specs"].tolist()):
        shape, total, cyc, a, b = spec.split("|")
        total, cyc = int(total), int(cyc)
        for mode in ("anneal", "cyclical"):
            want = d["%s.%d" % (mode, i)]
            got = [scheduled_value("%s_%s$%s$%s" % (mode, shape, a, b), s, total, cyc) for s in range(len(want))]
            assert np.allclose(got, want, rtol=0, atol=1e-12), (mode, spec)
    for i, spec in enumerate(d["lr_specs"].tolist()):
        name, warm, total, cyc, floor = spec.split("|")
        want = d["lr.%d" % i]
        got = [lr_factor(name, s, int(warm), int(total), float(cyc), float(floor)) for s in range(len(want))]
        assert np.allclose(got, want, rtol=0, atol=1e-12), spec
    assert scheduled_value(0.25, 3, 10) == 0.25


def test_native_enumeration_equals_python_twin_and_batch_counts():
    """csrc/dmp_subiso.cpp against the plain-Python search: the same matches in the same (lexicographic) order, on
    multigraphs with parallel edges of different labels and self loops; the threaded batch counter; the match limit."""
    import numpy as np
    from dualmessagepassing_amd.harness import (count_subisomorphisms_batch, enumerate_subisomorphisms, enumerate_subisomorphisms_py)
    rng = np.random.default_rng(4)
    pairs, total = [], 0
    for i in range(40):
        pn, gn = int(rng.integers(1, 5)), int(rng.integers(3, 10))
        pe, ge = int(rng.integers(0, 7)), int(rng.integers(2, 40))
        ps, pd = rng.integers(0, pn, pe), rng.integers(0, pn, pe)          # loops and parallel edges allowed
        gs, gd = rng.integers(0, gn, ge), rng.integers(0, gn, ge)
        pair = (ps, pd, rng.integers(0, 2, pn), rng.integers(0, 2, pe), gs, gd, rng.integers(0, 2, gn), rng.integers(0, 2, ge))
        a, b = enumerate_subisomorphisms(*pair), enumerate_subisomorphisms_py(*pair)
        assert a.shape == b.shape and np.array_equal(a, b), i
        if len(a) > 3:
            assert np.array_equal(enumerate_subisomorphisms(*pair, limit=3), b[:3])
        pairs.append(pair)
        total += len(a)
    assert total > 100
    counts = count_subisomorphisms_batch(pairs, threads=4)
    assert counts.tolist() == [len(enumerate_subisomorphisms_py(*p)) for p in pairs]
    big = (np.array([0, 1, 2]), np.array([1, 2, 0]), np.zeros(3, int), np.zeros(3, int),
           np.repeat(np.arange(12), 11), np.concatenate([np.delete(np.arange(12), i) for i in range(12)]), np.zeros(12, int), np.zeros(132, int))
    assert len(enumerate_subisomorphisms(*big)) == 12 * 11 * 10       # beyond the first buffer: directed triangles in K12


def test_validate_samples_names_the_bad_sample():
    import pytest
    """harness.validate_samples: the host-side dataset check harness.fit runs once (the device index builds only
    flag an out-of-range endpoint in a status word)."""
    import torch as th
    from dualmessagepassing_amd.graph import BatchedGraph
    from dualmessagepassing_amd.harness import validate_samples

    def g(src, dst, n):
        return BatchedGraph(th.tensor(src), th.tensor(dst), n)

    good = {"id": "a", "pattern": g([0, 1], [1, 2], 3), "graph": g([0, 3], [3, 1], 4), "counts": 0}
    assert validate_samples([good, good]) == 2
    bad = dict(good, id="b", graph=g([0, 4], [3, 1], 4))
    with pytest.raises(ValueError, match=r"sample 1 \(b\): graph"):
        validate_samples([good, bad])
    stored = {"id": "d", "pattern": {"num_nodes": 2, "src": np.array([0, 1]), "dst": np.array([1, 0]), "rev": np.array([False, True])},
              "graph": {"num_nodes": 3, "src": np.array([0, 3]), "dst": np.array([1, 2])}, "counts": 1}     # PairDataset's stored form
    with pytest.raises(ValueError, match=r"sample 0 \(d\): graph"):
        validate_samples([stored])
    flagged = dict(good, id="c")
    flagged["pattern"] = g([0, 1], [1, 2], 3)
    flagged["pattern"].edata["is_reversed"] = th.zeros(3, dtype=th.bool)
    with pytest.raises(ValueError, match="is_reversed"):
        validate_samples([flagged])


def test_link_file_round_trip(tmp_path):
    from dualmessagepassing_amd.unc_harness import load_links, save_embeddings
    trip = np.array([[0, 0, 1], [2, 1, 0], [1, 0, 2]], np.int64)
    p = tmp_path / "link.dat"
    p.write_text("3 2\n" + "".join("%d %d %d\n" % tuple(r) for r in trip))
    got, n, r = load_links(str(p))
    assert (n, r) == (3, 2) and np.array_equal(got, trip)
    out = tmp_path / "emb.dat"
    save_embeddings(str(out), np.array([[0.5, 1.0], [2.0, -1.0]], np.float32), index=[2, 0], header="args")
    lines = out.read_text().splitlines()
    assert lines[0] == "args" and lines[1].split("\t")[0] == "2" and lines[1].split("\t")[1].split() == ["0.5", "1.0"]
    assert lines[2].startswith("0\t2.0 -1.0")


def test_gate_compact_is_refused_under_data_parallelism():
    """the gate-compaction overflow veto is rank-local -- with a multi-rank gradient sync the overflowing rank's
    wrong gradients would still be all-reduced into every replica.  ``fit`` refuses the combination before touching data."""
    import pytest
    from dualmessagepassing_amd import harness

    class TwoRanks:
        world = 2

    with pytest.raises(ValueError, match="single-rank"):
        harness.fit(None, None, [], [], 1, 4, "cpu", sync=TwoRanks(), gate_compact=True)


def test_veto_word_needs_one_flat_parameter():
    """the veto word is consumed by the launch of ONE parameter tensor; with several, the first launch would
    clear it and the others would apply the flagged gradients.  ``set_veto`` refuses such an optimizer."""
    import pytest
    import torch
    from dualmessagepassing_amd.dp import FlatAdamW
    a, b = torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(8))
    word = torch.zeros(4, dtype=torch.int32)
    with pytest.raises(ValueError, match="exactly one"):
        FlatAdamW([a, b]).set_veto(word)
    opt = FlatAdamW([a]).set_veto(word)
    assert opt.veto is not None and FlatAdamW([a, b]).set_veto(None).veto is None

"""GPU parity (bit-exact) of the integer graph transforms: line graph, reversed edges,
collate, degree/eigenvalue bounds -- against the reference's golden vectors and the C oracle."""
import numpy as np
import pytest
import torch as th

import graph_oracle as GO
from conftest import golden_files, load_golden
from util_graphs import er_edges

pytestmark = pytest.mark.gpu


def _t(a):
    return th.from_numpy(np.ascontiguousarray(a))


def _frames(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def _bg(src, dst, n, ndata, edata, dev, bnn=None, bne=None):
    from dualmessagepassing_amd.graph import BatchedGraph
    return BatchedGraph(_t(src).to(dev), _t(dst).to(dev), n, None if bnn is None else _t(bnn).to(dev),
                        None if bne is None else _t(bne).to(dev),
                        {k: _t(v).to(dev) for k, v in ndata.items()}, {k: _t(v).to(dev) for k, v in edata.items()})


@pytest.mark.parametrize("path", golden_files("linegraph_"))
def test_line_graph_matches_reference_golden(path, gpu):
    from dualmessagepassing_amd.linegraph import convert_to_dual_graph
    d = load_golden(path)
    ndata, edata = _frames(d, "ndata."), _frames(d, "edata.")
    g = _bg(d["src"].astype(np.int64), d["dst"].astype(np.int64), int(d["num_nodes"]), ndata, edata, gpu)
    dg = convert_to_dual_graph(g)
    assert dg.number_of_nodes() == int(d["dual_num_nodes"])
    u, v = dg.all_edges()
    assert np.array_equal(u.cpu().numpy(), d["dual_src"]) and np.array_equal(v.cpu().numpy(), d["dual_dst"])
    ref_nd, ref_ed = _frames(d, "dual_ndata."), _frames(d, "dual_edata.")
    assert sorted(dg.ndata) == sorted(ref_nd) and sorted(dg.edata) == sorted(ref_ed)
    for k in ref_nd:
        assert np.array_equal(dg.ndata[k].cpu().numpy(), ref_nd[k]), k
    for k in ref_ed:
        assert np.array_equal(dg.edata[k].cpu().numpy().reshape(ref_ed[k].shape), ref_ed[k]), k


@pytest.mark.parametrize("mode", ["plain", "idrev", "dupid"])
def test_line_graph_batched_equals_per_graph_oracle(mode, gpu):
    """A whole batch converted at once == the oracle applied graph by graph (ragged sizes,
    one empty graph, duplicate ids, id holes)."""
    from dualmessagepassing_amd.collate import collate_device
    from dualmessagepassing_amd.linegraph import convert_to_dual_graph
    rng = np.random.default_rng(8)
    sizes = [(5, 7), (1, 0), (9, 30), (3, 6), (12, 40)]
    per, ls, ld, nd_all, ed_all = [], [], [], {}, {}
    for n, m in sizes:
        u, v = er_edges(n, m, rng) if m else (np.zeros(0, np.int64), np.zeros(0, np.int64))
        ndata, edata = {}, {}
        if mode != "plain":
            nl = rng.integers(0, 3, size=n)
            el = rng.integers(0, 3, size=len(u))
            ndata = {"id": np.arange(n), "label": nl}
            if mode == "idrev":
                e = len(u)
                u, v = np.concatenate([u, v]), np.concatenate([v, u])
                edata = {"id": np.concatenate([np.arange(e), e + 2 + np.arange(e)]), "label": np.concatenate([el, el + 3]),
                         "is_reversed": np.concatenate([np.zeros(e, bool), np.ones(e, bool)])}
            else:
                edata = {"id": rng.integers(0, max(1, m // 2 + 1), size=len(u)), "label": el}
        per.append((u, v, n, ndata, edata))
        ls.append(u); ld.append(v)
        for k, x in ndata.items():
            nd_all.setdefault(k, []).append(x)
        for k, x in edata.items():
            ed_all.setdefault(k, []).append(x)
    nn = np.array([p[2] for p in per], np.int64)
    ne = np.array([len(p[0]) for p in per], np.int64)
    g = collate_device(_t(np.concatenate(ls)).to(gpu), _t(np.concatenate(ld)).to(gpu), _t(nn).to(gpu), _t(ne).to(gpu),
                       int(nn.sum()), int(ne.sum()),
                       {k: _t(np.concatenate(x)).to(gpu) for k, x in nd_all.items()},
                       {k: _t(np.concatenate(x)).to(gpu) for k, x in ed_all.items()})
    dg = convert_to_dual_graph(g)
    # expected: oracle per graph, then dgl.batch semantics
    exp_s, exp_t, exp_nd, exp_ed, off, dnn, dne = [], [], {}, {}, 0, [], []
    for u, v, n, ndata, edata in per:
        s, t, dn, dnd, ded = GO.convert_to_dual_graph(u, v, n, ndata, edata)
        exp_s.append(s + off); exp_t.append(t + off)
        off += dn
        dnn.append(dn); dne.append(len(s))
        for k, x in dnd.items():
            exp_nd.setdefault(k, []).append(x)
        for k, x in ded.items():
            exp_ed.setdefault(k, []).append(x)
    u, v = dg.all_edges()
    assert np.array_equal(u.cpu().numpy(), np.concatenate(exp_s)) and np.array_equal(v.cpu().numpy(), np.concatenate(exp_t))
    assert dg.batch_num_nodes().cpu().tolist() == dnn and dg.batch_num_edges().cpu().tolist() == dne
    for k, x in exp_nd.items():
        assert np.array_equal(dg.ndata[k].cpu().numpy(), np.concatenate(x)), k
    for k, x in exp_ed.items():
        assert np.array_equal(dg.edata[k].cpu().numpy(), np.concatenate(x)), k
    # |E_L| = sum_v indeg(v) * outdeg(v) in the plain branch
    if mode == "plain":
        for (uu, vv, n, _, _), m in zip(per, dne):
            assert m == int((np.bincount(vv, minlength=n) * np.bincount(uu, minlength=n)).sum())


@pytest.mark.parametrize("path", golden_files("addrev_"))
def test_add_reversed_edges_matches_reference_golden(path, gpu):
    from dualmessagepassing_amd.preprocess import add_reversed_edges
    d = load_golden(path)
    n = int(d["num_nodes"])
    g = _bg(d["src"], d["dst"], n, {"id": np.arange(n)}, {"id": d["eid"], "label": d["elabel"]}, gpu)
    g.in_degrees(); g.out_degrees()  # cached before, updated by the pass (dataset.py:1289-1293)
    r = add_reversed_edges(g, int(d["max_ne"]), int(d["max_nel"]))
    u, v = r.all_edges()
    for got, key in ((u, "o_src"), (v, "o_dst"), (r.edata["id"], "o_eid"), (r.edata["label"], "o_elabel"),
                     (r.edata["is_reversed"], "o_rev"), (r.ndata["in_deg"], "o_in_deg"), (r.ndata["out_deg"], "o_out_deg")):
        assert np.array_equal(got.cpu().numpy(), d[key]), key
    assert add_reversed_edges(r, 1, 1) is r  # train.py:302: no-op when REVFLAG is present


def test_collate_addrev_degrees_batched(gpu):
    """collate -> add_reversed_edges on a ragged batch == oracle per graph, in dgl.batch order."""
    from dualmessagepassing_amd.collate import collate_device
    from dualmessagepassing_amd.preprocess import add_reversed_edges, compute_largest_eigenvalues
    rng = np.random.default_rng(21)
    sizes = [(6, 9), (2, 1), (4, 0), (10, 33)]
    graphs = [(er_edges(n, m, rng) if m else (np.zeros(0, np.int64), np.zeros(0, np.int64))) + (n,) for n, m in sizes]
    ls = np.concatenate([g[0] for g in graphs]); ld = np.concatenate([g[1] for g in graphs])
    nn = np.array([g[2] for g in graphs], np.int64); ne = np.array([len(g[0]) for g in graphs], np.int64)
    eid = np.concatenate([np.arange(len(g[0])) for g in graphs]); el = rng.integers(0, 4, size=len(ls))
    src, dst, no, eo, eg, ng = GO.collate(ls, ld, nn, ne)
    g = collate_device(_t(ls).to(gpu), _t(ld).to(gpu), _t(nn).to(gpu), _t(ne).to(gpu), int(nn.sum()), int(ne.sum()),
                       None, {"id": _t(eid).to(gpu), "label": _t(el).to(gpu)})
    u, v = g.all_edges()
    assert np.array_equal(u.cpu().numpy(), src) and np.array_equal(v.cpu().numpy(), dst)
    assert np.array_equal(g.node_offsets.cpu().numpy(), no) and np.array_equal(g.edge_offsets.cpu().numpy(), eo)
    assert np.array_equal(g.edge_graph.cpu().numpy(), eg) and np.array_equal(g.node_graph.cpu().numpy(), ng)
    assert g.batch_size == 4 and g.batch_num_nodes().cpu().tolist() == nn.tolist()
    r = add_reversed_edges(g, 50, 10)
    exp = [GO.add_reversed_edges(src[eo[i]:eo[i + 1]], dst[eo[i]:eo[i + 1]], eid[eo[i]:eo[i + 1]], el[eo[i]:eo[i + 1]], 50, 10)
           for i in range(4)]
    u, v = r.all_edges()
    for got, j in ((u, 0), (v, 1), (r.edata["id"], 2), (r.edata["label"], 3), (r.edata["is_reversed"], 4)):
        assert np.array_equal(got.cpu().numpy(), np.concatenate([x[j] for x in exp]))
    assert r.batch_num_edges().cpu().tolist() == (2 * ne).tolist()
    # per-graph eigenvalue bounds (utils/graph.py:40-71) on the doubled graphs
    nd, ed = compute_largest_eigenvalues(r)
    uu, vv = u.cpu().numpy(), v.cpu().numpy()
    for i in range(4):
        if ne[i] == 0:
            continue
        sl = slice(2 * eo[i], 2 * eo[i + 1])
        a, b = GO.eigen_bounds(uu[sl] - no[i], vv[sl] - no[i], int(nn[i]))
        assert float(nd[i]) == a and float(ed[i]) == b


@pytest.mark.gpu
def test_concat_pairs_bit_exact(gpu):
    """dmp_concat_pairs (the structure arrays of a union of two batches in one launch) against torch.cat: int64 with
    and without the node shift, bool / uint8 flags, int32, fp32 with a constant first block, empty halves."""
    import torch as th
    from dualmessagepassing_amd.collate import concat_pairs
    g = th.Generator().manual_seed(11)
    ri = lambda n, dt=th.int64: th.randint(0, 1000, (n,), generator=g).to(dt)
    pairs_cpu = [(ri(24576), ri(524288), 8192), (ri(5), ri(0), 0), (ri(0), ri(7), 3), (ri(1024), ri(1024), 0),
                 (ri(300) > 500, ri(70001) > 500, 0), (ri(9, th.uint8), ri(11, th.uint8), 0),
                 (ri(13, th.int32), ri(17, th.int32), 0), ((8192, 1.0), th.rand(65536, generator=g), 0),
                 (th.rand(3, generator=g), th.rand(5, generator=g), 0)]
    to = lambda t: t.cuda() if th.is_tensor(t) else t
    got = concat_pairs([(to(a), to(b), add) for a, b, add in pairs_cpu])
    want = concat_pairs(pairs_cpu)                               # CPU tensors: the torch.cat path
    for w, o in zip(want, got):
        assert o.is_cuda and o.dtype == w.dtype and th.equal(o.cpu(), w)


@pytest.mark.gpu
@pytest.mark.parametrize("N,E", [(1, 0), (5, 40), (2049, 30000), (73728, 200001)])
def test_csr_build_pair_equals_two_builds(N, E, gpu):
    """dmp_csr_build_pair against two dmp_csr_build calls: identical rowptr / entries / 32-bit keys / degrees / status,
    including out-of-range endpoints (flagged, skipped)."""
    import torch as th
    from dualmessagepassing_amd import _lib
    lib = _lib.load()
    g = th.Generator().manual_seed(N + E)
    src = th.randint(0, N, (E,), generator=g)
    dst = th.randint(0, N, (E,), generator=g)
    if E > 10:
        dst[3], src[7] = N + 5, -1                               # out of range: status bit, no entry
    flag = (th.rand(E, generator=g) < 0.5).to(th.uint8)
    src, dst, flag = src.cuda(), dst.cuda(), flag.cuda()
    i32 = dict(dtype=th.int32, device="cuda")
    mk = lambda: (th.full((N + 1,), -1, **i32), th.full((E,), -1, **i32), th.full((E,), -1, **i32),
                  th.full((N,), -1, dtype=th.int64, device="cuda"))
    P = _lib.ptr
    single, st1 = [mk(), mk()], th.full((2,), 7, **i32)
    ws = th.empty(lib.dmp_csr_workspace_words(N, E), **i32)
    for k, (key, o) in enumerate(zip((dst, src), single)):
        _lib.check(lib.dmp_csr_build(P(key), P(flag), E, N, P(o[0]), P(o[1]), P(o[2]), P(o[3]), P(st1[k:]), P(ws),
                                     _lib.stream_ptr()), "dmp_csr_build")
    pair, st2 = [mk(), mk()], th.full((2,), 7, **i32)
    ws2 = th.empty(lib.dmp_csr_pair_workspace_words(N), **i32)
    _lib.check(lib.dmp_csr_build_pair(P(dst), P(src), P(flag), E, N, *[P(t) for t in pair[0]], *[P(t) for t in pair[1]],
                                      P(st2), P(ws2), _lib.stream_ptr()), "dmp_csr_build_pair")
    assert th.equal(st1, st2)
    for a, b in zip(single, pair):
        valid = int(a[0][N])                                      # entries past the valid ones are never written
        assert th.equal(a[0], b[0]) and th.equal(a[1][:valid], b[1][:valid]) and th.equal(a[2], b[2]) and th.equal(a[3], b[3])


def test_dual_subisomorphisms_match_reference_and_oracle(gpu):
    """``linegraph.dual_subisomorphisms`` (dmp_dual_subisomorphisms) against the reference's own
    ``get_dual_subisomorphisms`` + ``g_eid`` mapping (tests/golden/dual_subiso.npz; utils/graph.py:277-316,
    train.py:417-446) -- every case alone and all of them as one batch -- and against the oracle restatement on patterns
    with repeated keys (parallel edges in one run, a key split into two runs: the later run replaces the earlier)."""
    import graph_oracle as GO
    from dualmessagepassing_amd.collate import collate_device
    from dualmessagepassing_amd.linegraph import dual_subisomorphisms
    d = load_golden(golden_files("dual_subiso")[0])
    cases = []
    for c in range(int(d["num_cases"])):
        k = "%d." % c
        cases.append(dict(p_u=d[k + "p_u"], p_v=d[k + "p_v"], p_el=d[k + "p_el"], g_u=d[k + "g_u"], g_v=d[k + "g_v"], g_el=d[k + "g_el"],
                          sub=d[k + "sub"], want=d[k + "dual_eids"], pn=d[k + "sub"].shape[1], gn=int(max(d[k + "g_u"].max(), d[k + "g_v"].max())) + 1))
    # the oracle restatement agrees on every case (the last three have repeated pattern keys on multigraphs)
    for c in cases:
        order = np.lexsort((np.arange(len(c["g_u"])), c["g_v"], c["g_u"]))
        idx = GO.dual_subisomorphisms(c["p_u"], c["p_v"], c["p_el"], c["g_u"][order], c["g_v"][order], c["g_el"][order], c["sub"])
        assert np.array_equal(order[idx], c["want"])
    assert len(cases) >= 7

    def run(batch):
        cat = lambda k: _t(np.concatenate([np.asarray(c[k], np.int64) for c in batch])).to(gpu)
        sizes = lambda k: _t(np.array([len(c[k]) for c in batch], np.int64)).to(gpu)
        pn_, gn_ = _t(np.array([c["pn"] for c in batch], np.int64)).to(gpu), _t(np.array([c["gn"] for c in batch], np.int64)).to(gpu)
        pattern = collate_device(cat("p_u"), cat("p_v"), pn_, sizes("p_u"), sum(c["pn"] for c in batch), sum(len(c["p_u"]) for c in batch),
                                 edata={"label": cat("p_el")})
        graph = collate_device(cat("g_u"), cat("g_v"), gn_, sizes("g_u"), sum(c["gn"] for c in batch), sum(len(c["g_u"]) for c in batch),
                               edata={"label": cat("g_el")})
        flat = _t(np.concatenate([c["sub"].reshape(-1) for c in batch])).to(gpu)
        sp = _t(np.concatenate([[0], np.cumsum([c["sub"].size for c in batch])]).astype(np.int64)).to(gpu)
        out, optr = dual_subisomorphisms(pattern, graph, flat, sp, validate=True)
        want = np.concatenate([np.asarray(c["want"], np.int64).reshape(-1) for c in batch])
        assert out.dtype == th.int64 and np.array_equal(out.cpu().numpy(), want)
        assert np.array_equal(optr.cpu().numpy(), np.concatenate([[0], np.cumsum([np.asarray(c["want"]).size for c in batch])]))

    for c in cases:
        run([c])
    run(cases)
    run(cases[::-1])


@pytest.mark.gpu
@pytest.mark.parametrize("batch,n,m", [(1024, 64, 256), (64, 512, 4096)])
def test_integer_transforms_bit_exact_at_baseline_sizes(batch, n, m, gpu):
    """BASELINE configs[1] target batch (1024 x (64, 256)) and 64 graphs of configs[3]'s target shape (512, 4096): device
    collate, add_reversed_edges and the line-graph transform of the whole batch, bit for bit against the sequential C
    oracle applied graph by graph (dgl.batch order)."""
    from dualmessagepassing_amd.collate import collate_device
    from dualmessagepassing_amd.linegraph import convert_to_dual_graph
    from dualmessagepassing_amd.preprocess import add_reversed_edges
    rng = np.random.default_rng(batch + n)
    per = [er_edges(n, m, rng) for _ in range(batch)]
    ls, ld = np.concatenate([p[0] for p in per]), np.concatenate([p[1] for p in per])
    nn, ne = np.full(batch, n, np.int64), np.full(batch, m, np.int64)
    eid = np.tile(np.arange(m), batch)
    el = rng.integers(0, 16, size=batch * m)
    nl = rng.integers(0, 16, size=batch * n)
    g = collate_device(_t(ls).to(gpu), _t(ld).to(gpu), _t(nn).to(gpu), _t(ne).to(gpu), batch * n, batch * m,
                       {"id": _t(np.tile(np.arange(n), batch)).to(gpu), "label": _t(nl).to(gpu)},
                       {"id": _t(eid).to(gpu), "label": _t(el).to(gpu)})
    src, dst, no, eo, eg, ng = GO.collate(ls, ld, nn, ne)
    u, v = g.all_edges()
    assert np.array_equal(u.cpu().numpy(), src) and np.array_equal(v.cpu().numpy(), dst)
    assert np.array_equal(g.edge_graph.cpu().numpy(), eg) and np.array_equal(g.node_graph.cpu().numpy(), ng)
    r = add_reversed_edges(g, m, 16)
    ru, rv = (t.cpu().numpy() for t in r.all_edges())
    rid, rl, rr = r.edata["id"].cpu().numpy(), r.edata["label"].cpu().numpy(), r.edata["is_reversed"].cpu().numpy()
    dg = convert_to_dual_graph(r)
    du, dv = (t.cpu().numpy() for t in dg.all_edges())
    dnn, dne = dg.batch_num_nodes().cpu().numpy(), dg.batch_num_edges().cpu().numpy()
    d_nd = {k: t.cpu().numpy() for k, t in dg.ndata.items()}
    d_ed = {k: t.cpu().numpy() for k, t in dg.edata.items()}
    n_off = e_off = dn_off = de_off = 0
    for i in range(batch):
        a = GO.add_reversed_edges(per[i][0], per[i][1], np.arange(m), el[i * m:(i + 1) * m], m, 16)
        sl = slice(e_off, e_off + 2 * m)
        assert np.array_equal(ru[sl] - n_off, a[0]) and np.array_equal(rv[sl] - n_off, a[1]), i
        assert np.array_equal(rid[sl], a[2]) and np.array_equal(rl[sl], a[3]) and np.array_equal(rr[sl], a[4]), i
        s, t, dn, dnd, ded = GO.convert_to_dual_graph(a[0], a[1], n, {"id": np.arange(n), "label": nl[i * n:(i + 1) * n]},
                                                      {"id": a[2], "label": a[3], "is_reversed": a[4]})
        assert int(dnn[i]) == dn and int(dne[i]) == len(s), i
        assert np.array_equal(du[de_off:de_off + len(s)] - dn_off, s) and np.array_equal(dv[de_off:de_off + len(s)] - dn_off, t), i
        for k, x in dnd.items():
            assert np.array_equal(d_nd[k][dn_off:dn_off + dn], x), (i, k)
        for k, x in ded.items():
            assert np.array_equal(d_ed[k][de_off:de_off + len(s)], x), (i, k)
        n_off += n; e_off += 2 * m; dn_off += dn; de_off += len(s)
    assert dn_off == dg.number_of_nodes() and de_off == dg.number_of_edges()

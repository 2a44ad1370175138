"""The oracle's restatements of the preprocessing / graph-construction functions against fixtures emitted by the
REFERENCE's own functions (oracle/make_golden.py: gen_preprocess, gen_unc, gen_dual_subiso): A14
``compute_largest_eigenvalues`` / ``calculate_eigenvalues`` (utils/graph.py:40-71, train.py:368-380), A17 UNC
``compute_edgenorm`` / ``build_graph_from_triplets`` (utils.py:437-491), ``get_dual_subisomorphisms``
(utils/graph.py:277-316).  No GPU."""
import numpy as np
import torch as th

import dmp_oracle as O
import graph_oracle as GO
from conftest import golden_files, load_golden


def test_eigenvalue_bounds_oracle_matches_reference():
    d = load_golden(golden_files("preprocess_eigen")[0])
    dropped = set(d["dropped"].tolist())
    checked = 0
    for i in range(int(d["num_samples"])):
        for t in ("p", "g"):
            src, dst, n = d["%d.%s.src" % (i, t)], d["%d.%s.dst" % (i, t)], len(d["%d.%s.vlabel" % (i, t)])
            if "plain.%d.%s.node_eigenv" % (i, t) in d:          # before add_rev: degrees of the plain graph
                a, b = GO.eigen_bounds(src, dst, n)
                assert a == float(d["plain.%d.%s.node_eigenv" % (i, t)]) and b == float(d["plain.%d.%s.edge_eigenv" % (i, t)])
                checked += 1
            if i in dropped:
                continue
            o_src, o_dst = d["%d.%s.o_src" % (i, t)], d["%d.%s.o_dst" % (i, t)]
            # add_reversed_edges keeps the cached degrees in step with the structure (dataset.py:1289-1293)
            assert np.array_equal(d["%d.%s.in_deg" % (i, t)], np.bincount(o_dst, minlength=n))
            assert np.array_equal(d["%d.%s.out_deg" % (i, t)], np.bincount(o_src, minlength=n))
            a, b = GO.eigen_bounds(o_src, o_dst, n)
            ne, ee = d["%d.%s.node_eigenv" % (i, t)], d["%d.%s.edge_eigenv" % (i, t)]
            assert ne.shape == (n, 1) and ee.shape == (len(o_src), 1)      # train.py:376-379: repeated over nodes / edges
            assert np.all(ne == max(a, 1.0)) and np.all(ee == max(b, 1.0))
            checked += 1
    assert checked >= 16
    # dataset-level bounds (train.py:1174-1186): max over the PATTERN graphs, floor 4.0
    mn = me = 4.0
    for i in range(int(d["num_samples"])):
        if i in dropped:
            continue
        a, b = GO.eigen_bounds(d["%d.p.o_src" % i], d["%d.p.o_dst" % i], len(d["%d.p.vlabel" % i]))
        mn, me = max(mn, max(a, 1.0)), max(me, max(b, 1.0))
    assert (mn, me) == (float(d["init_neigenv"]), float(d["init_eeigenv"]))


def test_unc_graph_build_oracle_matches_reference():
    d = load_golden(golden_files("unc_graph_build")[0])
    n, nrel = int(d["num_nodes"]), int(d["num_rels"])
    src, dst, typ, norm = O.unc_build_graph(n, nrel, d["triplets"])
    assert np.array_equal(src.numpy(), d["src"]) and np.array_equal(dst.numpy(), d["dst"]) and np.array_equal(typ.numpy(), d["type"])
    assert np.array_equal(norm.numpy(), d["norm"])                       # one division per edge: exact
    for mode in ("in", "out", "both"):
        got = O.unc_edge_norm(src, dst, n, mode).numpy()
        assert np.allclose(got, d["norm_" + mode], rtol=1e-6, atol=0)
        ds, dd = th.from_numpy(d["dir_src"]), th.from_numpy(d["dir_dst"])
        got = O.unc_edge_norm(ds, dd, 5, mode).numpy()                   # zero-degree endpoints: the Inf -> min branch
        assert np.array_equal(np.isnan(got), np.isnan(d["dir_norm_" + mode]))
        assert np.allclose(np.nan_to_num(got), np.nan_to_num(d["dir_norm_" + mode]), rtol=1e-6)
    ne, ee = O.unc_eigen_bounds(src, dst, n)
    assert float(ne) == float(d["node_eigenv"]) and float(ee) == float(d["edge_eigenv"])
    assert np.array_equal(d["in_deg"], np.bincount(d["dst"], minlength=n)) and np.array_equal(d["out_deg"], np.bincount(d["src"], minlength=n))


def test_dual_subisomorphism_oracle_matches_reference():
    d = load_golden(golden_files("dual_subiso")[0])
    assert int(d["num_cases"]) >= 3
    for c in range(int(d["num_cases"])):
        k = "%d." % c
        got = GO.dual_subisomorphisms(d[k + "p_u"], d[k + "p_v"], d[k + "p_el"], d[k + "g_u_sorted"], d[k + "g_v_sorted"],
                                      d[k + "g_el_sorted"], d[k + "sub"])
        assert np.array_equal(got, d[k + "dual_sorted_index"])
        assert np.array_equal(d[k + "g_eid_sorted"][got], d[k + "dual_eids"])
        # the sorted order the reference asks DGL for: by (src, dst), ties in edge-id order
        order = np.lexsort((np.arange(len(d[k + "g_u"])), d[k + "g_v"], d[k + "g_u"]))
        assert np.array_equal(order, d[k + "g_eid_sorted"])

"""The C integer oracle (oracle/graph_oracle.c) against golden vectors produced by the
reference's own convert_to_dual_graph / add_reversed_edges.  CPU only, bit-exact."""
import numpy as np
import pytest

import graph_oracle as GO
from conftest import golden_files, load_golden


def _frames(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


@pytest.mark.parametrize("path", golden_files("linegraph_"))
def test_line_graph_oracle_matches_reference(path):
    d = load_golden(path)
    ndata, edata = _frames(d, "ndata."), _frames(d, "edata.")
    dsrc, ddst, dn, dnd, ded = GO.convert_to_dual_graph(d["src"], d["dst"], int(d["num_nodes"]), ndata, edata)
    assert dn == int(d["dual_num_nodes"])
    assert np.array_equal(dsrc, d["dual_src"]) and np.array_equal(ddst, d["dual_dst"])
    ref_nd, ref_ed = _frames(d, "dual_ndata."), _frames(d, "dual_edata.")
    assert sorted(dnd) == sorted(ref_nd) and sorted(ded) == sorted(ref_ed)
    for k in ref_nd:
        assert np.array_equal(dnd[k], ref_nd[k]), k
    for k in ref_ed:
        assert np.array_equal(ded[k].reshape(ref_ed[k].shape), ref_ed[k]), k


@pytest.mark.parametrize("path", golden_files("addrev_"))
def test_add_reversed_edges_oracle_matches_reference(path):
    d = load_golden(path)
    s, t, i, l, r = GO.add_reversed_edges(d["src"], d["dst"], d["eid"], d["elabel"], int(d["max_ne"]), int(d["max_nel"]))
    assert np.array_equal(s, d["o_src"]) and np.array_equal(t, d["o_dst"])
    assert np.array_equal(i, d["o_eid"]) and np.array_equal(l, d["o_elabel"]) and np.array_equal(r, d["o_rev"])
    n = int(d["num_nodes"])
    assert np.array_equal(np.bincount(t, minlength=n), d["o_in_deg"])    # cached degrees, dataset.py:1289-1293
    assert np.array_equal(np.bincount(s, minlength=n), d["o_out_deg"])


def test_known_answers_from_survey_appendix_b():
    """Worked examples recorded from the reference."""
    s, t, n, nd, ed = GO.convert_to_dual_graph([0, 1, 2], [1, 2, 0], 3, {}, {})
    assert list(zip(s, t)) == [(2, 0), (0, 1), (1, 2)] and list(ed["id"]) == [0, 1, 2] and n == 3
    s, t, n, nd, ed = GO.convert_to_dual_graph([0, 1, 1], [1, 0, 1], 2, {}, {})
    assert list(zip(s, t)) == [(1, 0), (0, 1), (2, 1), (0, 2), (2, 2)] and list(ed["id"]) == [0, 1, 1, 1, 1]
    s, t, n, nd, ed = GO.convert_to_dual_graph(
        [0, 1, 1, 2], [1, 2, 0, 1], 3, {"id": np.arange(3), "label": np.array([7, 8, 9])},
        {"id": np.array([0, 1, 5, 6]), "label": np.array([3, 4, 13, 14]), "is_reversed": np.array([0, 0, 1, 1], bool)})
    assert n == 4 and list(nd["id"]) == [0, 1, 5, 6] and list(nd["label"]) == [3, 4, 13, 14]
    assert list(zip(s, t)) == [(2, 0), (0, 1), (3, 1), (0, 2), (3, 2), (1, 3)]
    assert list(ed["id"]) == [0, 1, 1, 1, 1, 2] and list(ed["label"]) == [7, 8, 8, 8, 8, 9]
    s, t, n, nd, ed = GO.convert_to_dual_graph(
        [0, 0, 1], [1, 1, 2], 3, {"label": np.array([1, 1, 1])}, {"id": np.array([0, 0, 1]), "label": np.array([2, 2, 3])})
    assert n == 2 and list(nd["label"]) == [2, 3] and list(zip(s, t)) == [(0, 1)]
    assert list(ed["id"]) == [1] and list(ed["label"]) == [1]


def test_collate_and_csr_oracle_basics():
    src, dst, no, eo, eg, ng = GO.collate([0, 1, 0, 2, 1], [1, 0, 1, 0, 2], [2, 3], [2, 3])
    assert list(src) == [0, 1, 2, 4, 3] and list(dst) == [1, 0, 3, 2, 4]
    assert list(no) == [0, 2, 5] and list(eo) == [0, 2, 5] and list(eg) == [0, 0, 1, 1, 1] and list(ng) == [0, 0, 1, 1, 1]
    ptr, ent = GO.csr_build([1, 0, 1, 1], [0, 1, 1, 0], 3)
    assert list(ptr) == [0, 1, 4, 4] and list(ent) == [3, 0, 5, 6]
    a, b = GO.eigen_bounds([0, 1, 2], [1, 2, 0], 3)
    assert (a, b) == (2.0, 2.0)


@pytest.mark.parametrize("path", golden_files("subiso_weights_"))
def test_subiso_weight_oracle_matches_reference(path):
    """batchify(return_weights="node,edge") of the reference (dataset.py:1604-1636) -- including the
    samples where a repeated (u, v) key replaces its earlier run -- against the C restatement."""
    d = load_golden(path)
    nw, ew = GO.batch_subiso_weights(d)
    assert nw.dtype == np.int64 and np.array_equal(nw, d["node_weights"])
    assert np.array_equal(ew, d["edge_weights"])
    # every subisomorphism row touches pattern_nodes target nodes
    rows = (np.diff(d["sample_ptr"]) // d["p_num_nodes"])
    assert np.array_equal(nw.sum(1), rows * d["p_num_nodes"]) and np.array_equal(rows, d["counts"])

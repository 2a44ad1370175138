"""GPU parity (bit-exact) of the batch's subisomorphism node / edge weights
(GraphAdjDataset.batchify(return_weights=...), dataset.py:1604-1636) against the reference's golden
vectors and, on larger random batches, against the C oracle applied sample by sample."""
import numpy as np
import pytest
import torch as th

import graph_oracle as GO
from conftest import golden_files, load_golden

pytestmark = pytest.mark.gpu


def _graphs(d, dev):
    from dualmessagepassing_amd.collate import collate_device
    out = []
    for t in ("p", "g"):
        nn, ne = d[t + "_num_nodes"].astype(np.int64), d[t + "_num_edges"].astype(np.int64)
        out.append(collate_device(th.from_numpy(d[t + "_src"]).to(dev), th.from_numpy(d[t + "_dst"]).to(dev),
                                  th.from_numpy(nn).to(dev), th.from_numpy(ne).to(dev), int(nn.sum()), int(ne.sum()),
                                  edata={"label": th.from_numpy(d[t + "_elabel"]).to(dev)}))
    return out


@pytest.mark.parametrize("path", golden_files("subiso_weights_"))
def test_subiso_weights_match_reference_golden(path, gpu):
    from dualmessagepassing_amd.collate import subiso_weights
    d = load_golden(path)
    pattern, graph = _graphs(d, gpu)
    nw, ew = subiso_weights(pattern, graph, th.from_numpy(d["sub_flat"]).to(gpu), th.from_numpy(d["sample_ptr"]).to(gpu),
                            "node,edge", validate=True)
    assert nw.dtype == th.int64 and ew.dtype == th.int64
    assert np.array_equal(nw.cpu().numpy(), d["node_weights"]) and np.array_equal(ew.cpu().numpy(), d["edge_weights"])
    only_n, none_e = subiso_weights(pattern, graph, th.from_numpy(d["sub_flat"]).to(gpu),
                                    th.from_numpy(d["sample_ptr"]).to(gpu), "node")
    assert none_e is None and th.equal(only_n, nw)


def test_subiso_weights_random_batch_equals_oracle(gpu):
    """64 ragged pairs with reversed edges, parallel edges and repeated keys; enumerated rows."""
    from dualmessagepassing_amd.collate import subiso_weights
    from dualmessagepassing_amd.harness import enumerate_subisomorphisms
    rng = np.random.default_rng(5)
    d = {k: [] for k in ("p_src", "p_dst", "p_elabel", "p_num_nodes", "p_num_edges", "g_src", "g_dst", "g_elabel",
                         "g_num_nodes", "g_num_edges", "sub")}
    for i in range(64):
        pn, gn = int(rng.integers(2, 5)), int(rng.integers(5, 12))
        pm, gm = int(rng.integers(1, 5)), int(rng.integers(12, 60))
        pu, pv = rng.integers(0, pn, pm), rng.integers(0, pn, pm)      # multigraph, self loops allowed
        gu, gv = rng.integers(0, gn, gm), rng.integers(0, gn, gm)
        pl, gl = rng.integers(0, 2, pm), rng.integers(0, 2, gm)
        pvl, gvl = rng.integers(0, 1 + i % 2, pn), rng.integers(0, 1 + i % 2, gn)
        sub = enumerate_subisomorphisms(pu, pv, pvl, pl, gu, gv, gvl, gl) if i % 7 else np.zeros((0, pn), np.int64)
        if i % 2:  # reversed copies (train.py:299-327)
            pu, pv, pl = np.r_[pu, pv], np.r_[pv, pu], np.r_[pl, pl + 2]
            gu, gv, gl = np.r_[gu, gv], np.r_[gv, gu], np.r_[gl, gl + 2]
        for k, v in (("p_src", pu), ("p_dst", pv), ("p_elabel", pl), ("g_src", gu), ("g_dst", gv), ("g_elabel", gl),
                     ("sub", sub.reshape(-1))):
            d[k].append(v.astype(np.int64))
        d["p_num_nodes"].append(pn), d["p_num_edges"].append(len(pu))
        d["g_num_nodes"].append(gn), d["g_num_edges"].append(len(gu))
    d["sample_ptr"] = np.r_[0, np.cumsum([len(s) for s in d["sub"]])].astype(np.int64)
    d["sub_flat"] = np.concatenate(d.pop("sub"))
    for k in list(d):
        if isinstance(d[k], list):
            d[k] = np.concatenate(d[k]) if k.endswith(("src", "dst", "elabel")) else np.array(d[k], np.int64)
    assert d["sub_flat"].size > 500
    want_n, want_e = GO.batch_subiso_weights(d)
    pattern, graph = _graphs(d, gpu)
    nw, ew = subiso_weights(pattern, graph, th.from_numpy(d["sub_flat"]).to(gpu), th.from_numpy(d["sample_ptr"]).to(gpu),
                            ("node", "edge"), validate=True)
    assert np.array_equal(nw.cpu().numpy(), want_n) and np.array_equal(ew.cpu().numpy(), want_e)
    assert want_e.sum() > 0


def test_subiso_weights_reject_rows_outside_the_graph(gpu):
    from dualmessagepassing_amd import _lib
    from dualmessagepassing_amd.collate import subiso_weights
    d = load_golden(golden_files("subiso_weights_plain")[0])
    pattern, graph = _graphs(d, gpu)
    bad = d["sub_flat"].copy()
    bad[0] = 10 ** 6
    with pytest.raises(_lib.DmpError):
        subiso_weights(pattern, graph, th.from_numpy(bad).to(gpu), th.from_numpy(d["sample_ptr"]).to(gpu), validate=True)


def test_batchify_with_weights(gpu):
    """``batchify(samples, return_weights="node,edge")`` over single graphs."""
    from dualmessagepassing_amd.collate import batchify
    from dualmessagepassing_amd.graph import BatchedGraph
    d = load_golden(golden_files("subiso_weights_rev")[0])
    samples, po, go = [], 0, 0
    for i in range(len(d["p_num_nodes"])):
        gs = {}
        for t, off in (("p", po), ("g", go)):
            n, e = int(d[t + "_num_nodes"][i]), int(d[t + "_num_edges"][i])
            gs[t] = BatchedGraph(th.from_numpy(d[t + "_src"][off:off + e]).to(gpu), th.from_numpy(d[t + "_dst"][off:off + e]).to(gpu),
                                 n, None, None, {}, {"label": th.from_numpy(d[t + "_elabel"][off:off + e]).to(gpu)})
        po, go = po + int(d["p_num_edges"][i]), go + int(d["g_num_edges"][i])
        sub = d["sub_flat"][d["sample_ptr"][i]:d["sample_ptr"][i + 1]].reshape(-1, int(d["p_num_nodes"][i]))
        samples.append({"id": "s-%d" % i, "pattern": gs["p"], "graph": gs["g"], "counts": len(sub),
                        "subisomorphisms": th.from_numpy(sub)})
    ids, pattern, graph, counts, (nw, ew) = batchify(samples, return_weights="node,edge", device=gpu)
    assert ids[3] == "s-3" and np.array_equal(counts.cpu().numpy(), d["counts"])
    assert np.array_equal(nw.cpu().numpy(), d["node_weights"]) and np.array_equal(ew.cpu().numpy(), d["edge_weights"])

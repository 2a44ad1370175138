"""Oracle restatement of LRPLayer / DMPLRPPoolLayer (oracle/dmp_oracle.py::lrp_layer, dmplrp_layer) against fixtures emitted
by the reference's own layers over the permutation matrices of its LRPDataset (oracle/make_golden.py::gen_lrp)."""
import numpy as np
import pytest
import torch as th

import dmp_oracle as O
from conftest import golden_files, load_golden


def _t(a):
    return th.from_numpy(np.asarray(a))


def _coo(d, key):
    return _t(d[key + ".indices"]), _t(d[key + ".values"]), d[key + ".shape"]


@pytest.mark.parametrize("path", golden_files("lrp_layer_"))
def test_lrp_layers_oracle_matches_reference(path):
    d = load_golden(path)
    kw = {k: eval(v) for k, v in zip(d["kw_keys"].tolist(), d["kw_vals"].tolist())}
    params = {k[2:]: _t(v).clone().requires_grad_(True) for k, v in d.items() if k.startswith("p.")}
    x, z = _t(d["x"]).clone().requires_grad_(True), _t(d["z"]).clone().requires_grad_(True)
    pool, n2p, e2p = _coo(d, "pool"), _coo(d, "n2p"), _coo(d, "e2p")
    if "dmplrp" in path:
        node_out, edge_out = O.dmplrp_layer(params, _t(d["g_src"]), _t(d["g_dst"]), _t(d["g_edata.is_reversed"]), _t(d["g_ndata.out_deg"]),
                                            pool, n2p, e2p, x, z, kw["act_func"], kw["lrp_seq_len"], kw["num_mlp_layers"])
    else:
        node_out, edge_out = O.lrp_layer(params, pool, n2p, e2p, _t(d["g_ndata.in_deg"]), x, z, kw["act_func"], kw["lrp_seq_len"])
    assert th.allclose(node_out, _t(d["node_out"]), rtol=1e-5, atol=1e-5) and th.allclose(edge_out, _t(d["edge_out"]), rtol=1e-5, atol=1e-5)
    ((node_out * _t(d["wn"])).sum() + (edge_out * _t(d["we"])).sum()).backward()
    assert th.allclose(x.grad, _t(d["dx"]), rtol=1e-4, atol=1e-5) and th.allclose(z.grad, _t(d["dz"]), rtol=1e-4, atol=1e-5)
    for k, p in params.items():
        if "g." + k in d:
            assert th.allclose(p.grad, _t(d["g." + k]), rtol=1e-4, atol=2e-5), k
    # structure of the matrices the product relies on: selection matrices with unit entries and at most one per row;
    # the pooling matrix averages CONTIGUOUS column ranges (dataset.py:1795-1811)
    for key in ("n2p", "e2p"):
        rows = d[key + ".indices"][0]
        assert np.all(d[key + ".values"] == 1.0) and len(np.unique(rows)) == len(rows)
    prow, pcol = d["pool.indices"]
    assert np.array_equal(pcol, np.arange(len(pcol))) and np.all(np.diff(prow) >= 0)

"""GPU parity of the local-relational-pooling layers and models (dualmessagepassing_amd/lrp.py: row gather + dense slot
contraction + segment-sum pooling) against fixtures from the reference's own ``LRPLayer`` / ``DMPLRPPoolLayer`` /
``LRP`` / ``DMPLRP`` over the matrices its ``LRPDataset`` builds (models/lrp.py, models/dmplrp.py, dataset.py:1751-1862).
Tolerances: outputs / input gradients 2e-5, parameter gradients 3e-4 of the largest reference value."""
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


def _sparse(d, key, dev):
    return th.sparse_coo_tensor(_t(d[key + ".indices"]), _t(d[key + ".values"]), tuple(int(x) for x in d[key + ".shape"])).to(dev)


def _graph(d, t, dev):
    from dualmessagepassing_amd.graph import BatchedGraph
    g = BatchedGraph(_t(d[t + "_src"]).to(dev), _t(d[t + "_dst"]).to(dev), int(d[t + "_num_nodes"]), _t(d[t + "_bnn"]).to(dev),
                     _t(d[t + "_bne"]).to(dev))
    for k, v in d.items():
        if k.startswith(t + "_ndata."):
            g.ndata[k.split(".", 1)[1]] = _t(v).to(dev)
        if k.startswith(t + "_edata."):
            g.edata[k.split(".", 1)[1]] = _t(v).to(dev)
    return g


def _host_graphs(d, t):
    """The batch's single graphs as (src, dst, n, is_reversed) with local node ids."""
    bnn, bne = d[t + "_bnn"], d[t + "_bne"]
    out, n0, e0 = [], 0, 0
    for n, e in zip(bnn.tolist(), bne.tolist()):
        out.append((d[t + "_src"][e0:e0 + e] - n0, d[t + "_dst"][e0:e0 + e] - n0, n, d[t + "_edata.is_reversed"][e0:e0 + e]))
        n0, e0 = n0 + n, e0 + e
    return out


@pytest.mark.parametrize("path", golden_files("lrp_layer_"))
@pytest.mark.parametrize("via", ["sparse", "index"])
def test_lrp_layers_match_reference(path, via, gpu):
    from dualmessagepassing_amd.lrp import DMPLRPPoolLayer, LRPLayer, PermIndex
    d = load_golden(path)
    kw = {k: eval(v) for k, v in zip(d["kw_keys"].tolist(), d["kw_vals"].tolist())}
    layer = (DMPLRPPoolLayer if "dmplrp" in path else LRPLayer)(**kw)
    layer.load_state_dict({k[2:]: _t(v) for k, v in d.items() if k.startswith("p.")}, strict=True)
    layer.to(gpu)
    g = _graph(d, "g", gpu)
    x, z = _t(d["x"]).to(gpu).requires_grad_(True), _t(d["z"]).to(gpu).requires_grad_(True)
    pool, n2p, e2p = _sparse(d, "pool", gpu), _sparse(d, "n2p", gpu), _sparse(d, "e2p", gpu)
    if via == "index":      # the permutation index built here from the graphs instead of from the reference's matrices
        perm = PermIndex.from_graphs(_host_graphs(d, "g"), gpu, kw["lrp_seq_len"])
        ref = PermIndex.from_sparse(pool, n2p, e2p, kw["lrp_seq_len"])
        assert th.equal(perm.slot, ref.slot) and th.equal(perm.sizes, ref.sizes) and th.equal(perm.scale, ref.scale)
        out = layer(g, x, z, perm)
    else:
        out = layer(g, x, z, pool, n2p, e2p)
    node_out, edge_out = out[0], out[1]
    _close(node_out, d["node_out"], 2e-5, "node_out")
    _close(edge_out, d["edge_out"], 2e-5, "edge_out")
    ((node_out * _t(d["wn"]).to(gpu)).sum() + (edge_out * _t(d["we"]).to(gpu)).sum()).backward()
    _close(x.grad, d["dx"], 2e-5, "dx")
    _close(z.grad, d["dz"], 2e-5, "dz")
    for k, p in layer.named_parameters():
        if "g." + k in d:
            _close(p.grad, d["g." + k], 3e-4, "grad " + k)
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k


@pytest.mark.parametrize("path", golden_files("lrp_model_"))
def test_lrp_models_match_reference(path, gpu):
    from dualmessagepassing_amd.basemodel import build_model
    d = load_golden(path)
    config = {str(k): eval(str(v)) for k, v in zip(d["config_keys"], d["config_vals"])}
    model = build_model(**config)
    missing, unexpected = model.load_state_dict({k[3:]: _t(v) for k, v in d.items() if k.startswith("sd.")}, strict=True)
    assert not missing and not unexpected
    model.to(gpu)
    pattern, graph = _graph(d, "p", gpu), _graph(d, "g", gpu)
    out = model(pattern, _sparse(d, "p_pool", gpu), _sparse(d, "p_n2p", gpu), _sparse(d, "p_e2p", gpu),
                graph, _sparse(d, "g_pool", gpu), _sparse(d, "g_n2p", gpu), _sparse(d, "g_e2p", gpu))
    for k in ("p_v_rep", "p_e_rep", "g_v_rep", "g_e_rep", "pred_c"):
        _close(out[k], d["out." + k], 2e-4, k)
    out["pred_c"].sum().backward()
    n = 0
    for k, p in model.named_parameters():
        if "grad." + k in d:
            _close(p.grad, d["grad." + k], 3e-4, "grad " + k)
            n += 1
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    assert n > 10

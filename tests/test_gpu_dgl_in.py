"""DGLGraph-in acceptance (north_star; reference models/dmpnn.py:96-109,158-166, train.py:606-611): layers and models
take any graph object with the DGL surface, give the results of the BatchedGraph path bit for bit, and leave the
reference's side effects (``out_deg``, ``node_feat``, ``node_agg``, ``edge_feat``) on the CALLER's graph."""
import numpy as np
import pytest
import torch as th

from conftest import golden_files, load_golden
from util_dglike import DGLike
from util_graphs import er_batch

pytestmark = pytest.mark.gpu


def _t(a):
    return th.from_numpy(np.asarray(a))


def _pair(gpu, rev=True, B=6, n=12, m=30, seed=3):
    from dualmessagepassing_amd.graph import BatchedGraph
    rng = np.random.default_rng(seed)
    src, dst, r, N, bnn, bne = er_batch(B, n, m, rng, add_rev=rev)
    ts, td = _t(src).to(gpu), _t(dst).to(gpu)
    a = BatchedGraph(ts, td, N, _t(bnn).to(gpu), _t(bne).to(gpu))
    b = DGLike(ts, td, N, _t(bnn).to(gpu), _t(bne).to(gpu))
    if rev:
        a.edata["is_reversed"] = _t(r).to(gpu)
        b.edata["is_reversed"] = _t(r).to(gpu)
    return a, b, N, len(src)


@pytest.mark.parametrize("h,act", [(128, "relu"), (64, "leaky_relu"), (20, "relu")])
def test_dmplayer_takes_dgl_like_graph(h, act, gpu):
    from dualmessagepassing_amd.dmpnn import DMPLayer
    a, b, N, E = _pair(gpu)
    th.manual_seed(h)
    layer = DMPLayer(h, h, batch_norm=False, act_func=act).to(gpu)
    x, z = th.randn(N, h, device=gpu), th.randn(E, h, device=gpu)
    outs = []
    for g in (a, b):
        xg, zg = x.clone().requires_grad_(True), z.clone().requires_grad_(True)
        layer.zero_grad()
        no, eo = layer(g, xg, zg)
        (no.sum() + (eo * eo).sum()).backward()
        outs.append((no, eo, xg.grad, zg.grad, layer.in_weight.grad.clone()))
    for p, q in zip(*outs):
        assert th.equal(p, q)
    # side effects of _node_init_func / _edge_init_func / update_all on the caller's own frames
    for k in ("out_deg", "node_feat", "node_agg"):
        assert k in b.ndata and k in b.ndata.writes
    assert "edge_feat" in b.edata
    assert th.equal(b.ndata["out_deg"], b.out_degrees())
    # a cached out_deg on the caller's graph is what the layer reads (dmpnn.py:100-101)
    b.ndata["out_deg"] = b.ndata["out_deg"] + 3
    no2, eo2 = layer(b, x, z)
    assert th.equal(no2, outs[1][0])                  # the node side does not depend on out_deg
    assert not th.equal(eo2, outs[1][1])              # the edge side does: 2 (1 + log2(1 + out_deg[dst]))
    # the same graph object is converted (and indexed) once
    from dualmessagepassing_amd.graph import BatchedGraph
    assert BatchedGraph.from_graph(b) is BatchedGraph.from_graph(b)


def test_compgcn_and_linegraph_take_dgl_like_graph(gpu):
    from dualmessagepassing_amd.compgcn import CompGCNLayer
    from dualmessagepassing_amd.linegraph import convert_to_dual_graph
    a, b, N, E = _pair(gpu)
    th.manual_seed(1)
    layer = CompGCNLayer(16, 16, comp_opt="mult", edge_norm="both", batch_norm=False).to(gpu)
    x, z = th.randn(N, 16, device=gpu), th.randn(E, 16, device=gpu)
    na, ea = layer(a, x, z)
    nb, eb = layer(b, x, z)
    assert th.equal(na, nb) and th.equal(ea, eb)
    da, db = convert_to_dual_graph(a), convert_to_dual_graph(b)
    for p, q in zip(da.all_edges(), db.all_edges()):
        assert th.equal(p, q)


def test_full_model_takes_dgl_like_graphs(gpu):
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.graph import BatchedGraph
    d = load_golden([p for p in golden_files("fullmodel_") if p.endswith("ragged.npz")][0])
    cfg = {k: eval(v) for k, v in zip(d["config_keys"].tolist(), d["config_vals"].tolist())}
    model = build_model(**cfg)
    model.load_state_dict({k[3:]: _t(v) for k, v in d.items() if k.startswith("sd.")}, strict=True)
    model.to(gpu)
    graphs = {}
    for kind in ("batched", "dgl"):
        pair = []
        for t in ("p", "g"):
            u, v, n = _t(d[t + "_src"]).to(gpu), _t(d[t + "_dst"]).to(gpu), int(d[t + "_num_nodes"])
            bnn, bne = _t(d[t + "_bnn"]).to(gpu), _t(d[t + "_bne"]).to(gpu)
            g = BatchedGraph(u, v, n, bnn, bne) if kind == "batched" else DGLike(u, v, n, bnn, bne)
            for k in ("id", "label", "in_deg", "out_deg"):
                g.ndata[k] = _t(d["%s_ndata.%s" % (t, k)]).to(gpu)
            for k in ("id", "label", "is_reversed"):
                g.edata[k] = _t(d["%s_edata.%s" % (t, k)]).to(gpu)
            pair.append(g)
        graphs[kind] = pair
    oa = model(*graphs["batched"])
    ob = model(*graphs["dgl"])
    for k in oa:
        if oa[k] is not None:
            assert th.equal(oa[k], ob[k]), k
    ref = _t(d["out.pred_c"]).double()
    assert float((ob["pred_c"].double().cpu() - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))

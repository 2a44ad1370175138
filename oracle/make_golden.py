"""Emit the golden fixtures under tests/golden/ by running the REFERENCE's own code.

Runs only in the development container (needs /root/reference).  The reference's
model files are imported over the DGL/numba/igraph stand-ins of
``oracle/ref_standin.py``; their arithmetic is the reference's own torch code.
Outputs are data only (inputs, parameters, expected outputs/gradients) -- no
reference source is copied.

    python oracle/make_golden.py            # writes tests/golden/*.npz

Fixtures
  dmplayer_*.npz     DMPLayer fwd+bwd (models/dmpnn.py:158-166)
  dmpnn_rep.npz      DMPNN.get_pattern_rep / get_graph_rep, L=3, gates, residual (dmpnn.py:215-277)
  compgcn_*.npz      CompGCNLayer fwd+bwd (models/compgcn.py:265-274)
  linegraph_*.npz    convert_to_dual_graph (utils/graph.py:74-169)
  addrev_*.npz       add_reversed_edges, GraphAdj branch (train.py:299-327)
  fullmodel_*.npz    DMPNN.forward(pattern, graph): 15 OutputDict entries + grads (basemodel.py:1500-1663)
  unc_dualconv_*.npz DualGraphConv (UNC Model/DMPNN/src/model.py:117-273) -- separate interpreter
"""
import os
import subprocess
import sys
import zlib

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
# DMP_GOLDEN_OUT: write somewhere else (tests/test_fixtures_regenerate.py regenerates into a temporary directory)
OUT = os.environ.get("DMP_GOLDEN_OUT") or os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, HERE)

# Reproducible to the bit: one CPU thread (multi-threaded reductions sum in a thread-count-dependent order; a chaotic
# five-epoch training run amplifies that to percents) and deterministic algorithms only.
th.set_num_threads(1)
th.use_deterministic_algorithms(True)


def er_edges(n, m, rng):
    """m distinct ordered pairs u != v, uniformly without replacement."""
    total = n * (n - 1)
    pick = rng.choice(total, size=m, replace=False)
    u = pick // (n - 1)
    r = pick % (n - 1)
    v = r + (r >= u)
    return u.astype(np.int64), v.astype(np.int64)


def named_graphs(rng):
    g = {}
    g["cycle3"] = (np.array([0, 1, 2]), np.array([1, 2, 0]), 3)
    g["pair2"] = (np.array([0]), np.array([1]), 2)
    g["star8"] = (np.arange(1, 9), np.zeros(8, dtype=np.int64), 9)
    u, v = er_edges(8, 12, rng)
    g["er8_12"] = (u, v, 8)
    u, v = er_edges(64, 256, rng)
    g["er64_256"] = (u, v, 64)
    # multigraph with a self loop and an isolated node
    g["multi"] = (np.array([0, 0, 1, 2, 2, 1]), np.array([1, 1, 2, 2, 0, 0]), 4)
    return {k: (np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64), n) for k, (a, b, n) in g.items()}


def with_rev(u, v):
    e = len(u)
    return np.concatenate([u, v]), np.concatenate([v, u]), np.concatenate([np.zeros(e, bool), np.ones(e, bool)])


def t2n(d):
    """tensors -> numpy; ``None`` entries (gradients of unused parameters) are dropped."""
    return {k: (v.detach().numpy() if isinstance(v, th.Tensor) else np.asarray(v)) for k, v in d.items()
            if v is not None}


def seed_of(*key):
    return zlib.crc32(repr(key).encode()) % (2 ** 31)


def gen_dmplayer():
    import dgl
    from models.dmpnn import DMPLayer
    rng = np.random.default_rng(1234)
    graphs = named_graphs(rng)
    cases = [
        ("cycle3", 4, "relu", False, 2), ("cycle3", 4, "relu", True, 2),
        ("pair2", 4, "leaky_relu", True, 2), ("star8", 8, "relu", True, 2),
        ("multi", 8, "leaky_relu", True, 2), ("multi", 6, "relu", False, 0),
        ("er8_12", 64, "relu", True, 2), ("er8_12", 64, "leaky_relu", False, 2),
        ("er64_256", 64, "relu", True, 2), ("er64_256", 128, "leaky_relu", True, 2),
    ]
    for name, h, act, use_rev, nmlp in cases:
        u, v, n = graphs[name]
        rev = None
        if use_rev:
            u, v, rev = with_rev(u, v)
        th.manual_seed(seed_of(name, h, act, use_rev))
        layer = DMPLayer(h, h, init_neigenv=4.0, init_eeigenv=4.0, num_mlp_layers=nmlp, batch_norm=False,
                         act_func=act, dropout=0.0)
        with th.no_grad():  # biases are zero-initialised; make them matter
            layer.nbias.uniform_(-0.1, 0.1)
            layer.ebias.uniform_(-0.1, 0.1)
            for m in list(layer.nmlp.modules()) + list(layer.emlp.modules()):
                if isinstance(m, th.nn.Linear):
                    m.bias.uniform_(-0.1, 0.1)
        g = dgl.DGLGraph.from_edges(u, v, n)
        if rev is not None:
            g.edata["is_reversed"] = th.from_numpy(rev)
        x = th.randn(n, h, requires_grad=True)
        z = th.randn(len(u), h, requires_grad=True)
        node_out, edge_out = layer(g, x, z)
        wn = th.randn_like(node_out)
        we = th.randn_like(edge_out)
        ((node_out * wn).sum() + (edge_out * we).sum()).backward()
        d = {"src": u, "dst": v, "num_nodes": n, "out_deg": g.ndata["out_deg"],
             "x": x, "z": z, "node_out": node_out, "edge_out": edge_out, "wn": wn, "we": we,
             "dx": x.grad, "dz": z.grad, "edge_agg": g.edata["edge_agg"], "node_agg": g.ndata["node_agg"],
             "act_func": act, "num_mlp_layers": nmlp}
        if rev is not None:
            d["rev"] = rev
        for k, p in layer.named_parameters():
            d["p." + k] = p
            d["g." + k] = p.grad
        fn = "dmplayer_%s_h%d_%s_%s_m%d.npz" % (name, h, act, "rev" if use_rev else "norev", nmlp)
        np.savez_compressed(os.path.join(OUT, fn), **t2n(d))
        print("wrote", fn)


def gen_dmplayer_bn():
    """DMPLayer with its constructor default ``batch_norm=True`` (models/dmpnn.py:17-28,45-60: Linear -> BatchNorm1d ->
    activation -> Linear in both MLPs): training mode (batch statistics, running-average update) and evaluation mode (the
    running statistics a training pass has left), forward + backward."""
    import dgl
    from models.dmpnn import DMPLayer
    rng = np.random.default_rng(4321)
    graphs = named_graphs(rng)
    cases = [("er8_12", 64, "relu"), ("er64_256", 64, "leaky_relu"), ("er64_256", 128, "relu"), ("star8", 8, "leaky_relu")]
    for name, h, act in cases:
        u, v, n = graphs[name]
        u, v, rev = with_rev(u, v)
        th.manual_seed(seed_of("bn", name, h, act))
        layer = DMPLayer(h, h, init_neigenv=4.0, init_eeigenv=4.0, num_mlp_layers=2, batch_norm=True, act_func=act, dropout=0.0)
        with th.no_grad():
            layer.nbias.uniform_(-0.1, 0.1)
            layer.ebias.uniform_(-0.1, 0.1)
            for m in list(layer.nmlp.modules()) + list(layer.emlp.modules()):
                if isinstance(m, th.nn.Linear):
                    m.bias.uniform_(-0.1, 0.1)
                if isinstance(m, th.nn.BatchNorm1d):              # affine parameters and running statistics that matter
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.uniform_(-0.2, 0.2)
                    m.running_mean.uniform_(-0.3, 0.3)
                    m.running_var.uniform_(0.5, 2.0)
        x0, z0 = th.randn(n, h), th.randn(len(u), h)
        for mode in ("train", "eval"):
            layer.train(mode == "train")
            g = dgl.DGLGraph.from_edges(u, v, n)
            g.edata["is_reversed"] = th.from_numpy(rev)
            d = {"src": u, "dst": v, "rev": rev, "num_nodes": n, "act_func": act, "mode": mode}
            for k, b in layer.named_buffers():                    # the statistics BEFORE the pass
                d["b0." + k] = b.clone()
            x, z = x0.clone().requires_grad_(True), z0.clone().requires_grad_(True)
            for p in layer.parameters():
                p.grad = None
            node_out, edge_out = layer(g, x, z)
            wn, we = th.randn_like(node_out), th.randn_like(edge_out)
            ((node_out * wn).sum() + (edge_out * we).sum()).backward()
            d.update({"out_deg": g.ndata["out_deg"], "x": x, "z": z, "node_out": node_out, "edge_out": edge_out, "wn": wn,
                      "we": we, "dx": x.grad, "dz": z.grad})
            for k, p in layer.named_parameters():
                d["p." + k] = p
                d["g." + k] = p.grad
            for k, b in layer.named_buffers():                    # ... and after it (training mode moves them)
                d["b1." + k] = b.clone()
            fn = "bnlayer_dmp_%s_h%d_%s_%s.npz" % (name, h, act, mode)
            np.savez_compressed(os.path.join(OUT, fn), **t2n(d))
            print("wrote", fn)


def gen_dmpnn_rep():
    """get_pattern_rep / get_graph_rep of the reference DMPNN class, called unbound on a
    light holder object (they only touch p_rep_net / g_rep_net / rep_residual)."""
    import dgl
    from models.dmpnn import DMPNN, DMPLayer
    from models.container import ModuleDict, ModuleList
    rng = np.random.default_rng(77)
    h, L, B = 32, 3, 4
    th.manual_seed(5)

    class Holder(th.nn.Module):
        pass

    hold = Holder()
    hold.rep_residual = True
    layers = ModuleList()
    for i in range(L):
        layers.add_module("graph_dmpnn_(%d)" % i,
                          DMPLayer(h, h, num_mlp_layers=2, batch_norm=False, act_func="relu", dropout=0.0))
    hold.g_rep_net = ModuleDict({"dmpnn": layers})
    hold.p_rep_net = hold.g_rep_net  # share_rep_net (dmpnn.py:186-188)

    def batch(nv, ne):
        gs = []
        for _ in range(B):
            u, v = er_edges(nv, ne, rng)
            u, v, rev = with_rev(u, v)
            g = dgl.DGLGraph.from_edges(u, v, nv)
            g.edata["is_reversed"] = th.from_numpy(rev)
            g.ndata["out_deg"] = g.out_degrees()
            gs.append(g)
        return dgl.batch(gs)

    pattern, graph = batch(4, 5), batch(16, 40)
    d = {}
    for tag, g in (("p", pattern), ("g", graph)):
        n, e = g.number_of_nodes(), g.number_of_edges()
        v_emb = th.randn(n, h, requires_grad=True)
        e_emb = th.randn(e, h, requires_grad=True)
        for p in hold.parameters():
            p.grad = None
        if tag == "p":
            v_rep, e_rep = DMPNN.get_pattern_rep(hold, g, v_emb, e_emb)
        else:
            v_gate = (th.rand(n, 1) > 0.3).float()
            e_gate = (th.rand(e, 1) > 0.3).float()
            d["g_v_gate"], d["g_e_gate"] = v_gate, e_gate
            v_rep, e_rep = DMPNN.get_graph_rep(hold, g, v_emb, e_emb, v_gate=v_gate, e_gate=e_gate)
        wv, we = th.randn_like(v_rep), th.randn_like(e_rep)
        ((v_rep * wv).sum() + (e_rep * we).sum()).backward()
        d.update({tag + "_src": g._u, tag + "_dst": g._v, tag + "_rev": g.edata["is_reversed"],
                  tag + "_num_nodes": n, tag + "_out_deg": g.ndata["out_deg"],
                  tag + "_bnn": g.batch_num_nodes(), tag + "_bne": g.batch_num_edges(),
                  tag + "_v_emb": v_emb, tag + "_e_emb": e_emb, tag + "_v_rep": v_rep, tag + "_e_rep": e_rep,
                  tag + "_wv": wv, tag + "_we": we, tag + "_dv_emb": v_emb.grad, tag + "_de_emb": e_emb.grad})
        for k, p in hold.g_rep_net.named_parameters():
            d["%s_grad.%s" % (tag, k)] = p.grad.clone()
    for k, p in hold.g_rep_net.named_parameters():
        d["p." + k] = p
    d["hid"], d["layers"] = h, L
    np.savez_compressed(os.path.join(OUT, "dmpnn_rep.npz"), **t2n(d))
    print("wrote dmpnn_rep.npz")


def gen_compgcn():
    import dgl
    from models.compgcn import CompGCNLayer
    rng = np.random.default_rng(4321)
    graphs = named_graphs(rng)
    cases = [("er8_12", 16, "sub", "none", True, True), ("er8_12", 16, "mult", "both", True, True),
             ("er8_12", 16, "corr", "in", True, True), ("multi", 8, "mult", "out", True, False),
             ("er64_256", 64, "sub", "both", True, True), ("er64_256", 64, "mult", "in", False, True),
             ("star8", 8, "sub", "both", True, False)]
    for name, h, comp, norm, use_rev, self_loop in cases:
        u, v, n = graphs[name]
        rev = None
        if use_rev:
            u, v, rev = with_rev(u, v)
        th.manual_seed(seed_of(name, h, comp, norm))
        layer = CompGCNLayer(h, h, self_loop=self_loop, comp_opt=comp, edge_norm=norm, bias=True,
                             batch_norm=False, act_func="relu", dropout=0.0)
        with th.no_grad():
            layer.bias.uniform_(-0.1, 0.1)
        g = dgl.DGLGraph.from_edges(u, v, n)
        if rev is not None:
            g.edata["is_reversed"] = th.from_numpy(rev)
        x = th.randn(n, h, requires_grad=True)
        z = th.randn(len(u), h, requires_grad=True)
        node_out, edge_out = layer(g, x, z)
        wn, we = th.randn_like(node_out), th.randn_like(edge_out)
        ((node_out * wn).sum() + (edge_out * we).sum()).backward()
        d = {"src": u, "dst": v, "num_nodes": n, "x": x, "z": z, "node_out": node_out, "edge_out": edge_out,
             "wn": wn, "we": we, "dx": x.grad, "dz": z.grad, "comp_opt": comp, "edge_norm": norm,
             "self_loop": self_loop}
        if rev is not None:
            d["rev"] = rev
        if "norm" in g.edata:
            d["norm"] = g.edata["norm"]
        for k, p in layer.named_parameters():
            d["p." + k] = p
            d["g." + k] = p.grad
        fn = "compgcn_%s_h%d_%s_%s_%s.npz" % (name, h, comp, norm, "rev" if use_rev else "norev")
        np.savez_compressed(os.path.join(OUT, fn), **t2n(d))
        print("wrote", fn)


def gen_linegraph():
    import dgl
    from utils.graph import convert_to_dual_graph
    rng = np.random.default_rng(99)
    graphs = named_graphs(rng)
    graphs["loop3"] = (np.array([0, 1, 1]), np.array([1, 0, 1]), 2)
    graphs["empty"] = (np.zeros(0, np.int64), np.zeros(0, np.int64), 3)
    u, v = er_edges(16, 60, rng)
    graphs["er16_60"] = (u, v, 16)

    def run(tag, u, v, n, ndata, edata):
        g = dgl.DGLGraph.from_edges(u, v, n)
        for k, val in ndata.items():
            g.ndata[k] = th.as_tensor(val)
        for k, val in edata.items():
            g.edata[k] = th.as_tensor(val)
        dg = convert_to_dual_graph(g)
        d = {"src": u, "dst": v, "num_nodes": n, "dual_src": dg._u, "dual_dst": dg._v,
             "dual_num_nodes": dg.number_of_nodes()}
        for k, val in ndata.items():
            d["ndata." + k] = np.asarray(val)
        for k, val in edata.items():
            d["edata." + k] = np.asarray(val)
        for k, val in dg.ndata.items():
            d["dual_ndata." + k] = val
        for k, val in dg.edata.items():
            d["dual_edata." + k] = val
        np.savez_compressed(os.path.join(OUT, "linegraph_%s.npz" % tag), **t2n(d))
        print("wrote linegraph_%s.npz" % tag, "dual nodes", dg.number_of_nodes(), "dual edges", dg.number_of_edges())

    for name, (u, v, n) in graphs.items():
        # plain branch: no id / label frames (graph.py:96-103,126-134)
        run(name + "_plain", u, v, n, {}, {})
        if len(u) == 0:
            continue
        # training-pipeline branch: ids = arange, labels, then add_reversed_edges' arrays
        e = len(u)
        nl = rng.integers(0, 4, size=n)
        el = rng.integers(0, 3, size=e)
        max_ne, max_nel = e + 3, 3  # holes between e and max_ne (train.py:306)
        uu, vv, rev = with_rev(u, v)
        run(name + "_idrev", uu, vv, n, {"id": np.arange(n), "label": nl},
            {"id": np.concatenate([np.arange(e), max_ne + np.arange(e)]),
             "label": np.concatenate([el, el + max_nel]), "is_reversed": rev})
        # duplicate edge ids (merge rule graph.py:80-95) + dedupe of (uid, label, vid) keys (:110-125)
        dup = rng.integers(0, max(1, e // 2), size=e)
        run(name + "_dupid", u, v, n, {"id": np.arange(n), "label": nl}, {"id": dup, "label": el[dup % e]})


def gen_addrev():
    """add_reversed_edges on the reference's dataset.Graph class (train.py:299-327 drives
    dataset.Graph.add_edges, dataset.py:1261-1293, which also updates cached degrees)."""
    import dgl
    import dataset as ref_dataset  # reference module
    rng = np.random.default_rng(11)
    graphs = named_graphs(rng)
    for name in ("cycle3", "er8_12", "multi"):
        u, v, n = graphs[name]
        e = len(u)
        g = ref_dataset.Graph()
        g.add_nodes(n)
        dgl.DGLGraph.add_edges(g, u, v)
        g.ndata["id"] = th.arange(n)
        g.ndata["label"] = th.from_numpy(rng.integers(0, 4, size=n))
        g.edata["id"] = th.arange(e)
        g.edata["label"] = th.from_numpy(rng.integers(0, 3, size=e))
        g.ndata["in_deg"] = dgl.DGLGraph.in_degrees(g)
        g.ndata["out_deg"] = dgl.DGLGraph.out_degrees(g)
        max_nge, max_ngel = e + 2, 5
        d = {"src": u, "dst": v, "num_nodes": n, "eid": g.edata["id"].clone(), "elabel": g.edata["label"].clone(),
             "max_ne": max_nge, "max_nel": max_ngel}
        # body of train.py:303-316 (GraphAdj branch) on this one graph
        num_ge = g.number_of_edges()
        uu, vv = g.all_edges(form="uv", order="eid")
        eid = th.arange(max_nge, max_nge + num_ge)
        g.add_edges(vv, uu, data={"id": eid, "label": g.edata["label"] + max_ngel,
                                  "is_reversed": th.ones((num_ge,), dtype=th.bool)})
        d.update({"o_src": g._u, "o_dst": g._v, "o_eid": g.edata["id"], "o_elabel": g.edata["label"],
                  "o_rev": g.edata["is_reversed"], "o_in_deg": g.ndata["in_deg"], "o_out_deg": g.ndata["out_deg"]})
        np.savez_compressed(os.path.join(OUT, "addrev_%s.npz" % name), **t2n(d))
        print("wrote addrev_%s.npz" % name)


def gen_full_model():
    """Full ``DMPNN.forward(pattern, graph)`` of the reference (models/basemodel.py:1500-1663 +
    models/dmpnn.py) -> the 15 OutputDict entries and the gradients of ``pred_c.sum()``.
    'uniform': BASELINE config-1 shape (B=32, pattern (8,12), target (64,256), add_rev, hid 64);
    'ragged' : different sizes per pair (exercises the padding / mask / filter-gate paths)."""
    import dgl
    from models.dmpnn import DMPNN
    from models.compgcn import CompGCN

    def make_batch(sizes, n_vl, n_el, rng):
        gs = []
        for n, m in sizes:
            u, v = er_edges(n, m, rng)
            e = len(u)
            uu, vv, rev = with_rev(u, v)
            g = dgl.DGLGraph.from_edges(uu, vv, n)
            el = rng.integers(0, n_el, size=e)
            g.ndata["id"] = th.arange(n)
            g.ndata["label"] = th.from_numpy(rng.integers(0, n_vl, size=n))
            g.edata["id"] = th.cat([th.arange(e), th.arange(e) + max(m for _, m in sizes)])
            g.edata["label"] = th.from_numpy(np.concatenate([el, el + n_el]))
            g.edata["is_reversed"] = th.from_numpy(rev)
            g.ndata["in_deg"] = g.in_degrees()
            g.ndata["out_deg"] = g.out_degrees()
            gs.append(g)
        return dgl.batch(gs)

    ragged = dict(p_sizes=[(3, 3), (5, 9), (4, 6), (8, 12), (2, 1), (6, 10)],
                  g_sizes=[(10, 30), (16, 50), (7, 12), (20, 64), (5, 8), (12, 40)])
    cases = {
        "uniform": dict(p_sizes=[(8, 12)] * 32, g_sizes=[(64, 256)] * 32, hid=64, layers=3, extra={}),
        "ragged": dict(hid=16, layers=2, extra={"pred_with_deg": True, "pred_with_enc": True}, **ragged),
        # CompGCN(**config) (models/compgcn.py:289-385) through the same skeleton
        "compgcn": dict(hid=16, layers=2, extra={"rep_net": "CompGCN", "rep_compgcn_comp_opt": "mult",
                                                 "rep_compgcn_edge_norm": "both", "pred_net": "MeanPredictNet"}, **ragged),
        # four optimizer steps of the count loss (train.py:624-686: MSE on pred_c, AdamW(amsgrad), grad clip)
        "train4": dict(hid=16, layers=2, extra={}, train_steps=4, **ragged),
        # per-node / per-edge matching outputs pred_v, pred_e (pred.py:118-131; train.py:627-649 consumes them)
        "matchw": dict(hid=16, layers=2, extra={"pred_return_weights": "node,edge"}, **ragged),
    }
    for tag, c in cases.items():
        rng = np.random.default_rng(31 if tag == "uniform" else 32)
        th.manual_seed(7 if tag == "uniform" else 8)
        pattern = make_batch(c["p_sizes"], 8, 8, rng)
        graph = make_batch(c["g_sizes"], 16, 16, rng)
        config = dict(max_ngv=64, max_ngvl=16, max_nge=512, max_ngel=32, max_npv=8, max_npvl=8, max_npe=24,
                      max_npel=16, base=2, hid_dim=c["hid"], share_emb_net=True, share_enc_net=True,
                      share_rep_net=True, rep_residual=True, enc_net="Multihot", emb_net="Orthogonal",
                      filter_net="ScalarFilter", rep_net="DMPNN", rep_num_graph_layers=c["layers"],
                      rep_num_pattern_layers=c["layers"], rep_dmpnn_num_mlp_layers=2, rep_dmpnn_batch_norm=False,
                      rep_act_func="relu", rep_dropout=0.0, init_neigenv=6.0, init_eeigenv=5.0,
                      pred_net="SumPredictNet", pred_hid_dim=c["hid"], pred_act_func="relu", pred_dropout=0.0,
                      node_pred=True, edge_pred=True)
        config.update(c["extra"])
        model = (CompGCN if config["rep_net"] == "CompGCN" else DMPNN)(**config)
        with th.no_grad():  # pred_fc2 is zero-initialised: make the head's output depend on its input
            for head in model.pred_net.values():
                head.pred_fc2.weight.uniform_(-0.3, 0.3)
                head.pred_fc2.bias.uniform_(-0.1, 0.1)
                if head.weight_fc2 is not None:
                    head.weight_fc2.weight.uniform_(-0.3, 0.3)
                    head.weight_fc2.bias.uniform_(-0.1, 0.1)
        d = {"config_keys": np.array(sorted(config.keys())),
             "config_vals": np.array([repr(config[k]) for k in sorted(config.keys())])}
        for k, v in model.state_dict().items():
            d["sd." + k] = v.clone()
        if c.get("train_steps"):
            counts = th.from_numpy(rng.integers(0, 20, size=len(c["p_sizes"]))).float()
            opt = th.optim.AdamW([q for q in model.parameters() if q.requires_grad], lr=1e-3, weight_decay=1e-5, amsgrad=True)
            losses = []
            for _ in range(c["train_steps"]):
                opt.zero_grad()
                o = model(pattern, graph)
                loss = th.nn.functional.mse_loss(o["pred_c"].view(-1), counts)
                loss.backward()
                th.nn.utils.clip_grad_norm_(model.parameters(), 8.0)   # train.py:679-681 (max_grad_norm)
                opt.step()
                losses.append(float(loss))
            d["train_counts"], d["train_losses"] = counts, np.array(losses)
            for k, v in model.state_dict().items():
                d["sd_after." + k] = v.clone()
            print("  train losses", losses)
        out = model(pattern, graph)
        total = out["pred_c"].sum()
        for k in ("pred_v", "pred_e"):
            if out[k] is not None:
                total = total + out[k].sum()
        total.backward()
        for t, g in (("p", pattern), ("g", graph)):
            d.update({t + "_src": g._u, t + "_dst": g._v, t + "_num_nodes": g.number_of_nodes(),
                      t + "_bnn": g.batch_num_nodes(), t + "_bne": g.batch_num_edges()})
            for k, v in g.ndata.items():
                if k in ("id", "label", "in_deg", "out_deg"):
                    d["%s_ndata.%s" % (t, k)] = v
            for k, v in g.edata.items():
                if k in ("id", "label", "is_reversed"):
                    d["%s_edata.%s" % (t, k)] = v
        if not c.get("train_steps"):
            for k, p in model.named_parameters():
                if p.grad is not None:
                    d["grad." + k] = p.grad
        for k, v in out.items():
            if v is not None:
                d["out." + k] = v
        np.savez_compressed(os.path.join(OUT, "fullmodel_%s.npz" % tag), **t2n(d))
        print("wrote fullmodel_%s.npz" % tag, "pred_c[:3] =", out["pred_c"].view(-1)[:3].tolist())


UNC_SCRIPT = r'''
import os, sys
import numpy as np, torch as th
sys.path.insert(0, %(here)r)
import ref_standin
ref_standin.install()
sys.path.insert(0, ref_standin.REF_UNC)
import dgl
from model import DualGraphConv
sys.path.insert(0, %(here)r)
from make_golden import er_edges, t2n
rng = np.random.default_rng(2024)
for tag, n, m, h, act, bn_train in (("small", 12, 30, 8, None, False), ("tanh", 40, 160, 32, "tanh", False),
                                    ("bntrain", 40, 160, 32, None, True)):
    u, v = er_edges(n, m, rng)
    u, v = np.concatenate([u, v]), np.concatenate([v, u])   # utils.py:486-487: forward + reversed copies
    th.manual_seed(len(tag) * 101 + n)
    layer = DualGraphConv(h, h, activation=(th.nn.Tanh() if act == "tanh" else None), dropout=0.0)
    with th.no_grad():
        layer.nbias.uniform_(-0.1, 0.1); layer.ebias.uniform_(-0.1, 0.1)
        for seq in (layer.nmlp, layer.emlp):
            seq[1].running_mean.uniform_(-0.2, 0.2); seq[1].running_var.uniform_(0.5, 1.5)
            seq[1].weight.uniform_(0.5, 1.5); seq[1].bias.uniform_(-0.1, 0.1)
    layer.train(bn_train)
    g = dgl.DGLGraph.from_edges(u, v, n)
    in_deg = g.in_degrees().float()
    norm = in_deg[g._v].reciprocal().unsqueeze(-1)            # utils.py:446 (norm="in")
    x = th.randn(n, h, requires_grad=True); z = th.randn(len(u), h, requires_grad=True)
    bn_before = {k: b.clone() for k, b in layer.named_buffers()}
    node_out, edge_out = layer(g, x, z, norm)
    wn, we = th.randn_like(node_out), th.randn_like(edge_out)
    ((node_out * wn).sum() + (edge_out * we).sum()).backward()
    d = {"src": u, "dst": v, "num_nodes": n, "x": x, "z": z, "norm": norm, "node_out": node_out,
         "edge_out": edge_out, "wn": wn, "we": we, "dx": x.grad, "dz": z.grad, "out_deg": g.ndata["out_deg"],
         "activation": act or "", "bn_train": bn_train}
    for k, p in layer.named_parameters():
        d["p." + k] = p
        if p.grad is not None:
            d["g." + k] = p.grad
    for k, b in bn_before.items():
        d["b." + k] = b
    for k, b in layer.named_buffers():
        d["b_after." + k] = b
    np.savez_compressed(os.path.join(%(out)r, "unc_dualconv_%%s.npz" %% tag), **t2n(d))
    print("wrote unc_dualconv_%%s.npz" %% tag)

# whole UNC DMPNN model (model.py:281-328): embeddings -> 2 x DualGraphConv -> per-relation means
from model import DMPNN
n, m, h, nrel = 30, 70, 16, 3
u, v = er_edges(n, m, rng)
rel = rng.integers(0, nrel, size=m)
u2, v2 = np.concatenate([u, v]), np.concatenate([v, u])
etype = np.concatenate([rel, rel + nrel])
th.manual_seed(77)
model = DMPNN(None, None, n, h, h, nrel * 2, 2, 0.0, False)
model.eval()
with th.no_grad():
    for layer in model.layers:
        for seq in (layer.nmlp, layer.emlp):
            seq[1].running_mean.uniform_(-0.2, 0.2); seq[1].running_var.uniform_(0.5, 1.5)
g = dgl.DGLGraph.from_edges(u2, v2, n)
norm = g.in_degrees().float()[g._v].reciprocal().unsqueeze(-1)
hid = th.arange(n)
r = th.from_numpy(etype)
hh, zz, rr = model(g, hid, r, norm)
w1, w2, w3 = th.randn_like(hh), th.randn_like(zz), th.randn_like(rr)
((hh * w1).sum() + (zz * w2).sum() + (rr * w3).sum()).backward()
d = {"src": u2, "dst": v2, "num_nodes": n, "etype": etype, "norm": norm, "h": hh, "z": zz, "r": rr,
     "w1": w1, "w2": w2, "w3": w3, "hid": h, "num_rels": nrel * 2}
for k, p in model.named_parameters():
    d["p." + k] = p
    if p.grad is not None:
        d["g." + k] = p.grad
for k, b in model.named_buffers():
    d["b." + k] = b
np.savez_compressed(os.path.join(%(out)r, "unc_model.npz"), **t2n(d))
print("wrote unc_model.npz")
# TrainModel (model.py:631-744): link-prediction head + regulariser, and the supervised head
from model import TrainModel
for tag, nlabel in (("unsup", 0), ("sup", 4)):
    th.manual_seed(91 + nlabel)
    tm = TrainModel(None, n, h, nrel, nlabel, num_hidden_layers=2, dropout=0.0, use_cuda=False, reg_param=0.01)
    tm.eval()
    if nlabel:  # the reference's regulariser reads w_relation, which only the unsupervised ctor creates
        tm.w_relation = th.nn.Parameter(th.randn(nrel, h) * 0.1)
    with th.no_grad():
        for layer in tm.model.layers:
            for seq in (layer.nmlp, layer.emlp):
                seq[1].running_mean.uniform_(-0.2, 0.2); seq[1].running_var.uniform_(0.5, 1.5)
    emb, pred = tm(g, hid, r, norm)
    d = {"src": u2, "dst": v2, "num_nodes": n, "etype": etype, "norm": norm, "hid": h, "num_rels": nrel, "nlabel": nlabel}
    if nlabel == 0:
        trip = np.stack([rng.integers(0, n, 50), rng.integers(0, nrel, 50), rng.integers(0, n, 50)], 1)
        lab = (rng.random(50) < 0.5).astype(np.float32)
        loss = tm.get_unsupervised_loss(g, emb, r, th.from_numpy(trip), th.from_numpy(lab))
        d.update({"triplets": trip, "labels": lab})
    else:
        midx = rng.choice(n, 12, replace=False)
        mlab = rng.integers(0, nlabel, 12)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            loss = tm.get_supervised_loss(g, emb, r, pred, th.from_numpy(mlab), th.from_numpy(midx), False)
        d.update({"matched_index": midx, "matched_labels": mlab, "pred": pred})
    loss.backward()
    d["loss"] = loss.detach()
    for k, p in tm.named_parameters():
        d["p." + k] = p
        if p.grad is not None:
            d["g." + k] = p.grad
    for k, b in tm.named_buffers():
        d["b." + k] = b
    np.savez_compressed(os.path.join(%(out)r, "unc_train_%%s.npz" %% tag), **t2n(d))
    print("wrote unc_train_%%s.npz" %% tag, float(loss))
# mini-batch helpers with exact integer semantics (utils.py:539-567)
import utils as unc_utils
np.random.seed(1234)
pos = np.stack([rng.integers(0, 40, 25), rng.integers(0, 3, 25), rng.integers(0, 40, 25)], 1).astype(np.int64)
neg = unc_utils.negative_sampling(pos, 40, 4)
np.random.seed(1234)                                   # the same draws, in the function's order
values = np.random.randint(40 - 1, size=100)
choices = np.random.uniform(size=100)
subg_nids = np.sort(rng.choice(200, 60, replace=False)).astype(np.int64)
ori = subg_nids[rng.integers(0, 60, 80)]
mapped = unc_utils.convert_subgraph_nids(ori, subg_nids)
np.savez_compressed(os.path.join(%(out)r, "unc_sampling.npz"), pos=pos, neg=neg, values=values, choices=choices,
                    subg_nids=subg_nids, ori=ori, mapped=mapped, num_entity=40, negative_rate=4)
print("wrote unc_sampling.npz")

# A17: graph construction and edge normalisation by the reference's own functions (utils.py:437-491)
trip = np.stack([rng.integers(0, 25, 60), rng.integers(0, 3, 60), rng.integers(0, 25, 60)], 1).astype(np.int64)
trip[5] = trip[4]                                        # a duplicate triplet; node 24 may stay isolated
gb = unc_utils.build_graph_from_triplets(25, 3, trip)
d = {"triplets": trip, "num_nodes": 25, "num_rels": 3, "src": gb._u, "dst": gb._v, "type": gb.edata["type"], "norm": gb.edata["norm"],
     "in_deg": gb.ndata["in_deg"], "out_deg": gb.ndata["out_deg"]}
for mode in ("in", "out", "both"):
    d["norm_" + mode] = unc_utils.compute_edgenorm(gb, mode)
ne, ee = unc_utils.compute_largest_eigenvalues(gb)
d["node_eigenv"], d["edge_eigenv"] = ne, ee
# a directed graph with zero in- / out-degree endpoints: the NaN / Inf branches of compute_edgenorm (utils.py:450-451)
gd = dgl.DGLGraph.from_edges(np.array([0, 0, 1, 3]), np.array([1, 2, 2, 0]), 5)
for mode in ("in", "out", "both"):
    d["dir_norm_" + mode] = unc_utils.compute_edgenorm(gd, mode)
d["dir_src"], d["dir_dst"] = gd._u, gd._v
np.savez_compressed(os.path.join(%(out)r, "unc_graph_build.npz"), **t2n(d))
print("wrote unc_graph_build.npz", tuple(gb.edata["norm"].shape))
'''


def gen_unc():
    # the UNC package's top-level module is also called ``utils`` -> separate interpreter
    code = "import torch as _th\n_th.set_num_threads(1)\n_th.use_deterministic_algorithms(True)\n" + UNC_SCRIPT % {"here": HERE, "out": OUT}
    subprocess.run([sys.executable, "-c", code], check=True)


def enumerate_subisomorphisms(pu, pv, pel, pvl, gu, gv, gel, gvl):
    """All injective label-preserving node maps under which every pattern edge (u, v, l) has an
    image edge with label l (tiny inputs only): the ``subisomorphisms`` field of a data sample."""
    import itertools
    have = set(zip(gu.tolist(), gv.tolist(), gel.tolist()))
    out = []
    for m in itertools.permutations(range(len(gvl)), len(pvl)):
        if all(gvl[m[i]] == pvl[i] for i in range(len(pvl))) and \
                all((m[u], m[v], l) in have for u, v, l in zip(pu.tolist(), pv.tolist(), pel.tolist())):
            out.append(m)
    return np.array(out, dtype=np.int64).reshape(-1, len(pvl))


def gen_subiso_weights():
    """``GraphAdjDataset.batchify(batch, return_weights="node,edge")`` (dataset.py:1604-1636) ->
    pre-padded node / edge subisomorphism weights, through the reference's own counters
    (``compute_nodeseq_subisoweights`` / ``compute_edgeseq_subisoweights``, dataset.py:54-107, and
    ``calculate_{node,edge}_weights``, dataset.py:1491-1520)."""
    import dgl
    import dataset as ref_dataset  # reference module

    def make_graph(u, v, el, vl, rev):
        n, e = len(vl), len(u)
        g = ref_dataset.Graph()
        g.add_nodes(n)
        dgl.DGLGraph.add_edges(g, u, v)
        g.ndata["id"], g.ndata["label"] = th.arange(n), th.from_numpy(np.asarray(vl, np.int64))
        g.edata["id"], g.edata["label"] = th.arange(e), th.from_numpy(np.asarray(el, np.int64))
        if rev:  # train.py:303-316
            g.edata["is_reversed"] = th.zeros((e,), dtype=th.bool)
            dgl.DGLGraph.add_edges(g, v, u, data={"id": th.arange(e) + 64, "label": g.edata["label"][:e] + 8,
                                                 "is_reversed": th.ones((e,), dtype=th.bool)})
        return g

    def sample(pu, pv, pel, pvl, gu, gv, gel, gvl, rev, name):
        pu, pv, gu, gv = (np.asarray(a, np.int64) for a in (pu, pv, gu, gv))
        pel, pvl, gel, gvl = (np.asarray(a, np.int64) for a in (pel, pvl, gel, gvl))
        sub = enumerate_subisomorphisms(pu, pv, pel, pvl, gu, gv, gel, gvl)
        return {"id": name, "pattern": make_graph(pu, pv, pel, pvl, rev), "graph": make_graph(gu, gv, gel, gvl, rev),
                "counts": len(sub), "subisomorphisms": th.from_numpy(sub)}

    def er_sample(rng, pn, pm, gn, gm, nvl, nel, rev, name):
        pu, pv = er_edges(pn, pm, rng)
        gu, gv = er_edges(gn, gm, rng)
        return sample(pu, pv, rng.integers(0, nel, pm), rng.integers(0, nvl, pn),
                      gu, gv, rng.integers(0, nel, gm), rng.integers(0, nvl, gn), rev, name)

    rng = np.random.default_rng(77)
    cases = {}
    for rev in (False, True):
        batch = [er_sample(rng, 3, 2, 7, 18, 2, 1, rev, "a-0"), er_sample(rng, 2, 1, 6, 12, 1, 2, rev, "b-1"),
                 er_sample(rng, 4, 4, 8, 30, 1, 1, rev, "c-2"), er_sample(rng, 3, 3, 5, 6, 3, 3, rev, "d-3"),
                 er_sample(rng, 3, 2, 9, 24, 2, 1, rev, "e-4")]
        # antiparallel pattern edges: with reversed copies the key (v, u) occurs twice, apart (dict overwrite, dataset.py:84)
        batch.append(sample([0, 1, 1], [1, 0, 2], [0, 0, 0], [0, 0, 0],
                            [0, 1, 1, 2, 2, 3, 3], [1, 0, 2, 1, 3, 2, 0], [0] * 7, [0] * 4, rev, "anti-5"))
        # parallel pattern edges in one run (labels 0, 1) and split in two runs (label 0 | other | label 1);
        # parallel target edges with equal and different labels
        batch.append(sample([0, 0, 1], [1, 1, 2], [0, 1, 0], [0, 0, 0],
                            [0, 0, 0, 1, 1, 2, 3], [1, 1, 1, 2, 2, 3, 0], [0, 1, 1, 0, 0, 0, 0], [0] * 4, rev, "par-6"))
        batch.append(sample([0, 1, 0], [1, 2, 1], [0, 0, 1], [0, 0, 0],
                            [0, 0, 0, 1, 1, 2, 3], [1, 1, 1, 2, 2, 3, 0], [0, 1, 1, 0, 0, 0, 0], [0] * 4, rev, "split-7"))
        cases["rev" if rev else "plain"] = batch
    for tag, batch in cases.items():
        assert any(x["counts"] == 0 for x in batch) and sum(x["counts"] > 0 for x in batch) >= 5, [x["counts"] for x in batch]
        _, pattern, graph, counts, (nw, ew) = ref_dataset.GraphAdjDataset.batchify(batch, return_weights="node,edge")
        d = {"counts": counts, "node_weights": nw, "edge_weights": ew,
             "sub_flat": th.cat([x["subisomorphisms"].reshape(-1) for x in batch]),
             "sample_ptr": th.tensor([0] + list(np.cumsum([x["subisomorphisms"].numel() for x in batch])))}
        for t, g in (("p", pattern), ("g", graph)):
            off_n = th.cat([th.zeros(1, dtype=th.long), th.cumsum(g.batch_num_nodes(), 0)])[:-1]
            eg = th.repeat_interleave(th.arange(len(batch)), g.batch_num_edges())
            d.update({t + "_src": g._u - off_n[eg], t + "_dst": g._v - off_n[eg], t + "_elabel": g.edata["label"],
                      t + "_num_nodes": g.batch_num_nodes(), t + "_num_edges": g.batch_num_edges()})
        np.savez_compressed(os.path.join(OUT, "subiso_weights_%s.npz" % tag), **t2n(d))
        print("wrote subiso_weights_%s.npz" % tag, "counts", counts.tolist(), "nw", tuple(nw.shape), "ew", tuple(ew.shape))


def gen_expand():
    """``BaseModel.expand(**kw)`` of the reference (models/basemodel.py:167-219 over
    utils/dl.py:157-191): state_dict before and after growing the vocabularies."""
    from models.dmpnn import DMPNN
    th.manual_seed(99)
    config = dict(max_ngv=16, max_ngvl=4, max_nge=40, max_ngel=6, max_npv=4, max_npvl=4, max_npe=8, max_npel=6,
                  base=2, hid_dim=8, share_emb_net=False, share_enc_net=False, share_rep_net=True, rep_residual=True,
                  enc_net="Multihot", emb_net="Orthogonal", filter_net="ScalarFilter", rep_net="DMPNN",
                  rep_num_graph_layers=1, rep_num_pattern_layers=1, rep_dmpnn_batch_norm=False,
                  pred_net="SumPredictNet", pred_hid_dim=8, pred_with_enc=True, node_pred=True, edge_pred=True)
    grow = dict(max_ngv=40, max_ngvl=9, max_nge=40, max_ngel=20, max_npv=4, max_npvl=5, max_npe=20, max_npel=3)
    model = DMPNN(**config)
    with th.no_grad():
        for head in model.pred_net.values():
            head.pred_fc2.weight.uniform_(-0.3, 0.3)
    d = {"config_keys": np.array(sorted(config.keys())), "config_vals": np.array([repr(config[k]) for k in sorted(config.keys())]),
         "grow_keys": np.array(sorted(grow.keys())), "grow_vals": np.array([grow[k] for k in sorted(grow.keys())])}
    for k, v in model.state_dict().items():
        d["sd." + k] = v.clone()
    model.expand(**dict(config, **grow))   # the callers pass the whole (new) config: create_*_net read their kinds from it
    for k, v in model.state_dict().items():
        d["sd_after." + k] = v.clone()
    d["max_after"] = np.array([getattr(model, k) for k in sorted(grow.keys())])
    np.savez_compressed(os.path.join(OUT, "expand_dmpnn.npz"), **t2n(d))
    print("wrote expand_dmpnn.npz", {k: tuple(v.shape) for k, v in model.state_dict().items() if "emb_net" in k})


def gen_rgnn():
    """RGCNLayer / RGINLayer (models/rgcn.py:14-213, models/rgin.py:16-172) and the full
    ``RGCN(**config)`` / ``RGIN(**config)`` models (GraphAdjModel skeleton, basemodel.py:619-962)."""
    import dgl
    from models.rgcn import RGCN, RGCNLayer
    from models.rgin import RGIN, RGINLayer
    rng = np.random.default_rng(404)
    layer_cases = [
        ("rgcn_full_in", RGCNLayer, dict(num_rels=5, regularizer="basis", num_bases=-1, edge_norm="in")),
        ("rgcn_basis2_both", RGCNLayer, dict(num_rels=5, regularizer="basis", num_bases=2, edge_norm="both")),
        ("rgcn_bdd_none", RGCNLayer, dict(num_rels=4, regularizer="bdd", num_bases=2, edge_norm="none", self_loop=False)),
        ("rgin_full", RGINLayer, dict(num_rels=5, regularizer="basis", num_bases=-1)),
        ("rgin_bdd", RGINLayer, dict(num_rels=6, regularizer="bdd", num_bases=4, act_func="leaky_relu")),
    ]
    for tag, cls, kw in layer_cases:
        n, m, h = 24, 90, 16
        u, v = er_edges(n, m, rng)
        th.manual_seed(seed_of("rgnn", tag))
        layer = cls(h, h, **kw)
        with th.no_grad():
            layer.bias.uniform_(-0.2, 0.2)
        g = dgl.DGLGraph.from_edges(u, v, n)
        etype = th.from_numpy(rng.integers(0, kw["num_rels"], size=m))
        x = th.randn(n, h, requires_grad=True)
        out, _ = layer(g, x, etype)
        w = th.randn_like(out)
        (out * w).sum().backward()
        d = {"src": u, "dst": v, "num_nodes": n, "etype": etype, "x": x, "w": w, "out": out, "dx": x.grad,
             "kw_keys": np.array(sorted(kw)), "kw_vals": np.array([repr(kw[k]) for k in sorted(kw)])}
        for k, p in layer.named_parameters():
            d["p." + k] = p
            d["g." + k] = p.grad
        np.savez_compressed(os.path.join(OUT, "rgnn_layer_%s.npz" % tag), **t2n(d))
        print("wrote rgnn_layer_%s.npz" % tag)

    def make_batch(sizes, n_vl, n_el, rng):
        gs = []
        for n, m in sizes:
            u, v = er_edges(n, m, rng)
            g = dgl.DGLGraph.from_edges(u, v, n)
            g.ndata["id"] = th.arange(n)
            g.ndata["label"] = th.from_numpy(rng.integers(0, n_vl, size=n))
            g.edata["id"] = th.arange(m)
            g.edata["label"] = th.from_numpy(rng.integers(0, n_el, size=m))
            gs.append(g)
        return dgl.batch(gs)

    p_sizes = [(3, 3), (5, 9), (4, 6), (8, 12), (2, 1), (6, 10)]
    g_sizes = [(10, 30), (16, 50), (7, 12), (20, 64), (5, 8), (12, 40)]
    for tag, cls, extra in (("rgcn", RGCN, {"rep_net": "RGCN", "rep_rgcn_edge_norm": "in", "pred_with_deg": True}),
                            ("rgin", RGIN, {"rep_net": "RGIN", "rep_rgin_num_bases": 2, "pred_net": "MaxPredictNet",
                                            "pred_with_enc": True, "add_node_id": True, "share_enc_net": True})):
        rng = np.random.default_rng(41)
        th.manual_seed(seed_of("rgnn-model", tag))
        pattern, graph = make_batch(p_sizes, 4, 3, rng), make_batch(g_sizes, 6, 5, rng)
        config = dict(max_ngv=20, max_ngvl=6, max_nge=64, max_ngel=5, max_npv=8, max_npvl=4, max_npe=12, max_npel=3,
                      base=2, hid_dim=16, share_emb_net=False, share_enc_net=False, share_rep_net=False,
                      rep_residual=True, enc_net="Multihot", emb_net="Orthogonal", filter_net="ScalarFilter",
                      rep_num_graph_layers=2, rep_num_pattern_layers=2, rep_act_func="relu", rep_dropout=0.0,
                      pred_net="SumPredictNet", pred_hid_dim=16, pred_act_func="relu", pred_dropout=0.0)
        config.update(extra)
        model = cls(**config)
        with th.no_grad():
            model.pred_net.pred_fc2.weight.uniform_(-0.3, 0.3)
            model.pred_net.pred_fc2.bias.uniform_(-0.1, 0.1)
        d = {"config_keys": np.array(sorted(config.keys())), "config_vals": np.array([repr(config[k]) for k in sorted(config.keys())])}
        for k, v in model.state_dict().items():
            d["sd." + k] = v.clone()
        out = model(pattern, graph)
        out["pred_c"].sum().backward()
        for t, g in (("p", pattern), ("g", graph)):
            d.update({t + "_src": g._u, t + "_dst": g._v, t + "_num_nodes": g.number_of_nodes(),
                      t + "_bnn": g.batch_num_nodes(), t + "_bne": g.batch_num_edges()})
            for k in ("id", "label"):
                d["%s_ndata.%s" % (t, k)] = g.ndata[k]
                d["%s_edata.%s" % (t, k)] = g.edata[k]
        for k, p in model.named_parameters():
            if p.grad is not None:
                d["grad." + k] = p.grad
        for k, v in out.items():
            if v is not None:
                d["out." + k] = v
        np.savez_compressed(os.path.join(OUT, "rgnn_model_%s.npz" % tag), **t2n(d))
        print("wrote rgnn_model_%s.npz" % tag, "pred_c[:3] =", out["pred_c"].view(-1)[:3].tolist())


# ----------------------------------------------------------------------------- round 2: pins for A7 / A14 / A17, the
# reference's shipped configuration, its training loop, and the dual-subisomorphism re-indexing
README_COMPLEX_ARGS = (
    "--add_rev True --hid_dim 64 --node_pred True --edge_pred False --match_weights node --enc_net Multihot --enc_base 2 "
    "--emb_net Equivariant --share_emb_net True --rep_net DMPNN --rep_num_pattern_layers 3 --rep_num_graph_layers 3 "
    "--rep_residual True --rep_dropout 0.0 --share_rep_net True --pred_net SumPredictNet --pred_hid_dim 64 --pred_dropout 0.0 "
    "--train_grad_steps 1 --lr 1e-3 --seed 0 --gpu_id -1")          # README.md:72-94 ("Complex"), sizes / dirs given per case


def reference_config(extra_args):
    """The reference's OWN argument parser (config.py: every default it ships) on the README "Complex" command line."""
    import config as refconfig
    argv, sys.argv = sys.argv, ["train.py"] + README_COMPLEX_ARGS.split() + list(extra_args)
    try:
        return refconfig.get_train_config()
    finally:
        sys.argv = argv


def _ref_graph(u, v, vl, el):
    import dgl
    import dataset as ref_dataset
    g = ref_dataset.Graph()
    g.add_nodes(len(vl))
    dgl.DGLGraph.add_edges(g, np.asarray(u, np.int64), np.asarray(v, np.int64))
    g.ndata["id"], g.ndata["label"] = th.arange(len(vl)), th.from_numpy(np.asarray(vl, np.int64))
    g.edata["id"], g.edata["label"] = th.arange(len(u)), th.from_numpy(np.asarray(el, np.int64))
    return g


def gen_preprocess():
    """A14: ``compute_largest_eigenvalues`` (utils/graph.py:40-71), ``calculate_degrees`` / ``calculate_eigenvalues``
    (train.py:330-380) and the dataset-level ``init_neigenv / init_eeigenv`` rule (train.py:1174-1186), run by the
    reference's own functions on a ``GraphAdjDataset`` after its ``add_reversed_edges`` (train.py:299-327)."""
    import dataset as ref_dataset
    import train as ref_train
    from utils.graph import compute_largest_eigenvalues
    rng = np.random.default_rng(2026)
    shapes = [(3, 3, 7, 12), (4, 6, 9, 30), (2, 1, 5, 4), (8, 12, 64, 256), (5, 20, 16, 90), (1, 0, 3, 2)]
    data = ref_dataset.GraphAdjDataset()
    raw = []
    for i, (pn, pm, gn, gm) in enumerate(shapes):
        pu, pv = er_edges(pn, pm, rng) if pm else (np.zeros(0, np.int64), np.zeros(0, np.int64))
        gu, gv = er_edges(gn, gm, rng)
        pvl, gvl = rng.integers(0, 3, pn), rng.integers(0, 3, gn)
        pel, gel = rng.integers(0, 2, pm), rng.integers(0, 2, gm)
        raw.append((pu, pv, pvl, pel, gu, gv, gvl, gel))
        data.data.append({"id": "s-%d" % i, "pattern": _ref_graph(pu, pv, pvl, pel), "graph": _ref_graph(gu, gv, gvl, gel),
                          "counts": 0, "subisomorphisms": th.zeros((0, pn), dtype=th.long)})
    d = {}
    # before add_rev, on the plain graphs (no cached degrees: the function computes them)
    for i, x in enumerate(data):
        for t in ("pattern", "graph"):
            if x[t].number_of_edges() == 0:
                continue
            ne, ee = compute_largest_eigenvalues(x[t])
            d["plain.%d.%s.node_eigenv" % (i, t[0])], d["plain.%d.%s.edge_eigenv" % (i, t[0])] = ne, ee
    max_npe, max_npel, max_nge, max_ngel = 20, 2, 256, 2
    ref_train.add_reversed_edges(data, max_npe, max_npel, max_nge, max_ngel)
    ref_train.calculate_degrees(data)
    drop = [i for i, x in enumerate(data) if x["pattern"].number_of_edges() == 0]   # .max() of an empty tensor raises there
    keep = ref_dataset.GraphAdjDataset()
    keep.data = [x for i, x in enumerate(data) if i not in drop]
    ref_train.calculate_eigenvalues(keep)
    max_neigenv = max_eeigenv = 4.0
    for x in keep:                                             # train.py:1183-1186
        max_neigenv = max(max_neigenv, x["pattern"].ndata["node_eigenv"][0].item())
        max_eeigenv = max(max_eeigenv, x["pattern"].edata["edge_eigenv"][0].item())
    d["init_neigenv"], d["init_eeigenv"] = max_neigenv, max_eeigenv
    d["num_samples"], d["dropped"] = len(shapes), np.array(drop, np.int64)
    d["max_npe"], d["max_npel"], d["max_nge"], d["max_ngel"] = max_npe, max_npel, max_nge, max_ngel
    for i, (x, r) in enumerate(zip(data, raw)):
        for t, (u, v, vl, el) in (("p", r[:4]), ("g", r[4:])):
            g = x["pattern" if t == "p" else "graph"]
            d.update({"%d.%s.src" % (i, t): u, "%d.%s.dst" % (i, t): v, "%d.%s.vlabel" % (i, t): vl, "%d.%s.elabel" % (i, t): el,
                      "%d.%s.in_deg" % (i, t): g.ndata["in_deg"], "%d.%s.out_deg" % (i, t): g.ndata["out_deg"],
                      "%d.%s.o_src" % (i, t): g._u, "%d.%s.o_dst" % (i, t): g._v})
            if i not in drop:
                d["%d.%s.node_eigenv" % (i, t)] = g.ndata["node_eigenv"]
                d["%d.%s.edge_eigenv" % (i, t)] = g.edata["edge_eigenv"]
    np.savez_compressed(os.path.join(OUT, "preprocess_eigen.npz"), **t2n(d))
    print("wrote preprocess_eigen.npz  init_neigenv %.1f init_eeigenv %.1f" % (max_neigenv, max_eeigenv))


def gen_init():
    """A7: seeded construction.  ``th.manual_seed(s); Layer(...)`` of the reference -> its ``state_dict`` (xavier with
    the activation's gain, utils/init.py:70-143, the eigenvalue re-parameterisation dmpnn.py:78-85, Linear defaults)."""
    from models.compgcn import CompGCNLayer
    from models.dmpnn import DMPLayer, DMPNN
    d = {}
    cases = [("dmp_relu", DMPLayer, dict(input_dim=12, hidden_dim=12, init_neigenv=4.0, init_eeigenv=4.0, num_mlp_layers=2, batch_norm=False, act_func="relu")),
             ("dmp_leaky", DMPLayer, dict(input_dim=16, hidden_dim=16, init_neigenv=7.0, init_eeigenv=5.0, num_mlp_layers=2, batch_norm=True, act_func="leaky_relu")),
             ("dmp_tanh_m0", DMPLayer, dict(input_dim=8, hidden_dim=8, init_neigenv=4.0, init_eeigenv=6.0, num_mlp_layers=0, batch_norm=False, act_func="tanh", bias=False)),
             ("compgcn_corr", CompGCNLayer, dict(input_dim=8, hidden_dim=8, comp_opt="corr", edge_norm="both", act_func="leaky_relu", batch_norm=True)),
             ("compgcn_sub", CompGCNLayer, dict(input_dim=6, hidden_dim=6, comp_opt="sub", edge_norm="none", act_func="relu", self_loop=False))]
    for tag, cls, kw in cases:
        th.manual_seed(seed_of("init", tag))
        layer = cls(**kw)
        d["%s.seed" % tag] = seed_of("init", tag)
        d["%s.kw_keys" % tag] = np.array(sorted(kw))
        d["%s.kw_vals" % tag] = np.array([repr(kw[k]) for k in sorted(kw)])
        for k, v in layer.state_dict().items():
            d["%s.sd.%s" % (tag, k)] = v.clone()
    # the whole model at the reference's shipped defaults (Equivariant embeddings, leaky_relu, node head with weights)
    cfg = reference_config("--max_npv 8 --max_npe 8 --max_npvl 8 --max_npel 8 --max_ngv 64 --max_nge 256 --max_ngvl 16 --max_ngel 16".split())
    import train as ref_train
    mc = ref_train.process_model_config(cfg)
    th.manual_seed(cfg["seed"])
    model = ref_train.build_model(mc, init_neigenv=6.0, init_eeigenv=5.0)
    d["model.seed"] = cfg["seed"]
    d["model.config_json"] = np.array(json_dumps(mc))
    for k, v in model.state_dict().items():
        d["model.sd.%s" % k] = v.clone()
    np.savez_compressed(os.path.join(OUT, "init_state_dicts.npz"), **t2n(d))
    print("wrote init_state_dicts.npz", sum(1 for k in d if ".sd." in k), "tensors")


def json_dumps(cfg):
    import json
    return json.dumps({k: v for k, v in cfg.items() if isinstance(v, (int, float, str, bool, type(None)))}, sort_keys=True)


def _ref_sample(rng, name, pn, pm, gn, gm, nvl, nel):
    """One (pattern, graph) sample with its exact subisomorphisms, as the reference's loader would hold it."""
    pu, pv = er_edges(pn, pm, rng)
    gu, gv = er_edges(gn, gm, rng)
    pvl, gvl = rng.integers(0, nvl, pn), rng.integers(0, nvl, gn)
    pel, gel = rng.integers(0, nel, pm), rng.integers(0, nel, gm)
    sub = enumerate_subisomorphisms(pu, pv, pel, pvl, gu, gv, gel, gvl)
    raw = dict(pu=pu, pv=pv, pvl=pvl, pel=pel, gu=gu, gv=gv, gvl=gvl, gel=gel, sub=sub)
    return {"id": name, "pattern": _ref_graph(pu, pv, pvl, pel), "graph": _ref_graph(gu, gv, gvl, gel), "counts": len(sub),
            "subisomorphisms": th.from_numpy(sub)}, raw


class _ScalarLog:
    """Stands in for the SummaryWriter of train.py: keeps every ``add_scalar(tag, value, step)`` (the per-step losses,
    learning rate and annealed coefficients train_epoch reports, train.py:688-760)."""

    def __init__(self):
        self.rows = {}

    def add_scalar(self, tag, value, step):
        self.rows.setdefault(tag, []).append((int(step), float(value)))


class _FixedLoader:
    """What train_epoch / evaluate_epoch need from a DataLoader: ``dataset``, ``len`` and batches in a fixed order."""

    def __init__(self, dataset, batches, return_weights):
        self.dataset, self.batches, self.return_weights = dataset, batches, return_weights

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        for idx in self.batches:
            yield self.dataset.batchify([self.dataset[i] for i in idx], return_weights=self.return_weights)


def gen_train_run():
    _gen_train_run("train_run_default.npz")


def gen_train_run_compgcn():
    """The same run with the other rep-net of BASELINE config 3 ("repo's DMPNN+CompGCN config"): ``--rep_net CompGCN`` at
    the shipped settings (composition ``corr``, no batch norm) except ``--rep_compgcn_edge_norm both`` -- with the
    shipped ``none`` the reference itself diverges on a dataset of this size (training MSE 7e10 after the first epoch at
    lr 1e-3, 3e9 at 1e-4, then a constant predictor), which pins nothing."""
    _gen_train_run("train_run_compgcn.npz", "--rep_net CompGCN --rep_compgcn_edge_norm both")


def _gen_train_run(out_name, extra=""):
    """BASELINE config 3 in miniature: the reference's OWN pipeline on a small synthetic dataset at its shipped settings
    (README "Complex" command: leaky_relu, Equivariant, hid 64, node head with matching weights, AdamW(amsgrad), cosine
    warm-up / restart schedule, annealed neg_pred_slp / match_loss_w / rep_reg_w) -- ``add_reversed_edges`` ->
    ``calculate_degrees`` -> ``calculate_eigenvalues`` -> ``build_model`` -> epochs of ``train_epoch`` +
    ``evaluate_epoch`` (train.py:449-1061) over fixed batch orders.  Stored: the samples, the initial and final
    ``state_dict``, per-epoch train metrics and dev MAE / predictions."""
    import dataset as ref_dataset
    import train as ref_train
    from utils.scheduler import map_scheduler_str_to_scheduler
    rng = np.random.default_rng(303)
    n_train, n_dev, bsz, epochs = 96, 32, 32, 5
    sets = {"train": ref_dataset.GraphAdjDataset(), "dev": ref_dataset.GraphAdjDataset()}
    raws = {"train": [], "dev": []}
    for split, n in (("train", n_train), ("dev", n_dev)):
        for i in range(n):
            pn = int(rng.integers(3, 6)); pm = int(rng.integers(pn - 1, min(8, pn * (pn - 1)) + 1))
            gn = int(rng.integers(8, 21)); gm = int(rng.integers(2 * gn, min(64, gn * (gn - 1)) + 1))
            x, raw = _ref_sample(rng, "%s-%d" % (split, i), pn, pm, gn, gm, 2, 2)
            sets[split].data.append(x)
            raws[split].append(raw)
    args = ("--max_npv 8 --max_npe 8 --max_npvl 2 --max_npel 2 --max_ngv 20 --max_nge 64 --max_ngvl 2 --max_ngel 2 "
            "--train_batch_size %d --eval_batch_size %d --train_epochs %d --train_log_steps 1000 %s" % (bsz, bsz, epochs, extra)).split()
    config = reference_config(args)
    random_seed = config["seed"]
    import random
    random.seed(random_seed); th.manual_seed(random_seed); np.random.seed(random_seed)
    # train.py:1111-1186, in its order
    max_ngv, max_nge, max_ngvl, max_ngel = config["max_ngv"], config["max_nge"], config["max_ngvl"], config["max_ngel"]
    max_npv, max_npe, max_npvl, max_npel = (max_ngv, max_nge, max_ngvl, max_ngel) if config["share_emb_net"] else \
        (config["max_npv"], config["max_npe"], config["max_npvl"], config["max_npel"])
    for ds in sets.values():
        for x in ds:
            x["g_len"], x["p_len"] = len(x["graph"]), len(x["pattern"])
        ref_train.add_reversed_edges(ds, max_npe, max_npel, max_nge, max_ngel)
    max_neigenv = max_eeigenv = 4.0
    for ds in sets.values():
        ref_train.calculate_degrees(ds)
        ref_train.calculate_eigenvalues(ds)
        for x in ds:
            max_neigenv = max(max_neigenv, x["pattern"].ndata["node_eigenv"][0].item())
            max_eeigenv = max(max_eeigenv, x["pattern"].edata["edge_eigenv"][0].item())
    model = ref_train.build_model(ref_train.process_model_config(config), init_neigenv=max_neigenv, init_eeigenv=max_eeigenv)
    d = {"config_json": np.array(json_dumps(config)), "model_config_json": np.array(json_dumps(ref_train.process_model_config(config))),
         "init_neigenv": max_neigenv, "init_eeigenv": max_eeigenv, "rev_max_npe": max_npe, "rev_max_npel": max_npel,
         "rev_max_nge": max_nge, "rev_max_ngel": max_ngel}
    for k, v in model.state_dict().items():
        d["sd0." + k] = v.clone()
    optimizer = th.optim.AdamW(model.parameters(), lr=config["lr"], weight_decay=config["weight_decay"], amsgrad=True)
    optimizer.zero_grad()
    # train.py:1233-1253
    num_warmup_steps = int(n_train / config["train_batch_size"] * 0.5 * min(config["train_epochs"] * 0.06, config["early_stop_rounds"]))
    num_schedule_steps = int(n_train / config["train_batch_size"] * config["train_epochs"])
    min_percent = max(1e-3, config["weight_decay"])
    if min_percent > 1e-8:
        num_schedule_steps -= num_warmup_steps
    num_cycles = max(1, num_schedule_steps / 20000)
    scheduler = map_scheduler_str_to_scheduler(config["scheduler"], num_warmup_steps=num_warmup_steps,
                                               num_schedule_steps=num_schedule_steps, num_cycles=num_cycles, min_percent=min_percent)
    scheduler.set_optimizer(optimizer)
    d.update({"num_warmup_steps": num_warmup_steps, "num_schedule_steps": num_schedule_steps, "num_cycles": num_cycles,
              "min_percent": min_percent})
    orders = np.stack([np.random.default_rng(1000 + e).permutation(n_train) for e in range(epochs)])
    d["train_orders"] = orders
    dev_batches = [list(range(i, min(i + bsz, n_dev))) for i in range(0, n_dev, bsz)]
    hist = {"train_eval": [], "train_bp": [], "dev_eval": [], "lr": []}
    device = th.device("cpu")
    log = _ScalarLog()
    for epoch in range(epochs):
        batches = [orders[epoch][i:i + bsz].tolist() for i in range(0, n_train, bsz)]
        hist["lr"].append(scheduler.get_last_lr()[0])
        ev, bp = ref_train.train_epoch(model, optimizer, scheduler, "train", _FixedLoader(sets["train"], batches, config["match_weights"]),
                                       device, config, epoch, None, log)
        dev_ev, dev_res = ref_train.evaluate_epoch(model, "dev", _FixedLoader(sets["dev"], dev_batches, config["match_weights"]),
                                                   device, config, epoch, None, None)
        hist["train_eval"].append(ev); hist["train_bp"].append(bp); hist["dev_eval"].append(dev_ev)
        print("  epoch %d  train %s %.4f  bp %.4f  dev %s %.4f" % (epoch, config["eval_metric"], ev, bp, config["eval_metric"], dev_ev))
    for k, v in hist.items():
        d["hist." + k] = np.array(v, np.float64)
    for tag, rows in log.rows.items():                          # per optimisation step
        if tag.startswith("train/"):
            d["step." + tag[6:]] = np.array([v for _, v in sorted(rows)], np.float64)
    d["dev_pred_c"] = np.array(dev_res["prediction"]["pred_c"], np.float64).reshape(-1)
    d["dev_counts"] = np.array(dev_res["data"]["counts"], np.float64).reshape(-1)
    d["dev_MAE"], d["dev_MSE"] = dev_res["error"]["MAE"], dev_res["error"]["MSE"]
    for k, v in model.state_dict().items():
        d["sd1." + k] = v.clone()
    # ---- the SAME run from initial parameters moved by one / eight units in the last place (every float parameter times
    # 1 +- 2^-23 / 2^-20, alternating signs): how far the reference's own trajectory moves under a perturbation of the size of an fp32
    # rounding -- the yardstick for a re-associated implementation's per-step deviations
    sd0 = {k[4:]: v for k, v in d.items() if k.startswith("sd0.")}
    for tag2, eps in (("twin", 2.0 ** -23), ("twin8", 2.0 ** -20)):       # one unit in the last place; eight of them (~1e-6)
        random.seed(random_seed); th.manual_seed(random_seed); np.random.seed(random_seed)
        twin = ref_train.build_model(ref_train.process_model_config(config), init_neigenv=max_neigenv, init_eeigenv=max_eeigenv)
        with th.no_grad():
            moved = {}
            for k, v in sd0.items():
                if v.is_floating_point():
                    sign = th.where(th.arange(v.numel()).view(v.shape) % 2 == 0, 1.0, -1.0)
                    moved[k] = v * (1.0 + sign * eps)
                else:
                    moved[k] = v.clone()
        twin.load_state_dict(moved)
        opt2 = th.optim.AdamW(twin.parameters(), lr=config["lr"], weight_decay=config["weight_decay"], amsgrad=True)
        opt2.zero_grad()
        sched2 = map_scheduler_str_to_scheduler(config["scheduler"], num_warmup_steps=num_warmup_steps,
                                                num_schedule_steps=num_schedule_steps, num_cycles=num_cycles, min_percent=min_percent)
        sched2.set_optimizer(opt2)
        log2 = _ScalarLog()
        twin_dev = []
        for epoch in range(epochs):
            batches = [orders[epoch][i:i + bsz].tolist() for i in range(0, n_train, bsz)]
            ref_train.train_epoch(twin, opt2, sched2, "train", _FixedLoader(sets["train"], batches, config["match_weights"]),
                                  device, config, epoch, None, log2)
            dev_ev2, _ = ref_train.evaluate_epoch(twin, "dev", _FixedLoader(sets["dev"], dev_batches, config["match_weights"]),
                                                  device, config, epoch, None, None)
            twin_dev.append(dev_ev2)
        d[tag2 + ".hist.dev_eval"] = np.array(twin_dev, np.float64)
        for tag, rows in log2.rows.items():
            if tag.startswith("train/"):
                d[tag2 + ".step." + tag[6:]] = np.array([v for _, v in sorted(rows)], np.float64)
    for split in ("train", "dev"):
        for i, r in enumerate(raws[split]):
            for k, v in r.items():
                d["%s.%d.%s" % (split, i, k)] = v
        d[split + ".n"] = len(raws[split])
    np.savez_compressed(os.path.join(OUT, out_name), **t2n(d))
    print("wrote %s  dev MAE %.4f" % (out_name, d["dev_MAE"]))


def gen_default_model():
    """``fullmodel_default``: one forward / backward of the model the README "Complex" command builds (train.py:68-87 over
    config.py's defaults: leaky_relu 1/5.5 in rep-net and head, Equivariant embeddings, hid 64, node head only, per-node
    matching weights) on a ragged batch -> all OutputDict entries and the gradients of pred_c.sum() + pred_v.sum()."""
    import dgl
    import train as ref_train
    rng = np.random.default_rng(64)
    cfg = reference_config("--max_npv 8 --max_npe 8 --max_npvl 8 --max_npel 8 --max_ngv 64 --max_nge 256 --max_ngvl 16 --max_ngel 16".split())
    mc = ref_train.process_model_config(cfg)

    def make_batch(sizes, n_vl, n_el, max_ne):
        gs = []
        for n, m in sizes:
            u, v = er_edges(n, m, rng)
            uu, vv, rev = with_rev(u, v)
            g = dgl.DGLGraph.from_edges(uu, vv, n)
            el = rng.integers(0, n_el, size=m)
            g.ndata["id"] = th.arange(n)
            g.ndata["label"] = th.from_numpy(rng.integers(0, n_vl, size=n))
            g.edata["id"] = th.cat([th.arange(m), th.arange(m) + max_ne])
            g.edata["label"] = th.from_numpy(np.concatenate([el, el + n_el]))
            g.edata["is_reversed"] = th.from_numpy(rev)
            g.ndata["in_deg"], g.ndata["out_deg"] = g.in_degrees(), g.out_degrees()
            gs.append(g)
        return dgl.batch(gs)

    for tag, p_sizes, g_sizes in (("default", [(3, 3), (5, 8), (4, 6), (8, 8), (2, 1), (6, 8), (8, 7), (3, 2)],
                                   [(10, 30), (16, 50), (7, 12), (64, 256), (5, 8), (12, 40), (40, 128), (33, 77)]),
                                  ("default_uniform", [(8, 8)] * 16, [(64, 256)] * 16)):
        th.manual_seed(11)
        pattern, graph = make_batch(p_sizes, 8, 8, 256), make_batch(g_sizes, 16, 16, 256)   # share_emb_net: pattern ids in the graph vocabulary
        model = ref_train.build_model(mc, init_neigenv=6.0, init_eeigenv=5.0)
        with th.no_grad():
            for head in model.pred_net.values():
                if head is not None:
                    head.pred_fc2.weight.uniform_(-0.3, 0.3); head.pred_fc2.bias.uniform_(-0.1, 0.1)
                    head.weight_fc2.weight.uniform_(-0.3, 0.3); head.weight_fc2.bias.uniform_(-0.1, 0.1)
        d = {"config_json": np.array(json_dumps(mc))}
        for k, v in model.state_dict().items():
            d["sd." + k] = v.clone()
        out = model(pattern, graph)
        (out["pred_c"].sum() + out["pred_v"].sum()).backward()
        for t, g in (("p", pattern), ("g", graph)):
            d.update({t + "_src": g._u, t + "_dst": g._v, t + "_num_nodes": g.number_of_nodes(),
                      t + "_bnn": g.batch_num_nodes(), t + "_bne": g.batch_num_edges()})
            for k in ("id", "label", "in_deg", "out_deg"):
                d["%s_ndata.%s" % (t, k)] = g.ndata[k]
            for k in ("id", "label", "is_reversed"):
                d["%s_edata.%s" % (t, k)] = g.edata[k]
        for k, p in model.named_parameters():
            if p.grad is not None:
                d["grad." + k] = p.grad
        for k, v in out.items():
            if v is not None:
                d["out." + k] = v
        np.savez_compressed(os.path.join(OUT, "fullmodel_%s.npz" % tag), **t2n(d))
        print("wrote fullmodel_%s.npz" % tag, "pred_c[:3] =", out["pred_c"].view(-1)[:3].tolist())


def gen_lrp():
    """(f)3: ``LRPLayer`` (models/lrp.py:18-104) and ``DMPLRPPoolLayer`` (models/dmplrp.py:20-213) forward / backward with the
    permutation matrices the reference's own ``LRPDataset`` builds (dataset.py:1751-1862: ego-net neighbour permutations of
    length <= lrp_seq_len, node / edge -> permutation-slot selection matrices, mean pooling over a node's permutations),
    and the full ``LRP(**config)`` / ``DMPLRP(**config)`` models (8-argument forward, lrp.py:222-390)."""
    import dgl
    import dataset as ref_dataset
    from models.dmplrp import DMPLRP, DMPLRPPoolLayer
    from models.lrp import LRP, LRPLayer
    rng = np.random.default_rng(808)

    def make(sizes, n_vl, n_el, max_ne):
        gs = []
        for n, m in sizes:
            u, v = er_edges(n, m, rng)
            uu, vv, rev = with_rev(u, v)
            g = dgl.DGLGraph.from_edges(uu, vv, n)
            el = rng.integers(0, n_el, size=m)
            g.ndata["id"], g.ndata["label"] = th.arange(n), th.from_numpy(rng.integers(0, n_vl, size=n))
            g.edata["id"] = th.cat([th.arange(m), th.arange(m) + max_ne])
            g.edata["label"] = th.from_numpy(np.concatenate([el, el + n_el]))
            g.edata["is_reversed"] = th.from_numpy(rev)
            g.ndata["in_deg"], g.ndata["out_deg"] = g.in_degrees(), g.out_degrees()
            gs.append(g)
        return gs

    def perm_inputs(gs):
        ego = [ref_dataset.LRPDataset.graph_to_egonet_seq(g) for g in gs]
        split = np.asarray([len(node) for seq in ego for node in seq], dtype=np.int64)
        pool = ref_dataset.LRPDataset.build_perm_pooling_matrix(split, "mean")
        n2p, e2p = ref_dataset.LRPDataset.build_batch_graph_to_perm_matrices(gs, ego)
        return pool, n2p, e2p

    def coo(d, key, m):
        m = m.coalesce()
        d[key + ".indices"], d[key + ".values"], d[key + ".shape"] = m.indices(), m.values(), np.array(m.shape)

    def graph_fields(d, t, g):
        d.update({t + "_src": g._u, t + "_dst": g._v, t + "_num_nodes": g.number_of_nodes(), t + "_bnn": g.batch_num_nodes(),
                  t + "_bne": g.batch_num_edges()})
        for k in ("id", "label", "in_deg", "out_deg"):
            d["%s_ndata.%s" % (t, k)] = g.ndata[k]
        for k in ("id", "label", "is_reversed"):
            d["%s_edata.%s" % (t, k)] = g.edata[k]

    sizes = [(5, 8), (7, 14), (4, 4), (6, 9)]
    for tag, cls, kw in (("lrp_relu", LRPLayer, dict(input_dim=8, hidden_dim=8, lrp_seq_len=4, act_func="relu", batch_norm=False, mlp=False)),
                         ("lrp_leaky_mlp", LRPLayer, dict(input_dim=12, hidden_dim=12, lrp_seq_len=4, act_func="leaky_relu", batch_norm=False, mlp=True)),
                         ("dmplrp_relu", DMPLRPPoolLayer, dict(input_dim=8, hidden_dim=8, lrp_seq_len=4, num_mlp_layers=2, batch_norm=False, act_func="relu")),
                         ("dmplrp_leaky", DMPLRPPoolLayer, dict(input_dim=16, hidden_dim=16, lrp_seq_len=4, num_mlp_layers=2, batch_norm=False,
                                                                act_func="leaky_relu", init_neigenv=6.0, init_eeigenv=5.0))):
        gs = make(sizes, 3, 2, 20)
        pool, n2p, e2p = perm_inputs(gs)
        g = dgl.batch(gs)
        th.manual_seed(seed_of("lrp", tag))
        layer = cls(**kw)
        with th.no_grad():
            for k, p in layer.named_parameters():
                if k.endswith("bias"):
                    p.uniform_(-0.1, 0.1)
        h = kw["input_dim"]
        x = th.randn(g.number_of_nodes(), h, requires_grad=True)
        z = th.randn(g.number_of_edges(), h, requires_grad=True)
        out = layer(g, x, z, pool, n2p, e2p)
        node_out, edge_out = out[0], out[1]
        wn, we = th.randn_like(node_out), th.randn_like(edge_out)
        ((node_out * wn).sum() + (edge_out * we).sum()).backward()
        d = {"x": x, "z": z, "node_out": node_out, "edge_out": edge_out, "wn": wn, "we": we, "dx": x.grad, "dz": z.grad,
             "kw_keys": np.array(sorted(kw)), "kw_vals": np.array([repr(kw[k]) for k in sorted(kw)])}
        graph_fields(d, "g", g)
        coo(d, "pool", pool); coo(d, "n2p", n2p); coo(d, "e2p", e2p)
        for k, p in layer.named_parameters():
            d["p." + k] = p
            if p.grad is not None:
                d["g." + k] = p.grad
        np.savez_compressed(os.path.join(OUT, "lrp_layer_%s.npz" % tag), **t2n(d))
        print("wrote lrp_layer_%s.npz" % tag, "perm rows", n2p.shape[0])

    p_sizes, g_sizes = [(3, 3), (4, 5), (2, 1), (4, 6)], [(6, 12), (8, 20), (5, 8), (7, 16)]
    for tag, cls in (("lrp", LRP), ("dmplrp", DMPLRP)):
        rng2 = np.random.default_rng(99)
        th.manual_seed(seed_of("lrp-model", tag))
        ps, gs = make(p_sizes, 3, 2, 20), make(g_sizes, 4, 3, 20)
        config = dict(max_ngv=8, max_ngvl=4, max_nge=40, max_ngel=6, max_npv=4, max_npvl=4, max_npe=40, max_npel=6, base=2, hid_dim=8,
                      share_emb_net=True, share_enc_net=True, share_rep_net=True, rep_residual=True, enc_net="Multihot",
                      emb_net="Orthogonal", filter_net="ScalarFilter", rep_num_graph_layers=2, rep_num_pattern_layers=2,
                      rep_act_func="relu", rep_dropout=0.0, lrp_seq_len=4, rep_dmpnn_num_mlp_layers=2, rep_dmpnn_batch_norm=False,
                      init_neigenv=5.0, init_eeigenv=4.0, pred_net="SumPredictNet", pred_hid_dim=8, pred_act_func="relu",
                      pred_dropout=0.0, node_pred=True, edge_pred=True, rep_net=tag.upper())
        model = cls(**config)
        with th.no_grad():
            for head in model.pred_net.values():
                head.pred_fc2.weight.uniform_(-0.3, 0.3); head.pred_fc2.bias.uniform_(-0.1, 0.1)
        pp, pn, pe = perm_inputs(ps)
        gp, gn, ge = perm_inputs(gs)
        pattern, graph = dgl.batch(ps), dgl.batch(gs)
        d = {"config_keys": np.array(sorted(config.keys())), "config_vals": np.array([repr(config[k]) for k in sorted(config.keys())])}
        for k, v in model.state_dict().items():
            d["sd." + k] = v.clone()
        out = model(pattern, pp, pn, pe, graph, gp, gn, ge)
        out["pred_c"].sum().backward()
        graph_fields(d, "p", pattern); graph_fields(d, "g", graph)
        for key, m in (("p_pool", pp), ("p_n2p", pn), ("p_e2p", pe), ("g_pool", gp), ("g_n2p", gn), ("g_e2p", ge)):
            coo(d, key, m)
        for k, p in model.named_parameters():
            if p.grad is not None:
                d["grad." + k] = p.grad
        for k, v in out.items():
            if v is not None:
                d["out." + k] = v
        np.savez_compressed(os.path.join(OUT, "lrp_model_%s.npz" % tag), **t2n(d))
        print("wrote lrp_model_%s.npz" % tag, "pred_c[:3] =", out["pred_c"].view(-1)[:3].tolist())


def gen_schedules():
    """Tables of the reference's step-dependent quantities: ``anneal_fn`` / ``cyclical_fn`` (utils/anneal.py, utils/cyclical.py,
    as train.py:499-600 calls them: num_init_steps = 0) and every ``lr_lambda`` of utils/scheduler.py."""
    from utils.anneal import anneal_fn
    from utils.cyclical import cyclical_fn
    from utils.scheduler import map_scheduler_str_to_scheduler, supported_schedulers
    d, specs = {}, []
    for shape in ("linear", "cosine", "none", "constant"):
        for total, cyc, a, b in ((15, 2, 1.0, 0.01), (100, 2, 0.01, 0.0), (7, 3, 0.0, 1.0), (40, 1, 2.0, -1.0)):
            steps = np.arange(0, total + 6)
            specs.append("%s|%d|%d|%r|%r" % (shape, total, cyc, a, b))
            d["anneal.%d" % (len(specs) - 1)] = np.array([anneal_fn(shape, int(s), num_init_steps=0, num_anneal_steps=total, num_cycles=cyc,
                                                                    value1=a, value2=b) for s in steps], np.float64)
            d["cyclical.%d" % (len(specs) - 1)] = np.array([cyclical_fn(shape, int(s), num_init_steps=0, num_cyclical_steps=total,
                                                                        num_cycles=cyc, value1=a, value2=b) for s in steps], np.float64)
    d["specs"] = np.array(specs)
    lrs = []
    for name in sorted(supported_schedulers):
        for warm, total, cyc, floor in ((0, 15, 1, 1e-3), (3, 40, 2, 1e-3), (5, 30, 1.5, 0.01)):
            sd = map_scheduler_str_to_scheduler(name, num_warmup_steps=warm, num_schedule_steps=total, num_cycles=cyc, min_percent=floor)
            lrs.append("%s|%d|%d|%r|%r" % (name, warm, total, cyc, floor))
            d["lr.%d" % (len(lrs) - 1)] = np.array([sd.lr_lambda(int(s)) for s in range(total + 8)], np.float64)
    d["lr_specs"] = np.array(lrs)
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **d)
    print("wrote schedules.npz", len(specs), "anneal/cyclical tables,", len(lrs), "lr tables")


def gen_dual_subiso():
    """``get_dual_subisomorphisms`` (utils/graph.py:277-316) as ``convert_to_dual_data`` drives it (train.py:417-446):
    pattern edges in eid order, graph edges in (src, dst)-sorted order, node maps -> per pattern edge the index of the
    matched graph edge in that sorted order, then mapped back through ``g_eid``."""
    from utils.graph import get_dual_subisomorphisms
    rng = np.random.default_rng(515)
    d, n_cases = {}, 0
    shapes = [(3, 3, 7, 20, 2, 2), (4, 5, 8, 40, 1, 2), (2, 1, 6, 12, 2, 1), (4, 4, 9, 30, 1, 1), (3, 2, 5, 10, 2, 3), (5, 6, 10, 60, 1, 2)]
    for pn, pm, gn, gm, nvl, nel in shapes:
        x, raw = _ref_sample(rng, "c", pn, pm, gn, gm, nvl, nel)
        if x["counts"] == 0:
            continue
        p, g = x["pattern"], x["graph"]
        p_uid, p_vid, p_eid = p.all_edges(form="all", order="eid")
        p_elabel = p.edata["label"][p_eid]
        g_uid, g_vid, g_eid = g.all_edges(form="all", order="srcdst")
        g_elabel = g.edata["label"][g_eid]
        dual = get_dual_subisomorphisms(p_uid.numpy(), p_vid.numpy(), p_elabel.numpy(), g_uid.numpy(), g_vid.numpy(),
                                        g_elabel.numpy(), x["subisomorphisms"].numpy())
        k = "%d." % n_cases
        d.update({k + "p_u": p_uid, k + "p_v": p_vid, k + "p_el": p_elabel, k + "g_u_sorted": g_uid, k + "g_v_sorted": g_vid,
                  k + "g_el_sorted": g_elabel, k + "g_eid_sorted": g_eid, k + "sub": x["subisomorphisms"], k + "dual_sorted_index": dual,
                  k + "dual_eids": g_eid.numpy()[dual], k + "g_u": raw["gu"], k + "g_v": raw["gv"], k + "g_el": raw["gel"]})
        n_cases += 1
    # repeated keys: parallel pattern edges in one run, a key split into two runs (the later replaces the earlier), on
    # multigraphs with parallel edges and loops; arbitrary injective node maps (the function does not need real matches)
    for p_u, p_v, p_el in (([0, 0, 1], [1, 1, 2], [0, 1, 0]), ([0, 1, 0, 1], [1, 2, 1, 0], [0, 0, 1, 1]), ([0, 0, 0], [1, 1, 1], [2, 0, 1])):
        gn, gm = 7, 60
        g_u, g_v, g_el = rng.integers(0, gn, gm), rng.integers(0, gn, gm), rng.integers(0, 3, gm)
        sub = np.stack([rng.permutation(gn)[:3] for _ in range(25)]).astype(np.int64)
        order = np.lexsort((np.arange(gm), g_v, g_u))
        dual = get_dual_subisomorphisms(np.array(p_u), np.array(p_v), np.array(p_el), g_u[order], g_v[order], g_el[order], sub)
        k = "%d." % n_cases
        d.update({k + "p_u": np.array(p_u), k + "p_v": np.array(p_v), k + "p_el": np.array(p_el), k + "g_u_sorted": g_u[order],
                  k + "g_v_sorted": g_v[order], k + "g_el_sorted": g_el[order], k + "g_eid_sorted": order, k + "sub": sub,
                  k + "dual_sorted_index": dual, k + "dual_eids": order[dual], k + "g_u": g_u, k + "g_v": g_v, k + "g_el": g_el})
        n_cases += 1
    d["num_cases"] = n_cases
    np.savez_compressed(os.path.join(OUT, "dual_subiso.npz"), **t2n(d))
    print("wrote dual_subiso.npz", n_cases, "cases")



GENERATORS = ["dmplayer", "dmplayer_bn", "dmpnn_rep", "compgcn", "linegraph", "addrev", "full_model", "unc", "subiso_weights", "expand", "rgnn",
              "preprocess", "init", "default_model", "train_run", "train_run_compgcn", "dual_subiso", "schedules", "lrp"]


def main():
    """``python oracle/make_golden.py [generator ...]`` -- all of them without arguments."""
    os.makedirs(OUT, exist_ok=True)
    import ref_standin
    ref_standin.import_scm()
    want = sys.argv[1:] or GENERATORS
    for name in want:
        if name not in GENERATORS:
            raise SystemExit("unknown generator %r (have: %s)" % (name, ", ".join(GENERATORS)))
    sys.argv = sys.argv[:1]
    for name in want:
        globals()["gen_" + name]()


if __name__ == "__main__":
    main()

"""CPU restatement of the reference's whole DMPNN counting model -- TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file; the product
path never does.  It restates, in plain torch on whatever device its inputs live on (CPU in practice), what
``SubgraphCountingMatching/models/basemodel.py::GraphAdjModelV2.forward`` (lines 1493-1661) computes around the rep-net
of ``oracle/dmp_oracle.py``, in the reference's own operation order (padded ``[B, L, D]`` tensors, per-sample Python
loops and all), from a ``state_dict`` with the reference's key names:

* pre-padded length masks                      utils/dl.py:113-127
* ScalarFilter gates on padded label matrices  models/filter.py:6-16, basemodel.py:1394-1423
* Multihot encodings, label embeddings         models/embed.py:103-118,197-205, basemodel.py:1425-1467
* DMPNN pattern / graph rep-nets               models/dmpnn.py:215-277  (``dmp_oracle.dmpnn_graph_rep``)
* reversed edges leave the edge masks          basemodel.py:1521-1531
* encodings / degrees joined to the reps       basemodel.py:1533-1630
* SumPredictNet node / edge heads, their blend models/pred.py:87-156,198-214, basemodel.py:1469-1491

Pinned by ``tests/test_oracle_golden_model.py`` against the reference-generated ``tests/golden/fullmodel_*.npz``
(outputs and every parameter gradient).  Scope: ``rep_net == "DMPNN"``, Multihot encoder, Sum / Mean heads."""
import torch as th

import dmp_oracle as O


def len_mask(lens, pre_pad=True):
    """utils/dl.py:113-127: ``mask[i, -l:] = 1`` (pre-padding)."""
    lens = [int(l) for l in lens]
    m = max(lens) if lens else 0
    mask = th.ones((len(lens), m), dtype=th.bool)
    for i, l in enumerate(lens):
        if pre_pad:
            mask[i, :m - l] = False
        else:
            mask[i, l:] = False
    return mask


def split_pad(feats, sizes, pre_pad=True):
    """utils/dl.py:51-81: rows of sample i in row i of a ``[B, max, D]`` tensor, zeros in front (pre-padding)."""
    sizes = [int(l) for l in sizes]
    m = max(sizes) if sizes else 0
    rows, idx = [], 0
    for l in sizes:
        pad = th.zeros((m - l,) + tuple(feats.shape[1:]), dtype=feats.dtype)
        part = feats[idx:idx + l]
        rows.append(th.cat([pad, part], 0) if pre_pad else th.cat([part, pad], 0))
        idx += l
    return th.stack(rows, 0) if rows else feats.new_zeros((0, 0) + tuple(feats.shape[1:]))


def scalar_filter_gate(p_labels, p_sizes, g_labels, g_sizes):
    """basemodel.py:1394-1423 with filter.py:6-16: a target row passes when its label occurs among the pattern's labels --
    computed on the zero-padded label matrices, so label 0 also matches the padding of a shorter pattern."""
    p = split_pad(p_labels.view(-1, 1), p_sizes).squeeze(-1)          # [B, Lp]
    g = split_pad(g_labels.view(-1, 1), g_sizes).squeeze(-1)          # [B, Lg]
    gate = ((g.unsqueeze(2) - p.unsqueeze(1)) == 0).max(dim=2)[0]     # [B, Lg]
    out = [gate[i, gate.size(1) - int(l):] for i, l in enumerate(g_sizes)]
    return th.cat(out).view(-1, 1)


def _embed(sd, prefix, enc):
    """embed.py:109-118: ``enc @ weight`` for float encodings."""
    return enc @ sd[prefix + ".weight"]


def _layer_params(sd, prefix):
    names = ("in_weight", "out_weight", "src_weight", "dst_weight", "nloop_weight", "eloop_weight", "nbias", "ebias",
             "nmlp.0.weight", "nmlp.0.bias", "nmlp.2.weight", "nmlp.2.bias", "emlp.0.weight", "emlp.0.bias", "emlp.2.weight",
             "emlp.2.bias")
    return {n: sd[prefix + "." + n] for n in names}


def _head(sd, prefix, act, p_out, p_mask, g_out, g_mask, pool, return_weights):
    """pred.py:87-156 (``PredictNet.forward``) with Sum / Mean pooling over the PADDED positions (pred.py:176-214)."""
    bsz, g_len = g_mask.size(0), g_mask.size(1)
    pl = p_mask.float().sum(1).view(bsz, 1)
    gl = g_mask.float().sum(1).view(bsz, 1)
    agg = (lambda t: t.sum(1)) if pool == "sum" else (lambda t: t.mean(1))
    lin = lambda name, x: x @ sd["%s.%s.weight" % (prefix, name)].t() + sd["%s.%s.bias" % (prefix, name)]
    p = agg(lin("p_fc", p_out))                                       # init_pattern + agg_pattern
    g_rows = lin("g_fc", g_out)                                       # init_graph
    w = None
    if return_weights:
        pe = p.unsqueeze(1).expand(bsz, g_len, -1)
        ple, plie = pl.expand(bsz, g_len).unsqueeze(-1), (1.0 / pl).expand(bsz, g_len).unsqueeze(-1)
        w = act(O.probe(prefix + ".weight_fc1", lin("weight_fc1", th.cat([pe, g_rows, g_rows - pe, g_rows * pe, ple, plie], 2))))
        w = lin("weight_fc2", th.cat([w, ple, plie], 2)).squeeze(-1)
    g = agg(g_rows)
    y = th.cat([p, g, g - p, g * p, pl, gl, 1.0 / pl, 1.0 / gl], 1)
    y = act(O.probe(prefix + ".pred_fc1", lin("pred_fc1", y)))
    return lin("pred_fc2", th.cat([y, pl, gl, 1.0 / pl, 1.0 / gl], 1)), w


def model_forward(sd, cfg, pattern, graph):
    """``GraphAdjModelV2.forward`` (basemodel.py:1493-1661) of a DMPNN model.  ``sd``: reference ``state_dict``;
    ``cfg``: its construction kwargs (the keys read below); ``pattern`` / ``graph``: dicts with ``src``, ``dst``
    (global ids, eid order), ``bnn``, ``bne`` (per-graph sizes), ``id``, ``label`` (nodes), ``eid``, ``elabel``,
    ``rev`` (edges; ``rev`` may be None).  Returns the 15-entry output dict."""
    act_rep = cfg.get("rep_act_func", "relu")
    act_pred = O.activation(cfg.get("pred_act_func", "relu"))
    node_pred, edge_pred = cfg.get("node_pred", True), cfg.get("edge_pred", True)
    weights = str(cfg.get("pred_return_weights", cfg.get("match_weights", "none")))
    out = {}
    sides = {}
    gates = (None, None)
    if cfg.get("filter_net", "None") == "ScalarFilter":
        gates = (scalar_filter_gate(pattern["label"], pattern["bnn"], graph["label"], graph["bnn"]).float(),
                 scalar_filter_gate(pattern["elabel"], pattern["bne"], graph["elabel"], graph["bne"]).float())
    for tag, gr in (("p", pattern), ("g", graph)):
        n = int(sum(int(x) for x in gr["bnn"]))
        enc = {"v": sd["%s_enc_net.v.weight" % tag][gr["id"]], "vl": sd["%s_enc_net.vl.weight" % tag][gr["label"]],
               "el": sd["%s_enc_net.el.weight" % tag][gr["elabel"]]}
        v_emb = _embed(sd, "%s_emb_net.vl" % tag, enc["vl"])
        if cfg.get("add_node_id", False):
            v_emb = v_emb + _embed(sd, "%s_emb_net.v" % tag, enc["v"])
        e_emb = _embed(sd, "%s_emb_net.el" % tag, enc["el"])
        if cfg.get("add_edge_id", False):
            e_emb = e_emb + _embed(sd, "%s_emb_net.v" % tag, enc["v"][gr["src"]]) + _embed(sd, "%s_emb_net.v" % tag, enc["v"][gr["dst"]])
        L = int(cfg["rep_num_%s_layers" % ("pattern" if tag == "p" else "graph")])
        net = "%s_rep_net.dmpnn" % tag
        child = "pattern" if (tag == "p" and not cfg.get("share_rep_net", True)) else "graph"
        layers = [_layer_params(sd, "%s.%s_dmpnn_(%d)" % (net, child, i)) for i in range(L)]
        out_deg = O.out_degrees(gr["src"], n)
        vg, eg = gates if tag == "g" else (None, None)
        v_rep, e_rep = O.dmpnn_graph_rep(layers, gr["src"], gr["dst"], gr["rev"], out_deg, v_emb, e_emb, vg, eg,
                                         residual=cfg.get("rep_residual", True), act_func=act_rep)
        v_mask = len_mask(gr["bnn"]).view(len(gr["bnn"]), -1, 1)
        e_mask = len_mask(gr["bne"]).view(len(gr["bne"]), -1, 1)
        if gr["rev"] is not None:                                          # basemodel.py:1521-1531
            e_mask = e_mask.masked_fill(split_pad(gr["rev"].view(-1, 1), gr["bne"]).bool(), False)
        sides[tag] = dict(enc=enc, v_rep=v_rep, e_rep=e_rep, v_mask=v_mask, e_mask=e_mask, n=n, gr=gr,
                          out_deg=out_deg.float().view(-1, 1), in_deg=O.in_degrees(gr["dst"], n).float().view(-1, 1))
        out["%s_v_emb" % tag], out["%s_e_emb" % tag], out["%s_v_rep" % tag], out["%s_e_rep" % tag] = v_emb, e_emb, v_rep, e_rep

    def node_output(s):                                                    # basemodel.py:1541-1576
        add = []
        if cfg.get("pred_with_enc", False):
            add += [s["enc"]["v"], s["enc"]["vl"]]
        if cfg.get("pred_with_deg", False):
            add += [s["out_deg"], s["in_deg"]]
        rows = th.cat(add + [s["v_rep"]], -1) if add else s["v_rep"]
        return split_pad(rows, s["gr"]["bnn"]).masked_fill(~s["v_mask"], 0)

    def edge_output(s):                                                    # basemodel.py:1581-1627
        u, v = s["gr"]["src"], s["gr"]["dst"]
        add = []
        if cfg.get("pred_with_enc", False):
            add += [s["enc"]["v"][u], s["enc"]["v"][v], s["enc"]["vl"][u], s["enc"]["el"], s["enc"]["vl"][v]]
        if cfg.get("pred_with_deg", False):
            add += [s["out_deg"][u], s["in_deg"][v]]
        rows = th.cat(add + [s["e_rep"]], -1) if add else s["e_rep"]
        return split_pad(rows, s["gr"]["bne"]).masked_fill(~s["e_mask"], 0)

    pool = {"SumPredictNet": "sum", "MeanPredictNet": "mean"}[cfg.get("pred_net", "SumPredictNet")]
    masks = {k: sides[k[0]][k[2:]].view(sides[k[0]][k[2:]].size(0), -1) for k in ("p_v_mask", "p_e_mask", "g_v_mask", "g_e_mask")}
    v_c = v_w = e_c = e_w = None
    if node_pred:
        v_c, v_w = _head(sd, "pred_net.v", act_pred, node_output(sides["p"]), masks["p_v_mask"], node_output(sides["g"]),
                         masks["g_v_mask"], pool, "node" in weights)
    if edge_pred:
        e_c, e_w = _head(sd, "pred_net.e", act_pred, edge_output(sides["p"]), masks["p_e_mask"], edge_output(sides["g"]),
                         masks["g_e_mask"], pool, "edge" in weights)
    if node_pred and edge_pred:                                            # basemodel.py:1477-1486
        g_v_len = masks["g_v_mask"].float().sum(1).view(-1, 1)
        g_e_len = masks["g_e_mask"].float().sum(1).view(-1, 1)
        pred_c = (g_v_len / (g_v_len + g_e_len)) * v_c + (g_e_len / (g_v_len + g_e_len)) * e_c
    else:
        pred_c = v_c if node_pred else e_c
    out.update(masks)
    out.update(pred_c=pred_c, pred_v=v_w, pred_e=e_w)
    return out

"""ORACLE -- TEST INFRASTRUCTURE, NOT THE PRODUCT.

CPU restatement of the reference's dual-message-passing path in the reference's
own operation order (gather-then-project, both reversed/non-reversed branches
computed and masked, ``index_add_`` segment sum in eid order).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file; nothing under ``dualmessagepassing_amd/`` does.

Pinning: checked against golden vectors emitted by the reference's real
``models/dmpnn.py`` / ``models/compgcn.py`` / UNC ``model.py`` imported over the
DGL stand-in (``oracle/make_golden.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.  The third-party piece, DGL (``dgl >= 0.6.0``,
README.md:19; absent here), contributes only gathers, the unordered fp32
segment sum and integer bookkeeping; the reference holds no tests or golden
vectors of its own for this path.

Every function cites the reference lines it follows (paths under
/root/reference/SubgraphCountingMatching unless prefixed UNC).
"""
import math

import torch as th
import torch.nn.functional as F

LEAKY_RELU_A = 1 / 5.5  # constants.py:10


KINKED = ("relu", "leaky_relu")  # activations whose derivative jumps at 0


def activation(name):
    # utils/act.py:457-474
    fn = _activation(name)
    if name in KINKED:
        return fn
    smooth = lambda x: fn(x)
    smooth.smooth = True          # the probe below records "no kink here" for these
    return smooth


def _activation(name):
    return {
        "none": lambda x: x,
        "relu": F.relu,
        "leaky_relu": lambda x: F.leaky_relu(x, LEAKY_RELU_A),
        "tanh": th.tanh,
        "sigmoid": th.sigmoid,
        "elu": F.elu,
        "gelu": F.gelu,
    }[name]


# Test hook: when a list, every pre-activation of a piecewise-linear activation the oracle evaluates is appended to it
# as (site, tensor): the parity tests use the values to find the few activations that lie within rounding of the kink
# (tests/util_flips.py).  Not part of the arithmetic.
PROBE = None


def probe(site, pre, act=None):
    if PROBE is not None:
        # a smooth activation has no ambiguous elements: its layer is still recorded (the tracing needs every layer)
        PROBE.append((site, th.full_like(pre[:, :1], float("inf")) if getattr(act, "smooth", False) else pre.detach()))
    return pre


def seg_sum(msg, dst, num_nodes):
    """``fn.sum`` by destination (models/dmpnn.py:92,163): zeros for nodes with no
    in-edge, fp32 adds in eid order."""
    out = th.zeros((num_nodes,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
    return out.index_add(0, dst, msg)


def out_degrees(src, num_nodes):
    return th.bincount(src, minlength=num_nodes)  # graph.out_degrees(), models/dmpnn.py:101


def in_degrees(dst, num_nodes):
    return th.bincount(dst, minlength=num_nodes)


def mlp(x, params, prefix, act, num_layers=2, bn=None, training=False):
    """nmlp / emlp: Linear (-> BN) -> act -> ... -> Linear (models/dmpnn.py:45-60).
    ``params[prefix + '.<i>.weight']`` with the Sequential indices of the reference."""
    idx = 0
    for i in range(num_layers):
        x = F.linear(x, params["%s.%d.weight" % (prefix, idx)], params["%s.%d.bias" % (prefix, idx)])
        idx += 1
        if i != num_layers - 1:
            if bn is not None:
                b = bn["%s.%d" % (prefix, idx)]
                x = F.batch_norm(x, b["running_mean"], b["running_var"], params["%s.%d.weight" % (prefix, idx)],
                                 params["%s.%d.bias" % (prefix, idx)], training, 0.1, 1e-5)
                idx += 1
            x = act(probe(prefix, x, act))
            idx += 1
    return x


def dmp_layer(params, src, dst, rev, out_deg, x, z, act_func="relu", num_mlp_layers=2, edge_norm=None,
              bn=None, training=False, unc_order=False):
    """One DMPLayer forward (models/dmpnn.py:111-166) -> (node_out, edge_out, edge_msg, node_agg).

    params: dict with in/out/src/dst/nloop/eloop ``_weight`` [in,out], nbias/ebias, nmlp.*, emlp.*
    rev:    bool [E] or None (REVFLAG absent)
    out_deg: int64 [N] = ndata["out_deg"]
    edge_norm: per-edge [E,1] scale of node messages (UNC model.py:234-235) or None
    """
    act = activation(act_func)
    n = x.size(0)
    xs, xd = x[src], x[dst]  # edges.src[NODEFEAT], edges.dst[NODEFEAT]
    # _node_message_func, models/dmpnn.py:111-127
    edge_msg = th.matmul(xd, params["dst_weight"]) - th.matmul(xs, params["src_weight"])
    node_msg = -th.matmul(z, params["in_weight"])
    if rev is not None:
        rmask = rev.view(-1, 1)
        mask = ~rmask
        rev_edge_msg = th.matmul(xs, params["dst_weight"]) - th.matmul(xd, params["src_weight"])
        rev_node_msg = th.matmul(z, params["out_weight"])
        edge_msg = edge_msg.masked_fill(rmask, 0.0) + rev_edge_msg.masked_fill(mask, 0.0)
        node_msg = node_msg.masked_fill(rmask, 0.0) + rev_node_msg.masked_fill(mask, 0.0)
    if edge_norm is not None:
        node_msg = node_msg * edge_norm  # UNC Model/DMPNN/src/model.py:234-235
    # fn.sum
    agg = seg_sum(node_msg, dst, n)
    # _node_update_func, models/dmpnn.py:129-140
    out = th.matmul(x, params["nloop_weight"]) + agg
    if params.get("nbias") is not None:
        out = out + params["nbias"]
    out = mlp(out, params, "nmlp", act, num_mlp_layers, bn, training) if num_mlp_layers > 0 else act(out)
    node_out = out
    # _edge_update_func, models/dmpnn.py:142-156
    d = out_deg[dst].unsqueeze(-1).float()
    d = (1 + d).log2()
    add = 2 * (1 + d) * th.matmul(z, (params["src_weight"] - params["dst_weight"]))
    if unc_order:  # UNC Model/DMPNN/src/model.py:254: eloop + agg + add
        out = th.matmul(z, params["eloop_weight"]) + edge_msg + add
    else:          # models/dmpnn.py:147: eloop + add + agg
        out = th.matmul(z, params["eloop_weight"]) + add + edge_msg
    if params.get("ebias") is not None:
        out = out + params["ebias"]
    out = mlp(out, params, "emlp", act, num_mlp_layers, bn, training) if num_mlp_layers > 0 else act(out)
    edge_out = out
    return node_out, edge_out, edge_msg, agg


def dmpnn_graph_rep(layers, src, dst, rev, out_deg, v_emb, e_emb, v_gate=None, e_gate=None, residual=True,
                    act_func="relu"):
    """``DMPNN.get_graph_rep`` (models/dmpnn.py:245-277); with gates None it is also
    ``get_pattern_rep`` without masks (models/dmpnn.py:215-243, basemodel.py:1515)."""
    v = v_emb * v_gate if v_gate is not None else v_emb
    e = e_emb * e_gate if e_gate is not None else e_emb
    for params in layers:
        nv, ne, _, _ = dmp_layer(params, src, dst, rev, out_deg, v, e, act_func)
        if v_gate is not None:
            nv = nv * v_gate
        if e_gate is not None:
            ne = ne * e_gate
        if residual and nv.size() == v.size() and ne.size() == e.size():
            v, e = v + nv, e + ne
        else:
            v, e = nv, ne
    return v, e


# ----------------------------------------------------------------------------- CompGCN
def compgcn_comp(h, r, comp_opt):
    # models/compgcn.py:213-224
    if comp_opt == "sub":
        return h - r
    if comp_opt == "mult":
        return h * r
    if comp_opt == "corr":
        return th.fft.irfft(th.conj(th.fft.rfft(h, dim=-1)) * th.fft.rfft(r, dim=-1), n=h.size(-1), dim=-1)
    raise NotImplementedError(comp_opt)


def compgcn_norms(src, dst, n, edge_norm, self_loop=True):
    """Degree norms of CompGCNLayer (models/compgcn.py:173-211) -> per-edge [E,1] or None."""
    if edge_norm == "none":
        return None
    in_deg = in_degrees(dst, n)
    out_deg = out_degrees(src, n)
    if self_loop:  # compgcn.py:183-184,193-194
        in_norm = (in_deg + 1).reciprocal().unsqueeze(-1)
        out_norm = (out_deg + 1).reciprocal().unsqueeze(-1)
    else:          # compgcn.py:185-186,195-196
        in_norm = in_deg.reciprocal().masked_fill_(in_deg == 0, 1.0).unsqueeze(-1)
        out_norm = out_deg.reciprocal().masked_fill_(out_deg == 0, 1.0).unsqueeze(-1)
    if edge_norm == "in":
        return in_norm[dst]
    if edge_norm == "out":
        return out_norm[src]
    if edge_norm == "both":
        return (out_norm[src] * in_norm[dst]) ** 0.5
    raise ValueError(edge_norm)


def compgcn_layer(params, src, dst, rev, x, z, comp_opt="sub", edge_norm="none", act_func="relu", bn=None,
                  training=False):
    """CompGCNLayer forward (models/compgcn.py:226-274) -> (node_out, edge_out).
    self_loop is on iff ``params`` holds ``loop_weight`` (compgcn.py:170-171)."""
    act = activation(act_func)
    n = x.size(0)
    self_loop = params.get("loop_weight") is not None
    norm = compgcn_norms(src, dst, n, edge_norm, self_loop)
    # _node_message_func, compgcn.py:226-238
    data = compgcn_comp(x[src], z, comp_opt)
    msg = th.matmul(data, params["in_weight"])
    if rev is not None:
        rmask = rev.view(-1, 1)
        mask = ~rmask
        rev_msg = th.matmul(data, params["out_weight"])
        msg = msg.masked_fill(rmask, 0.0) + rev_msg.masked_fill(mask, 0.0)
    if norm is not None:
        msg = msg * norm
    agg = seg_sum(msg, dst, n)
    # _node_update_func, compgcn.py:240-258
    if self_loop:
        out = agg + th.matmul(compgcn_comp(x, params["loop_rel"], comp_opt), params["loop_weight"])
        out = out * 0.3333333
    else:
        out = agg * 0.5
    if params.get("bias") is not None:
        out = out + params["bias"]
    if bn is not None:
        out = F.batch_norm(out, bn["running_mean"], bn["running_var"], params["bn.weight"], params["bn.bias"],
                           training, 0.1, 1e-5)
    node_out = act(out)
    # _edge_update_func, compgcn.py:260-263
    edge_out = th.matmul(z, params["rel_weight"])
    return node_out, edge_out


# ----------------------------------------------------------------------------- UNC DualGraphConv
def unc_edge_norm(src, dst, n, norm="in"):
    """``compute_edgenorm`` (UNC Model/DMPNN/src/utils.py:437-453)."""
    in_deg = in_degrees(dst, n).float()
    out_deg = out_degrees(src, n).float()
    if norm == "in":
        w = in_deg[dst].reciprocal().unsqueeze(-1)
    elif norm == "out":
        w = out_deg[src].reciprocal().unsqueeze(-1)
    else:
        w = th.pow(out_deg[src] * in_deg[dst], 0.5).reciprocal().unsqueeze(-1)
    w.masked_fill_(th.isnan(w), w.min())
    w.masked_fill_(th.isinf(w), w.min())
    return w


def unc_build_graph(num_nodes, num_rels, triplets):
    """``build_graph_from_triplets`` (UNC Model/DMPNN/src/utils.py:473-491): triplets (src, rel, dst) sorted by
    (src, dst, rel) -- numpy's structured sort with ``order=["src", "dst", "rel"]`` --, the E forward edges then their
    E reversed copies, ``type = rel | rel + num_rels``, ``norm = 1 / in_degree[dst]``.
    Returns (src, dst, type, norm [2E, 1])."""
    import numpy as np
    t = np.asarray(triplets, dtype=np.int64)
    rows = sorted(range(len(t)), key=lambda i: (t[i, 0], t[i, 2], t[i, 1]))
    t = t[rows]
    src = th.from_numpy(np.concatenate([t[:, 0], t[:, 2]]))
    dst = th.from_numpy(np.concatenate([t[:, 2], t[:, 0]]))
    typ = th.from_numpy(np.concatenate([t[:, 1], t[:, 1] + num_rels]))
    return src, dst, typ, unc_edge_norm(src, dst, num_nodes, "in")


def unc_eigen_bounds(src, dst, n):
    """UNC ``compute_largest_eigenvalues`` (utils.py:456-470): the SCM rule on structure degrees."""
    ind, outd = in_degrees(dst, n).float(), out_degrees(src, n).float()
    return (outd[src] + ind[dst]).max(), (ind[src] + outd[dst]).max()


def dual_graph_conv(params, src, dst, out_deg, x, z, edge_norm=None, rev=None, bn=None, training=False,
                    activation_name=None):
    """UNC ``DualGraphConv.forward`` (UNC Model/DMPNN/src/model.py:222-273): DMPLayer math with
    node messages scaled by ``edge_norm`` (:234-235), edge update summed as eloop + agg + add
    (:254), dropout result discarded (:245,260), MLP = Linear -> BN -> act -> Linear where the
    inner act is LeakyReLU(1/5.5) unless ``activation`` is given (:145-164), and the same
    ``activation`` applied to the outputs (:247-248,262-263).  In the UNC model the activation is
    Tanh for all but the last layer (:300-306)."""
    inner = "leaky_relu" if activation_name is None else activation_name
    nv, ne, _, _ = dmp_layer(params, src, dst, rev, out_deg, x, z, inner, 2, edge_norm, bn, training,
                             unc_order=True)
    if activation_name is not None:
        a = activation(activation_name)
        nv, ne = a(nv), a(ne)
    return nv, ne


# ----------------------------------------------------------------------------- helpers shared by tests/bench
def xavier_bound(fan_a, fan_b, act_func):
    gain = th.nn.init.calculate_gain(
        {"relu": "relu", "leaky_relu": "leaky_relu", "none": "linear", "tanh": "tanh"}[act_func], LEAKY_RELU_A)
    return math.sqrt(3.0) * gain * math.sqrt(2.0 / float(fan_a + fan_b))  # utils/init.py:71-76


def random_dmp_params(h_in, h, gen, act_func="relu", init_neigenv=4.0, init_eeigenv=4.0, num_mlp_layers=2,
                      dtype=th.float32):
    """Random parameters with the reference's distribution (models/dmpnn.py:64-85)."""
    a = xavier_bound(h_in, h, act_func)
    p = {}
    for k in ("in", "out", "src", "dst", "nloop", "eloop"):
        w = (th.rand(h_in, h, generator=gen, dtype=th.float64) * 2 - 1) * a
        w = w / (init_neigenv if k in ("in", "out", "nloop") else init_eeigenv)
        p[k + "_weight"] = w.to(dtype)
    p["nbias"] = ((th.rand(h, generator=gen, dtype=th.float64) - 0.5) * 0.1).to(dtype)
    p["ebias"] = ((th.rand(h, generator=gen, dtype=th.float64) - 0.5) * 0.1).to(dtype)
    am = xavier_bound(h, h, act_func)
    for m in ("nmlp", "emlp"):
        for i in range(num_mlp_layers):
            p["%s.%d.weight" % (m, 2 * i)] = ((th.rand(h, h, generator=gen, dtype=th.float64) * 2 - 1) * am).to(dtype)
            p["%s.%d.bias" % (m, 2 * i)] = ((th.rand(h, generator=gen, dtype=th.float64) - 0.5) * 0.1).to(dtype)
    return p


def rel_dense_weight(params, num_rels, in_dim, out_dim, regularizer, num_bases):
    """Per-type [in, out] matrices of RGCNLayer / RGINLayer (rgcn.py:98-104 basis, :112-116 bdd)."""
    w = params["weight"]
    if regularizer in ("none", "basis"):
        if "w_comp" in params and params["w_comp"] is not None:
            return th.matmul(params["w_comp"], w.view(w.size(0), in_dim * out_dim)).view(num_rels, in_dim, out_dim)
        return w
    si, so = in_dim // num_bases, out_dim // num_bases
    blocks = w.view(num_rels, num_bases, si, so)
    return th.stack([th.block_diag(*[blocks[r, b] for b in range(num_bases)]) for r in range(num_rels)])


def rel_layer(params, src, dst, etype, x, kind, num_rels, regularizer="basis", num_bases=-1, edge_norm="in",
              self_loop=True, act_func="relu", num_mlp_layers=2):
    """RGCNLayer.forward (rgcn.py:125-199) for kind == "rgcn", RGINLayer.forward (rgin.py:124-160)
    for kind == "rgin", in the reference's order: per-edge matrix (bmm), norm, fn.sum, update."""
    n, in_dim = x.shape
    out_dim = params["bias"].numel()
    if regularizer == "none" or num_bases is None or num_bases > num_rels or num_bases <= 0:
        num_bases = num_rels
    w = rel_dense_weight(params, num_rels, in_dim, out_dim, regularizer, num_bases).index_select(0, etype)
    msg = th.bmm(x[src].unsqueeze(1), w).squeeze(1)
    act = activation(act_func)
    if kind == "rgcn":
        add = 1.0 if self_loop else 0.0
        ind, outd = in_degrees(dst, n).float(), out_degrees(src, n).float()
        inn = ((1.0 / (ind + add)) if self_loop else (1.0 / ind).masked_fill(ind == 0, 0.0)).view(-1, 1)
        outn = ((1.0 / (outd + add)) if self_loop else (1.0 / outd).masked_fill(outd == 0, 0.0)).view(-1, 1)
        if edge_norm == "in":
            msg, node_norm = msg * inn[dst], inn
        elif edge_norm == "both":
            msg, node_norm = msg * (outn[src] * inn[dst]) ** 0.5, (inn * outn) ** 0.5
        else:
            node_norm = None
        out = seg_sum(msg, dst, n)
        if self_loop:
            loop = x @ params["loop_weight"]
            out = out + (loop * node_norm if node_norm is not None else loop)
        return act(out + params["bias"])
    out = seg_sum(msg, dst, n)
    if self_loop:
        out = out + x @ params["loop_weight"]
    out = out + params["bias"]
    out = mlp(out, params, "mlp", act, num_mlp_layers) if num_mlp_layers > 0 else act(out)
    return act(out)


# ----------------------------------------------------------------------------- LRP / DMPLRP (models/lrp.py, models/dmplrp.py)
def coo_mm(indices, values, shape, x):
    """``torch.sparse.mm(A, x)`` for a COO matrix given by its arrays: out[r] += v * x[c], nonzeros in storage order."""
    out = th.zeros((int(shape[0]), x.size(1)), dtype=x.dtype, device=x.device)
    return out.index_add(0, indices[0], values.to(x.dtype).unsqueeze(-1) * x[indices[1]])


def lrp_contract(perm_rows, weight, seq_len):
    """``einsum('dab,bca->dc', rows.view(-1, L*L, in), weight)`` (models/lrp.py:68-69): the L x L slots of every
    permutation against ``weight`` [in, out, L*L]."""
    return th.einsum("dab,bca->dc", perm_rows.view(-1, seq_len * seq_len, weight.size(0)), weight)


def lrp_layer(params, pool, n2p, e2p, in_deg, x, z, act_func="relu", seq_len=4):
    """``LRPLayer.forward`` (models/lrp.py:66-88).  pool / n2p / e2p: (indices, values, shape) of the pooling and the
    node / edge -> permutation-slot matrices (dataset.py:1795-1862).  Returns (node_out, edge_out = z)."""
    act = activation(act_func)
    out = coo_mm(*n2p, x) + coo_mm(*e2p, z)
    out = lrp_contract(out, params["weight"], seq_len)
    if params.get("bias") is not None:
        out = out + params["bias"]
    out = coo_mm(*pool, act(out))
    deg = in_deg.to(x.dtype).unsqueeze(1)
    factor = F.linear(act(F.linear(deg, params["degnet_0.weight"], params["degnet_0.bias"])), params["degnet_1.weight"],
                      params["degnet_1.bias"])
    out = act(out * factor)
    if "mlp.weight" in params:
        out = act(F.linear(out, params["mlp.weight"], params["mlp.bias"]))
    return out, z


def dmplrp_layer(params, src, dst, rev, out_deg, pool, n2p, e2p, x, z, act_func="relu", seq_len=4, num_mlp_layers=2):
    """``DMPLRPPoolLayer.forward`` (models/dmplrp.py:180-196): one DMPLayer, then its node and edge outputs through the
    permutation slots, the slot contraction (+ lrp_bias) and the pooling -- without the activation / degree net of LRPLayer."""
    node_out, edge_out, _, _ = dmp_layer(params, src, dst, rev, out_deg, x, z, act_func, num_mlp_layers)
    out = coo_mm(*n2p, node_out) + coo_mm(*e2p, edge_out)
    out = lrp_contract(out, params["lrp_weight"], seq_len)
    if params.get("lrp_bias") is not None:
        out = out + params["lrp_bias"]
    return coo_mm(*pool, out), edge_out

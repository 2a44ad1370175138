"""DMPLayer and the DMPNN representation network on the MI355X kernels.

Drop-in for ``SubgraphCountingMatching/models/dmpnn.py``: same constructor
arguments, parameter names / shapes (``state_dict`` compatible), the same
``forward(graph, node_feat, edge_feat) -> (node_out, edge_out)`` and the same
``create_rep_net`` / ``get_pattern_rep`` / ``get_graph_rep`` used by
``GraphAdjModelV2.forward`` (basemodel.py:1515,1519).

How one layer runs here (math identical row by row to dmpnn.py:111-156, see
SURVEY.md Appendix A; only the association of sums and products differs):

  node side   S   = seg_sum2(Z)                 [N,2H]  one HIP kernel (the scatter-add)
              T   = X W_nloop + S [W_in ; W_out] + b_n          (N-row GEMMs -> MFMA)
              out = drop(nmlp(T))
  edge side   P   = X [W_dst | W_src]           [N,2H]          (N-row GEMM)
              G   = Z [W_eloop | W_src - W_dst] [E,2H]          (E-row GEMM)
              Y   = edge_combine(G, P, b_e)     [E,H]   one HIP kernel
              out = drop(emlp(Y))

The reference multiplies every edge row by W_in AND W_out (and gathers node rows
before projecting them): 10 [E,H]x[H,H] products per layer.  Summing first and
projecting node rows first leaves 4 of them and is the same linear map.
"""
import torch as th
import torch.nn as nn

from . import ops
from ._lib import on_input_device
from .act import init_module, init_weight, map_activation_str_to_layer
from .constants import (EDGEAGG, EDGEFEAT, NODEAGG, NODEFEAT, OUTDEGREE, REVFLAG)
from .graph import leave_detached, BatchedGraph, as_batched


def dual_message_passing(graph, x, z, in_weight, out_weight, src_weight, dst_weight, nloop_weight, eloop_weight,
                         nbias, ebias, has_rev, edge_norm=None, edge_msg_out=None):
    """The linear part of one dual-message-passing layer (dmpnn.py:111-151; UNC model.py:222-257):
    returns ``(node_pre [N,H], edge_pre [E,H], node_agg [N,H])`` with

        node_pre = X W_nloop + sum_{e->v} n_e (r_e ? Z_e W_out : -Z_e W_in) + b_n
        edge_pre = Z W_eloop + 2(1+log2(1+outdeg[dst])) Z (W_src - W_dst) + edge_msg + b_e
        edge_msg = r_e ? X[src] W_dst - X[dst] W_src : X[dst] W_dst - X[src] W_src

    ``edge_norm`` [E] or [E,1]: per-edge scale n_e of the node messages (UNC only).
    ``edge_msg_out``: a dict that receives ``edge_msg`` [E,H] under EDGEAGG -- the reference leaves it in
    ``edata["edge_agg"]`` as a side effect nobody reads (dmpnn.py:126); materialised only on request.
    Two HIP kernels (seg_sum2 / edge_combine) + four GEMMs."""
    ix = graph.index()
    coef = ix.degree_coef(graph.ndata[OUTDEGREE])
    ew = None if edge_norm is None else edge_norm.reshape(-1)
    h = nloop_weight.size(1)
    if has_rev:
        s = ops.seg_sum2(z, ix, ew, -1.0, 1.0)                       # [-S_fwd | S_rev]
        agg = ops.matmul_xw(s, th.cat([in_weight, out_weight], dim=0))
    else:
        agg = ops.matmul_xw(ops.seg_sum(z, ix, ew), -in_weight)
    # one N-row GEMM for the three node-side projections: [X W_nloop | X W_dst | X W_src]
    xp = ops.matmul_xw(x, th.cat([nloop_weight, dst_weight, src_weight], dim=1))
    node_pre = xp[:, :h] + agg
    if nbias is not None:
        node_pre = node_pre + nbias
    gm = ops.matmul_xw(z, th.cat([eloop_weight, src_weight - dst_weight], dim=1))
    edge_pre = ops.edge_combine(gm, xp[:, h:], ebias, coef, ix)
    if edge_msg_out is not None:
        with th.no_grad():
            u, v = graph.all_edges(form="uv", order="eid")
            pd, ps = xp[:, h:2 * h], xp[:, 2 * h:]
            fwd = pd[v] - ps[u]
            if has_rev:
                fwd = th.where(graph.edata[REVFLAG].view(-1, 1).bool(), pd[u] - ps[v], fwd)
            edge_msg_out[EDGEAGG] = fwd
    return node_pre, edge_pre, agg


class DMPLayer(nn.Module):
    def __init__(
        self,
        input_dim,
        hidden_dim,
        init_neigenv=4.0,  # dmpnn.py:21-22: empirical value of triangles
        init_eeigenv=4.0,
        bias=True,
        num_mlp_layers=2,
        batch_norm=True,
        act_func="relu",
        dropout=0.0
    ):
        super(DMPLayer, self).__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim

        # dmpnn.py:33-38 -- [in, out] layout, used as x @ W
        self.in_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        self.out_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        self.src_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        self.dst_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        self.nloop_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        self.eloop_weight = nn.Parameter(th.empty(input_dim, hidden_dim))
        if bias:
            self.nbias = nn.Parameter(th.empty(hidden_dim))
            self.ebias = nn.Parameter(th.empty(hidden_dim))
        else:
            self.register_parameter("nbias", None)
            self.register_parameter("ebias", None)
        self.nmlp = self._make_mlp(hidden_dim, num_mlp_layers, batch_norm, act_func)
        self.emlp = self._make_mlp(hidden_dim, num_mlp_layers, batch_norm, act_func)
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)
        # edata["edge_agg"] (the [E,H] edge message the reference's UDF leaves behind, dmpnn.py:126) is written
        # only when asked for: no caller reads it, and it is one more [E,H] tensor per layer
        self.write_edge_agg = False

        # dmpnn.py:64-75
        for w in (self.in_weight, self.out_weight, self.src_weight, self.dst_weight,
                  self.nloop_weight, self.eloop_weight):
            init_weight(w, activation=act_func, init="uniform")
        for module in self.nmlp.modules():
            init_module(module, activation=act_func, init="uniform")
        for module in self.emlp.modules():
            init_module(module, activation=act_func, init="uniform")
        if bias:
            nn.init.zeros_(self.nbias)
            nn.init.zeros_(self.ebias)

        # dmpnn.py:78-85 -- re-parameterisation by the largest-eigenvalue bounds
        with th.no_grad():
            self.in_weight.data.div_(init_neigenv)
            self.out_weight.data.div_(init_neigenv)
            self.nloop_weight.data.div_(init_neigenv)
            self.src_weight.data.div_(init_eeigenv)
            self.dst_weight.data.div_(init_eeigenv)
            self.eloop_weight.data.div_(init_eeigenv)

    @staticmethod
    def _make_mlp(hidden_dim, num_mlp_layers, batch_norm, act_func):
        # dmpnn.py:45-60: Linear (-> BN) -> act between layers, plain Linear last
        mods = []
        for i in range(num_mlp_layers):
            mods.append(nn.Linear(hidden_dim, hidden_dim))
            if i != num_mlp_layers - 1:
                if batch_norm:
                    mods.append(nn.BatchNorm1d(hidden_dim))
                mods.append(map_activation_str_to_layer(act_func))
        return nn.Sequential(*mods)

    @on_input_device
    def forward(self, graph, node_feat, edge_feat):
        g = as_batched(graph)   # DGLGraph-in (dmpnn.py:158): any graph object with the DGL surface, frames shared
        # _node_init_func / _edge_init_func (dmpnn.py:96-109)
        if node_feat is not None:
            g.ndata[NODEFEAT] = node_feat
        if OUTDEGREE not in g.ndata:
            g.ndata[OUTDEGREE] = g.out_degrees()
        if edge_feat is not None:
            g.edata[EDGEFEAT] = edge_feat
        x, z = g.ndata[NODEFEAT], g.edata[EDGEFEAT]
        node_pre, edge_pre, agg = dual_message_passing(
            g, x, z, self.in_weight, self.out_weight, self.src_weight, self.dst_weight, self.nloop_weight,
            self.eloop_weight, self.nbias, self.ebias, has_rev=REVFLAG in g.edata,
            edge_msg_out=g.edata if self.write_edge_agg else None)
        g.ndata[NODEAGG] = agg
        # _node_update_func / _edge_update_func tails (dmpnn.py:135-140,151-156)
        out = ops.apply_mlp(self.nmlp, node_pre) if len(self.nmlp) > 0 else self.act(node_pre)
        node_out = self.drop(out)
        out = ops.apply_mlp(self.emlp, edge_pre) if len(self.emlp) > 0 else self.act(edge_pre)
        edge_out = self.drop(out)
        leave_detached(g.ndata, NODEFEAT, NODEAGG)
        leave_detached(g.edata, EDGEFEAT)
        return node_out, edge_out

    # ---- single-node fused path (fused.py): layer + gate + residual in one autograd node
    def fused_ok(self, graph, node_feat, edge_feat, v_gate=None, e_gate=None):
        """True when the hand-orchestrated fused path computes exactly this layer's function."""
        if not (hasattr(graph, "edata") and REVFLAG in graph.edata):
            return False
        if self.input_dim != self.hidden_dim or self.hidden_dim % 4 != 0 or self.nbias is None:
            return False
        if self.drop.p > 0.0 and self.training:
            return False
        from .fused import activation_slope
        for mlp in (self.nmlp, self.emlp):   # Linear -> ReLU | LeakyReLU (the reference's default, config.py:298-301) -> Linear
            if len(mlp) != 3 or not isinstance(mlp[0], nn.Linear) or activation_slope(mlp[1]) is None \
                    or not isinstance(mlp[2], nn.Linear) or mlp[0].bias is None or mlp[2].bias is None:
                return False
        if activation_slope(self.nmlp[1]) != activation_slope(self.emlp[1]):
            return False
        for t in (node_feat, edge_feat):
            if t is None or not t.is_cuda or t.dtype != th.float32 or t.dim() != 2 or t.size(1) != self.hidden_dim:
                return False
        for g in (v_gate, e_gate):
            if g is not None and (g.requires_grad or g.dtype != th.float32 or g.numel() != g.size(0)):
                return False
        return True

    @on_input_device
    def forward_fused(self, graph, node_feat, edge_feat, v_gate=None, e_gate=None, residual=True, folded=None, pools=None, l0=None,
                      inner=0):
        """``(node_feat + v_gate * node_out, edge_feat + e_gate * edge_out)`` (without the
        ``node_feat +`` / ``edge_feat +`` terms if ``residual`` is False) -- one layer of the
        loops in ``get_pattern_rep`` / ``get_graph_rep`` (dmpnn.py:229-241,262-275)."""
        from . import fused
        g = as_batched(graph)
        g.ndata[NODEFEAT] = node_feat
        if OUTDEGREE not in g.ndata:
            g.ndata[OUTDEGREE] = g.out_degrees()
        g.edata[EDGEFEAT] = edge_feat
        ix = g.index()
        coef = ix.degree_coef(g.ndata[OUTDEGREE])
        vg = None if v_gate is None else v_gate.reshape(-1).contiguous()
        eg = None if e_gate is None else e_gate.reshape(-1).contiguous()
        out = fused.fused_dmp_layer(ix, coef, residual, node_feat, edge_feat, vg, eg, self, folded, pools, l0, inner)
        leave_detached(g.ndata, NODEFEAT)
        leave_detached(g.edata, EDGEFEAT)
        return out

    def extra_repr(self):
        return "in=%s, out=%s" % (self.input_dim, self.hidden_dim)

    def get_output_dim(self):
        return self.hidden_dim


class DMPNNRepMixin:
    """``create_rep_net`` / ``get_pattern_rep`` / ``get_graph_rep`` of the reference's
    ``DMPNN`` (dmpnn.py:183-277); mixed into the model skeleton (``basemodel``) and into
    the stand-alone ``DMPNNRep`` below.  Expects ``hid_dim``, ``share_rep_net``,
    ``rep_residual``, ``g_rep_net`` / ``p_rep_net`` attributes."""

    rep_key = "dmpnn"

    def get_joint_rep(self, pattern, graph, p_v_emb, p_e_emb, g_v_emb, g_e_emb, v_gate=None, e_gate=None, pools=None):
        return joint_rep(self, pattern, graph, p_v_emb, p_e_emb, g_v_emb, g_e_emb, v_gate, e_gate, pools)

    def _make_layer(self, **kw):
        return DMPLayer(self.hid_dim, self.hid_dim, init_neigenv=kw.get("init_neigenv", 4.0), init_eeigenv=kw.get("init_eeigenv", 4.0),
                        num_mlp_layers=kw.get("rep_dmpnn_num_mlp_layers", 2), batch_norm=kw.get("rep_dmpnn_batch_norm", False),
                        act_func=kw.get("rep_act_func", "relu"), dropout=kw.get("rep_dropout", 0.0))

    def create_rep_net(self, type, **kw):
        """``ModuleDict({rep_key: ModuleList})`` with children ``"<type>_<rep_key>_(<i>)"`` (the reference's ``state_dict``
        names, dmpnn.py:183-213); the pattern side IS the graph side under ``share_rep_net``."""
        if type not in ("graph", "pattern"):
            raise ValueError(type)
        if type == "pattern" and self.share_rep_net:
            return self.g_rep_net
        layers = nn.ModuleList()
        for i in range(kw.get("rep_num_%s_layers" % type, 1)):
            layers.add_module("%s_%s_(%d)" % (type, self.rep_key, i), self._make_layer(**kw))
        return nn.ModuleDict({self.rep_key: layers})

    def _rep_loop(self, net, graph, v, e, v_gate=None, e_gate=None, v_zero=None, e_zero=None):
        """The layer loop shared by the pattern and the graph side (dmpnn.py:229-241,262-275): after every layer the
        outputs are zero-masked (pattern side) or gated (graph side) and, with ``rep_residual`` and matching shapes, added
        to the layer's inputs.  Eligible layers take the fused single-node path (layer + gate + residual in one)."""
        for layer in net[self.rep_key]:
            if v_zero is None and e_zero is None and getattr(self, "use_fused", True) and hasattr(layer, "fused_ok") \
                    and layer.fused_ok(graph, v, e, v_gate, e_gate):
                v, e = layer.forward_fused(graph, v, e, v_gate, e_gate, self.rep_residual)
                continue
            nv, ne = layer(graph, v, e)
            if v_zero is not None:
                nv = nv.masked_fill(v_zero, 0.0)
            if e_zero is not None:
                ne = ne.masked_fill(e_zero, 0.0)
            if v_gate is not None:
                nv = nv * v_gate
            if e_gate is not None:
                ne = ne * e_gate
            if self.rep_residual and v.size() == nv.size() and e.size() == ne.size():
                v, e = v + nv, e + ne
            else:
                v, e = nv, ne
        return v, e

    def get_pattern_rep(self, pattern, p_v_emb, p_e_emb, v_mask=None, e_mask=None):
        # dmpnn.py:215-243: masks zero the rows before the first and after every layer
        v_zero = None if v_mask is None else ~v_mask
        e_zero = None if e_mask is None else ~e_mask
        v = p_v_emb if v_zero is None else p_v_emb.masked_fill(v_zero, 0.0)
        e = p_e_emb if e_zero is None else p_e_emb.masked_fill(e_zero, 0.0)
        return self._rep_loop(self.p_rep_net, pattern, v, e, v_zero=v_zero, e_zero=e_zero)

    def get_graph_rep(self, graph, g_v_emb, g_e_emb, v_mask=None, e_mask=None, v_gate=None, e_gate=None):
        # dmpnn.py:245-277: a mask acts as (or multiplies into) the gate
        if v_mask is not None:
            v_gate = v_mask.float() if v_gate is None else v_mask.float() * v_gate
        if e_mask is not None:
            e_gate = e_mask.float() if e_gate is None else e_mask.float() * e_gate
        v = g_v_emb if v_gate is None else g_v_emb * v_gate
        e = g_e_emb if e_gate is None else g_e_emb * e_gate
        return self._rep_loop(self.g_rep_net, graph, v, e, v_gate=v_gate, e_gate=e_gate)


def _union_of(pattern, graph):
    from .collate import union_graphs
    u = getattr(pattern, "_union_cache", None)
    if u is None or u[0] is not graph:
        u = (graph, union_graphs(pattern, graph))
        pattern._union_cache = u
    return u[1]


def prepare_joint(pattern, graph, hidden_dim=128, backward=True, class_tiles=True):
    """Everything of the joint pattern + target pass that depends on the batch's STRUCTURE only -- the union graph, its
    CSR index, degrees, coefficient vector, per-edge selectors, degree-class tiles and (``backward``) the incidence CSR --
    built ahead of the forward pass.  A data-parallel step calls this between launching the gradient all-reduce of the
    previous batch and waiting for it: none of it reads a parameter, so it overlaps the collective.
    ``class_tiles=False``: the degree-class tiles are left to the forward pass (under a 0 / 1 edge gate it builds them over
    the kept edges, ``fused.live_tiles``: the list over all edges would not be used)."""
    from . import fused
    pattern, graph = as_batched(pattern), as_batched(graph)
    if (REVFLAG in pattern.edata) != (REVFLAG in graph.edata):
        return None
    union = _union_of(pattern, graph)
    ix = union.index()
    if OUTDEGREE not in union.ndata:
        union.ndata[OUTDEGREE] = union.out_degrees()
    coef = ix.degree_coef(union.ndata[OUTDEGREE])
    if fused.mfma_ok(ix, hidden_dim):
        ix.edge_select(coef)
    if class_tiles and fused.typed_ok(ix, hidden_dim):
        ix.class_tiles(coef)
    if backward and not (ops.USE_GRAPH_SEG_SUM and ix.node_tiling is not None and hidden_dim in (64, 128)):
        ix.incidence()                     # the one-pass endpoint sums (ops.endpoint_sums) need no incidence CSR
    return union


def _joint_gates(v_gate, e_gate, np_, ep_, dtype, device):
    """``(vg, eg)``: the union's gates -- ones for the pattern rows, the target's gates for its rows -- with the marks the
    kernels read (0 / 1, wiped input rows).  Memoised on the target's gate tensors (``prefetch_joint_indexes`` makes them
    ahead of the pass, on the side stream)."""
    owner = e_gate if e_gate is not None else v_gate
    if owner is None:
        return None, None
    key = (None if v_gate is None else (v_gate.data_ptr(), v_gate._version), None if e_gate is None else (e_gate.data_ptr(), e_gate._version),
           np_, ep_, dtype)
    hit = getattr(owner, "_dmp_joint_gates", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    vg = eg = None
    if v_gate is not None and e_gate is not None:            # ones for the pattern rows, the gates for the target rows
        from .collate import concat_pairs
        vg, eg = concat_pairs([((np_, 1.0), v_gate.reshape(-1).to(dtype), 0), ((ep_, 1.0), e_gate.reshape(-1).to(dtype), 0)])
    elif v_gate is not None:
        vg = th.cat([th.ones(np_, dtype=dtype, device=device), v_gate.reshape(-1)])
    elif e_gate is not None:
        eg = th.cat([th.ones(ep_, dtype=dtype, device=device), e_gate.reshape(-1)])
    for made, src in ((vg, v_gate), (eg, e_gate)):           # ones for the pattern rows + a 0 / 1 gate: still 0 / 1
        if made is not None and getattr(src, "_dmp_binary", False):
            made._dmp_binary = True
    if eg is not None:
        eg._dmp_zero_rows = True         # e's target rows are multiplied by this gate (_gate_concat / the packed codes)
    if vg is not None:
        vg._dmp_zero_rows = True         # ... and v's target rows by this one: a gated-out node is a zero row in every layer
        vg._dmp_ones_prefix = np_        # (its first np_ entries are ones by construction: the pattern's nodes are all kept)
    if eg is not None and getattr(e_gate, "_dmp_dense_gate", False):
        eg._dmp_dense_gate = True        # the kept edges of a compacted batch (collate.compact_gated_edges): ones but for the padding
    try:
        owner._dmp_joint_gates = (key, vg, eg)
    except Exception:
        pass
    return vg, eg


import os as _os
PREFETCH_UNGATED = _os.environ.get("DMP_DEV_PREFETCH_UNGATED", "1") == "1"   # the side-stream index branch also without gates


def prefetch_joint_indexes(model, pattern, graph, v_gate, e_gate, pool_kinds=(), skip_rev=True):
    """Everything ``joint_rep`` and its layers derive from the batch's STRUCTURE and its two filter gates alone, issued on the
    side stream (``side.fork``) as soon as the gates exist: the union graph and its CSR index, the degree coefficients, the
    edge selectors; the union's gates, their row masks, the kept nodes' list / tiles / selectors, the kept edge rows of
    the first layer; the kept edges' class tiles, their CSR by destination, the pooling indexes, the kept incidence CSR of
    the backward.  None of it reads a parameter or a feature row, so it runs beside the embedding kernels and the first
    layer's node side; every product lands in the memo its consumer looks it up in, and the consumer ``side.wait``s for its
    stage first.  Nothing here decides anything: a product that turns out unused costs a few microseconds of the side
    stream; one that is missing is built by its consumer on the main stream as before."""
    from . import fused, side
    if not side.USE_SIDE_STREAM or not getattr(model, "use_fused", True) or not hasattr(model, "g_rep_net"):
        return
    if model.p_rep_net is not model.g_rep_net or (v_gate is None) != (e_gate is None) or (e_gate is not None and not e_gate.is_cuda):
        return
    layers = list(model.g_rep_net[model.rep_key])
    if not layers or not all(hasattr(l, "fused_ok") for l in layers):
        return
    pattern, graph = as_batched(pattern), as_batched(graph)
    if (REVFLAG in pattern.edata) != (REVFLAG in graph.edata):
        return
    H = layers[0].hidden_dim
    np_, ep_ = pattern.number_of_nodes(), pattern.number_of_edges()
    side.join()              # (a step that never joined -- an exception on the way: nothing of it may stay dangling)
    if e_gate is None:
        # no filter net (the all-rows step): what is left is the structure itself -- the union's CSR, the coefficients and
        # selectors, the degree-class tiles of the typed kernels, the pooling indexes
        if not PREFETCH_UNGATED:
            return
        with side.fork() as forked:
            if not forked:
                return
            union = _union_of(pattern, graph)
            ix = union.index()
            if OUTDEGREE not in union.ndata:
                union.ndata[OUTDEGREE] = union.out_degrees()
            coef = ix.degree_coef(union.ndata[OUTDEGREE])
            if fused.mfma_ok(ix, H):
                ix.edge_select(coef)
            side.mark("index")
            side.mark("nodes")
            side.mark("erows")
            if fused.typed_ok(ix, H):
                ix.class_tiles(coef)
            side.mark("tiles")
            if pool_kinds:
                from .basemodel import _pool_indexes_union
                _pool_indexes_union(pattern, graph, pool_kinds, skip_rev)
            side.mark("pools")
        return
    with side.fork() as forked:
        if not forked:
            return
        union = _union_of(pattern, graph)
        ix = union.index()
        if OUTDEGREE not in union.ndata:
            union.ndata[OUTDEGREE] = union.out_degrees()
        coef = ix.degree_coef(union.ndata[OUTDEGREE])
        if fused.mfma_ok(ix, H):
            ix.edge_select(coef)
        side.mark("index")
        vg, eg = _joint_gates(v_gate, e_gate, np_, ep_, th.float32, e_gate.device)
        N, E = ix.num_nodes, ix.num_edges
        typed = fused.typed_ok(ix, H)
        # both gates' row masks, the three kept-row lists below and the kept nodes' selectors: three launches instead of ten
        # (fused.gate_bundle plants each product in the memo its own function reads; the conditions are those tested below)
        want_nodes = bool(N >= 4096 and fused.onepanel_ok(H) and fused.USE_NODE_ROWS and fused.USE_PLAIN_ATB and fused.zero_rows_gate(vg) and typed and E > 0)
        want_l0 = bool(typed and fused.SKIP_DEAD_ROWS and fused.USE_ROW_MASKS and fused.USE_L0_ROW_LISTS and len(layers) > 1 and ep_ % 32 == 0
                       and E - ep_ >= fused.L0_LIST_MIN_ROWS and th.is_grad_enabled())
        want_asc = bool(typed and fused.zero_rows_gate(eg) and fused.USE_PLAIN_ATB and getattr(eg, "_dmp_binary", False)
                        and (fused.PLAIN_ROWS_ASCENDING or (fused.PLAIN_ATB_ASCENDING and th.is_grad_enabled())))
        fused.gate_bundle(ix, vg, eg, want_nodes, (ep_, E) if want_l0 else None, want_asc)
        nd = None
        if N >= 4096 and fused.onepanel_ok(H):
            nd = fused.node_rows(ix, vg, H)
        side.mark("nodes")
        if typed and fused.SKIP_DEAD_ROWS and fused.USE_ROW_MASKS:
            mask = fused.gate_row_mask(eg)
            if (mask is not None and fused.USE_L0_ROW_LISTS and len(layers) > 1 and ep_ % 32 == 0 and E - ep_ >= fused.L0_LIST_MIN_ROWS
                    and th.is_grad_enabled()):
                fused.kept_rows(mask, ep_, E)          # the first layer's target rows (fused.l0_edge_fwd / l0_bwd_w)
        side.mark("erows")
        if typed and fused.zero_rows_gate(eg):
            fused.live_tiles(ix, coef, eg)
            if fused.PLAIN_ROWS_ASCENDING or (fused.PLAIN_ATB_ASCENDING and th.is_grad_enabled()):
                fused.ascending_tiles(eg)                  # the plain-panel launches' row order
            side.mark("tiles")
            fused.keep_in_csr(ix, eg)
            side.mark("keepcsr")
        if pool_kinds:
            from .basemodel import _pool_indexes_union
            built = dict(zip(pool_kinds, _pool_indexes_union(pattern, graph, pool_kinds, skip_rev)))
            if built.get("edge") is not None and typed and fused.zero_rows_gate(eg) and th.is_grad_enabled():
                fused.keep_pool_csr(built["edge"], eg)     # the pooled passes of the last layer over the kept edges
                if getattr(built["edge"], "rows_in_order", False):
                    fused.pool_weight_sums(built["edge"], eg)     # ... and the gate's per-graph sums (the pooled second Linear's bias term)
        side.mark("pools")
        if (nd is not None and typed and fused.USE_MASKED_SUMS and fused.USE_KEPT_INCIDENCE and fused.zero_rows_gate(eg) and H % 4 == 0
                and fused.gate_row_mask(eg) is not None and th.is_grad_enabled()):
            nd.kept_incidence(ix, eg)
        side.mark("incidence")


def joint_rep(model, pattern, graph, p_v_emb, p_e_emb, g_v_emb, g_e_emb, v_gate=None, e_gate=None, pools=None):
    """``get_pattern_rep`` + ``get_graph_rep`` (dmpnn.py:215-277) in ONE pass over the union of
    the two batched graphs, when the rep-net is shared (``share_rep_net``, dmpnn.py:186-188) and
    every layer is eligible for the fused path.  Pattern rows get gate 1 (the pattern side has no
    gate, basemodel.py:1515).  Returns ``(p_v_rep, p_e_rep, g_v_rep, g_e_rep)`` or ``None`` if not
    applicable (callers then run the two loops separately)."""
    from . import fused
    from .collate import union_graphs
    if not getattr(model, "use_fused", True) or model.p_rep_net is not model.g_rep_net:
        return None
    layers = list(model.g_rep_net[model.rep_key])
    if not layers or not all(hasattr(l, "fused_ok") for l in layers):
        return None
    pattern, graph = as_batched(pattern), as_batched(graph)
    if (REVFLAG in pattern.edata) != (REVFLAG in graph.edata):
        return None
    from .embed import materialize
    p_v_emb, p_e_emb = materialize(p_v_emb), materialize(p_e_emb)       # the pattern side is small: plain tensors
    np_, ep_ = pattern.number_of_nodes(), pattern.number_of_edges()
    union = _union_of(pattern, graph)
    l0 = _layer0_codes(union, layers, p_v_emb, p_e_emb, g_v_emb, g_e_emb, v_gate, e_gate, np_)
    v = None
    if l0 is not None:        # the first layer works on the label codes: its [E, H] input rows are only its residual term
        from . import fused
        # ... which the layer reads over the kept edges' / nodes' tiles only: the rows under a zero of a 0 / 1 gate are not even stored
        # (the degree coefficients the test below asks about: memoised, the layers ask for the same -- made here when the
        # prefetch did not, so that a step without the side stream takes the same launches as one with it)
        if OUTDEGREE not in union.ndata:
            union.ndata[OUTDEGREE] = union.out_degrees()
        union.index().degree_coef(union.ndata[OUTDEGREE])
        live_e, live_v = fused.l0_dead_inputs(union.index(), layers[0].hidden_dim, *_joint_gates(v_gate, e_gate, np_, ep_, p_e_emb.dtype, p_e_emb.device),
                                              l0)
        l0.z_from_codes = fused.l0_z_from_codes(union.index(), layers[0].hidden_dim, _joint_gates(v_gate, e_gate, np_, ep_, p_e_emb.dtype, p_e_emb.device)[1],
                                                l0, live_e, model.rep_residual)
        with th.no_grad():
            if l0.z_from_codes:     # ... or not stored at all: the layer's second Linear forms the kept rows from the codes in registers
                l0.z_head = p_e_emb.detach()          # (the pattern's rows: few, embedded by their own table)
                e = fused.dead_rows_buffer((ep_ + g_e_emb._dmp_src[0].size(0), layers[0].hidden_dim), p_e_emb.device)
            else:
                e = _GateConcat.apply(p_e_emb, None, e_gate, g_e_emb._dmp_src[0], g_e_emb._dmp_src[1], live_e)
            if l0.venc is not None:
                v = _GateConcat.apply(p_v_emb, None, v_gate, g_v_emb._dmp_src[0], g_v_emb._dmp_src[1], live_v)
    else:
        e = _gate_concat(p_e_emb, g_e_emb, e_gate)
    if v is None:
        v = _gate_concat(p_v_emb, g_v_emb, v_gate)   # [pattern rows | gate * target rows] in one pass
    vg, eg = _joint_gates(v_gate, e_gate, np_, ep_, v.dtype, v.device)
    from . import side
    if not all(l.fused_ok(union, v, e, vg, eg) for l in layers):     # e.g. dropout in training: the callers run the two loops
        side.join()
        return None
    from . import fused
    folded = fused.fold_layers(layers)                       # the parameter algebra of all layers: one launch
    side.wait("index")                                       # (prefetch_joint_indexes: the union's CSR, coefficients, selectors)
    sums = (None, None)
    lazy_e = None
    for i, (layer, fw) in enumerate(zip(layers, folded)):
        if i == len(layers) - 1:
            side.wait("pools")
        if pools is not None and i == len(layers) - 1:
            # the last layer also pools its outputs per graph (``pools``: PoolIndex over the union's node / edge rows): a
            # gradient that comes back only through the edge sums never becomes an [E, H] tensor (fused._FusedDMPLayer) --
            # and the edge rows themselves are only formed if somebody reads them (``embed.DeferredRows``): their sums come
            # from two pooled passes.  A reader gets the rows, differentiable, from the layer's ordinary form.
            if pools[1] is not None and getattr(model, "lazy_edge_rep", True) and th.is_grad_enabled():
                v_in, e_in = v, e
                v, _, vs, es = layer.forward_fused(union, v, e, vg, eg, model.rep_residual, fw, tuple(pools[:2]) + (False,),
                                                   inner=2 if i > 0 else 0)
                lazy_e = _LazyEdgeRows(lambda: layer.forward_fused(union, v_in, e_in, vg, eg, model.rep_residual, fw)[1],
                                       ep_, (e_in.size(0), layer.hidden_dim), e_in.dtype, e_in.device)
            else:
                v, e, vs, es = layer.forward_fused(union, v, e, vg, eg, model.rep_residual, fw, pools, inner=2 if i > 0 else 0)
            sums = (vs, es)
        else:
            # (inner: the next layer of this loop, under the same gates, is the only reader of this layer's edge rows)
            v, e = layer.forward_fused(union, v, e, vg, eg, model.rep_residual, fw, None, l0 if i == 0 else None,
                                       inner=(1 if i < len(layers) - 1 else 0) | (2 if i > 0 else 0))
    side.join()              # (the kept incidence CSR of the backward: long done; nothing of the side stream outlives the pass)
    p_v, g_v = _SplitRows.apply(v, np_)
    if lazy_e is not None:
        return p_v, lazy_e.part(0), g_v, lazy_e.part(1), v, lazy_e.whole(), sums
    p_e, g_e = _SplitRows.apply(e, ep_)
    return p_v, p_e, g_v, g_e, v, e, sums


class _LazyEdgeRows:
    """The union's edge rows of the last layer, computed on first access (``fn``), and their pattern / target parts."""

    def __init__(self, fn, split, shape, dtype, device):
        self._fn, self._split, self._shape, self._dtype, self._device = fn, split, shape, dtype, device
        self._rows = self._parts = None

    def _get(self):
        if self._rows is None:
            self._rows = self._fn()
            self._parts = _SplitRows.apply(self._rows, self._split)
            self._fn = None
        return self._rows, self._parts

    def whole(self):
        from .embed import DeferredRows
        return DeferredRows(lambda: self._get()[0], self._shape, self._dtype, self._device)

    def part(self, i):
        from .embed import DeferredRows
        rows = self._split if i == 0 else self._shape[0] - self._split
        return DeferredRows(lambda: self._get()[1][i], (rows, self._shape[1]), self._dtype, self._device)


def _layer0_codes(union, layers, p_v_emb, p_e_emb, g_v_emb, g_e_emb, v_gate, e_gate, p_nodes):
    """``fused.Layer0Codes`` when the first layer can run on the label codes of its edge rows (``fused.l0_ok``: both edge
    embeddings are ``codes @ table`` with tables of the same shape -- one shared table, or the pattern's stacked over the
    target's) and is not also the pooled last layer; with the node rows' codes as well when they qualify.  Else None."""
    from . import fused
    if len(layers) < 2 or not th.is_grad_enabled():
        return None
    ps, gs = getattr(p_e_emb, "_dmp_src", None), getattr(g_e_emb, "_dmp_src", None)
    if ps is None or gs is None or not gs[1].requires_grad or not ps[1].requires_grad:
        return None
    H = layers[0].hidden_dim
    if layers[0].input_dim != H or not fused.l0_ok(union.index(), H, ps[0], gs[0], ps[1], gs[1]):
        return None
    if e_gate is not None and (e_gate.requires_grad or e_gate.numel() != gs[0].size(0)):
        return None
    shared = ps[1] is gs[1]
    pv, gv = getattr(p_v_emb, "_dmp_src", None), getattr(g_v_emb, "_dmp_src", None)
    nodes = bool(fused.USE_LAYER0_NODES and pv is not None and gv is not None and gv[1].requires_grad and pv[1].requires_grad
                 and (pv[1] is gv[1]) == shared and pv[0].size(0) == p_nodes and fused.l0_nodes_ok(H, pv[0], gv[0], pv[1], gv[1])
                 and (v_gate is None or (not v_gate.requires_grad and v_gate.numel() == gv[0].size(0))))
    specs = [(ps[0], gs[0], e_gate, False)]
    if nodes:
        stacked = not shared and 2 * gv[0].size(1) <= fused.SMALLK_MAX    # two tables as ONE of 2 VK rows: no second launches
        specs.append((pv[0], gv[0], v_gate, stacked))
    packed = fused.l0_pack_many(specs)             # the edge rows' codes and the node rows' codes: one launch
    # the pattern's table over the target's, for the edge rows and for the node rows: one launch for both stacks
    stacks = None if shared else _StackTables.apply(*([ps[1], gs[1]] + ([pv[1], gv[1]] if nodes else [])))
    l0 = fused.Layer0Codes(packed[0], gs[0].size(1), gs[1] if shared else stacks[0],
                           0 if shared else ps[0].size(0), 0 if shared else p_nodes)
    if nodes:
        l0.venc, l0.VK = packed[1], gv[0].size(1) * (2 if stacked else 1)
        l0.WV = gv[1] if shared else stacks[1]
    # the code rows a zero gate wiped, as row masks: the BACKWARD skips their gradient rows -- built on the side stream (behind the
    # index builds; joined with them when the pass ends)
    from . import side
    with side.fork():
        if l0.venc is not None and v_gate is not None:
            l0.venc_mask = fused.code_row_mask(l0.venc, l0.VK)
        if e_gate is not None and not getattr(e_gate, "_dmp_dense_gate", False):
            l0.enc_mask = fused.code_row_mask(l0.enc, l0.K)
    return l0


class _StackTables(th.autograd.Function):
    """``(cat([a, b]), cat([c, d]))`` (one or two pairs of [K, H] tables, row-wise) in ONE launch (``collate.concat_pairs``); the
    gradients go back as the row blocks they belong to (views: no launch)."""

    @staticmethod
    def forward(ctx, *tables):
        from .collate import concat_pairs
        pairs = [(tables[i], tables[i + 1]) for i in range(0, len(tables), 2)]
        ctx.rows = [a.size(0) for a, _ in pairs]
        outs = concat_pairs([(a.detach().contiguous(), b.detach().contiguous(), 0) for a, b in pairs])
        return tuple(o.view(a.size(0) + b.size(0), a.size(1)) for o, (a, b) in zip(outs, pairs))

    @staticmethod
    def backward(ctx, *grads):
        out = []
        for g, n in zip(grads, ctx.rows):
            out += [None, None] if g is None else [g[:n], g[n:]]
        return tuple(out)


class _GateConcat(th.autograd.Function):
    """``cat([p, gate * g])`` written once: the pattern rows are copied, the gated target rows go
    straight into their place in the union buffer (instead of a multiply pass plus a concatenation
    pass over the E-row tensors).  ``gate`` ([rows, 1] or None) carries no gradient.
    ``enc`` / ``W`` (optional): ``g`` is ``enc @ W`` (a label embedding, passed DETACHED): the backward then
    returns ``dW = enc^T (gate * d[n:])`` from one pass over the upstream gradient instead of materialising
    ``gate * d[n:]`` for the embedding's own backward product."""

    @staticmethod
    def forward(ctx, p, g, gate, enc=None, W=None, live_only=False):
        """``live_only`` (with ``enc`` and a 0 / 1 ``gate``): the rows under a zero of the gate are not stored -- every reader of
        the result leaves them out (the first layer of a joint pass over the kept edges' / nodes' tiles)."""
        from . import _lib
        lib = _lib.load()
        if g is None:                      # the embedding itself was never materialised (embed.DeferredEmbedding): rows from enc
            _lib.require_gpu(p, enc)
            p = p.contiguous()
            rows_g = enc.size(0)
        else:
            _lib.require_gpu(p, g)
            p, g = p.contiguous(), g.contiguous()
            rows_g = g.size(0)
        n, H = p.size(0), p.size(1)
        live_only = bool(live_only and enc is not None and gate is not None and p.dtype == th.float32)
        if live_only:
            from . import fused
            out = fused.dead_rows_buffer((n + rows_g, H), p.device)
        else:
            out = th.empty((n + rows_g, H), dtype=p.dtype, device=p.device)
        out[:n].copy_(p)
        if rows_g > 0:
            gt = None if gate is None else gate.reshape(-1).contiguous()
            if enc is not None:   # the gated rows from the K inputs per row instead of from the [rows, H] embedding
                Wd = W.detach()
                fn = lib.dmp_smallk_embed_live if live_only else lib.dmp_smallk_embed_gate
                _lib.check(fn(_lib.ptr(enc), enc.stride(0), enc.size(1), _lib.ptr(Wd), Wd.stride(0),
                              _lib.ptr(gt), rows_g, H, _lib.ptr(out[n:]), H, _lib.stream_ptr()), "dmp_smallk_embed_gate")
            else:
                _lib.check(lib.dmp_gate_residual(None, H, _lib.ptr(g), H, _lib.ptr(gt), g.size(0), H,
                                                 _lib.ptr(out[n:]), H, _lib.stream_ptr()), "dmp_gate_residual")
        ctx.n, ctx.gate, ctx.enc = n, gate, enc
        return out

    @staticmethod
    def backward(ctx, d):
        from . import fused
        d = d.contiguous()
        dg = d[ctx.n:]
        gate = None if ctx.gate is None else ctx.gate.reshape(-1).contiguous()
        if ctx.enc is not None:
            dW = fused.smallk_atb(ctx.enc, dg, gate) if dg.size(0) > 0 else th.zeros((ctx.enc.size(1), d.size(1)), device=d.device)
            return d[:ctx.n], None, None, None, dW, None
        if gate is not None and dg.size(0) > 0:
            dg = fused.gate_residual(None, dg, gate)
        return d[:ctx.n], dg, None, None, None, None


def _gate_concat(p, g, gate):
    """``_GateConcat`` with the embedding-aware backward when ``g`` is a plain label embedding."""
    from . import fused
    from .embed import DeferredEmbedding
    src = getattr(g, "_dmp_src", None)
    if isinstance(g, DeferredEmbedding):
        if (g.dim() == 2 and g.size(1) in fused.MFMA_WIDTHS and g.is_cuda and g.dtype == th.float32 and src[0].dtype == th.float32
                and src[0].size(1) <= fused.SMALLK_MAX and src[0].stride(1) == 1 and src[1].requires_grad and th.is_grad_enabled()):
            return _GateConcat.apply(p, None, gate, src[0], src[1])
        g = g.materialize()
        src = getattr(g, "_dmp_src", None)
    if (src is not None and g.dim() == 2 and g.size(1) in fused.MFMA_WIDTHS and g.is_cuda and g.dtype == th.float32
            and src[0].dtype == th.float32 and src[0].size(1) <= fused.SMALLK_MAX and src[0].size(0) == g.size(0)
            and src[0].stride(1) == 1 and src[1].requires_grad and th.is_grad_enabled()):
        return _GateConcat.apply(p, g.detach(), gate, src[0], src[1])
    return _GateConcat.apply(p, g, gate)


class _SplitRows(th.autograd.Function):
    """(x[:n], x[n:]) whose backward is ONE concatenation (autograd's slice backward would
    allocate and add two zero-padded full-size tensors)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        ctx.n, ctx.shape = n, x.shape
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, da, db):
        if da is None and db is None:
            return None, None
        if da is None or db is None:  # only one side is used downstream
            out = (da if da is not None else db).new_zeros(ctx.shape)
            (out[:ctx.n] if da is not None else out[ctx.n:]).copy_(da if da is not None else db)
            return out, None
        return th.cat([da, db], dim=0), None


class DMPNNRep(DMPNNRepMixin, nn.Module):
    """The representation stage of ``DMPNN`` on its own: ``g_rep_net`` / ``p_rep_net``
    with the reference's child names (``g_rep_net.dmpnn.graph_dmpnn_(i).*``), so a
    reference checkpoint's rep-net entries load with ``strict=False``."""

    def __init__(self, **kw):
        super(DMPNNRep, self).__init__()
        self.hid_dim = kw.get("hid_dim", 64)
        self.share_rep_net = kw.get("share_rep_net", True)
        self.rep_residual = kw.get("rep_residual", True)
        self.g_rep_net = self.create_rep_net(type="graph", **kw)
        self.p_rep_net = self.create_rep_net(type="pattern", **kw)

    @on_input_device
    def forward(self, pattern, graph, p_v_emb, p_e_emb, g_v_emb, g_e_emb, v_gate=None, e_gate=None):
        joint = self.get_joint_rep(pattern, graph, p_v_emb, p_e_emb, g_v_emb, g_e_emb, v_gate, e_gate)
        if joint is not None:
            return joint[:4]
        p_v_rep, p_e_rep = self.get_pattern_rep(pattern, p_v_emb, p_e_emb)
        g_v_rep, g_e_rep = self.get_graph_rep(graph, g_v_emb, g_e_emb, v_gate=v_gate, e_gate=e_gate)
        return p_v_rep, p_e_rep, g_v_rep, g_e_rep

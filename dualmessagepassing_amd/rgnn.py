"""Relational GNN layers on the segment-sum / gather kernels: ``RGCNLayer`` and ``RGINLayer``
(SubgraphCountingMatching/models/rgcn.py:14-213, rgin.py:16-172) and the rep-net parts of the
``RGCN`` / ``RGIN`` models (rgcn.py:215-300, rgin.py:175-260).

Reference per edge (rgcn.py:98-123): ``msg_e = X[src_e] W[type_e] (* norm_e)`` with
``weight.index_select(0, edge_type)`` -- an ``[E, in, out]`` copy of the weights (36 GB at BASELINE
config 2) -- then ``fn.sum`` by destination.  Here the edges are sorted by type once per batched
graph; the gathered (and norm-scaled) source rows then form one contiguous row block per type, each
multiplied by its ONE ``[in, out]`` matrix (a grouped GEMM, no per-edge weight copy), and the
messages are summed by destination with the fixed-order segment-sum kernel over the permuted edge
list.  ``basis`` / ``bdd`` regularisers produce the dense per-type matrices with a few small
differentiable tensor ops (rgcn.py:98-104,112-116).
"""
from collections import OrderedDict

import numpy as np
import torch as th
import torch.nn as nn
from torch.autograd.function import once_differentiable

from . import _lib, ops
from .act import init_weight, map_activation_str_to_layer
from .constants import EDGETYPE, INDEGREE, INNORM, NODEFEAT, NORM, OUTDEGREE, OUTNORM
from .graph import GraphIndex, as_batched, leave_detached

_TYPED_CACHE = OrderedDict()
_TYPED_CACHE_MAX = 16
USE_REL_KERNELS = True      # 128 -> 128 typed products on dmp_rel_gemm / dmp_rel_atb (False: one library GEMM per type)


class TypedIndex:
    """Edges of a (batched) graph sorted by relation type: ``perm`` (edge ids in type order, ties in
    edge-id order), host-side row ranges per type, and the in/out CSR of the permuted edge list.
    Built once per (graph structure, type tensor); the one host sync (segment sizes) is paid then."""

    def __init__(self, index, etype, num_rels):
        _lib.require_gpu(etype)
        etype = etype.view(-1).to(th.int64)
        if etype.numel() != index.num_edges:
            raise _lib.DmpError("edge_type must have one entry per edge")
        self.num_rels, self.num_nodes, self.num_edges = num_rels, index.num_nodes, index.num_edges
        self.perm = th.sort(etype, stable=True)[1]
        counts = th.bincount(etype, minlength=num_rels)
        if counts.numel() > num_rels:
            raise ValueError("edge type %d >= num_rels %d" % (counts.numel() - 1, num_rels))
        self.bounds = [0]
        for c in counts.tolist():                                   # host sync, memoised by typed_index()
            self.bounds.append(self.bounds[-1] + c)
        self.src32p = index.src32[self.perm].contiguous()
        self.dst32p = index.dst32[self.perm].contiguous()
        self.graph = GraphIndex(self.src32p.long(), self.dst32p.long(), index.num_nodes)
        self._tiles = None

    def permuted(self, edge_w):
        """Per-edge weights (edge-id order) in type order, or None."""
        return None if edge_w is None else edge_w.view(-1)[self.perm].contiguous()

    def tiles(self):
        """The 32-slot tiles of the relation-typed MFMA kernels (``dmp_rel_gemm`` / ``dmp_rel_atb``): every
        type's row block of the permuted edge list padded to whole tiles.  ``slot_row`` (row of the permuted
        list per slot, -1 = padding), ``slot_src`` / ``slot_dst`` (its end nodes), ``tile_type``,
        ``type_tile_ptr`` and the tile count, built once from the host-side bounds."""
        if self._tiles is None:
            dev = self.src32p.device
            counts = np.diff(np.asarray(self.bounds, dtype=np.int64))
            per_type = (counts + 31) // 32
            tile_ptr = np.concatenate([[0], np.cumsum(per_type)])
            total = int(tile_ptr[-1])
            shift = np.repeat(tile_ptr[:-1] * 32 - np.asarray(self.bounds[:-1], dtype=np.int64), counts)
            slot_row = np.full(max(total, 1) * 32, -1, dtype=np.int32)
            rows = np.arange(self.num_edges, dtype=np.int64)
            slot_row[rows + shift] = rows
            slot_row = th.from_numpy(slot_row).to(dev)
            valid = slot_row >= 0
            at = slot_row.clamp(min=0).long()
            neg = th.full_like(slot_row, -1)
            self._tiles = dict(
                slot_row=slot_row, valid=valid, at=at,
                slot_src=th.where(valid, self.src32p[at], neg), slot_dst=th.where(valid, self.dst32p[at], neg),
                tile_type=th.from_numpy(np.repeat(np.arange(self.num_rels, dtype=np.int32), per_type)).to(dev)
                if total else th.zeros(1, dtype=th.int32, device=dev),
                type_tile_ptr=th.from_numpy(tile_ptr.astype(np.int32)).to(dev),
                num_tiles=th.tensor([total], dtype=th.int32, device=dev), total=total)
        return self._tiles


def typed_index(graph, etype, num_rels):
    """Memoised ``TypedIndex`` of ``graph`` for the type tensor ``etype``."""
    index = graph.index()
    key = (id(index), etype.data_ptr(), etype._version, int(etype.numel()), num_rels)
    hit = _TYPED_CACHE.get(key)
    if hit is None:
        hit = (TypedIndex(index, etype, num_rels), index, etype)  # keep the keyed objects alive
        _TYPED_CACHE[key] = hit
        while len(_TYPED_CACHE) > _TYPED_CACHE_MAX:
            _TYPED_CACHE.popitem(last=False)
    else:
        _TYPED_CACHE.move_to_end(key)
    return hit[0]


def rel_ok(x, weight):
    """The relation-typed MFMA kernels take 128 -> 128 layers (contiguous fp32 rows)."""
    return (x.dim() == 2 and x.size(1) == 128 and weight.dim() == 3 and weight.size(1) == 128 and weight.size(2) == 128
            and x.dtype == th.float32 and weight.dtype == th.float32)


def rel_gemm(a, weight, slot_arow, tix, row_scale=None, transposed=False):
    """``out[r] = row_scale[r] * a[slot_arow -> r] @ weight[type r]`` (or its transpose) for the rows ``r`` of
    the permuted edge list (``dmp_rel_gemm``)."""
    t = tix.tiles()
    lib = _lib.load()
    a, weight = a.contiguous(), weight.contiguous()
    out = th.empty((tix.num_edges, 128), dtype=th.float32, device=a.device)
    with _lib.timed("rel_gemm[E=%d]", (tix.num_edges,), 8 * 128 * tix.num_edges):
        _lib.check(lib.dmp_rel_gemm(_lib.ptr(a), a.stride(0), a.size(0), _lib.ptr(weight), weight.stride(1), weight.size(0),
                                    1 if transposed else 0, _lib.ptr(slot_arow), _lib.ptr(t["slot_row"]), _lib.ptr(t["tile_type"]),
                                    _lib.ptr(t["num_tiles"]), t["total"], _lib.ptr(row_scale) if row_scale is not None else None,
                                    tix.num_edges, 128, _lib.ptr(out), out.stride(0), _lib.stream_ptr()), "dmp_rel_gemm")
    return out


def rel_atb(x, d, tix, row_scale=None):
    """``dW[t] = sum over the edges e of type t of row_scale_e * x[src e]^T d[dst e]`` (``dmp_rel_atb`` and a
    fixed-order sum of its per-workgroup partials)."""
    t = tix.tiles()
    lib = _lib.load()
    x, d = x.contiguous(), d.contiguous()
    blocks = int(lib.dmp_rel_atb_blocks(tix.num_rels))
    part = th.empty((tix.num_rels, blocks, 128, 128), dtype=th.float32, device=x.device)
    slot_scale = None
    if row_scale is not None:
        slot_scale = th.where(t["valid"], row_scale[t["at"]], row_scale.new_zeros(()))
    with _lib.timed("rel_atb[E=%d]", (tix.num_edges,), 8 * 128 * tix.num_edges):
        _lib.check(lib.dmp_rel_atb(_lib.ptr(x), x.stride(0), x.size(0), _lib.ptr(d), d.stride(0), d.size(0), _lib.ptr(t["slot_src"]),
                                   _lib.ptr(t["slot_dst"]), _lib.ptr(slot_scale) if slot_scale is not None else None,
                                   _lib.ptr(t["type_tile_ptr"]), tix.num_rels, t["total"], 128, _lib.ptr(part),
                                   _lib.stream_ptr()), "dmp_rel_atb")
    return part.sum(1) if blocks > 1 else part[:, 0]


class _TypedLinearAgg(th.autograd.Function):
    """``agg[v] = sum_{e: dst(e)=v} w_e * X[src e] @ W[type e]``  (rgcn.py:98-123 + fn.sum).

    128 -> 128 layers run on the relation-typed MFMA kernels: the type-sorted edge list in 32-slot tiles that
    never mix types, the source (backward: destination) rows gathered inside the kernel, the weight panel of the
    tile's type in registers -- one launch each for the messages, the input gradient and all the weight gradients
    (``USE_REL_KERNELS``; profiles/r02_rgnn.txt).  Other widths: one library GEMM per type over that type's
    contiguous row block (a single batched GEMM over equal, padded slices was measured 2.5x slower at BASELINE
    config 2: profiles/r01_rgnn.txt)."""

    @staticmethod
    def forward(ctx, x, weight, tix, w_p):
        _lib.require_gpu(x, weight)
        x = x.contiguous()
        g = tix.graph
        ctx.tix, ctx.w_p = tix, w_p
        ctx.rel = USE_REL_KERNELS and rel_ok(x, weight) and tix.num_edges > 0
        if ctx.rel:
            msg = rel_gemm(x, weight, tix.tiles()["slot_src"], tix, w_p)
            ctx.save_for_backward(x, weight)
            return ops.seg_sum_raw(msg, g.in_ptr, g.in_ent, tix.num_nodes)
        xg = ops.gather_rows_raw(x, tix.src32p, w_p)                  # [E, in] in type order, w_e applied
        msg = th.empty((tix.num_edges, weight.size(2)), dtype=th.float32, device=x.device)
        for t in range(tix.num_rels):
            lo, hi = tix.bounds[t], tix.bounds[t + 1]
            if hi > lo:
                th.mm(xg[lo:hi], weight[t], out=msg[lo:hi])
        ctx.save_for_backward(xg, weight)
        return ops.seg_sum_raw(msg, g.in_ptr, g.in_ent, tix.num_nodes)

    @staticmethod
    @once_differentiable
    def backward(ctx, d_agg):
        tix, g = ctx.tix, ctx.tix.graph
        d_agg = d_agg.contiguous()
        if ctx.rel:
            x, weight = ctx.saved_tensors
            d_xg = rel_gemm(d_agg, weight, tix.tiles()["slot_dst"], tix, ctx.w_p, transposed=True)
            d_x = ops.seg_sum_raw(d_xg, g.out_ptr, g.out_ent, tix.num_nodes)
            return d_x, rel_atb(x, d_agg, tix, ctx.w_p), None, None
        xg, weight = ctx.saved_tensors
        d_msg = ops.gather_rows_raw(d_agg, tix.dst32p)               # [E, out] in type order
        d_xg = th.empty_like(xg)
        d_w = th.zeros_like(weight)
        for t in range(tix.num_rels):
            lo, hi = tix.bounds[t], tix.bounds[t + 1]
            if hi > lo:
                th.mm(d_msg[lo:hi], weight[t].t(), out=d_xg[lo:hi])
                th.mm(xg[lo:hi].t(), d_msg[lo:hi], out=d_w[t])
        d_x = ops.seg_sum_raw(d_xg, g.out_ptr, g.out_ent, tix.num_nodes, ctx.w_p)
        return d_x, d_w, None, None


def typed_linear_agg(x, weight, tix, edge_w=None):
    """``edge_w``: optional per-EDGE weights (edge-id order), e.g. the RGCN normaliser."""
    return _TypedLinearAgg.apply(x, weight, tix, tix.permuted(edge_w))


class _RelLayer(nn.Module):
    """Weights and message part shared by RGCNLayer / RGINLayer (rgcn.py:31-96, rgin.py:31-98)."""

    def _init_rel(self, input_dim, hidden_dim, num_rels, regularizer, num_bases, self_loop, bias, act_func):
        assert regularizer in ["none", "basis", "bdd"]
        self.input_dim, self.hidden_dim, self.num_rels, self.regularizer = input_dim, hidden_dim, num_rels, regularizer
        if regularizer == "none" or num_bases is None or num_bases > num_rels or num_bases <= 0:
            self.num_bases = num_rels
        else:
            self.num_bases = num_bases
        if self_loop:
            self.loop_weight = nn.Parameter(th.Tensor(input_dim, hidden_dim))
        else:
            self.register_parameter("loop_weight", None)
        if bias:
            self.bias = nn.Parameter(th.Tensor(hidden_dim))
        else:
            self.register_parameter("bias", None)

    def _init_rel_weights(self, act_func):
        if self.regularizer in ("none", "basis"):
            self.weight = nn.Parameter(th.Tensor(self.num_bases, self.input_dim, self.hidden_dim))
            if self.num_bases < self.num_rels:
                self.w_comp = nn.Parameter(th.Tensor(self.num_rels, self.num_bases))
            else:
                self.register_parameter("w_comp", None)
        else:
            if self.input_dim % self.num_bases != 0 or self.hidden_dim % self.num_bases != 0:
                raise ValueError("Feature size must be a multiplier of num_bases (%d)." % self.num_bases)
            submat_in, submat_out = self.input_dim // self.num_bases, self.hidden_dim // self.num_bases
            self.weight = nn.Parameter(th.Tensor(self.num_rels, self.num_bases * submat_in * submat_out))
            self.register_parameter("w_comp", None)
        init_weight(self.weight, activation=act_func, init="uniform")
        if self.w_comp is not None:
            init_weight(self.w_comp, activation=act_func, init="uniform")
        if self.loop_weight is not None:
            init_weight(self.loop_weight, activation=act_func, init="uniform")
        nn.init.zeros_(self.bias)   # as the reference: a layer without bias fails here (rgcn.py:84)

    @property
    def self_loop(self):
        return hasattr(self, "loop_weight") and self.loop_weight is not None

    def dense_weight(self):
        """The ``[num_rels, in, out]`` matrices the message functions apply per edge type."""
        if self.regularizer in ("none", "basis"):
            if self.num_bases < self.num_rels:  # rgcn.py:99-102
                w = self.weight.view(self.num_bases, self.input_dim * self.hidden_dim)
                return th.matmul(self.w_comp, w).view(self.num_rels, self.input_dim, self.hidden_dim)
            return self.weight
        # bdd (rgcn.py:112-116): block b maps input slice b to output slice b
        nb, si, so = self.num_bases, self.input_dim // self.num_bases, self.hidden_dim // self.num_bases
        blocks = self.weight.view(self.num_rels, nb, si, so)
        dense = self.weight.new_zeros((self.num_rels, nb, si, nb, so))
        idx = th.arange(nb, device=self.weight.device)
        dense[:, idx, :, idx, :] = blocks.transpose(0, 1)
        return dense.view(self.num_rels, self.input_dim, self.hidden_dim)

    def get_output_dim(self):
        return self.hidden_dim


class RGCNLayer(_RelLayer):
    """rgcn.py:14-213."""

    def __init__(self, input_dim, hidden_dim, num_rels=1, regularizer="basis", num_bases=-1, edge_norm="in",
                 self_loop=True, bias=True, batch_norm=False, act_func="relu", dropout=0.0):
        super(RGCNLayer, self).__init__()
        assert edge_norm in ["none", "in", "both"]
        self._init_rel(input_dim, hidden_dim, num_rels, regularizer, num_bases, self_loop, bias, act_func)
        self.edge_norm = edge_norm
        self.bn = nn.BatchNorm1d(hidden_dim) if batch_norm else None
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)
        self._init_rel_weights(act_func)

    def _norms(self, graph):
        """rgcn.py:129-151: cached ``in_norm`` / ``out_norm`` node frames."""
        g, add = graph, (1.0 if self.self_loop else 0.0)
        if self.edge_norm in ("in", "both"):
            if INDEGREE not in g.ndata:
                g.ndata[INDEGREE] = g.in_degrees()
            if INNORM not in g.ndata:
                d = g.ndata[INDEGREE].float()
                g.ndata[INNORM] = ((1.0 / (d + add)) if self.self_loop else (1.0 / d).masked_fill_(d == 0, 0.0)).view(-1, 1)
        if self.edge_norm in ("out", "both"):
            if OUTDEGREE not in g.ndata:
                g.ndata[OUTDEGREE] = g.out_degrees()
            if OUTNORM not in g.ndata:
                d = g.ndata[OUTDEGREE].float()
                g.ndata[OUTNORM] = ((1.0 / (d + add)) if self.self_loop else (1.0 / d).masked_fill_(d == 0, 0.0)).view(-1, 1)

    @_lib.on_input_device
    def forward(self, g, node_feat, edge_type):
        g = as_batched(g)   # DGLGraph-in (rgcn.py:182)
        if node_feat is not None:
            g.ndata[NODEFEAT] = node_feat
        self._norms(g)
        if edge_type is not None:
            g.edata[EDGETYPE] = edge_type
        u, v = g.all_edges(form="uv", order="eid")
        if self.edge_norm == "in":      # rgcn.py:157-163
            g.edata[NORM] = g.ndata[INNORM][v]
        elif self.edge_norm == "out":
            g.edata[NORM] = g.ndata[OUTNORM][u]
        elif self.edge_norm == "both":
            g.edata[NORM] = (g.ndata[OUTNORM][u] * g.ndata[INNORM][v]) ** 0.5
        tix = typed_index(g, g.edata[EDGETYPE], self.num_rels)
        agg = typed_linear_agg(g.ndata[NODEFEAT], self.dense_weight(), tix,
                               None if self.edge_norm == "none" else g.edata[NORM])
        if self.self_loop:               # rgcn.py:167-180
            loop_msg = th.matmul(g.ndata[NODEFEAT], self.loop_weight)
            if self.edge_norm == "in":
                out = agg + loop_msg * g.ndata[INNORM]
            elif self.edge_norm == "out":
                out = agg + loop_msg * g.ndata[OUTNORM]
            elif self.edge_norm == "both":
                out = agg + loop_msg * (g.ndata[INNORM] * g.ndata[OUTNORM]) ** 0.5
            else:
                out = agg + loop_msg
        else:
            out = agg
        if self.bias is not None:
            out = out + self.bias
        if self.bn is not None:
            out = self.bn(out)
        out = self.drop(self.act(out))
        leave_detached(g.ndata, NODEFEAT)
        return out, edge_type

    def extra_repr(self):
        return "in=%d, out=%d, num_rels=%d, regularizer=%s, num_bases=%d, edge_norm=%s, self_loop=%s, bias=%s" % (
            self.input_dim, self.hidden_dim, self.num_rels, self.regularizer, self.num_bases, self.edge_norm,
            self.self_loop, self.bias is not None)


class RGINLayer(_RelLayer):
    """rgin.py:16-172: RGCN messages without normalisation, GIN-style MLP update."""

    def __init__(self, input_dim, hidden_dim, num_rels=1, regularizer="basis", num_bases=-1, num_mlp_layers=2,
                 self_loop=True, bias=True, batch_norm=False, act_func="relu", dropout=0.0):
        super(RGINLayer, self).__init__()
        self._init_rel(input_dim, hidden_dim, num_rels, regularizer, num_bases, self_loop, bias, act_func)
        mlp = []
        for i in range(num_mlp_layers):
            mlp.append(nn.Linear(hidden_dim, hidden_dim))
            if i != num_mlp_layers - 1:
                if batch_norm:
                    mlp.append(nn.BatchNorm1d(hidden_dim))
                mlp.append(map_activation_str_to_layer(act_func))
        self.mlp = nn.Sequential(*mlp)
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)
        self._init_rel_weights(act_func)

    @_lib.on_input_device
    def forward(self, g, node_feat, edge_type):
        g = as_batched(g)   # DGLGraph-in (rgin.py:124)
        if node_feat is not None:
            g.ndata[NODEFEAT] = node_feat
        if edge_type is not None:
            g.edata[EDGETYPE] = edge_type
        tix = typed_index(g, g.edata[EDGETYPE], self.num_rels)
        agg = typed_linear_agg(g.ndata[NODEFEAT], self.dense_weight(), tix, None)
        out = agg + th.matmul(g.ndata[NODEFEAT], self.loop_weight) if self.self_loop else agg
        if self.bias is not None:
            out = out + self.bias
        out = ops.apply_mlp(self.mlp, out) if len(self.mlp) > 0 else self.act(out)
        out = self.drop(self.act(out))   # rgin.py:147-152: activation after the MLP as well
        leave_detached(g.ndata, NODEFEAT)
        return out, edge_type

    def extra_repr(self):
        return "in=%d, out=%d, num_rels=%d, regularizer=%s, num_bases=%d, self_loop=%s, bias=%s" % (
            self.input_dim, self.hidden_dim, self.num_rels, self.regularizer, self.num_bases, self.self_loop,
            self.bias is not None)


class _RelRepMixin:
    """``create_rep_net`` / ``get_pattern_rep`` / ``get_graph_rep`` of RGCN / RGIN
    (rgcn.py:219-300, rgin.py:179-260); ``rep_key`` names the ModuleDict entry and the children."""

    rep_key = None

    def _make_layer(self, num_rels, **kw):
        raise NotImplementedError

    def create_rep_net(self, type, **kw):
        if type == "graph":
            num_layers, num_rels = kw.get("rep_num_graph_layers", 1), self.max_ngel
        elif type == "pattern":
            if self.share_rep_net:
                return self.g_rep_net
            num_layers, num_rels = kw.get("rep_num_pattern_layers", 1), self.max_npel
        else:
            raise ValueError(type)
        layers = nn.ModuleList()
        for i in range(num_layers):
            layers.add_module("%s_%s_(%d)" % (type, self.rep_key, i), self._make_layer(num_rels, **kw))
        return nn.ModuleDict({self.rep_key: layers})

    def get_pattern_rep(self, pattern, p_emb, mask=None):
        etype = pattern.edata["label"]
        if mask is not None:
            zero = ~mask
            out = p_emb.masked_fill(zero, 0.0)
            for layer in self.p_rep_net[self.rep_key]:
                o, etype = layer(pattern, out, etype)
                out = o.masked_fill(zero, 0.0)
            return out
        out = p_emb
        for layer in self.p_rep_net[self.rep_key]:
            o, etype = layer(pattern, out, etype)
            out = out + o if (self.rep_residual and out.size() == o.size()) else o
        return out

    def get_graph_rep(self, graph, g_emb, mask=None, gate=None):
        etype = graph.edata["label"]
        if mask is None and gate is None:
            out = g_emb
            for layer in self.g_rep_net[self.rep_key]:
                o, etype = layer(graph, out, etype)
                out = out + o if (self.rep_residual and out.size() == o.size()) else o
            return out
        if gate is None:
            gate = mask.float()
        elif mask is not None:
            gate = mask.float() * gate
        out = g_emb * gate
        for layer in self.g_rep_net[self.rep_key]:
            o, etype = layer(graph, out, etype)
            o = o * gate
            out = out + o if (self.rep_residual and out.size() == o.size()) else o
        return out


class RGCNRepMixin(_RelRepMixin):
    rep_key = "rgcn"

    def _make_layer(self, num_rels, **kw):
        return RGCNLayer(self.hid_dim, self.hid_dim, num_rels=num_rels,
                         regularizer=kw.get("rep_rgcn_regularizer", "basis"), num_bases=kw.get("rep_rgcn_num_bases", -1),
                         edge_norm=kw.get("rep_rgcn_edge_norm", "in"), batch_norm=kw.get("rep_rgcn_batch_norm", False),
                         act_func=kw.get("rep_act_func", "relu"), dropout=kw.get("rep_dropout", 0.0))


class RGINRepMixin(_RelRepMixin):
    rep_key = "rgin"

    def _make_layer(self, num_rels, **kw):
        return RGINLayer(self.hid_dim, self.hid_dim, num_rels=num_rels,
                         regularizer=kw.get("rep_rgin_regularizer", "basis"), num_bases=kw.get("rep_rgin_num_bases", -1),
                         num_mlp_layers=kw.get("rep_rgin_num_mlp_layers", 2), batch_norm=kw.get("rep_rgin_batch_norm", False),
                         act_func=kw.get("rep_act_func", "relu"), dropout=kw.get("rep_dropout", 0.0))

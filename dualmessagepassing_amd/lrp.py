"""Local relational pooling on the gather / segment-sum kernels: ``LRPLayer`` / ``LRP`` and ``DMPLRPPoolLayer`` / ``DMPLRP``
(SubgraphCountingMatching/models/lrp.py:18-390, models/dmplrp.py:20-470).

What the reference does per layer with three ``torch.sparse.mm`` calls (lrp.py:66-73, dmplrp.py:188-194):

    rows = node_to_perm @ node_feat + edge_to_perm @ edge_feat      [P * L*L, in]   P permutations, L = lrp_seq_len
    out  = einsum('dab,bca->dc', rows.view(P, L*L, in), weight)     [P, hid]
    node = pooling @ out                                            [N, hid]

The two selection matrices have unit entries and at most one entry per row between them (slot (i, i) of a permutation
holds its i-th node, slot (i, j) the edge from its i-th to its j-th node if there is one, dataset.py:1759-1776), and
the pooling matrix averages a CONTIGUOUS range of permutations per node (dataset.py:1795-1811).  Here therefore:

    rows = take_rows([node_feat ; edge_feat ; 0], slot)            one row gather (backward: fixed-order segment sum)
    out  = rows.view(P, L*L*in) @ weight.permute(2, 0, 1).reshape(L*L*in, hid)     one dense product, K = 16 in
    node = scale[:, None] * seg_pool(out, ranges)                  the segment-sum kernel over contiguous ranges

``PermIndex`` holds the slot vector / ranges / scales of a batch; it is built from the reference's three sparse
tensors (``from_sparse``: what ``LRPDataset.batchify`` hands over) or from the ego-net permutations directly.
"""
from itertools import permutations

import numpy as np
import torch as th
import torch.nn as nn

from . import _lib, ops
from .act import init_module, init_weight, map_activation_str_to_layer
from .basemodel import GraphAdjModelV2
from .constants import INDEGREE
from .dmpnn import DMPLayer
from .graph import as_batched


class PermIndex:
    """Slots, pooling ranges and scales of the permutations of one batched graph (device tensors).

    slot  int64 [P * L*L]  row of ``[node rows ; edge rows ; one zero row]`` that fills the slot
    sizes int64 [N]        permutations of node v: the next ``sizes[v]`` of them
    scale float [N]        factor on node v's sum (1 / sizes[v] for the reference's mean pooling)"""

    def __init__(self, slot, sizes, scale, num_nodes, num_edges, seq_len):
        self.slot, self.sizes, self.scale = slot, sizes, scale
        self.num_nodes, self.num_edges, self.seq_len = int(num_nodes), int(num_edges), int(seq_len)
        self.num_perms = int(slot.numel()) // (seq_len * seq_len)
        self._pool = None

    @classmethod
    def from_sparse(cls, pooling_matrix, node_to_perm, edge_to_perm, seq_len=4):
        """From the reference's ``(p_perm_pool, (p_n_perm_matrix, p_e_perm_matrix))`` sparse COO tensors."""
        n2p, e2p, pool = node_to_perm.coalesce(), edge_to_perm.coalesce(), pooling_matrix.coalesce()
        _lib.require_gpu(n2p.values(), e2p.values(), pool.values())
        rows, N, E = n2p.shape[0], n2p.shape[1], e2p.shape[1]
        dev = n2p.values().device
        slot = th.full((rows,), N + E, dtype=th.int64, device=dev)            # empty slots read the zero row
        slot[n2p.indices()[0]] = n2p.indices()[1]
        slot[e2p.indices()[0]] = N + e2p.indices()[1]
        prow = pool.indices()[0]
        marks = th.searchsorted(prow, th.arange(pool.shape[0] + 1, device=dev))   # rows are sorted after coalesce()
        sizes = marks[1:] - marks[:-1]
        first = marks[:-1].clamp(max=max(int(prow.numel()) - 1, 0))
        scale = th.where(sizes > 0, pool.values()[first] if prow.numel() else th.zeros_like(sizes, dtype=th.float32),
                         th.zeros((), dtype=th.float32, device=dev))
        return cls(slot, sizes, scale.float(), N, E, seq_len)

    @classmethod
    def from_graphs(cls, graphs, device, seq_len=4, pooling="mean"):
        """From single graphs given as ``(src, dst, num_nodes[, is_reversed])`` host arrays: the ego-net permutations of
        ``LRPDataset.graph_to_egonet_seq`` (dataset.py:1751-1793: for every node, the permutations of up to
        ``seq_len - 1`` of its out-neighbours over the non-reversed edges, in ``itertools.permutations`` order) laid out
        like ``build_batch_graph_to_perm_matrices`` / ``build_perm_pooling_matrix`` do for a batch."""
        L, slots, sizes = seq_len, [], []
        n_off = e_off = 0
        total_n = sum(int(g[2]) for g in graphs)
        total_e = sum(len(g[0]) for g in graphs)
        zero = total_n + total_e
        for g in graphs:
            src, dst, n = np.asarray(g[0]), np.asarray(g[1]), int(g[2])
            keep = np.ones(len(src), bool) if len(g) < 4 or g[3] is None else ~np.asarray(g[3], bool)
            eid = {}
            adj = [[] for _ in range(n)]
            for e in np.nonzero(keep)[0]:
                u, v = int(src[e]), int(dst[e])
                if (u, v) not in eid:          # scipy's csr_matrix merges duplicate (u, v) entries into one neighbour
                    adj[u].append(v)
                eid[(u, v)] = int(e)           # the dict keeps the LAST edge of a pair
            for v in range(n):
                nbrs = sorted(adj[v])          # csr indices come out sorted by column
                perms = [(v,) + p for p in permutations(nbrs, min(L - 1, len(nbrs)))]
                sizes.append(len(perms))
                for perm in perms:
                    block = np.full(L * L, zero, np.int64)
                    for i, a in enumerate(perm):
                        block[i * (L + 1)] = n_off + a
                        for j, b in enumerate(perm):
                            if (a, b) in eid:
                                block[i * L + j] = total_n + e_off + eid[(a, b)]
                    slots.append(block)
            n_off += n
            e_off += len(src)
        slot = th.from_numpy(np.concatenate(slots) if slots else np.zeros(0, np.int64)).to(device)
        sizes = th.tensor(sizes, dtype=th.int64, device=device)
        scale = th.where(sizes > 0, 1.0 / sizes.clamp(min=1).float(), th.zeros((), device=device)) if pooling == "mean" \
            else th.ones(sizes.numel(), device=device)
        return cls(slot, sizes, scale, total_n, total_e, seq_len)

    def pool_index(self):
        if self._pool is None:
            self._pool = ops.PoolIndex(self.sizes, num_rows=self.num_perms)
        return self._pool


def as_perm_index(graph, pooling_matrix, node_to_perm, edge_to_perm, seq_len):
    """The layer's three matrix arguments -> a ``PermIndex`` (a PermIndex passed in their first position is taken as is;
    the conversion of sparse tensors is cached on the pooling matrix object: every layer of a rep-net gets the same three)."""
    if isinstance(pooling_matrix, PermIndex):
        return pooling_matrix
    cached = getattr(pooling_matrix, "_dmp_perm_index", None)
    if cached is None or cached[0] is not node_to_perm or cached[1] is not edge_to_perm:
        cached = (node_to_perm, edge_to_perm, PermIndex.from_sparse(pooling_matrix, node_to_perm, edge_to_perm, seq_len))
        try:
            pooling_matrix._dmp_perm_index = cached
        except Exception:
            pass
    return cached[2]


def perm_pool(perm, node_feat, edge_feat, weight, bias=None, act=None):
    """``pooling @ act(einsum(slots(node_feat, edge_feat), weight) + bias)`` (lrp.py:66-73 / dmplrp.py:188-194 with
    ``act`` None): gather, one dense product over the L*L slots, per-node segment sum, scale."""
    L2 = perm.seq_len * perm.seq_len
    table = th.cat([node_feat, edge_feat, node_feat.new_zeros((1, node_feat.size(1)))], dim=0)
    rows = ops.take_rows(table, perm.slot)                                        # [P * L*L, in]
    w = weight.permute(2, 0, 1).reshape(L2 * weight.size(0), weight.size(1))      # [(slot, in), hid]
    out = ops.matmul_xw(rows.view(perm.num_perms, L2 * weight.size(0)), w)
    if bias is not None:
        out = out + bias
    if act is not None:
        out = act(out)
    return ops.seg_pool(out, perm.pool_index()) * perm.scale.view(-1, 1)


class LRPLayer(nn.Module):
    """models/lrp.py:18-104."""

    def __init__(self, input_dim=2, hidden_dim=128, lrp_seq_len=4, bias=True, act_func="relu", batch_norm=False, mlp=False,
                 dropout=0.0):
        super(LRPLayer, self).__init__()
        self.lrp_seq_len, self.input_dim, self.hidden_dim = lrp_seq_len, input_dim, hidden_dim
        self.weight = nn.Parameter(th.empty(input_dim, hidden_dim, lrp_seq_len * lrp_seq_len))
        self.degnet_0 = nn.Linear(1, 2 * hidden_dim)
        self.degnet_1 = nn.Linear(2 * hidden_dim, hidden_dim)
        if bias:
            self.bias = nn.Parameter(th.empty(hidden_dim))
        else:
            self.register_parameter("bias", None)
        self.act = map_activation_str_to_layer(act_func)
        self.bn = nn.BatchNorm1d(hidden_dim) if batch_norm else None
        self.mlp = nn.Linear(hidden_dim, hidden_dim) if mlp else None
        self.drop = nn.Dropout(dropout)
        init_weight(self.weight, activation=act_func, init="uniform")
        init_module(self.degnet_0, activation=act_func, init="uniform")
        init_module(self.degnet_1, activation=act_func, init="uniform")
        if bias:
            nn.init.zeros_(self.bias)
        if mlp:
            init_module(self.mlp, activation=act_func, init="uniform")

    @_lib.on_input_device
    def forward(self, graph, node_feat, edge_feat, pooling_matrix, node_to_perm_matrix=None, edge_to_perm_matrix=None):
        g = as_batched(graph)
        perm = as_perm_index(g, pooling_matrix, node_to_perm_matrix, edge_to_perm_matrix, self.lrp_seq_len)
        node_out = perm_pool(perm, node_feat, edge_feat, self.weight, self.bias, self.act)
        deg = g.ndata[INDEGREE] if INDEGREE in g.ndata else g.in_degrees()
        factor = self.degnet_1(self.act(self.degnet_0(deg.float().unsqueeze(1))))
        node_out = self.act(node_out * factor)
        if self.bn is not None:
            node_out = self.bn(node_out)
        if self.mlp is not None:
            node_out = self.act(self.mlp(node_out))
        return self.drop(node_out), edge_feat

    def get_output_dim(self):
        return self.hidden_dim


class DMPLRPPoolLayer(DMPLayer):
    """models/dmplrp.py:20-213: a DMPLayer whose node and edge outputs go through the permutation pooling."""

    def __init__(self, input_dim, hidden_dim, init_neigenv=4.0, init_eeigenv=4.0, lrp_seq_len=4, bias=True, num_mlp_layers=2,
                 batch_norm=True, act_func="relu", dropout=0.0):
        nn.Module.__init__(self)
        self.input_dim, self.hidden_dim, self.lrp_seq_len = input_dim, hidden_dim, lrp_seq_len
        names = ("in_weight", "out_weight", "src_weight", "dst_weight", "nloop_weight", "eloop_weight")
        for name in names:                                     # registration order of dmplrp.py:39-45
            setattr(self, name, nn.Parameter(th.empty(input_dim, hidden_dim)))
        self.lrp_weight = nn.Parameter(th.empty(input_dim, hidden_dim, lrp_seq_len * lrp_seq_len))
        for name in ("nbias", "ebias", "lrp_bias"):
            if bias:
                setattr(self, name, nn.Parameter(th.empty(hidden_dim)))
            else:
                self.register_parameter(name, None)
        self.nmlp = self._make_mlp(hidden_dim, num_mlp_layers, batch_norm, act_func)
        self.emlp = self._make_mlp(hidden_dim, num_mlp_layers, batch_norm, act_func)
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)
        self.write_edge_agg = False
        for name in names:                                     # initialisation order of dmplrp.py:73-90
            init_weight(getattr(self, name), activation=act_func, init="uniform")
        init_weight(self.lrp_weight, init="uniform")
        for module in list(self.nmlp.modules()) + list(self.emlp.modules()):
            init_module(module, activation=act_func, init="uniform")
        if bias:
            for name in ("nbias", "ebias", "lrp_bias"):
                nn.init.zeros_(getattr(self, name))
        with th.no_grad():
            for name in ("in_weight", "out_weight", "nloop_weight"):
                getattr(self, name).div_(init_neigenv)
            for name in ("src_weight", "dst_weight", "eloop_weight"):
                getattr(self, name).div_(init_eeigenv)

    def fused_ok(self, *a, **k):                               # the pooled node output replaces the layer's: modular path
        return False

    @_lib.on_input_device
    def forward(self, graph, node_feat, edge_feat, pooling_matrix, node_to_perm_matrix=None, edge_to_perm_matrix=None):
        g = as_batched(graph)
        perm = as_perm_index(g, pooling_matrix, node_to_perm_matrix, edge_to_perm_matrix, self.lrp_seq_len)
        node_out, edge_out = DMPLayer.forward.__wrapped__(self, g, node_feat, edge_feat)
        node_out = perm_pool(perm, node_out, edge_out, self.lrp_weight, self.lrp_bias, None)
        return node_out, edge_out, pooling_matrix, node_to_perm_matrix, edge_to_perm_matrix


class _PermRepMixin:
    """``create_rep_net`` / ``get_pattern_rep`` / ``get_graph_rep`` of LRP / DMPLRP (lrp.py:108-220, dmplrp.py:216-330):
    the layer loop with the three permutation inputs handed through; the residual sum is taken by DMPLRP only (the LRP
    loop appends ``v`` on both branches, lrp.py:162-167)."""

    rep_key, residual_sum = None, False

    def _make_layer(self, **kw):
        raise NotImplementedError

    def create_rep_net(self, type, **kw):
        if type == "pattern" and self.share_rep_net:
            return self.g_rep_net
        num_layers = kw.get("rep_num_graph_layers" if type == "graph" else "rep_num_pattern_layers", 1)
        layers = nn.ModuleList()
        for i in range(num_layers):
            layers.add_module("%s_%s_(%d)" % (type, self.rep_key, i), self._make_layer(**kw))
        return nn.ModuleDict({self.rep_key: layers})

    def _run(self, net, graph, v, e, perm, v_gate=None, e_gate=None, v_zero=None, e_zero=None):
        for layer in net[self.rep_key]:
            nv, ne = layer(graph, v, e, *perm)[:2]
            if v_zero is not None:
                nv = nv.masked_fill(v_zero, 0.0)
            if e_zero is not None:
                ne = ne.masked_fill(e_zero, 0.0)
            if v_gate is not None:
                nv = nv * v_gate
            if e_gate is not None:
                ne = ne * e_gate
            if self.residual_sum and self.rep_residual and nv.size() == v.size() and ne.size() == e.size():
                v, e = v + nv, e + ne
            else:
                v, e = nv, ne
        return v, e

    def get_pattern_rep(self, pattern, p_v_emb, p_e_emb, p_perm_pool, p_n_perm_matrix=None, p_e_perm_matrix=None, v_mask=None,
                        e_mask=None):
        v_zero = None if v_mask is None else ~v_mask
        e_zero = None if e_mask is None else ~e_mask
        v = p_v_emb if v_zero is None else p_v_emb.masked_fill(v_zero, 0.0)
        e = p_e_emb if e_zero is None else p_e_emb.masked_fill(e_zero, 0.0)
        return self._run(self.p_rep_net, pattern, v, e, (p_perm_pool, p_n_perm_matrix, p_e_perm_matrix), v_zero=v_zero, e_zero=e_zero)

    def get_graph_rep(self, graph, g_v_emb, g_e_emb, g_perm_pool, g_n_perm_matrix=None, g_e_perm_matrix=None, v_mask=None,
                      e_mask=None, v_gate=None, e_gate=None):
        if v_mask is not None:
            v_gate = v_mask.float() if v_gate is None else v_mask.float() * v_gate
        if e_mask is not None:
            e_gate = e_mask.float() if e_gate is None else e_mask.float() * e_gate
        v = g_v_emb if v_gate is None else g_v_emb * v_gate
        e = g_e_emb if e_gate is None else g_e_emb * e_gate
        return self._run(self.g_rep_net, graph, v, e, (g_perm_pool, g_n_perm_matrix, g_e_perm_matrix), v_gate=v_gate, e_gate=e_gate)


class _PermModel(_PermRepMixin, GraphAdjModelV2):
    """The 8-argument forward of lrp.py:222-390 / dmplrp.py:332-470 on the common skeleton: the permutation inputs ride
    along on the graph objects while ``GraphAdjModelV2.forward`` runs."""

    edge_head_skips_reversed = False      # lrp.py:240-243 / dmplrp.py:350-353: plain length masks

    def forward(self, pattern, p_perm_pool, p_n_perm_matrix, p_e_perm_matrix, graph, g_perm_pool=None, g_n_perm_matrix=None,
                g_e_perm_matrix=None):
        pattern, graph = as_batched(pattern), as_batched(graph)
        self._perm = {id(pattern): (p_perm_pool, p_n_perm_matrix, p_e_perm_matrix), id(graph): (g_perm_pool, g_n_perm_matrix, g_e_perm_matrix)}
        try:
            return GraphAdjModelV2.forward(self, pattern, graph)
        finally:
            self._perm = None

    def get_pattern_rep(self, pattern, p_v_emb, p_e_emb, *perm, **kw):
        perm = perm or self._perm[id(pattern)]
        return _PermRepMixin.get_pattern_rep(self, pattern, p_v_emb, p_e_emb, *perm, **kw)

    def get_graph_rep(self, graph, g_v_emb, g_e_emb, *perm, **kw):
        perm = perm or self._perm[id(graph)]
        return _PermRepMixin.get_graph_rep(self, graph, g_v_emb, g_e_emb, *perm, **kw)


class LRP(_PermModel):
    rep_key, residual_sum = "lrp", False

    def _make_layer(self, **kw):
        return LRPLayer(self.hid_dim, self.hid_dim, lrp_seq_len=kw.get("lrp_seq_len", 4), batch_norm=kw.get("rep_lrp_batch_norm", False),
                        act_func=kw.get("rep_act_func", "relu"), dropout=kw.get("rep_dropout", 0.0))


class DMPLRP(_PermModel):
    rep_key, residual_sum = "DMPLRP", True

    def _make_layer(self, **kw):
        return DMPLRPPoolLayer(self.hid_dim, self.hid_dim, init_neigenv=kw.get("init_neigenv", 4.0), init_eeigenv=kw.get("init_eeigenv", 4.0),
                               lrp_seq_len=kw.get("lrp_seq_len", 4), num_mlp_layers=kw.get("rep_dmpnn_num_mlp_layers", 2),
                               batch_norm=kw.get("rep_dmpnn_batch_norm", False), act_func=kw.get("rep_act_func", "relu"),
                               dropout=kw.get("rep_dropout", 0.0))

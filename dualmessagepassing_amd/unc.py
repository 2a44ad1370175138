"""UnsupervisedNodeClassification twin of the DMPLayer on the MI355X kernels.

Drop-in for ``DualGraphConv`` (UnsupervisedNodeClassification/Model/DMPNN/src/model.py:117-281)
plus the two graph helpers that define its inputs, ``compute_edgenorm`` and
``build_graph_from_triplets`` (.../src/utils.py:437-453,473-491).

Differences from the SCM layer, all preserved:
  * node messages are scaled by ``edata["norm"]`` when an ``edge_norm`` is passed (model.py:234-235);
  * frame keys are "h" / "out_deg" / "norm" / "is_rev" (model.py:207-221,228,234);
  * MLP = Linear -> BN -> act -> Linear with act = LeakyReLU(1/5.5) unless an ``activation`` module
    is given, which is then ALSO applied to the outputs (model.py:145-165,247-248,262-263);
  * dropout is called but its result discarded (model.py:245,260) -> no effect;
  * ``nfc`` / ``efc`` exist but are never used (model.py:137-138) -> their grads stay None.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from ._lib import on_input_device
from .constants import OUTDEGREE, REVFLAG
from .dmpnn import dual_message_passing
from .graph import BatchedGraph, as_batched, leave_detached
from .ops import PoolIndex, seg_pool


class DualGraphConv(nn.Module):
    def __init__(
            self,
            input_dim,
            hidden_dim,
            init_neigenv=4.0,
            init_eeigenv=4.0,
            bias=True,
            batch_norm=True,
            activation=None,
            dropout=0.0
    ):
        super(DualGraphConv, self).__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.in_weight = nn.Parameter(torch.empty(input_dim, hidden_dim))
        self.out_weight = nn.Parameter(torch.empty(input_dim, hidden_dim))
        self.src_weight = nn.Parameter(torch.empty(input_dim, hidden_dim))
        self.dst_weight = nn.Parameter(torch.empty(input_dim, hidden_dim))
        self.nloop_weight = nn.Parameter(torch.empty(input_dim, hidden_dim))
        self.eloop_weight = nn.Parameter(torch.empty(input_dim, hidden_dim))
        self.nfc = nn.Linear(hidden_dim, hidden_dim)
        self.efc = nn.Linear(hidden_dim, hidden_dim)
        if bias:
            self.nbias = nn.Parameter(torch.zeros((hidden_dim)))
            self.ebias = nn.Parameter(torch.zeros((hidden_dim)))
        else:
            self.register_parameter("nbias", None)
            self.register_parameter("ebias", None)

        def mlp():
            inner = nn.LeakyReLU(1 / 5.5) if activation is None else activation
            mods = [nn.Linear(hidden_dim, hidden_dim)]
            if batch_norm:
                mods.append(nn.BatchNorm1d(hidden_dim))
            mods += [inner, nn.Linear(hidden_dim, hidden_dim)]
            return nn.Sequential(*mods)

        self.nmlp = mlp()
        self.emlp = mlp()
        self.act = activation
        self.drop = nn.Dropout(dropout)

        # model.py:168-181
        for w in (self.in_weight, self.out_weight, self.src_weight, self.dst_weight, self.nloop_weight,
                  self.eloop_weight, self.nmlp[0].weight, self.nmlp[-1].weight, self.emlp[0].weight,
                  self.emlp[-1].weight):
            nn.init.xavier_uniform_(w)
        for b in (self.nmlp[0].bias, self.nmlp[-1].bias, self.emlp[0].bias, self.emlp[-1].bias):
            nn.init.zeros_(b)
        # model.py:185-191
        with torch.no_grad():
            self.in_weight.data.div_(init_neigenv)
            self.out_weight.data.div_(init_neigenv)
            self.nloop_weight.data.div_(init_neigenv)
            self.src_weight.data.div_(init_eeigenv)
            self.dst_weight.data.div_(init_eeigenv)
            self.eloop_weight.data.div_(init_eeigenv)

    @on_input_device
    def forward(self, graph, node_feat, edge_feat, edge_norm=None):
        g = as_batched(graph)   # DGLGraph-in (model.py:267): frames shared with the caller's graph
        # _node_init_func / _edge_init_func (model.py:207-221)
        g.ndata["h"] = node_feat
        if OUTDEGREE not in g.ndata:
            g.ndata[OUTDEGREE] = g.out_degrees()
        g.edata["h"] = edge_feat
        if edge_norm is not None:
            g.edata["norm"] = edge_norm
        if "is_rev" in g.edata and REVFLAG not in g.edata:
            g.edata[REVFLAG] = g.edata["is_rev"]  # the index reads the flag under the SCM key
        node_pre, edge_pre, _ = dual_message_passing(
            g, node_feat, edge_feat, self.in_weight, self.out_weight, self.src_weight, self.dst_weight,
            self.nloop_weight, self.eloop_weight, self.nbias, self.ebias, has_rev="is_rev" in g.edata,
            edge_norm=g.edata.get("norm"))
        # (a sub-graph padded to a capacity, ``PaddedRows`` on the graph object: BatchNorm over the real rows only)
        valid = getattr(graph, "_dmp_valid", None) or getattr(g, "_dmp_valid", None)
        node_out = ops.apply_mlp(self.nmlp, node_pre, None if valid is None else valid.n_dev)
        edge_out = ops.apply_mlp(self.emlp, edge_pre, None if valid is None else valid.e_dev)
        if self.act:
            node_out = self.act(node_out)
            edge_out = self.act(edge_out)
        leave_detached(g.ndata, "h")          # UNC trains on ONE graph object: nothing of a step may stay alive on it
        leave_detached(g.edata, "h")
        return node_out, edge_out

    def extra_repr(self):
        return "in=%s, out=%s," % (self.input_dim, self.hidden_dim)


class PaddedRows:
    """What a graph PADDED to fixed capacities carries under ``graph._dmp_valid`` (``unc_harness.SampledStep``: a sampled
    sub-graph with inert nodes and edges behind the real ones, so that its step replays from a recording): the real
    row counts on the device (int64 [1] each) and the 0 / 1 masks of the real rows.  Every reduction over the rows of the
    model -- BatchNorm statistics, the per-relation means, the regularisers' means -- runs over the real rows only."""

    def __init__(self, n_dev, e_dev, num_nodes, num_edges):
        self.n_dev, self.e_dev = n_dev.view(1), e_dev.view(1)
        dev = n_dev.device
        self.node_mask = (torch.arange(num_nodes, device=dev) < self.n_dev).to(torch.float32).unsqueeze(1)
        self.edge_bool = torch.arange(num_edges, device=dev) < self.e_dev
        self.edge_mask = self.edge_bool.to(torch.float32).unsqueeze(1)


def compute_edgenorm(g, norm="in"):
    """utils.py:437-453: per-edge ``1/in_deg[dst]`` (or out / both) as [E,1]; NaN and Inf
    entries are replaced by the minimum (in that order)."""
    in_deg = g.in_degrees().float()
    out_deg = g.out_degrees().float()
    u, v = g.all_edges(form="uv", order="eid")
    if norm == "in":
        w = in_deg[v].reciprocal().unsqueeze(-1)
    elif norm == "out":
        w = out_deg[u].reciprocal().unsqueeze(-1)
    elif norm == "both":
        w = torch.pow(out_deg[u] * in_deg[v], 0.5).reciprocal().unsqueeze(-1)
    else:
        raise ValueError(norm)
    w.masked_fill_(torch.isnan(w), w.min())
    w.masked_fill_(torch.isinf(w), w.min())
    return w


def build_graph_from_triplets(num_nodes, num_rels, triplets, device):
    """utils.py:473-491: sort the (src, rel, dst) triplets by (src, dst, rel), add E forward and E
    reversed edges, ``type`` = rel | rel + num_rels, ``norm`` = compute_edgenorm (1 / in-degree)."""
    t = np.asarray(triplets).copy()
    order = np.lexsort((t[:, 1], t[:, 2], t[:, 0]))  # keys: src, then dst, then rel
    t = t[order]
    src = np.concatenate([t[:, 0], t[:, 2]]).astype(np.int64)
    dst = np.concatenate([t[:, 2], t[:, 0]]).astype(np.int64)
    rel = np.concatenate([t[:, 1], t[:, 1] + num_rels]).astype(np.int64)
    g = BatchedGraph(torch.from_numpy(src).to(device), torch.from_numpy(dst).to(device), num_nodes)
    g.edata["type"] = torch.from_numpy(rel).to(device)
    g.edata["norm"] = compute_edgenorm(g)
    return g


class EmbeddingLayer(nn.Module):
    """UNC model.py:41-52: learned table, uniform(-1/sqrt(d), 1/sqrt(d))."""

    def __init__(self, num_emb, emb_dim):
        super(EmbeddingLayer, self).__init__()
        self.embedding = nn.Embedding(num_emb, emb_dim)
        scale = 1 / (emb_dim) ** 0.5
        nn.init.uniform_(self.embedding.weight, -scale, scale)

    def forward(self, g, x):
        w, ids = self.embedding.weight, x.squeeze()
        if w.is_cuda and w.requires_grad and torch.is_grad_enabled() and ids.dim() == 1:
            # the row gather with the kernels' backward (fixed-order segment sums: torch's embedding backward sorts the
            # indices with a library call that cannot be recorded in a HIP graph, and serialises on repeated indices)
            return ops.take_rows_small_table(w, ids) if w.size(0) <= 64 else ops.take_rows(w, ids)
        return self.embedding(ids)

    @property
    def weight(self):
        return self.embedding.weight


class EmbeddingLayerAttri(nn.Module):
    """UNC model.py:55-65: frozen pre-trained attributes."""

    def __init__(self, attri):
        super(EmbeddingLayerAttri, self).__init__()
        self.embedding = nn.Embedding.from_pretrained(torch.as_tensor(attri))

    def forward(self, g, x):
        return self.embedding(x.squeeze())

    @property
    def weight(self):
        return self.embedding.weight


class DMPNN(nn.Module):
    """UNC ``DMPNN(BaseModel)`` (model.py:68-115,281-328): node / relation embeddings ->
    ``num_hidden_layers`` x DualGraphConv (Tanh between layers, none after the last) -> per-relation
    mean of the edge representations.  ``forward(g, h, r, norm) -> (h, z, r_rep)``."""

    def __init__(self, node_attri, rel_attri, num_nodes, h_dim, out_dim, num_rels, num_hidden_layers=1,
                 dropout=0, use_cuda=False):
        super(DMPNN, self).__init__()
        self.num_nodes, self.h_dim, self.out_dim = num_nodes, h_dim, out_dim
        self.num_rels, self.num_hidden_layers, self.dropout, self.use_cuda = num_rels, num_hidden_layers, dropout, use_cuda
        self.node_emb = EmbeddingLayerAttri(node_attri) if node_attri is not None else EmbeddingLayer(num_nodes, h_dim)
        self.rel_emb = EmbeddingLayerAttri(rel_attri) if rel_attri is not None else EmbeddingLayer(num_rels, h_dim)
        self.layers = nn.ModuleList()
        for idx in range(num_hidden_layers):
            in_dim = h_dim if idx == 0 else out_dim
            act = nn.Tanh() if idx < num_hidden_layers - 1 else None
            self.layers.append(DualGraphConv(in_dim, out_dim, activation=act, dropout=dropout))

    @on_input_device
    def forward(self, g, h, r, norm):
        h = self.node_emb(g, h)
        z = self.rel_emb(g, r)
        for layer in self.layers:
            h, z = layer(g, h, z, norm)
        # model.py:319-325 (one masked full-size sum per relation type there): mean of the edge
        # representations per relation type as ONE keyed segment sum; the index is memoised per
        # relation tensor (UNC trains on one fixed graph)
        valid = getattr(g, "_dmp_valid", None)
        if valid is not None:      # a padded sub-graph: its inert edges under a key of their own, dropped from the result
            pool = PoolIndex.from_keys(torch.where(valid.edge_bool, r.view(-1), torch.full_like(r.view(-1), self.num_rels)), self.num_rels + 1)
            r_rep = (seg_pool(z, pool) / (pool.sizes.float().view(-1, 1) + 1e-8))[:self.num_rels]
            return h, z, r_rep
        pool = self._rel_pool(r)
        r_rep = seg_pool(z, pool) / (pool.sizes.float().view(-1, 1) + 1e-8)
        return h, z, r_rep

    def _rel_pool(self, r):
        key = (r.data_ptr(), r._version, int(r.numel()), r.device)
        cached = getattr(self, "_rel_pool_cache", None)
        if cached is not None and not ops._memo_usable(r):
            cached = None                                      # being recorded and not marked immutable (ops.mark_immutable)
        if cached is None or cached[0] != key:
            self._rel_pool_cache = cached = (key, PoolIndex.from_keys(r, self.num_rels), r)  # r kept alive: ptr stays unique
        return cached[1]


class TrainModel(nn.Module):
    """UNC ``TrainModel`` (model.py:631-744): the DMPNN encoder with ``2 * num_rels`` edge types plus
    the link-prediction (DistMult score) or node-classification head and their regularisers."""

    def __init__(self, node_attri, num_nodes, o_dim, num_rels, nlabel, num_hidden_layers=1, dropout=0,
                 use_cuda=False, reg_param=0):
        super(TrainModel, self).__init__()
        i_dim = o_dim if node_attri is None else node_attri.shape[1]
        self.model = DMPNN(node_attri, None, num_nodes, i_dim, o_dim, num_rels * 2, num_hidden_layers, dropout, use_cuda)
        self.reg_param = reg_param
        if nlabel == 0:
            self.supervised = False
            self.w_relation = nn.Parameter(torch.Tensor(num_rels, o_dim))
            nn.init.xavier_uniform_(self.w_relation, gain=nn.init.calculate_gain("relu"))
        else:
            self.supervised = True
            self.node_fc = nn.Linear(o_dim, nlabel)
            nn.init.xavier_uniform_(self.node_fc.weight, gain=nn.init.calculate_gain("sigmoid"))
            nn.init.zeros_(self.node_fc.bias)
        self.edge_fc = nn.Linear(o_dim, o_dim)
        nn.init.xavier_uniform_(self.edge_fc.weight, gain=nn.init.calculate_gain("sigmoid"))
        nn.init.zeros_(self.edge_fc.bias)

    def calc_score(self, embedding, triplets):
        node_emb = embedding[0] if isinstance(embedding, (tuple, list)) else embedding
        # model.py:669-677; the two endpoint lookups as ONE row gather whose backward is a segment sum
        so = ops.take_rows(node_emb, torch.cat([triplets[:, 0], triplets[:, 2]]), key=triplets, tag="subject|object")
        s, o = so[:triplets.size(0)], so[triplets.size(0):]
        r = ops.take_rows_small_table(self.w_relation, triplets[:, 1])
        return torch.sum(s * r * o, dim=1)

    @on_input_device
    def forward(self, g, h, edge_type, edge_norm):
        output = self.model.forward(g, h, edge_type, edge_norm)
        pred = None
        if self.supervised:
            pred = self.node_fc(output[0] if isinstance(output, (tuple, list)) else output)
        return output, pred

    def unsupervised_regularization_loss(self, embedding, edge_type=None, valid=None):
        # model.py:692-714 (needs ``w_relation``: as in the reference, only defined for nlabel == 0)
        # ``valid`` (``PaddedRows``): the node and edge representations (the first two entries) carry padding rows -- means
        # over the real rows only
        reg = torch.mean(self.w_relation.pow(2))
        embs = list(embedding) if isinstance(embedding, (tuple, list)) else [embedding]
        for k, emb in enumerate(embs):
            if valid is not None and k < 2:
                m, n = (valid.node_mask, valid.n_dev) if k == 0 else (valid.edge_mask, valid.e_dev)
                reg = reg + (emb.pow(2) * m).sum() / (n.to(emb.dtype) * emb.size(1)).squeeze()
                continue
            reg = reg + torch.mean(emb.pow(2))
        if edge_type is not None:
            for emb in embs:
                if emb.size(0) == edge_type.size(0):
                    # model.py:703-707 selects the rows with ``emb[mask]`` (a host sync for the row count and a
                    # serialised indexing backward); same mean as a mask-weighted sum over all rows
                    num_rels = self.w_relation.size(0)
                    # per edge-type tensor: the mask and the clamped types (a fresh clamp result every step would also
                    # defeat the keyed pool index's memo: ~25 launches of sorting per step on the one fixed graph)
                    ck = (edge_type.data_ptr(), edge_type._version, int(edge_type.numel()), emb.dtype)
                    cached = getattr(self, "_etype_cache", None)
                    if cached is not None and not ops._memo_usable(edge_type):
                        cached = None                          # being recorded and not marked immutable: rebuilt in the recording
                    if cached is None or cached[0] != ck:
                        cached = self._etype_cache = (ck, (edge_type < num_rels).to(emb.dtype).unsqueeze(1),
                                                      edge_type.clamp(max=num_rels - 1), edge_type)
                        if ops.is_immutable(edge_type):       # what is derived from a fixed tensor is fixed
                            ops.mark_immutable(cached[1], cached[2])
                    mask, clamped = cached[1], cached[2]
                    if valid is not None:
                        mask = mask * valid.edge_mask
                    emb_diff = self.edge_fc(emb) - ops.take_rows_small_table(self.w_relation, clamped)
                    reg = reg + (torch.pow(emb_diff, 2) * mask).sum() / (mask.sum() * emb_diff.size(1))
        return reg

    def get_unsupervised_loss(self, g, embedding, edge_type, triplets, labels):
        score = self.calc_score(embedding, triplets)
        predict_loss = F.binary_cross_entropy_with_logits(score, labels)
        return predict_loss + self.reg_param * self.unsupervised_regularization_loss(embedding, edge_type=edge_type,
                                                                                    valid=getattr(g, "_dmp_valid", None))

    def supervised_regularization_loss(self, embedding, edge_type=None):
        return self.unsupervised_regularization_loss(embedding, edge_type=edge_type)

    def get_supervised_loss(self, g, embedding, edge_type, pred, matched_labels, matched_index, multi):
        if multi:
            predict_loss = F.binary_cross_entropy(torch.sigmoid(pred[matched_index]), matched_labels)
        else:
            predict_loss = F.nll_loss(F.log_softmax(pred[matched_index], dim=1), matched_labels)
        return predict_loss + self.reg_param * self.supervised_regularization_loss(embedding, edge_type=edge_type)

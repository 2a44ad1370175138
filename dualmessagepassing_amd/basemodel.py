"""Model skeleton around the hot path: enc -> filter -> emb -> rep -> pred, on the device.

Mirror of ``BaseModel`` / ``GraphAdjModelV2`` (SubgraphCountingMatching/models/basemodel.py:15-160,
965-1663) for the configurations the DMPNN / CompGCN training commands use (Multihot / Position
encodings, the four embedding kinds, ``ScalarFilter``, Mean / Sum / Max pooling heads).  Same
constructor keywords, sub-module names (hence ``state_dict`` keys) and ``forward(pattern, graph)
-> OutputDict`` with the reference's 15 keys.

The reference's per-sample Python loops (``utils/dl.py:51-81,113-127``, ``basemodel.py:1411,1421``)
are replaced by index arithmetic on the device; uniform-size batches take the same reshape fast
paths as the reference.  ``model.expand(**config)`` grows the vocabulary-dependent parts for fine-tuning.
"""
from collections import OrderedDict

import torch as th
import torch.nn as nn

from . import ops
from ._lib import on_input_device
from .compgcn import CompGCNRepMixin
from .embed import materialize
from .constants import REVFLAG
from .graph import as_batched
from .dmpnn import DMPNNRepMixin
from .embed import (EquivariantEmbedding, MultihotEmbedding, NormalEmbedding, OrthogonalEmbedding,
                    PositionEmbedding, UniformEmbedding, get_enc_len, lookup_rows)
from .pred import PRED_NETS
from .rgnn import RGCNRepMixin, RGINRepMixin


import os as _os
# the order in which the forward pass ISSUES the forked index builds and the encoding / embedding kernels (both depend on the
# gates only): a development switch for measuring how a replayed HIP graph schedules its two branches
PREFETCH_AFTER_EMBEDDINGS = _os.environ.get("DMP_DEV_PREFETCH_LATE") == "1"

class OutputDict(OrderedDict):
    """Ordered mapping with attribute access (container.py:14-100, the part callers use).  An entry may be deferred
    (``embed.DeferredEmbedding``: the target embeddings, which the joint rep-net pass does not need as tensors): it becomes
    its tensor the first time it is read, by whatever accessor.  Likewise ``embed.DeferredRows`` (the last layer's edge rows
    under pooling heads)."""

    def _resolve(self, k, v):
        from .embed import DeferredEmbedding, DeferredRows
        if isinstance(v, (DeferredEmbedding, DeferredRows)):
            v = v.materialize()
            OrderedDict.__setitem__(self, k, v)
        return v

    def __getitem__(self, k):
        return self._resolve(k, OrderedDict.__getitem__(self, k))

    def get(self, k, default=None):
        return self[k] if k in self else default

    def pop(self, k, *default):
        if k in self:
            v = self[k]                                    # a deferred entry leaves the dictionary as its tensor
            OrderedDict.__delitem__(self, k)
            return v
        if default:
            return default[0]
        raise KeyError(k)

    def popitem(self, last=True):
        k = next(reversed(self.keys())) if last else next(iter(self.keys()))
        return k, self.pop(k)

    def setdefault(self, k, default=None):
        if k not in self:
            OrderedDict.__setitem__(self, k, default)
        return self[k]

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


class _Blend(th.autograd.Function):
    """``wa * a + wb * b`` with constant weights (basemodel.py:1489-1492: the node / edge head predictions weighted by
    the share of target nodes / edges): one autograd node, two launches each way."""

    @staticmethod
    def forward(ctx, a, b, wa, wb):
        ctx.save_for_backward(wa, wb)
        return th.addcmul(wa * a, wb, b)

    @staticmethod
    def backward(ctx, d):
        wa, wb = ctx.saved_tensors
        return wa * d, wb * d, None, None


class ScalarFilter(nn.Module):
    """filter.py:6-16: gate[b, j] = any_i (g_x[b, j] == p_x[b, i])."""

    def forward(self, p_x, g_x):
        matrix = g_x.unsqueeze(2) - p_x.unsqueeze(1)
        return th.max(matrix == 0, dim=2)[0]


def scalar_filter_gate(p_pad, p_label, g_pad, g_label, num_labels):
    """``ScalarFilter`` on the pre-padded label matrices (filter.py:10-16 after
    ``split_and_batchify_graph_feats(..., pre_pad=True)``, basemodel.py:1394-1423), evaluated per row
    instead of as a ``[B, g_len, p_len]`` comparison cube: a row of the target passes iff its label
    occurs among the labels of the pattern of the same pair -- or is 0 and that pattern is shorter
    than the longest one of the batch (its pre-padding zeros take part in the reference's comparison)."""
    B, L = p_pad.bsz, int(num_labels)
    present = th.zeros(B * L, dtype=th.bool, device=p_label.device)
    present.index_fill_(0, p_pad.seg.long() * L + p_label.view(-1), True)
    if not p_pad.uniform:
        present.view(B, L)[:, 0] |= p_pad.sizes < p_pad.max
    return present.index_select(0, g_pad.seg.long() * L + g_label.view(-1)).view(-1, 1)


_FILTER_JOB = None


def scalar_filter_gates(jobs):
    """``scalar_filter_gate`` for several element kinds at once -- ``jobs`` = list of ``(p_pad, p_label, g_pad, g_label,
    num_labels)`` -- as float ``[rows, 1]`` gates.  Device tensors: three dispatches for all jobs together
    (csrc/dmp_graph.hip::dmp_scalar_filter_gates) instead of nine small torch launches per kind."""
    global _FILTER_JOB
    import ctypes
    from . import _lib
    lib = _lib.load()
    _lib.require_gpu(*[t for j in jobs for t in (j[1], j[3])])   # no host implementation: device tensors only
    if _FILTER_JOB is None:
        class _Job(ctypes.Structure):
            _fields_ = [("p_seg", ctypes.c_void_p), ("p_label", ctypes.c_void_p), ("num_p", ctypes.c_int64),
                        ("p_sizes", ctypes.c_void_p), ("p_max", ctypes.c_int64),
                        ("g_seg", ctypes.c_void_p), ("g_label", ctypes.c_void_p), ("num_g", ctypes.c_int64),
                        ("num_labels", ctypes.c_int64), ("present_off", ctypes.c_int64), ("gate", ctypes.c_void_p),
                        ("seg_is_i32", ctypes.c_int64)]
        _FILTER_JOB = _Job
    dev = jobs[0][1].device
    B = jobs[0][0].bsz
    J, keep, off, gates = (_FILTER_JOB * len(jobs))(), [], 0, []
    for i, (p_pad, p_label, g_pad, g_label, L) in enumerate(jobs):
        if p_pad.bsz != B or g_pad.bsz != B:
            raise ValueError("filter jobs must cover the same pairs")
        L = int(L)
        ps, pl = p_pad.seg.contiguous(), p_label.reshape(-1).long().contiguous()
        gs, gl = g_pad.seg.contiguous(), g_label.reshape(-1).long().contiguous()
        if ps.dtype != gs.dtype or ps.dtype not in (th.int32, th.int64):
            ps, gs = ps.long(), gs.long()
        sizes = None if p_pad.uniform else p_pad.sizes.long().contiguous()
        gate = th.empty((gl.numel(), 1), dtype=th.float32, device=dev)
        J[i].p_seg, J[i].p_label, J[i].num_p = ps.data_ptr(), pl.data_ptr(), pl.numel()
        J[i].p_sizes, J[i].p_max = (None if sizes is None else sizes.data_ptr()), int(p_pad.max)
        J[i].g_seg, J[i].g_label, J[i].num_g = gs.data_ptr(), gl.data_ptr(), gl.numel()
        J[i].num_labels, J[i].present_off, J[i].gate = L, off, gate.data_ptr()
        J[i].seg_is_i32 = 1 if ps.dtype == th.int32 else 0
        off += (B * L + 15) // 16 * 16
        keep.append((ps, pl, gs, gl, sizes))
        gates.append(gate)
    present = th.empty(max(off, 1), dtype=th.uint8, device=dev)
    _lib.check(lib.dmp_scalar_filter_gates(J, len(jobs), B, present.data_ptr(), off, _lib.stream_ptr()),
               "dmp_scalar_filter_gates")
    for g in gates:
        g._dmp_binary = True          # exactly 0.0 / 1.0: kernels may treat such a gate as a row mask
    return gates


_MASK_JOB = None


def len_masks(jobs):
    """Pre-padding masks ``[B, max, 1]`` (bool) and their row counts ``[B, 1]`` (float) for several element kinds --
    ``jobs`` = list of ``(padder, rev)`` with ``rev`` the per-row is-reversed flags or None (reversed edges leave
    the mask, basemodel.py:1521-1531).  Device tensors: ONE launch for all kinds (csrc/dmp_graph.hip::len_masks_k)
    instead of the arange / compare / masked_fill / sum chain per kind."""
    global _MASK_JOB
    import ctypes
    from . import _lib
    lib = _lib.load()
    _lib.require_gpu(*[p.sizes for p, _ in jobs])            # no host implementation: device tensors only
    if _MASK_JOB is None:
        class _Job(ctypes.Structure):
            _fields_ = [("sizes", ctypes.c_void_p), ("off", ctypes.c_void_p), ("max_len", ctypes.c_int64),
                        ("rev", ctypes.c_void_p), ("mask", ctypes.c_void_p), ("count", ctypes.c_void_p)]
        _MASK_JOB = _Job
    B, dev = jobs[0][0].bsz, jobs[0][0].sizes.device
    J, keep, out = (_MASK_JOB * len(jobs))(), [], []
    for i, (p, rev) in enumerate(jobs):
        if p.bsz != B:
            raise ValueError("mask jobs must cover the same graphs")
        sizes = p.sizes.long().contiguous()
        off = None if p.uniform else p.off.contiguous()
        if rev is not None:
            rev = rev.reshape(-1).contiguous()
            rev = rev.view(th.uint8) if rev.element_size() == 1 else (rev != 0).view(th.uint8)
        mask = th.empty((B, p.max, 1), dtype=th.bool, device=dev)
        count = th.empty((B, 1), dtype=th.float32, device=dev)
        J[i].sizes, J[i].off, J[i].max_len = sizes.data_ptr(), (None if off is None else off.data_ptr()), p.max
        J[i].rev, J[i].mask, J[i].count = (None if rev is None else rev.data_ptr()), mask.data_ptr(), count.data_ptr()
        keep.append((sizes, off, rev))
        out.append((mask, count))
    _lib.check(lib.dmp_len_masks(J, len(jobs), B, _lib.stream_ptr()), "dmp_len_masks")
    return out


# ----------------------------------------------------------------------------- padding helpers
def _segments(graph, kind):
    seg = graph.node_graph if kind == "node" else graph.edge_graph
    sizes = graph.batch_num_nodes() if kind == "node" else graph.batch_num_edges()
    if seg is None:
        seg = th.repeat_interleave(th.arange(sizes.numel(), device=sizes.device), sizes)
    return seg, sizes      # int32 from the device collate, int64 otherwise: the consumers take both


def _max_len(sizes):
    return int(sizes.max().item()) if sizes.numel() else 0


def len_to_mask(lens, max_len):
    """``batch_convert_len_to_mask(lens, pre_pad=True)`` (utils/dl.py:113-127), vectorised."""
    return th.arange(max_len, device=lens.device).unsqueeze(0) >= (max_len - lens).unsqueeze(1)


class _Padder:
    """``split_and_batchify_graph_feats(x, sizes, pre_pad=True)[0]`` (utils/dl.py:51-81) for one
    graph and one element kind: rows of graph i go to ``[i, max - size_i :]``, zeros in front."""

    def __init__(self, graph, kind):
        self.seg, self.sizes = _segments(graph, kind)
        self.bsz = int(self.sizes.numel())
        # the padded length: from the collate when the dataset supplied it (host int), else one device round trip
        hint = getattr(graph, "max_num_nodes" if kind == "node" else "max_num_edges", None)
        self.max = int(hint) if hint is not None else _max_len(self.sizes)
        n = int(self.seg.numel())
        self.uniform = self.bsz * self.max == n
        if not self.uniform:
            off = th.zeros(self.bsz + 1, dtype=th.int64, device=self.sizes.device)
            th.cumsum(self.sizes, 0, out=off[1:])
            self.off = off[:-1]
            seg = self.seg.long()
            pos = th.arange(n, device=self.sizes.device) - off[seg]
            self.idx = seg * self.max + (self.max - self.sizes[seg]) + pos

    def pad(self, x):
        x = x.reshape(x.size(0), -1)
        if self.uniform:
            return x.view(self.bsz, self.max, -1)
        out = th.zeros((self.bsz * self.max, x.size(1)), dtype=x.dtype, device=x.device)
        return out.index_copy(0, self.idx, x).view(self.bsz, self.max, -1)

    def unpad(self, y):
        """inverse gather: [B, max, ...] -> rows in graph order (basemodel.py:1411,1421)."""
        y = y.reshape(self.bsz * self.max, -1)
        return y if self.uniform else y[self.idx]

    def mask(self):
        return len_to_mask(self.sizes, self.max).view(self.bsz, -1, 1)


# Size-derived helpers (padding maps, pooling indexes) depend only on the per-graph size vectors
# (and the reversed-edge flags).  Building them needs the padded length on the host (a device
# sync, like the reference's ``graph_sizes.max()``), so they are memoised on the identity of those
# tensors: a loader that hands over the same size tensors again (fixed-shape batches, repeated
# epochs) pays the sync once and the step stays asynchronous.
_SIZE_CACHE = OrderedDict()
_SIZE_CACHE_MAX = 12   # a loader that makes new size tensors for every batch never hits: keep the retained set (and the
                       # allocator growth it causes, a dozen device allocations over the first steps) small


def _tensor_key(t):
    return None if t is None else (t.data_ptr(), t._version, t.numel())


def _memo(key, build):
    hit = _SIZE_CACHE.get(key)
    if hit is not None:
        _SIZE_CACHE.move_to_end(key)
        return hit[0]
    val = build()
    _SIZE_CACHE[key] = (val,)
    if len(_SIZE_CACHE) > _SIZE_CACHE_MAX:
        _SIZE_CACHE.popitem(last=False)
    return val


def _pool_index(graph, kind, skip_reversed=True):
    if kind == "node":
        sizes, flag = graph.batch_num_nodes(), None
    else:
        sizes, flag = graph.batch_num_edges(), (graph.edata.get(REVFLAG) if skip_reversed else None)
    # the cached object keeps its source tensors alive, so a data_ptr cannot be recycled under it
    rows = graph.number_of_nodes() if kind == "node" else graph.number_of_edges()      # host ints: no sync in the build
    return _memo(("pool", _tensor_key(sizes), _tensor_key(flag)),
                 lambda: _Keep(ops.PoolIndex(sizes, flag, num_rows=rows), sizes, flag)).obj


def _pool_union_spec(pattern, graph, kind, skip_reversed):
    if kind == "node":
        a, b, fa, fb = pattern.batch_num_nodes(), graph.batch_num_nodes(), None, None
    else:
        a, b = pattern.batch_num_edges(), graph.batch_num_edges()
        fa, fb = (pattern.edata.get(REVFLAG), graph.edata.get(REVFLAG)) if skip_reversed else (None, None)
    rows = (pattern.number_of_nodes() + graph.number_of_nodes() if kind == "node"
            else pattern.number_of_edges() + graph.number_of_edges())                     # host ints: no sync in the build
    key = ("upool", _tensor_key(a), _tensor_key(b), _tensor_key(fa), _tensor_key(fb))
    return key, ((a, b), None if fa is None or fb is None else (fa, fb), rows), (a, b, fa, fb)


def _pool_index_union(pattern, graph, kind, skip_reversed=True):
    """PoolIndex over [pattern graphs | target graphs] (2B segments) of the union row order."""
    return _pool_indexes_union(pattern, graph, (kind,), skip_reversed)[0]


def _pool_indexes_union(pattern, graph, kinds, skip_reversed=True):
    """``_pool_index_union`` for several element kinds: the indexes that are not memoised yet are built together (sizes /
    flags as (pattern, target) pairs, on the device without concatenating them first: one pair of launches for all)."""
    specs = [_pool_union_spec(pattern, graph, k, skip_reversed) for k in kinds]
    missing = [i for i, (key, _, _) in enumerate(specs) if key not in _SIZE_CACHE]
    if missing:
        if all(th.is_tensor(specs[i][2][0]) and specs[i][2][0].is_cuda for i in missing):
            built = ops.pool_indexes([specs[i][1] for i in missing])
        else:
            built = [ops.PoolIndex(specs[i][1][0], specs[i][1][1], num_rows=specs[i][1][2]) for i in missing]
        for i, obj in zip(missing, built):
            _memo(specs[i][0], lambda obj=obj, i=i: _Keep(obj, *specs[i][2]))
    return [_memo(key, None).obj for key, _, _ in specs]


class _Keep:
    def __init__(self, obj, *tensors):
        self.obj, self.tensors = obj, tensors


def _padder(graph, kind):
    sizes = graph.batch_num_nodes() if kind == "node" else graph.batch_num_edges()
    seg = graph.node_graph if kind == "node" else graph.edge_graph
    if seg is None:  # no collate segments: nothing stable to key on
        return _Padder(graph, kind)
    return _memo(("pad", _tensor_key(sizes)), lambda: _Keep(_Padder(graph, kind), sizes)).obj


def _expanded(comp, rows_c):
    """The target's [E, H] edge rows from the compacted batch's (``collate.CompactedEdges.expand``); rows that have not
    been formed (``embed.DeferredRows``) stay unformed until somebody reads them."""
    from .embed import DeferredRows
    if isinstance(rows_c, DeferredRows):
        return DeferredRows(lambda: comp.expand(rows_c.materialize()), (comp.num_edges, rows_c.size(1)), rows_c.dtype, rows_c.device)
    return comp.expand(rows_c)


class BaseModel(nn.Module):
    """basemodel.py:15-160."""

    # vocabulary sizes every configuration must name, and the optional switches with the reference's defaults
    # (basemodel.py:25-41): the attribute names are part of the interface (expand(), the drivers and the mixins read them)
    REQUIRED = ("max_ngv", "max_ngvl", "max_nge", "max_ngel", "max_npv", "max_npvl", "max_npe", "max_npel")
    OPTIONAL = (("base", 2), ("hid_dim", 64), ("share_emb_net", True), ("share_enc_net", True), ("share_rep_net", True),
                ("rep_residual", True), ("pred_with_enc", False), ("pred_with_deg", False))
    # sub-networks in the order the reference registers them (``state_dict`` order, seeded-init RNG order); the pattern
    # side follows the graph side because it may alias it (share_*_net)
    STAGES = (("enc", True), ("filter", False), ("emb", True), ("rep", True), ("pred", False))

    def __init__(self, **kw):
        super(BaseModel, self).__init__()
        missing = [k for k in self.REQUIRED if k not in kw]
        if missing:
            raise KeyError("model configuration lacks %s" % ", ".join(missing))
        for name in self.REQUIRED:
            setattr(self, name, kw[name])
        for name, default in self.OPTIONAL:
            setattr(self, name, kw.get(name, default))
        for stage, two_sided in self.STAGES:
            make = getattr(self, "create_%s_net" % stage)
            if two_sided:
                setattr(self, "g_%s_net" % stage, make(type="graph", **kw))
                setattr(self, "p_%s_net" % stage, make(type="pattern", **kw))
            else:
                setattr(self, "%s_net" % stage, make(**kw))

    def refine_node_weights(self, weights, use_max=False):
        return weights

    def refine_edge_weights(self, weights, use_max=False):
        return weights

    def expand(self, **kw):
        """basemodel.py:167-219: grow the vocabulary-dependent parts (encodings, embeddings and, with
        ``pred_with_enc``, the heads) to larger ``max_*`` sizes; trained weights land in the trailing
        corner of the new, otherwise zero, tensors (multi-hot codes are right-aligned)."""
        if "base" in kw and kw["base"] != self.base:
            raise ValueError("expand: base must not change")
        kw = dict(kw)
        keys = ["max_npv", "max_npvl", "max_npe", "max_npel", "max_ngv", "max_ngvl", "max_nge", "max_ngel"]
        bak = {k: getattr(self, k) for k in keys}
        for k in keys:
            setattr(self, k, max(kw.get(k, -1), bak[k]))
        old = {k: getattr(self, k) for k in ("g_enc_net", "p_enc_net", "filter_net", "g_emb_net", "p_emb_net", "pred_net")}
        try:
            dev = next(self.parameters()).device
            self.g_enc_net = self.create_enc_net(type="graph", **kw).to(dev)
            self.p_enc_net = self.g_enc_net if self.share_enc_net else self.create_enc_net(type="pattern", **kw).to(dev)
            new_filter = self.create_filter_net(**kw)
            if new_filter is not None:
                expand_dimensions(old["filter_net"], new_filter.to(dev), pre_pad=True)
            self.filter_net = new_filter
            new_g_emb = self.create_emb_net(type="graph", **kw).to(dev)
            expand_dimensions(old["g_emb_net"], new_g_emb, pre_pad=True)
            self.g_emb_net = new_g_emb
            if self.share_emb_net:
                self.p_emb_net = self.g_emb_net
            else:
                new_p_emb = self.create_emb_net(type="pattern", **kw).to(dev)
                expand_dimensions(old["p_emb_net"], new_p_emb, pre_pad=True)
                self.p_emb_net = new_p_emb
            if self.pred_with_enc:
                new_pred = self.create_pred_net(**kw).to(dev)
                expand_dimensions(old["pred_net"], new_pred, pre_pad=True)
                self.pred_net = new_pred
        except Exception:
            for k, v in bak.items():
                setattr(self, k, v)
            for k, v in old.items():
                setattr(self, k, v)
            raise


def expand_dimensions(old_module, new_module, pre_pad=True):
    """utils/dl.py:157-191: zero the new tensors and copy the old ones into their trailing
    (``pre_pad``) or leading corner, parameter by parameter of equal name."""
    with th.no_grad():
        if isinstance(old_module, th.Tensor):
            new_module.zero_()
            idx = tuple(slice(-n, None) if pre_pad else slice(0, n) for n in old_module.size())
            new_module[idx].copy_(old_module)
            return
        old_params = dict(old_module.named_parameters())
        for name, param in new_module.named_parameters():
            if name in old_params:
                expand_dimensions(old_params[name], param, pre_pad)


class GraphAdjModel(BaseModel):
    """basemodel.py:619-962: the node-only skeleton (RGCN / RGIN): vertex id + label encodings,
    vertex-label filter gate, one pooling head; edge outputs are None."""

    def __init__(self, **kw):
        self.add_node_id = kw.get("add_node_id", kw.get("gnn_add_node_id", False))
        super(GraphAdjModel, self).__init__(**kw)

    def _enc_set(self, enc_net, nv, nvl):
        if enc_net == "Multihot":
            return OrderedDict({"v": MultihotEmbedding(nv, self.base), "vl": MultihotEmbedding(nvl, self.base)})
        if enc_net == "Position":
            return OrderedDict({"v": PositionEmbedding(int(get_enc_len(nv - 1, self.base)) * self.base, nv),
                                "vl": PositionEmbedding(int(get_enc_len(nvl - 1, self.base)) * self.base, nvl)})
        raise NotImplementedError(enc_net)

    def create_enc_net(self, type, **kw):
        enc_net = kw.get("enc_net", "Multihot")
        if type == "graph":
            nets = self._enc_set(enc_net, self.max_ngv, self.max_ngvl)
        elif type == "pattern":
            if self.share_enc_net:
                return self.g_enc_net
            nets = self._enc_set(enc_net, self.max_npv, self.max_npvl)
        else:
            raise ValueError(type)
        for net in nets.values():
            net.weight.requires_grad = False
        return nn.ModuleDict(nets)

    def create_filter_net(self, **kw):
        filter_net = kw.get("filter_net", "None")
        if filter_net == "None":
            return None
        if filter_net == "ScalarFilter":
            return nn.ModuleDict({"vl": ScalarFilter()})
        raise ValueError(filter_net)

    def create_emb_net(self, type, **kw):  # basemodel.py:69-91 (no rescaling in this skeleton)
        emb_net = kw.get("emb_net", "Orthogonal")
        dims = self.get_graph_enc_dims() if type == "graph" else self.get_pattern_enc_dims()
        cls = {"Orthogonal": OrthogonalEmbedding, "Normal": NormalEmbedding, "Uniform": UniformEmbedding,
               "Equivariant": EquivariantEmbedding}.get(emb_net)
        if cls is None:
            raise ValueError(emb_net)
        return nn.ModuleDict(OrderedDict({k: cls(v, self.hid_dim) for k, v in dims.items()}))

    def create_pred_net(self, **kw):
        name = kw.get("pred_net", "SumPredictNet")
        if name not in PRED_NETS:
            raise NotImplementedError("pred_net=%s is outside the MI355X hot-path scope" % name)
        return PRED_NETS[name](self.get_rep_dim(), hidden_dim=kw.get("pred_hid_dim", 64),
                               act_func=kw.get("pred_act_func", "relu"), dropout=kw.get("pred_dropout", 0.0),
                               return_weights="node" in kw.get("pred_return_weights", "none"))

    def _enc_dims(self, nv, nvl):
        return OrderedDict({"v": int(get_enc_len(nv - 1, self.base)) * self.base,
                            "vl": int(get_enc_len(nvl - 1, self.base)) * self.base})

    def get_graph_enc_dims(self):
        return self._enc_dims(self.max_ngv, self.max_ngvl)

    def get_pattern_enc_dims(self):
        return self.get_graph_enc_dims() if self.share_enc_net else self._enc_dims(self.max_npv, self.max_npvl)

    def get_rep_dim(self):  # basemodel.py:117-123
        rep_dim = self.hid_dim
        if self.pred_with_enc:
            rep_dim += sum(self.get_graph_enc_dims().values())
        if self.pred_with_deg:
            rep_dim += 2
        return rep_dim

    def get_filter_gate(self, pattern, graph, pv, gv):
        if self.filter_net is None or len(self.filter_net) == 0:
            return None
        if type(self.filter_net["vl"]) is ScalarFilter:
            return scalar_filter_gates([(pv, pattern.ndata["label"], gv, graph.ndata["label"],
                                         max(self.max_ngvl, self.max_npvl))])[0]
        p_vl = pv.pad(pattern.ndata["label"].view(-1, 1))
        g_vl = gv.pad(graph.ndata["label"].view(-1, 1))
        return gv.unpad(self.filter_net["vl"](p_vl, g_vl)).view(-1, 1)

    def _enc(self, net, g):
        v, vl = lookup_rows([net["v"], net["vl"]], [g.ndata["id"].view(-1), g.ndata["label"].view(-1)])
        return OrderedDict({"v": v, "vl": vl})

    def _emb(self, net, enc):
        emb = net["vl"](enc["vl"])
        if self.add_node_id:
            emb = emb + net["v"](enc["v"])
        return emb

    def get_subiso_pred(self, p_v_rep, p_v_mask, g_v_rep, g_v_mask):
        v_pred_c, v_pred_w = self.pred_net(p_v_rep, p_v_mask, g_v_rep, g_v_mask)
        return v_pred_c, (v_pred_w, None)

    @on_input_device
    def forward(self, pattern, graph):  # basemodel.py:877-962
        pattern, graph = as_batched(pattern), as_batched(graph)   # DGLGraph-in (train.py:606-611)
        bsz = pattern.batch_size
        pv, gv = _padder(pattern, "node"), _padder(graph, "node")
        p_v_mask, g_v_mask = pv.mask(), gv.mask()
        vl_gate = self.get_filter_gate(pattern, graph, pv, gv)
        if vl_gate is not None:
            vl_gate = vl_gate.float()
        p_enc = self._enc(self.p_enc_net, pattern)
        p_v_emb = self._emb(self.p_emb_net, p_enc)
        p_v_rep = self.get_pattern_rep(pattern, p_v_emb)
        g_enc = self._enc(self.g_enc_net, graph)
        g_v_emb = self._emb(self.g_emb_net, g_enc)
        g_v_rep = self.get_graph_rep(graph, g_v_emb, gate=vl_gate)

        p_add, g_add = [], []
        if self.pred_with_enc:
            p_add += [p_enc["v"], p_enc["vl"]]
            g_add += [g_enc["v"], g_enc["vl"]]
        if self.pred_with_deg:
            p_add += [pattern.out_degrees().float().view(-1, 1), pattern.in_degrees().float().view(-1, 1)]
            g_add += [graph.out_degrees().float().view(-1, 1), graph.in_degrees().float().view(-1, 1)]
        p_v_output = th.cat([self.refine_node_weights(th.cat(p_add, dim=-1)), p_v_rep], dim=-1) if p_add else p_v_rep
        g_v_output = th.cat([self.refine_node_weights(th.cat(g_add, dim=-1)), g_v_rep], dim=-1) if g_add else g_v_rep
        p_v_mask, g_v_mask = self.refine_node_weights(p_v_mask), self.refine_node_weights(g_v_mask)
        p_v_mask2, g_v_mask2 = p_v_mask.view(bsz, -1), g_v_mask.view(bsz, -1)
        if self.pred_net.poolable():  # pool-then-project on per-graph sums (see GraphAdjModelV2.forward)
            cnt = lambda m: m.float().sum(dim=1).view(-1, 1)
            pred_c, pred_v = self.pred_net.forward_pooled(
                ops.seg_pool(p_v_output, _pool_index(pattern, "node")), p_v_mask2.size(1), cnt(p_v_mask2),
                ops.seg_pool(g_v_output, _pool_index(graph, "node")), g_v_mask2.size(1), cnt(g_v_mask2))
        else:
            p_pad = pv.pad(p_v_output).masked_fill(~p_v_mask, 0)
            g_pad = gv.pad(g_v_output).masked_fill(~g_v_mask, 0)
            pred_c, (pred_v, _) = self.get_subiso_pred(p_pad, p_v_mask2, g_pad, g_v_mask2)
        return OutputDict(p_v_emb=p_v_emb, p_e_emb=None, g_v_emb=g_v_emb, g_e_emb=None,
                          p_v_rep=p_v_rep, p_e_rep=None, g_v_rep=g_v_rep, g_e_rep=None,
                          p_v_mask=p_v_mask2, p_e_mask=None, g_v_mask=g_v_mask2, g_e_mask=None,
                          pred_c=pred_c, pred_v=pred_v, pred_e=None)


class GraphAdjModelV2(BaseModel):
    """basemodel.py:965-1663."""

    # GraphAdjModelV2.forward takes the reversed edges out of the edge head's masks (basemodel.py:1521-1531); the LRP /
    # DMPLRP forwards (lrp.py:222-390, dmplrp.py:332-470) are copies of it WITHOUT that step
    edge_head_skips_reversed = True

    def __init__(self, **kw):
        self.add_node_id = kw.get("add_node_id", kw.get("gnn_add_node_id", False))
        self.add_edge_id = kw.get("add_edge_id", kw.get("gnn_add_edge_id", False))
        self.node_pred = kw.get("node_pred", True)
        self.edge_pred = kw.get("edge_pred", True)
        super(GraphAdjModelV2, self).__init__(**kw)

    # ---- construction (basemodel.py:973-1340)
    def _enc_set(self, enc_net, nv, nvl, nel):
        if enc_net == "Multihot":
            return OrderedDict({"v": MultihotEmbedding(nv, self.base), "vl": MultihotEmbedding(nvl, self.base),
                                "el": MultihotEmbedding(nel, self.base)})
        if enc_net == "Position":
            return OrderedDict({
                "v": PositionEmbedding(int(get_enc_len(nv - 1, self.base)) * self.base, nv),
                "vl": PositionEmbedding(int(get_enc_len(nvl - 1, self.base)) * self.base, nvl),
                "el": PositionEmbedding(int(get_enc_len(nel - 1, self.base)) * self.base, nel)})
        raise NotImplementedError(enc_net)

    def create_enc_net(self, type, **kw):
        enc_net = kw.get("enc_net", "Multihot")
        if type == "graph":
            nets = self._enc_set(enc_net, self.max_ngv, self.max_ngvl, self.max_ngel)
        elif type == "pattern":
            if self.share_enc_net:
                return self.g_enc_net
            nets = self._enc_set(enc_net, self.max_npv, self.max_npvl, self.max_npel)
        else:
            raise ValueError(type)
        for net in nets.values():
            net.weight.requires_grad = False
        return nn.ModuleDict(nets)

    def create_filter_net(self, **kw):
        filter_net = kw.get("filter_net", "None")
        if filter_net == "None":
            return None
        if filter_net == "ScalarFilter":
            return nn.ModuleDict({"vl": ScalarFilter(), "el": ScalarFilter()})
        raise ValueError(filter_net)

    def create_emb_net(self, type, **kw):
        emb_net = kw.get("emb_net", "Orthogonal")
        enc_dims = self.get_graph_enc_dims() if type == "graph" else self.get_pattern_enc_dims()
        cls = {"Orthogonal": OrthogonalEmbedding, "Normal": NormalEmbedding, "Uniform": UniformEmbedding,
               "Equivariant": EquivariantEmbedding}.get(emb_net)
        if cls is None:
            raise ValueError(emb_net)
        nets = OrderedDict({k: cls(enc_dims[k], self.hid_dim) for k in ("v", "vl", "el")})
        with th.no_grad():  # basemodel.py:1066-1070: rescale because of multi-hot inputs
            for k in nets:
                nets[k].weight.div_(enc_dims[k] // self.base)
        return nn.ModuleDict(nets)

    def create_pred_net(self, **kw):
        name = kw.get("pred_net", "SumPredictNet")
        if name not in PRED_NETS:
            raise NotImplementedError("pred_net=%s is outside the MI355X hot-path scope" % name)
        return_weights = kw.get("pred_return_weights", "none")
        rep_v_dim, rep_e_dim = self.get_rep_dim()
        args = dict(hidden_dim=kw.get("pred_hid_dim", 64), act_func=kw.get("pred_act_func", "relu"),
                    dropout=kw.get("pred_dropout", 0.0))
        return nn.ModuleDict({
            "v": PRED_NETS[name](rep_v_dim, return_weights="node" in return_weights, **args) if self.node_pred else None,
            "e": PRED_NETS[name](rep_e_dim, return_weights="edge" in return_weights, **args) if self.edge_pred else None,
        })

    def _enc_dims(self, nv, nvl, nel):
        return OrderedDict({"v": int(get_enc_len(nv - 1, self.base)) * self.base,
                            "vl": int(get_enc_len(nvl - 1, self.base)) * self.base,
                            "el": int(get_enc_len(nel - 1, self.base)) * self.base})

    def get_graph_enc_dims(self):
        return self._enc_dims(self.max_ngv, self.max_ngvl, self.max_ngel)

    def get_pattern_enc_dims(self):
        if self.share_enc_net:
            return self.get_graph_enc_dims()
        return self._enc_dims(self.max_npv, self.max_npvl, self.max_npel)

    def get_graph_enc_dim(self):
        d = self.get_graph_enc_dims()
        return d["v"] + d["vl"], (d["v"] + d["vl"]) * 2 + d["el"]

    def get_rep_dim(self):
        rep_v_dim, rep_e_dim = self.hid_dim, self.hid_dim
        if self.pred_with_enc:
            enc_v_dim, enc_e_dim = self.get_graph_enc_dim()
            rep_v_dim += enc_v_dim
            rep_e_dim += enc_e_dim
        if self.pred_with_deg:
            rep_v_dim += 2
            rep_e_dim += 2
        return rep_v_dim, rep_e_dim

    # ---- gate compaction (collate.compact_gated_edges): the rep-net on the edges the filter gate keeps
    gate_capacity = None        # edge rows of the compacted target batch; None: the rep-net runs on every edge

    def set_gate_capacity(self, capacity):
        """Run the rep-net on the target edges the filter gate keeps, as a batch of ``capacity`` edge rows (kept edges +
        inert padding; ``None`` / 0 turns it off).  A gate-0 edge is a zero row through the reference's whole rep-net
        (basemodel.py:1515-1531, dmpnn.py:262-275), so every output is what it was -- ``g_e_rep`` gets its zero rows back on
        first read.  A batch that keeps MORE than ``capacity`` edges cannot be represented: bit 0 of the status word goes up
        (``compaction_status()``; ``compaction_word`` for ``dp.FlatAdamW.set_veto``, which then drops that step) and the
        caller runs such a batch with the capacity off; ``calibrate_gate_capacity`` picks a capacity from a batch."""
        self.gate_capacity = int(capacity) if capacity else None
        if self.gate_capacity and getattr(self, "_compact_status", None) is None:
            # [0] flags since the last guarded optimizer step, [1] flags it has collected, [2] last step dropped, [3] steps dropped
            self._compact_status = th.zeros(4, dtype=th.int32, device=next(self.parameters()).device)
        return self

    @property
    def compaction_word(self):
        return getattr(self, "_compact_status", None)

    def compaction_status(self, clear=True):
        """Flags raised by the forward passes since the last clear (one host sync): 1 = a batch kept more edges than
        ``gate_capacity`` (its outputs and gradients were wrong; a ``FlatAdamW`` with ``set_veto(model.compaction_word)``
        dropped that step), 2 = padding fell on a graph without nodes.  ``compaction_dropped_steps()`` counts the drops."""
        st = getattr(self, "_compact_status", None)
        if st is None:
            return 0
        host = st.tolist()
        bits = host[0] | host[1]
        if clear and bits:
            st[:2].zero_()
        return bits

    def compaction_dropped_steps(self):
        st = getattr(self, "_compact_status", None)
        return 0 if st is None else int(st[3].item())

    def calibrate_gate_capacity(self, pattern, graph, margin=1.15, multiple=1024):
        """``set_gate_capacity`` from one batch: the edges its gate keeps (one host sync), times ``margin``, rounded up to a
        multiple of ``multiple``; left off where that would not shrink the batch by a tenth.  Returns the capacity or None."""
        kept, E = self.gate_kept_edges(pattern, graph)
        if kept is None:
            return self.set_gate_capacity(None).gate_capacity
        cap = -(-int(kept * margin + as_batched(graph).batch_size) // multiple) * multiple
        return self.set_gate_capacity(cap if cap <= 0.9 * E else None).gate_capacity

    def gate_kept_edges(self, pattern, graph):
        """``(target edges the filter's edge gate keeps, target edges)`` of one batch (one host sync); ``(None, E)`` without an edge gate."""
        pattern, graph = as_batched(pattern), as_batched(graph)
        pads = {"pv": _padder(pattern, "node"), "pe": _padder(pattern, "edge"), "gv": _padder(graph, "node"), "ge": _padder(graph, "edge")}
        el_gate = self.get_filter_gate(pattern, graph, pads)[1]
        E = graph.number_of_edges()
        return (None if el_gate is None else int((el_gate != 0).sum().item())), E

    def gate_kept_rows(self, pattern, graph):
        """``{"edges": (kept, all), "nodes": (kept, all)}`` of the TARGET rows under the filter's two gates for one batch (one host
        sync); ``kept`` is None without a gate."""
        pattern, graph = as_batched(pattern), as_batched(graph)
        pads = {"pv": _padder(pattern, "node"), "pe": _padder(pattern, "edge"), "gv": _padder(graph, "node"), "ge": _padder(graph, "edge")}
        vl_gate, el_gate = self.get_filter_gate(pattern, graph, pads)
        out = {"edges": (None if el_gate is None else int((el_gate != 0).sum().item()), graph.number_of_edges()),
               "nodes": (None if vl_gate is None else int((vl_gate != 0).sum().item()), graph.number_of_nodes())}
        if el_gate is not None and vl_gate is not None:
            # kept edges with a kept endpoint, and (kept edge, kept endpoint) pairs: what the backward's endpoint sums over the
            # kept edges' incidence CSR fetch / add (fused.NodeRows.kept_incidence)
            u, v = graph.all_edges(form="uv", order="eid")
            ke, kv = el_gate.reshape(-1) != 0, vl_gate.reshape(-1) != 0
            ku, kw = kv[u.long()] & ke, kv[v.long()] & ke
            out["edges_with_kept_endpoint"] = int((ku | kw).sum().item())
            out["kept_incidences"] = int(ku.sum().item() + kw.sum().item())
            out["kept_in_edges"] = int(kw.sum().item())      # kept edges INTO a kept node: what the forward aggregation over the kept nodes' rows fetches
        return out

    def _compact_gated(self, pattern, graph, el_gate):
        """``collate.CompactedEdges`` for this batch, or None (capacity off, no gate, options that read per-edge extras)."""
        if not self.gate_capacity or el_gate is None or not hasattr(self, "get_joint_rep") or self.pred_with_enc or self.pred_with_deg:
            return None
        if el_gate.requires_grad or self.add_edge_id or not graph.is_batched_on_device():
            return None
        from .collate import compact_gated_edges, out_degrees
        comp = compact_gated_edges(graph, el_gate, self.gate_capacity, self._compact_status[:1])
        if comp is not None:
            out_degrees(pattern)                   # the union graph then carries both sides' degrees: none is derived from the kept edges
        return comp

    # ---- forward pieces (basemodel.py:1394-1498)
    def get_filter_gate(self, pattern, graph, pads):
        if self.filter_net is None or len(self.filter_net) == 0:
            return None, None
        if type(self.filter_net["vl"]) is ScalarFilter and type(self.filter_net["el"]) is ScalarFilter:
            return tuple(scalar_filter_gates([
                (pads["pv"], pattern.ndata["label"], pads["gv"], graph.ndata["label"], max(self.max_ngvl, self.max_npvl)),
                (pads["pe"], pattern.edata["label"], pads["ge"], graph.edata["label"], max(self.max_ngel, self.max_npel))]))
        p_vl = pads["pv"].pad(pattern.ndata["label"].view(-1, 1))
        g_vl = pads["gv"].pad(graph.ndata["label"].view(-1, 1))
        vl_gate = pads["gv"].unpad(self.filter_net["vl"](p_vl, g_vl)).view(-1, 1)
        p_el = pads["pe"].pad(pattern.edata["label"].view(-1, 1))
        g_el = pads["ge"].pad(graph.edata["label"].view(-1, 1))
        el_gate = pads["ge"].unpad(self.filter_net["el"](p_el, g_el)).view(-1, 1)
        return vl_gate, el_gate

    def _enc(self, net, g):
        v, vl, el = lookup_rows([net["v"], net["vl"], net["el"]],
                                [g.ndata["id"].view(-1), g.ndata["label"].view(-1), g.edata["label"].view(-1)])
        enc = OrderedDict({"v": v, "vl": vl, "el": el})
        if self.add_edge_id:
            u, v = g.all_edges(form="uv", order="eid")
            enc["src"] = enc["v"][u]
            enc["dst"] = enc["v"][v]
        return enc

    def get_encs(self, pattern, graph):
        """``(get_pattern_enc(pattern), get_graph_enc(graph))`` with all six table lookups in ONE launch (``embed.lookup_rows``
        takes up to eight; a table it cannot take sends all of them through the modules, as two calls would)."""
        pn, gn = self.p_enc_net, self.g_enc_net
        rows = lookup_rows([pn["v"], pn["vl"], pn["el"], gn["v"], gn["vl"], gn["el"]],
                           [pattern.ndata["id"].view(-1), pattern.ndata["label"].view(-1), pattern.edata["label"].view(-1),
                            graph.ndata["id"].view(-1), graph.ndata["label"].view(-1), graph.edata["label"].view(-1)])
        encs = []
        for g, (v, vl, el) in ((pattern, rows[:3]), (graph, rows[3:])):
            enc = OrderedDict({"v": v, "vl": vl, "el": el})
            if self.add_edge_id:
                u, w = g.all_edges(form="uv", order="eid")
                enc["src"] = enc["v"][u]
                enc["dst"] = enc["v"][w]
            encs.append(enc)
        return encs[0], encs[1]

    def get_pattern_enc(self, pattern):
        return self._enc(self.p_enc_net, pattern)

    def get_graph_enc(self, graph):
        return self._enc(self.g_enc_net, graph)

    def _emb(self, net, enc):
        v_emb = net["vl"](enc["vl"])
        if self.add_node_id:
            v_emb = v_emb + net["v"](enc["v"])
        e_emb = net["el"](enc["el"])
        if self.add_edge_id:
            e_emb = e_emb + net["v"](enc["src"]) + net["v"](enc["dst"])
        return v_emb, e_emb

    def get_pattern_emb(self, p_enc):
        return self._emb(self.p_emb_net, p_enc)

    def get_graph_emb(self, g_enc):
        return self._emb(self.g_emb_net, g_enc)

    def get_graph_emb_deferred(self, g_enc):
        """``get_graph_emb`` whose label embeddings (float encodings @ table, no id terms) are left uncomputed
        (``embed.DeferredEmbedding``): the joint rep-net pass makes its gated input rows from the encodings themselves."""
        from .embed import DeferredEmbedding, Embedding
        net = self.g_emb_net
        if self.add_node_id or self.add_edge_id or not th.is_grad_enabled():
            return self.get_graph_emb(g_enc)
        out = []
        for key in ("vl", "el"):
            enc, m = g_enc[key], net[key]
            ok = (isinstance(m, Embedding) and th.is_tensor(enc) and enc.dtype == th.float32 and enc.dim() == 2 and enc.is_cuda
                  and enc.size(-1) == m.num_embeddings)
            out.append(DeferredEmbedding(m, enc) if ok else m(enc))
        return tuple(out)

    def get_subiso_pred(self, p_v_rep, p_v_mask, p_e_rep, p_e_mask, g_v_rep, g_v_mask, g_e_rep, g_e_mask):
        v_pred_c = v_pred_w = e_pred_c = e_pred_w = None
        if self.node_pred:
            v_pred_c, v_pred_w = self.pred_net["v"](p_v_rep, p_v_mask, g_v_rep, g_v_mask)
        if self.edge_pred:
            e_pred_c, e_pred_w = self.pred_net["e"](p_e_rep, p_e_mask, g_e_rep, g_e_mask)
        if self.node_pred and self.edge_pred:
            g_v_len = g_v_mask.float().sum(dim=1).view(-1, 1)
            g_e_len = g_e_mask.float().sum(dim=1).view(-1, 1)
            g_len = g_v_len + g_e_len
            return (g_v_len / g_len) * v_pred_c + (g_e_len / g_len) * e_pred_c, (v_pred_w, e_pred_w)
        if self.node_pred:
            return v_pred_c, (v_pred_w, e_pred_w)
        if self.edge_pred:
            return e_pred_c, (v_pred_w, e_pred_w)
        raise ValueError

    def _hip_heads_ok(self, v_sums, e_sums):
        if not (self.node_pred or self.edge_pred):
            return False
        if self.node_pred and not self.pred_net["v"].hip_head_ok(v_sums):
            return False
        if self.edge_pred and not self.pred_net["e"].hip_head_ok(e_sums):
            return False
        return True

    def get_subiso_pred_hip(self, v_sums, p_v_mask, g_v_mask, e_sums, p_e_mask, g_e_mask, counts=None):
        """``get_subiso_pred`` (basemodel.py:1477-1498) on the pooled union rows: every head and their blend in one
        autograd node / three HIP launches (``pred._PooledHeadsHIP``)."""
        from .pred import _PooledHeadsHIP
        cnt = lambda m: m.sum(dim=1, dtype=th.float32).view(-1, 1)
        args, g_lens = [], []
        for key, on, sums, pm, gm in (("v", self.node_pred, v_sums, p_v_mask, g_v_mask), ("e", self.edge_pred, e_sums, p_e_mask, g_e_mask)):
            if on:
                gl = cnt(gm) if counts is None else counts["g" + key]       # the mask kernel counted the rows already
                pl = cnt(pm) if counts is None else counts["p" + key]
                g_lens.append(gl)
                args.append([sums, pl, gl, float(pm.size(1)), float(gm.size(1)), None] + list(self.pred_net[key].head_params()))
        if len(args) == 2:                                   # blend weights g_i_len / (g_v_len + g_e_len): inside the op
            args[0][5] = args[1][5] = "len"
        flat = [a for head in args for a in head]
        slopes = {self.pred_net[k].act_slope() for k, on in (("v", self.node_pred), ("e", self.edge_pred)) if on}
        return _PooledHeadsHIP.apply(len(args), slopes.pop(), *flat), (None, None)

    def get_subiso_pred_pooled(self, p_v_sum, p_v_mask, p_e_sum, p_e_mask, g_v_sum, g_v_mask, g_e_sum, g_e_mask):
        """``get_subiso_pred`` (basemodel.py:1477-1498) on per-graph sums instead of padded rows."""
        cnt = lambda m: m.sum(dim=1, dtype=th.float32).view(-1, 1)
        v_pred_c = e_pred_c = None
        if self.node_pred:
            g_v_len = cnt(g_v_mask)
            v_pred_c, _ = self.pred_net["v"].forward_pooled(p_v_sum, p_v_mask.size(1), cnt(p_v_mask),
                                                            g_v_sum, g_v_mask.size(1), g_v_len)
        if self.edge_pred:
            g_e_len = cnt(g_e_mask)
            e_pred_c, _ = self.pred_net["e"].forward_pooled(p_e_sum, p_e_mask.size(1), cnt(p_e_mask),
                                                            g_e_sum, g_e_mask.size(1), g_e_len)
        if self.node_pred and self.edge_pred:
            g_len = g_v_len + g_e_len
            return _Blend.apply(v_pred_c, e_pred_c, g_v_len / g_len, g_e_len / g_len), (None, None)
        if self.node_pred:
            return v_pred_c, (None, None)
        if self.edge_pred:
            return e_pred_c, (None, None)
        raise ValueError

    # ---- forward (basemodel.py:1500-1663)
    @on_input_device
    def forward(self, pattern, graph):
        pattern, graph = as_batched(pattern), as_batched(graph)   # DGLGraph-in (train.py:606-611)
        bsz = pattern.batch_size
        pads = {"pv": _padder(pattern, "node"), "pe": _padder(pattern, "edge"),
                "gv": _padder(graph, "node"), "ge": _padder(graph, "edge")}
        # reversed edges do not take part in the edge head (basemodel.py:1521-1531)
        skip_rev = self.edge_head_skips_reversed
        (p_v_mask, p_v_cnt), (p_e_mask, p_e_cnt), (g_v_mask, g_v_cnt), (g_e_mask, g_e_cnt) = len_masks(
            [(pads["pv"], None), (pads["pe"], pattern.edata.get(REVFLAG) if skip_rev else None),
             (pads["gv"], None), (pads["ge"], graph.edata.get(REVFLAG) if skip_rev else None)])
        counts = {"pv": p_v_cnt, "pe": p_e_cnt, "gv": g_v_cnt, "ge": g_e_cnt}
        vl_gate, el_gate = self.get_filter_gate(pattern, graph, pads)
        if vl_gate is not None:  # bool gate * float features == float gate * float features
            vl_gate, el_gate = vl_gate.float(), el_gate.float()

        pooled = all(h is None or h.poolable() for h in self.pred_net.values())

        def prefetch():
            if hasattr(self, "get_joint_rep") and not self.gate_capacity and (vl_gate is None) == (el_gate is None):
                # the index arrays the joint pass derives from the structure and the gates alone: on the side stream from here on,
                # beside the encoding / embedding kernels and the first layer's node side (side.py)
                from .dmpnn import prefetch_joint_indexes
                kinds = ()
                if pooled and not self.pred_with_enc and not self.pred_with_deg:
                    kinds = tuple(k for k, on in (("node", self.node_pred), ("edge", self.edge_pred)) if on)
                prefetch_joint_indexes(self, pattern, graph, vl_gate, el_gate, kinds, skip_rev)

        if not PREFETCH_AFTER_EMBEDDINGS:
            prefetch()
        p_enc, g_enc = self.get_encs(pattern, graph)          # (both sides' encodings: one launch)
        p_v_emb, p_e_emb = self.get_pattern_emb(p_enc)
        g_v_emb, g_e_emb = self.get_graph_emb_deferred(g_enc) if hasattr(self, "get_joint_rep") else self.get_graph_emb(g_enc)
        if PREFETCH_AFTER_EMBEDDINGS:
            prefetch()
        joint = None
        comp = self._compact_gated(pattern, graph, el_gate)
        rep_graph = graph                                           # the graph the rep-net runs on
        if hasattr(self, "get_joint_rep"):
            import inspect
            with_pools = "pools" in inspect.signature(self.get_joint_rep).parameters

            def run_joint(rg, e_emb, e_gate):
                # heads that pool over the un-augmented representations: the last layer pools its own outputs (and its backward
                # then never builds the [E, H] gradient of the edge representation, fused._FusedDMPLayer)
                pools = None
                if pooled and not self.pred_with_enc and not self.pred_with_deg and (self.node_pred or self.edge_pred):
                    kinds = [k for k, on in (("node", self.node_pred), ("edge", self.edge_pred)) if on]
                    built = dict(zip(kinds, _pool_indexes_union(pattern, rg, kinds, skip_rev)))      # one pair of launches
                    pools = (built.get("node"), built.get("edge"))
                if with_pools:
                    return self.get_joint_rep(pattern, rg, p_v_emb, p_e_emb, g_v_emb, e_emb, vl_gate, e_gate, pools=pools)
                return self.get_joint_rep(pattern, rg, p_v_emb, p_e_emb, g_v_emb, e_emb, vl_gate, e_gate)   # no pooled form

            if comp is not None:
                # the rep-net on the kept edges (+ inert padding): their encodings' rows, their gates, their graph
                g_enc_c = OrderedDict(g_enc)
                g_enc_c["el"] = comp.take(g_enc["el"])
                joint = run_joint(comp.graph, self.get_graph_emb_deferred(g_enc_c)[1], comp.gate)
                if joint is not None:
                    rep_graph = comp.graph
                    joint = list(joint)
                    joint[3] = _expanded(comp, joint[3])            # g_e_rep: the gated-out edges' zero rows back in place
            if joint is None:
                comp = None
                joint = run_joint(graph, g_e_emb, el_gate)
            from . import side
            side.join()          # (whatever path the pass took: nothing of the side stream outlives it)
        v_union = e_union = None
        union_sums = (None, None)
        if joint is not None:
            p_v_rep, p_e_rep, g_v_rep, g_e_rep, v_union, e_union = joint[:6]
            if len(joint) > 6:
                union_sums = joint[6]
        else:
            p_v_rep, p_e_rep = self.get_pattern_rep(pattern, p_v_emb, p_e_emb)
            g_v_emb, g_e_emb = materialize(g_v_emb), materialize(g_e_emb)
            g_v_rep, g_e_rep = self.get_graph_rep(graph, g_v_emb, g_e_emb, v_gate=vl_gate, e_gate=el_gate)

        if self.pred_with_deg:
            p_out_deg = pattern.out_degrees().float().view(-1, 1)
            p_in_deg = pattern.in_degrees().float().view(-1, 1)
            g_out_deg = graph.out_degrees().float().view(-1, 1)
            g_in_deg = graph.in_degrees().float().view(-1, 1)

        # pool-then-project (pred.py heads that sum/average rows): per-graph sums straight from the
        # un-padded representations with the segment-sum kernel; otherwise the reference's padded path
        p_v_output = g_v_output = p_e_output = g_e_output = None
        v_sums = e_sums = None      # [2B, H] pooled rows of the union pass (pattern graphs, then target graphs)
        if self.node_pred:
            p_add, g_add = [], []
            if self.pred_with_enc:
                p_add += [p_enc["v"], p_enc["vl"]]
                g_add += [g_enc["v"], g_enc["vl"]]
            if self.pred_with_deg:
                p_add += [p_out_deg, p_in_deg]
                g_add += [g_out_deg, g_in_deg]
            p_v_output = th.cat([self.refine_node_weights(th.cat(p_add, dim=-1)), p_v_rep], dim=-1) if p_add else p_v_rep
            g_v_output = th.cat([self.refine_node_weights(th.cat(g_add, dim=-1)), g_v_rep], dim=-1) if g_add else g_v_rep
            p_v_mask = self.refine_node_weights(p_v_mask)
            g_v_mask = self.refine_node_weights(g_v_mask)
            if pooled and v_union is not None and not p_add:
                # the shared rep-net ran over the union of both batches: pool the union rows once
                # (its backward is the union gradient itself, no concatenation of two halves)
                sums = v_sums = (union_sums[0] if union_sums[0] is not None
                                 else ops.seg_pool(v_union, _pool_index_union(pattern, graph, "node")))
                p_v_output, g_v_output = sums[:bsz], sums[bsz:]
            elif pooled:
                p_v_output = ops.seg_pool(p_v_output, _pool_index(pattern, "node"))
                g_v_output = ops.seg_pool(g_v_output, _pool_index(graph, "node"))
            else:
                p_v_output = pads["pv"].pad(p_v_output).masked_fill(~p_v_mask, 0)
                g_v_output = pads["gv"].pad(g_v_output).masked_fill(~g_v_mask, 0)
        if self.edge_pred:
            p_u, p_v = pattern.all_edges(form="uv", order="eid")
            g_u, g_v = graph.all_edges(form="uv", order="eid")
            p_add, g_add = [], []
            if self.pred_with_enc:
                p_add += [p_enc["v"][p_u], p_enc["v"][p_v], p_enc["vl"][p_u], p_enc["el"], p_enc["vl"][p_v]]
                g_add += [g_enc["v"][g_u], g_enc["v"][g_v], g_enc["vl"][g_u], g_enc["el"], g_enc["vl"][g_v]]
            if self.pred_with_deg:
                p_add += [p_out_deg[p_u], p_in_deg[p_v]]
                g_add += [g_out_deg[g_u], g_in_deg[g_v]]
            p_e_output = th.cat([self.refine_edge_weights(th.cat(p_add, dim=-1)), p_e_rep], dim=-1) if p_add else p_e_rep
            g_e_output = th.cat([self.refine_edge_weights(th.cat(g_add, dim=-1)), g_e_rep], dim=-1) if g_add else g_e_rep
            p_e_mask = self.refine_edge_weights(p_e_mask)
            g_e_mask = self.refine_edge_weights(g_e_mask)
            if pooled and e_union is not None and not p_add:
                sums = e_sums = (union_sums[1] if union_sums[1] is not None
                                 else ops.seg_pool(e_union, _pool_index_union(pattern, rep_graph, "edge", skip_rev)))[:, :e_union.size(1)]
                p_e_output, g_e_output = sums[:bsz], sums[bsz:]
            elif pooled:  # reversed edges are masked out of the edge head: keep the non-flagged half
                d = p_e_output.size(1)
                p_e_output = ops.seg_pool(p_e_output, _pool_index(pattern, "edge", skip_rev))[:, :d]
                g_e_output = ops.seg_pool(g_e_output, _pool_index(graph, "edge", skip_rev))[:, :d]
            else:
                p_e_output = pads["pe"].pad(p_e_output).masked_fill(~p_e_mask, 0)
                g_e_output = pads["ge"].pad(g_e_output).masked_fill(~g_e_mask, 0)

        p_v_mask, p_e_mask = p_v_mask.view(bsz, -1), p_e_mask.view(bsz, -1)
        g_v_mask, g_e_mask = g_v_mask.view(bsz, -1), g_e_mask.view(bsz, -1)
        if pooled and self._hip_heads_ok(v_sums, e_sums):
            pred_c, (pred_v, pred_e) = self.get_subiso_pred_hip(v_sums, p_v_mask, g_v_mask, e_sums, p_e_mask, g_e_mask,
                                                                counts=counts)
        elif pooled:
            pred_c, (pred_v, pred_e) = self.get_subiso_pred_pooled(p_v_output, p_v_mask, p_e_output, p_e_mask,
                                                                   g_v_output, g_v_mask, g_e_output, g_e_mask)
        else:
            pred_c, (pred_v, pred_e) = self.get_subiso_pred(p_v_output, p_v_mask, p_e_output, p_e_mask,
                                                            g_v_output, g_v_mask, g_e_output, g_e_mask)
        return OutputDict(p_v_emb=p_v_emb, p_e_emb=p_e_emb, g_v_emb=g_v_emb, g_e_emb=g_e_emb,
                          p_v_rep=p_v_rep, p_e_rep=p_e_rep, g_v_rep=g_v_rep, g_e_rep=g_e_rep,
                          p_v_mask=p_v_mask, p_e_mask=p_e_mask, g_v_mask=g_v_mask, g_e_mask=g_e_mask,
                          pred_c=pred_c, pred_v=pred_v, pred_e=pred_e)


class DMPNN(DMPNNRepMixin, GraphAdjModelV2):
    """``models/dmpnn.py:179-277``: ``DMPNN(**config)`` as built by ``train.py:68-87``."""


class CompGCN(CompGCNRepMixin, GraphAdjModelV2):
    """``models/compgcn.py:289-385``."""


class RGCN(RGCNRepMixin, GraphAdjModel):
    """``models/rgcn.py:215-300``."""


class RGIN(RGINRepMixin, GraphAdjModel):
    """``models/rgin.py:175-260``."""


def build_model(config=None, **kw):
    """``train.py:68-87`` for the rep-nets on the MI355X path.  Takes the reference's call form
    ``build_model(process_model_config(config), init_neigenv=..., init_eeigenv=...)`` (the ``match_weights`` entry of the
    run configuration becomes ``pred_return_weights``, train.py:70-86) as well as plain keywords."""
    config = dict(config or {}, **kw)
    if "match_weights" in config and "pred_return_weights" not in config:
        config["pred_return_weights"] = config["match_weights"]
    rep_net = config.get("rep_net", "DMPNN")
    if rep_net == "DMPNN":
        return DMPNN(**config)
    if rep_net == "CompGCN":
        return CompGCN(**config)
    if rep_net == "RGCN":
        return RGCN(**config)
    if rep_net == "RGIN":
        return RGIN(**config)
    if rep_net in ("LRP", "DMPLRP"):
        from . import lrp
        return getattr(lrp, rep_net)(**config)
    raise NotImplementedError("rep_net=%s is outside the MI355X hot-path scope" % rep_net)

"""Batched graph object: the part of the DGLGraph surface the reference's models
touch, backed by device arrays and the HIP index kernels.

Surface mirrored (measured by attribute tracing of
``GraphAdjModelV2.forward``, SubgraphCountingMatching/models/basemodel.py:1500-1663):
``batch_size``, ``batch_num_nodes()``, ``batch_num_edges()``, ``number_of_nodes()``,
``number_of_edges()``, ``ndata`` / ``edata``, ``in_degrees()``, ``out_degrees()``,
``update_all``, ``apply_edges``, ``all_edges(form, order="eid")``, ``.to(device)``.
Degree caching follows ``dataset.Graph.in_degrees/out_degrees``
(SubgraphCountingMatching/dataset.py:1222-1236).

Edges are kept in eid order; CSR arrays are an *index* into that order and never
reorder features (outputs of every layer stay in eid order).
"""
import os

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .constants import INDEGREE, OUTDEGREE, REVFLAG


class SumReducer:
    """``dgl.function.sum(msg, out)`` (dmpnn.py:92)."""

    def __init__(self, msg, out):
        self.msg = msg
        self.out = out


class function:  # namespace mirroring ``import dgl.function as fn``
    sum = SumReducer


USE_GRAPH_CSR = True   # block-diagonal batches: both CSRs from ONE launch (dmp_csr_build_graphs)


class GraphIndex:
    """Device-resident integer index of one (batched) graph.

    in_ptr/in_ent   CSR by destination, entries (eid<<1)|is_reversed, ascending eid
    out_ptr/out_ent CSR by source
    inc_ptr/inc_ent incidence CSR (in-edges ++ out-edges with flipped flag), lazy
    src32/dst32     int32 endpoints in eid order; rev8 uint8 is_reversed or None
    in_deg/out_deg  int64 degree vectors (structure-derived)
    """

    _status = {}     # device -> int32 [2] that the one-launch build ORs into (cleared when a validation reads it)

    def __init__(self, src, dst, num_nodes, rev=None, validate=False, offsets=None):
        lib = _lib.load()
        _lib.require_gpu(src, dst, rev)
        if src.dtype != torch.int64 or dst.dtype != torch.int64:
            raise _lib.DmpError("edge endpoints must be int64 (DGL default idtype)")
        src = src.contiguous()
        dst = dst.contiguous()
        dev = src.device
        E, N = src.numel(), int(num_nodes)
        self.num_nodes, self.num_edges, self.device = N, E, dev
        if rev is not None:
            if rev.dtype == torch.bool:
                rev = rev.contiguous().view(torch.uint8)      # same bytes (0 / 1): no conversion pass
            elif rev.dtype != torch.uint8:
                raise _lib.DmpError("is_reversed must be bool or uint8")
            rev = rev.contiguous().view(-1)
            if rev.numel() != E:
                raise _lib.DmpError("is_reversed must have one entry per edge")
        self.rev8 = rev
        i32 = dict(dtype=torch.int32, device=dev)
        self.in_ptr = torch.empty(N + 1, **i32)
        self.in_ent = torch.empty(E, **i32)
        self.dst32 = torch.empty(E, **i32)
        self.in_deg = torch.empty(N, dtype=torch.int64, device=dev)
        self.out_ptr = torch.empty(N + 1, **i32)
        self.out_ent = torch.empty(E, **i32)
        self.src32 = torch.empty(E, **i32)
        self.out_deg = torch.empty(N, dtype=torch.int64, device=dev)
        if offsets is not None and E > 0 and N > 0:
            # a block-diagonal batch whose graphs fit a workgroup's LDS counters (``offsets`` = node offsets, edge offsets,
            # number of graphs; the caller checked the largest graph): both CSRs in ONE launch
            node_off, edge_off, B = offsets
            status = GraphIndex._status.get(dev)
            if status is None:
                status = GraphIndex._status[dev] = torch.zeros(2, **i32)
            check(lib.dmp_csr_build_graphs(ptr(dst), ptr(src), ptr(rev), ptr(node_off), ptr(edge_off), int(B), E, N,
                                           ptr(self.in_ptr), ptr(self.in_ent), ptr(self.dst32), ptr(self.in_deg),
                                           ptr(self.out_ptr), ptr(self.out_ent), ptr(self.src32), ptr(self.out_deg),
                                           ptr(status), stream_ptr()), "dmp_csr_build_graphs")
        else:
            ws = torch.empty(lib.dmp_csr_pair_workspace_words(N), **i32)
            status = torch.empty(2, **i32)
            # the in-CSR (by destination) and the out-CSR (by source) side by side: one set of dispatches for both
            check(lib.dmp_csr_build_pair(ptr(dst), ptr(src), ptr(rev), E, N,
                                         ptr(self.in_ptr), ptr(self.in_ent), ptr(self.dst32), ptr(self.in_deg),
                                         ptr(self.out_ptr), ptr(self.out_ent), ptr(self.src32), ptr(self.out_deg),
                                         ptr(status), ptr(ws), stream_ptr()), "dmp_csr_build_pair")
        self._inc = None
        self._coef = {}
        self.tiling = None
        self.node_tiling = None      # ops.graph_node_tiling(...): whole graphs per tile of the one-pass endpoint sums
        if (validate or _lib.VALIDATE) and int(status.sum().item()) != 0:
            status.zero_()
            raise _lib.DmpError("edge endpoint outside [0, num_nodes)" if offsets is None else
                                "edge endpoint outside its graph's node range (not a block-diagonal batch)")

    def incidence(self):
        if self._inc is None:
            lib = _lib.load()
            N, E = self.num_nodes, self.num_edges
            inc_ptr = torch.empty(N + 1, dtype=torch.int32, device=self.device)
            inc_ent = torch.empty(2 * E, dtype=torch.int32, device=self.device)
            check(lib.dmp_incidence_build(ptr(self.in_ptr), ptr(self.in_ent), ptr(self.out_ptr),
                                          ptr(self.out_ent), N, E, ptr(inc_ptr), ptr(inc_ent), stream_ptr()),
                  "dmp_incidence_build")
            self._inc = (inc_ptr, inc_ent)
        return self._inc

    def degree_coef(self, out_deg):
        """2(1+log2(1+out_deg)) per node (dmpnn.py:144-146); cached per degree tensor."""
        key = (out_deg.data_ptr(), out_deg._version)
        c = self._coef.get(key)
        if c is None:
            lib = _lib.load()
            _lib.require_gpu(out_deg)
            if out_deg.dtype != torch.int64:
                out_deg = out_deg.to(torch.int64)
            out_deg = out_deg.contiguous().view(-1)
            c = torch.empty(self.num_nodes, dtype=torch.float32, device=self.device)
            check(lib.dmp_degree_coef(ptr(out_deg), self.num_nodes, ptr(c), stream_ptr()), "dmp_degree_coef")
            self._coef = {key: c}
            self._coef_deg = (c, out_deg)    # the integer degrees this coefficient tensor was computed from
        return c

    def edge_select(self, coef):
        """Per-edge selectors of the fused edge chain (``dmp_edge_select_build``): ``(selA, selB, coefE)``
        with selA = is_reversed ? src : dst, selB = the other endpoint, coefE = coef[dst]; cached per
        coefficient tensor (they depend only on the structure and the degrees)."""
        key = (coef.data_ptr(), coef._version)
        cached = getattr(self, "_esel", None)
        if cached is None or cached[0] != key:
            lib = _lib.load()
            E = self.num_edges
            sel_a = torch.empty(E, dtype=torch.int32, device=self.device)
            sel_b = torch.empty(E, dtype=torch.int32, device=self.device)
            coef_e = torch.empty(E, dtype=torch.float32, device=self.device)
            check(lib.dmp_edge_select_build(ptr(self.src32), ptr(self.dst32), ptr(self.rev8), ptr(coef), E, ptr(sel_a),
                                            ptr(sel_b), ptr(coef_e), stream_ptr()), "dmp_edge_select_build")
            self._esel = cached = (key, (sel_a, sel_b, coef_e), coef)
        return cached[1]

    def edge_select_nodes(self, nodemask):
        """``(selA, selB, dstM)`` of ``edge_select`` / the destinations with every node whose bit of ``nodemask`` (uint32
        words, ``fused.gate_row_mask`` of a 0 / 1 node gate) is clear replaced by -1 (``dmp_edge_select_nodes``): the tile
        kernels read such a node's rows as zeros without fetching them.  Memoised per mask tensor."""
        cached = getattr(self, "_esel_nodes", None)
        if cached is None or cached[0] is not nodemask:
            lib = _lib.load()
            E = self.num_edges
            out = torch.empty((3, E), dtype=torch.int32, device=self.device)
            check(lib.dmp_edge_select_nodes(ptr(self.src32), ptr(self.dst32), ptr(self.rev8), ptr(nodemask), E, ptr(out[0]), ptr(out[1]),
                                            ptr(out[2]), stream_ptr()), "dmp_edge_select_nodes")
            self._esel_nodes = cached = (nodemask, (out[0], out[1], out[2]))
        return cached[1]

    def endpoint_select(self):
        """``(selA, selB)`` of ``edge_select`` without the coefficient: selA = is_reversed ? src : dst, selB = the other
        endpoint (int32 [E]).  Taken from the selectors the forward pass built when there are any."""
        cached = getattr(self, "_esel", None)
        if cached is not None:
            return cached[1][0], cached[1][1]
        ep = getattr(self, "_epsel", None)
        if ep is None:
            if self.rev8 is None:
                ep = (self.dst32, self.src32)
            else:
                r = self.rev8.bool()
                ep = (torch.where(r, self.src32, self.dst32), torch.where(r, self.dst32, self.src32))
            self._epsel = ep
        return ep

    MAX_EDGE_CLASSES = 65536
    USE_DEGREE_HINT = True       # class tables sized by the batch's largest graph (``max_degree_hint``) where it is known
    max_degree_hint = None

    def _num_classes(self):
        """Classes the device build lays its tables out for: the integer degrees 0 .. hint (+ the overflow class, which a degree
        above the hint would land in and poison with NaN: loud, not silently wrong), or MAX_EDGE_CLASSES without a hint."""
        h = self.max_degree_hint
        if not self.USE_DEGREE_HINT or h is None:
            return self.MAX_EDGE_CLASSES
        return max(2, min(self.MAX_EDGE_CLASSES, int(h) + 2))

    def class_tiles(self, coef):
        """Tile list of the class-typed edge kernels (csrc/dmp_typed.hip): edges sorted by their
        ``coef[dst]`` value (a function of the destination's out-degree, dmpnn.py:144-146: few distinct
        values) and cut into tiles of 32 that never mix values.  Returns ``(slot_edge int32
        [tiles_bound * 32] (-1 = padding), tile_scale float [tiles_bound], num_tiles int32 [1] (device),
        tiles_bound (host int))``.  No host sync: the number of tiles in use stays on the device and the
        arrays are sized by a host-side bound.  More than MAX_EDGE_CLASSES classes poison the overflow
        class with NaN (loud, not silently wrong).  Cached per coefficient tensor.  When the coefficient
        is this index's own ``degree_coef`` the list is built by ``dmp_class_tiles`` from the integer
        degrees; otherwise by a value sort."""
        key = (coef.data_ptr(), coef._version)
        cached = getattr(self, "_ctiles", None)
        if cached is None or cached[0] != key:
            src = getattr(self, "_coef_deg", None)
            if src is not None and src[0] is coef and self.num_edges > 0:
                res = self._class_tiles_device(src[1])
            else:
                res = self._class_tiles_by_value(coef)
            self._ctiles = cached = (key, res, coef)
        return cached[1]

    def _class_tiles_device(self, deg):
        """``coef`` came from ``degree_coef(deg)``: classes are the integer degrees; five HIP launches
        (``dmp_class_tiles``), slot order = class, node id, edge id."""
        lib = _lib.load()
        E, N, C, dev = self.num_edges, self.num_nodes, self._num_classes(), self.device
        bound = E // 32 + C + 1
        slot_edge = torch.empty(bound * 32, dtype=torch.int32, device=dev)
        tile_scale = torch.empty(bound, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.dmp_class_tiles_workspace_words(N, C)) + 2, dtype=torch.int32, device=dev)
        num_tiles = ws[-1:]                                    # allocations are 8-byte aligned; the tail word is ours
        check(lib.dmp_class_tiles(ptr(deg), ptr(self.in_ptr), ptr(self.in_ent), N, E, C, bound, ptr(ws), ptr(slot_edge),
                                  ptr(tile_scale), ptr(num_tiles), stream_ptr()), "dmp_class_tiles")
        return slot_edge, tile_scale, num_tiles, bound

    def class_tiles_gated(self, coef, gate):
        """``class_tiles`` over the edges a 0 / 1 edge ``gate`` ([E] floats) keeps -- edges under a zero gate get no slot
        (``dmp_class_tiles_gated``) -- or None where the list cannot be built on the device from the integer degrees.
        Not cached here (the gate changes with every batch: ``fused.live_tiles`` memoises per gate)."""
        src = getattr(self, "_coef_deg", None)
        if src is None or src[0] is not coef or self.num_edges == 0:
            return None
        lib = _lib.load()
        E, N, C, dev = self.num_edges, self.num_nodes, self._num_classes(), self.device
        bound = E // 32 + C + 1
        slot_edge = torch.empty(bound * 32, dtype=torch.int32, device=dev)
        tile_scale = torch.empty(bound, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.dmp_class_tiles_workspace_words(N, C)) + 2, dtype=torch.int32, device=dev)
        row_cnt = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
        num_tiles = ws[-1:]
        check(lib.dmp_class_tiles_gated(ptr(src[1]), ptr(self.in_ptr), ptr(self.in_ent), ptr(gate), ptr(row_cnt), N, E, C, bound, ptr(ws),
                                        ptr(slot_edge), ptr(tile_scale), ptr(num_tiles), stream_ptr()), "dmp_class_tiles_gated")
        return slot_edge, tile_scale, num_tiles, bound

    def _class_tiles_by_value(self, coef):
        """Generic fallback (a coefficient tensor of unknown origin): classes = distinct values of
        ``coef[dst]``, found by a value sort with a handful of tensor ops."""
        E, dev = self.num_edges, self.device
        C = max(1, min(self.MAX_EDGE_CLASSES, E))
        bound = E // 32 + C + 1
        coef_e = self.edge_select(coef)[2]
        slot_edge = torch.full((bound * 32,), -1, dtype=torch.int32, device=dev)
        if E == 0:
            return (slot_edge, torch.zeros(bound, dtype=torch.float32, device=dev),
                    torch.zeros(1, dtype=torch.int32, device=dev), bound)
        vals, order = torch.sort(coef_e, stable=True)
        cidx = torch.zeros(E, dtype=torch.int64, device=dev)
        torch.cumsum(vals[1:] != vals[:-1], 0, out=cidx[1:])      # class of each sorted position
        over = cidx[-1] >= C - 1
        cidx.clamp_(max=C - 1)
        # cidx is sorted: class sizes from its boundaries (a bincount would serialise its atomics
        # on the handful of hot bins: 5 ms at E = 549 k)
        marks = torch.searchsorted(cidx, torch.arange(C + 1, device=dev))
        seg_start, cnt = marks[:-1], marks[1:] - marks[:-1]
        ntile = (cnt + 31) >> 5
        tile_end = torch.cumsum(ntile, 0)
        slot = (tile_end - ntile)[cidx] * 32 + (torch.arange(E, device=dev) - seg_start[cidx])
        slot_edge[slot] = order.to(torch.int32)
        tile_class = torch.searchsorted(tile_end, torch.arange(bound, device=dev), right=True).clamp_(max=C - 1)
        tile_scale = vals[seg_start.clamp(max=E - 1)][tile_class]
        tile_scale = torch.where((tile_class == C - 1) & over, torch.full_like(tile_scale, float("nan")), tile_scale)
        return slot_edge, tile_scale.contiguous(), tile_end[-1:].to(torch.int32), bound


class _Gather:
    """``edges.src`` / ``edges.dst`` views handed to message UDFs."""

    def __init__(self, graph, by_src):
        self._g, self._by_src = graph, by_src

    def __contains__(self, k):
        return k in self._g.ndata

    def __getitem__(self, k):
        from . import ops
        t = self._g.ndata[k]
        ix = self._g.index()
        if t.is_cuda and t.dtype == torch.float32 and t.dim() == 2:
            return ops.gather_src(t, ix) if self._by_src else ops.gather_dst(t, ix)
        idx = self._g._src if self._by_src else self._g._dst
        return t[idx]


class EdgeBatch:
    def __init__(self, graph):
        self.src = _Gather(graph, True)
        self.dst = _Gather(graph, False)
        self.data = graph.edata  # same dict: UDF side-effect writes persist (dmpnn.py:126)
        self._n = graph.number_of_edges()

    def __len__(self):
        return self._n


class NodeBatch:
    def __init__(self, graph):
        self.data = graph.ndata
        self._n = graph.number_of_nodes()

    def __len__(self):
        return self._n


def leave_detached(frame, *keys):
    """What a layer leaves on the graph -- the reference's ``ndata["node_feat"]``, ``edata["edge_feat"]``,
    ``ndata["node_agg"]`` side effects (dmpnn.py:96-109,163) -- as VALUES, without their autograd history: a graph object
    that outlives the step (UNC trains on one graph; a DGL graph a caller reuses) would otherwise keep the whole step
    alive, its activations and its gradient accumulators -- and accumulators made on one stream break the recording of a
    later step in a HIP graph on another (``dp.StepGraph``).  The reference never differentiates through these entries
    (every call passes its features); a call with ``node_feat=None`` now reads values only."""
    for k in keys:
        if k in frame:
            t = frame[k]
            if torch.is_tensor(t) and t.requires_grad:
                frame[k] = t.detach()


class BatchedGraph:
    """A block-diagonal batch of directed multigraphs in eid order."""

    def __init__(self, src, dst, num_nodes, batch_num_nodes=None, batch_num_edges=None, ndata=None,
                 edata=None, share_frames=False):
        if src.dtype != torch.int64 or dst.dtype != torch.int64:
            raise ValueError("src/dst must be int64")
        if src.shape != dst.shape or src.dim() != 1:
            raise ValueError("src/dst must be 1-D and of equal length")
        self._src, self._dst = src, dst
        self._n = int(num_nodes)
        self._bnn = batch_num_nodes
        self._bne = batch_num_edges
        if share_frames:   # the caller's own frame objects (``from_graph``): layer side effects land on the caller's graph
            self.ndata, self.edata = ndata, edata
        else:
            self.ndata = dict(ndata) if ndata else {}
            self.edata = dict(edata) if edata else {}
        self._index = None
        self._index_key = None
        self.node_graph = None  # int32 [N] owning graph of each node (set by collate)
        self.edge_graph = None
        self.tiling = None      # ops.graph_tiling(...) of a block-diagonal batch (set by collate / union_graphs)
        self.node_tiling = None  # ops.graph_node_tiling(...)

    # ---- DGLGraph-in: any graph object with the surface the reference's models touch
    @classmethod
    def from_graph(cls, obj):
        """Adapter for the reference's call convention ``layer(graph: dgl.DGLGraph, ...)`` (dmpnn.py:158-166;
        train.py:606-611 passes ``dataset.Graph``, dataset.py:1053-1134).  ``obj`` is duck-typed: it needs
        ``all_edges(form="uv", order="eid")``, ``number_of_nodes()``, ``ndata`` / ``edata`` (mutable mappings
        of device tensors) and, for batches, ``batch_num_nodes()`` / ``batch_num_edges()``.  The result
        SHARES the caller's frames, so what the reference's layer leaves on the graph (``ndata["out_deg"]``,
        ``"node_feat"``, ``"node_agg"``, ``edata["edge_feat"]``, dmpnn.py:96-109) lands on ``obj`` itself.
        Built once per graph object and edge tensor (the CSR index is cached with it)."""
        if isinstance(obj, cls):
            return obj
        for need in ("all_edges", "number_of_nodes", "ndata", "edata"):
            if not hasattr(obj, need):
                raise TypeError("expected a graph with all_edges / number_of_nodes / ndata / edata (DGLGraph surface); "
                                "got %s" % type(obj).__name__)
        u, v = obj.all_edges(form="uv", order="eid")
        cached = getattr(obj, "_dmp_batched", None)
        if cached is not None and cached[0] is u and cached[1] is v and cached[2]._n == int(obj.number_of_nodes()):
            return cached[2]
        if not (torch.is_tensor(u) and torch.is_tensor(v)):
            raise TypeError("all_edges must return tensors")
        _lib.require_gpu(u, v)                                    # no CPU path: move the graph first (graph.to(device))
        src, dst = u.to(torch.int64), v.to(torch.int64)           # DGL idtype int32 graphs: widened once
        n = int(obj.number_of_nodes())
        bnn = bne = None
        if hasattr(obj, "batch_num_nodes") and hasattr(obj, "batch_num_edges"):
            bnn, bne = obj.batch_num_nodes(), obj.batch_num_edges()
            bnn = torch.as_tensor(bnn, dtype=torch.int64).to(src.device)
            bne = torch.as_tensor(bne, dtype=torch.int64).to(src.device)
            if bnn.numel() <= 1:
                bnn = bne = None
        g = cls(src, dst, n, bnn, bne, obj.ndata, obj.edata, share_frames=True)
        if bnn is not None:   # owning graph per node / edge, as the device collate leaves them (no host sync: sizes are known)
            ids = torch.arange(bnn.numel(), dtype=torch.int32, device=src.device)
            g.node_graph = torch.repeat_interleave(ids, bnn, output_size=n)
            g.edge_graph = torch.repeat_interleave(ids, bne, output_size=int(src.numel()))
        try:
            obj._dmp_batched = (u, v, g)
        except Exception:   # objects with __slots__: converted again on every call
            pass
        return g

    # ---- sizes
    @property
    def batch_size(self):
        return 1 if self._bnn is None else int(self._bnn.numel())

    def batch_num_nodes(self, *a):
        if self._bnn is None:
            return torch.tensor([self._n], dtype=torch.int64, device=self._src.device)
        return self._bnn

    def batch_num_edges(self, *a):
        if self._bne is None:
            return torch.tensor([self._src.numel()], dtype=torch.int64, device=self._src.device)
        return self._bne

    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self._src.numel())

    num_nodes = number_of_nodes
    num_edges = number_of_edges

    def __len__(self):
        return self._n

    @property
    def device(self):
        return self._src.device

    def is_batched_on_device(self):
        """A block-diagonal batch whose per-graph offsets and size bounds came with it (``collate.collate_device``)."""
        return (getattr(self, "node_offsets", None) is not None and getattr(self, "edge_offsets", None) is not None
                and getattr(self, "max_num_edges", None) is not None and self._src.is_cuda)

    # ---- structure
    def all_edges(self, form="uv", order="eid"):
        if order != "eid":
            raise NotImplementedError("only order='eid' is supported")
        if form == "uv":
            return self._src, self._dst
        e = torch.arange(self._src.numel(), device=self._src.device)
        if form == "eid":
            return e
        if form == "all":
            return self._src, self._dst, e
        raise ValueError(form)

    edges = all_edges

    def index(self, validate=False):
        """Build (once) and return the device index; keyed on the is_reversed tensor."""
        rev = self.edata.get(REVFLAG)
        key = None if rev is None else (rev.data_ptr(), rev._version)
        if self._index is None or self._index_key != key:
            offsets = None
            if (USE_GRAPH_CSR and self.is_batched_on_device() and getattr(self, "max_num_nodes", None) is not None
                    and self.max_num_nodes <= _lib.load().dmp_csr_build_graphs_max_nodes()
                    and int(self.node_offsets.numel()) - 1 == int(self.edge_offsets.numel()) - 1 > 0):
                offsets = (self.node_offsets, self.edge_offsets, int(self.node_offsets.numel()) - 1)
            self._index = GraphIndex(self._src, self._dst, self._n, rev, validate=validate, offsets=offsets)
            self._index_key = key
        self._index.tiling = getattr(self, "tiling", None)
        self._index.node_tiling = getattr(self, "node_tiling", None)
        # a node's degree is at most the number of edges of ITS graph: where the batch knows its largest graph (a host int the
        # dataset keeps), the degree-class tile build sizes its class table by that instead of by MAX_EDGE_CLASSES
        me = getattr(self, "max_num_edges", None)
        self._index.max_degree_hint = int(me) if (me is not None and self.is_batched_on_device()) else None
        return self._index

    def in_degrees(self):
        # dataset.py:1222-1228: cached in ndata["in_deg"]
        if INDEGREE not in self.ndata:
            self.ndata[INDEGREE] = self.index().in_deg
        return self.ndata[INDEGREE]

    def out_degrees(self):
        # dataset.py:1230-1236
        if OUTDEGREE not in self.ndata:
            self.ndata[OUTDEGREE] = self.index().out_deg
        return self.ndata[OUTDEGREE]

    # ---- message passing (generic UDF path = DGL semantics, HIP kernels underneath)
    def update_all(self, message_func, reduce_func, apply_node_func=None):
        """DGL semantics: message UDF once on all E edges (eid order), ``fn.sum`` by
        destination (zero rows for nodes without in-edges), apply UDF once on all N nodes."""
        from . import ops
        if not isinstance(reduce_func, SumReducer):
            raise NotImplementedError("only fn.sum reduction is supported")
        msgs = message_func(EdgeBatch(self))
        m = msgs[reduce_func.msg]
        shape = m.shape
        agg = ops.seg_sum(m.reshape(shape[0], -1), self.index())
        self.ndata[reduce_func.out] = agg.reshape((self._n,) + tuple(shape[1:]))
        if apply_node_func is not None:
            self.ndata.update(apply_node_func(NodeBatch(self)))

    def apply_edges(self, func):
        self.edata.update(func(EdgeBatch(self)))

    def local_var(self):
        return self

    # ---- placement
    def to(self, device):
        device = torch.device(device) if device is not None else None
        if device is None or device == self._src.device:
            return self
        g = BatchedGraph(self._src.to(device), self._dst.to(device), self._n,
                         None if self._bnn is None else self._bnn.to(device),
                         None if self._bne is None else self._bne.to(device),
                         {k: v.to(device) for k, v in self.ndata.items()},
                         {k: v.to(device) for k, v in self.edata.items()})
        if self.node_graph is not None:
            g.node_graph = self.node_graph.to(device)
        if self.edge_graph is not None:
            g.edge_graph = self.edge_graph.to(device)
        return g


def as_batched(graph):
    """``BatchedGraph.from_graph`` under the name the layers call at the top of their ``forward``."""
    return graph if isinstance(graph, BatchedGraph) else BatchedGraph.from_graph(graph)

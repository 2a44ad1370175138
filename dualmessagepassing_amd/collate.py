"""Batch collate on the device: block-diagonal batching of B graphs.

Replaces ``Graph.batch`` -> ``dgl.batch`` (SubgraphCountingMatching/dataset.py:1320-1328)
and mirrors ``GraphAdjDataset.batchify`` (dataset.py:1604-1636).  ``dgl.batch`` semantics:
graphs are concatenated in list order, node ids of graph i are shifted by the number of
nodes of graphs 0..i-1, node/edge frames are concatenated, ``batch_num_nodes`` /
``batch_num_edges`` record the sizes.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .graph import BatchedGraph


def collate_device(local_src, local_dst, num_nodes, num_edges, total_nodes, total_edges, ndata=None,
                   edata=None, with_segments=True, max_nodes=None, max_edges=None):
    """Device collate of per-graph LOCAL edge lists laid out back to back.

    local_src/local_dst [E] int64 (device), num_nodes/num_edges [B] int64 (device);
    total_nodes/total_edges are host ints (the dataset knows them; no device sync here).
    max_nodes/max_edges (optional host ints): the largest graph of the batch, which the dataset also
    knows -- the model's padding helpers then need no ``sizes.max().item()`` round trip per batch.
    Returns a BatchedGraph with global endpoints, ``batch_num_nodes/edges`` and, if
    ``with_segments``, ``node_graph`` / ``edge_graph`` (owning graph per node / edge).
    """
    lib = _lib.load()
    _lib.require_gpu(local_src, local_dst, num_nodes, num_edges)
    for t in (local_src, local_dst, num_nodes, num_edges):
        if t.dtype != torch.int64:
            raise _lib.DmpError("collate inputs must be int64")
    local_src, local_dst = local_src.contiguous(), local_dst.contiguous()
    num_nodes, num_edges = num_nodes.contiguous(), num_edges.contiguous()
    B, N, E = int(num_nodes.numel()), int(total_nodes), int(total_edges)
    if local_src.numel() != E or local_dst.numel() != E or num_edges.numel() != B:
        raise _lib.DmpError("collate: inconsistent sizes")
    dev = local_src.device
    node_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    edge_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
    src = torch.empty(E, dtype=torch.int64, device=dev)
    dst = torch.empty(E, dtype=torch.int64, device=dev)
    eg = torch.empty(E, dtype=torch.int32, device=dev) if with_segments else None
    ng = torch.empty(N, dtype=torch.int32, device=dev) if with_segments else None
    check(lib.dmp_collate(ptr(local_src), ptr(local_dst), ptr(num_nodes), ptr(num_edges), B, N, E,
                          ptr(node_off), ptr(edge_off), ptr(src), ptr(dst), ptr(eg), ptr(ng), stream_ptr()),
          "dmp_collate")
    g = BatchedGraph(src, dst, N, num_nodes, num_edges, ndata, edata)
    g.node_graph, g.edge_graph = ng, eg
    g.node_offsets, g.edge_offsets = node_off, edge_off
    g.max_num_nodes = None if max_nodes is None else int(max_nodes)
    g.max_num_edges = None if max_edges is None else int(max_edges)
    from . import ops
    g.tiling = ops.graph_tiling(node_off, edge_off, B, g.max_num_edges)
    g.node_tiling = ops.graph_node_tiling(node_off, edge_off, B, g.max_num_nodes)
    return g


_COLLATE_JOB = None


def collate_device_many(batches):
    """``collate_device`` for several batches -- the pattern batch and the target batch of a step -- in ONE pair of launches
    (``dmp_collate_jobs``).  ``batches``: list of dicts with ``collate_device``'s arguments (``local_src``, ``local_dst``,
    ``num_nodes``, ``num_edges``, ``total_nodes``, ``total_edges`` and optionally ``ndata``, ``edata``, ``max_nodes``,
    ``max_edges``).  Returns the BatchedGraphs in order."""
    global _COLLATE_JOB
    import ctypes
    lib = _lib.load()
    if _COLLATE_JOB is None:
        P, I = ctypes.c_void_p, ctypes.c_int64

        class _Job(ctypes.Structure):
            _fields_ = [("local_src", P), ("local_dst", P), ("num_nodes", P), ("num_edges", P), ("B", I), ("N", I), ("E", I),
                        ("node_off", P), ("edge_off", P), ("src", P), ("dst", P), ("edge_graph", P), ("node_graph", P)]
        _COLLATE_JOB = _Job
    if not 0 < len(batches) <= 4 or any(int(b["num_nodes"].numel()) == 0 or int(b["num_nodes"].numel()) > 2048 for b in batches):
        return [collate_device(b["local_src"], b["local_dst"], b["num_nodes"], b["num_edges"], b["total_nodes"], b["total_edges"],
                               b.get("ndata"), b.get("edata"), True, b.get("max_nodes"), b.get("max_edges")) for b in batches]
    J = (_COLLATE_JOB * len(batches))()
    made, keep = [], []
    for k, b in enumerate(batches):
        ls, ld, nn, ne = b["local_src"], b["local_dst"], b["num_nodes"], b["num_edges"]
        _lib.require_gpu(ls, ld, nn, ne)
        for t in (ls, ld, nn, ne):
            if t.dtype != torch.int64:
                raise _lib.DmpError("collate inputs must be int64")
        ls, ld, nn, ne = ls.contiguous(), ld.contiguous(), nn.contiguous(), ne.contiguous()
        B, N, E = int(nn.numel()), int(b["total_nodes"]), int(b["total_edges"])
        if ls.numel() != E or ld.numel() != E or ne.numel() != B:
            raise _lib.DmpError("collate: inconsistent sizes")
        dev = ls.device
        offs = torch.empty(2 * (B + 1), dtype=torch.int64, device=dev)
        ends = torch.empty((2, E), dtype=torch.int64, device=dev)
        eg = torch.empty(E, dtype=torch.int32, device=dev)
        ng = torch.empty(N, dtype=torch.int32, device=dev)
        j = J[k]
        j.local_src, j.local_dst, j.num_nodes, j.num_edges, j.B, j.N, j.E = ptr(ls), ptr(ld), ptr(nn), ptr(ne), B, N, E
        j.node_off, j.edge_off, j.src, j.dst = ptr(offs[:B + 1]), ptr(offs[B + 1:]), ptr(ends[0]), ptr(ends[1])
        j.edge_graph, j.node_graph = ptr(eg), ptr(ng)
        keep.append((ls, ld, nn, ne))
        made.append((b, nn, ne, N, B, offs, ends, eg, ng))
    check(lib.dmp_collate_jobs(J, len(batches), stream_ptr()), "dmp_collate_jobs")
    from . import ops
    out = []
    for b, nn, ne, N, B, offs, ends, eg, ng in made:
        g = BatchedGraph(ends[0], ends[1], N, nn, ne, b.get("ndata"), b.get("edata"))
        g.node_graph, g.edge_graph = ng, eg
        g.node_offsets, g.edge_offsets = offs[:B + 1], offs[B + 1:]
        g.max_num_nodes = None if b.get("max_nodes") is None else int(b["max_nodes"])
        g.max_num_edges = None if b.get("max_edges") is None else int(b["max_edges"])
        g.tiling = ops.graph_tiling(g.node_offsets, g.edge_offsets, B, g.max_num_edges)
        g.node_tiling = ops.graph_node_tiling(g.node_offsets, g.edge_offsets, B, g.max_num_nodes)
        out.append(g)
    return out


def batch(graphs, device=None):
    """``Graph.batch(list_of_graphs)`` (dataset.py:1320-1328): list of single graphs -> one
    block-diagonal BatchedGraph on ``device`` (default: the graphs' device, which must be a GPU)."""
    assert isinstance(graphs, list) and len(graphs) > 0
    for g in graphs:
        if g.batch_size != 1:
            raise ValueError("batch() takes single graphs")
    dev = torch.device(device) if device is not None else graphs[0].device
    nn_host = [g.number_of_nodes() for g in graphs]
    ne_host = [g.number_of_edges() for g in graphs]
    ls = torch.cat([g._src for g in graphs]).to(dev)
    ld = torch.cat([g._dst for g in graphs]).to(dev)
    ndata = {k: torch.cat([g.ndata[k] for g in graphs], 0).to(dev) for k in graphs[0].ndata}
    edata = {k: torch.cat([g.edata[k] for g in graphs], 0).to(dev) for k in graphs[0].edata}
    nn = torch.tensor(nn_host, dtype=torch.int64).to(dev)
    ne = torch.tensor(ne_host, dtype=torch.int64).to(dev)
    return collate_device(ls, ld, nn, ne, sum(nn_host), sum(ne_host), ndata, edata,
                          max_nodes=max(nn_host), max_edges=max(ne_host))


def _offsets(sizes):
    off = torch.zeros(sizes.numel() + 1, dtype=torch.int64, device=sizes.device)
    torch.cumsum(sizes, 0, out=off[1:])
    return off


def _pre_pad(flat, sizes, seg):
    """``batch_convert_tensor_to_tensor(rows, pre_pad=True)`` (utils/dl.py:89-110): row i of the
    result holds sample i's values right-aligned, zeros in front.  One host sync (the padded length,
    as the reference's ``max(batch_lens)``)."""
    B, m = int(sizes.numel()), int(sizes.max().item()) if sizes.numel() else 0
    out = torch.zeros((B, m), dtype=flat.dtype, device=flat.device)
    if flat.numel():
        off = _offsets(sizes)
        seg = seg.long()
        pos = torch.arange(flat.numel(), device=flat.device) - off[seg]
        out.view(-1).index_copy_(0, seg * m + (m - sizes[seg]) + pos, flat)
    return out


def subiso_weights(pattern, graph, sub_flat, sample_ptr, return_weights=("node", "edge"), validate=False,
                   work_hint=0):
    """Node / edge subisomorphism weights of a whole batch on the device
    (``GraphAdjDataset.calculate_node_weights`` / ``calculate_edge_weights`` +
    ``compute_nodeseq_subisoweights`` / ``compute_edgeseq_subisoweights``, dataset.py:54-107,1491-1520),
    pre-padded like ``batchify`` does (dataset.py:1618-1634).

    pattern / graph: BatchedGraphs from ``collate_device`` / ``batch`` with ``edata["label"]``;
    sub_flat [T] int64: the samples' ``subisomorphisms`` tensors flattened back to back;
    sample_ptr [B+1] int64: first element of each sample.  Returns ``(node_weights [B, max_nodes] |
    None, edge_weights [B, max_edges] | None)``, int64."""
    lib = _lib.load()
    if isinstance(return_weights, str):
        return_weights = return_weights.split(",")
    _lib.require_gpu(sub_flat, sample_ptr)
    if sub_flat.dtype != torch.int64 or sample_ptr.dtype != torch.int64:
        raise _lib.DmpError("subiso_weights: int64 inputs expected")
    sub_flat, sample_ptr = sub_flat.contiguous(), sample_ptr.contiguous()
    B, T = graph.batch_size, int(sub_flat.numel())
    if pattern.batch_size != B or sample_ptr.numel() != B + 1:
        raise _lib.DmpError("subiso_weights: batch sizes disagree")
    dev = sub_flat.device
    g_node_off = getattr(graph, "node_offsets", None)
    if g_node_off is None:
        g_node_off = _offsets(graph.batch_num_nodes())
    status = torch.zeros(2, dtype=torch.int32, device=dev)
    st = stream_ptr()
    node_w = edge_w = None
    if "node" in return_weights:
        N = graph.number_of_nodes()
        flat = torch.empty(N, dtype=torch.int64, device=dev)
        check(lib.dmp_subiso_node_weights(ptr(sub_flat), T, ptr(sample_ptr), B, ptr(g_node_off), ptr(flat), N,
                                          ptr(status[0:]), st), "dmp_subiso_node_weights")
        node_w = _pre_pad(flat, graph.batch_num_nodes(), _segment_ids(graph, "node"))
    if "edge" in return_weights:
        from .constants import EDGELABEL
        E, PE = graph.number_of_edges(), pattern.number_of_edges()
        p_node_off = getattr(pattern, "node_offsets", None)
        p_edge_off = getattr(pattern, "edge_offsets", None)
        if p_node_off is None:
            p_node_off, p_edge_off = _offsets(pattern.batch_num_nodes()), _offsets(pattern.batch_num_edges())
        flat = torch.empty(E, dtype=torch.int64, device=dev)
        active = torch.empty(max(PE, 1), dtype=torch.uint8, device=dev)
        p_seg = _segment_ids(pattern, "edge")
        check(lib.dmp_pattern_edge_active(ptr(pattern._src), ptr(pattern._dst), ptr(p_edge_off), ptr(p_seg), PE,
                                          ptr(active), st), "dmp_pattern_edge_active")
        pn, pe = pattern.batch_num_nodes(), pattern.batch_num_edges()
        rows = (sample_ptr[1:] - sample_ptr[:-1]) // pn.clamp(min=1)
        work_ptr = _offsets(rows * pe)
        idx = graph.index()
        check(lib.dmp_subiso_edge_weights(ptr(sub_flat), T, ptr(sample_ptr), ptr(work_ptr), B, ptr(p_node_off),
                                          ptr(p_edge_off), ptr(pattern._src), ptr(pattern._dst),
                                          ptr(pattern.edata[EDGELABEL].contiguous()), ptr(active), ptr(g_node_off),
                                          ptr(idx.out_ptr), ptr(idx.out_ent), ptr(idx.dst32),
                                          ptr(graph.edata[EDGELABEL].contiguous()), ptr(flat), E, int(work_hint),
                                          ptr(status[1:]), st), "dmp_subiso_edge_weights")
        edge_w = _pre_pad(flat, graph.batch_num_edges(), _segment_ids(graph, "edge"))
    if validate and int(status.sum().item()) != 0:
        raise _lib.DmpError("subiso_weights: a subisomorphism row refers to a node outside its target graph")
    return node_w, edge_w


def _segment_ids(g, kind):
    seg = getattr(g, "node_graph" if kind == "node" else "edge_graph", None)
    if seg is None:
        sizes = g.batch_num_nodes() if kind == "node" else g.batch_num_edges()
        total = g.number_of_nodes() if kind == "node" else g.number_of_edges()
        seg = torch.repeat_interleave(torch.arange(sizes.numel(), device=sizes.device), sizes,
                                      output_size=total).to(torch.int32)
    return seg


def batchify(samples, return_weights=None, device=None):
    """``GraphAdjDataset.batchify`` (dataset.py:1604-1636) for samples
    ``{"id", "pattern", "graph", "counts"[, "subisomorphisms"]}`` ->
    ``(_id, pattern, graph, counts, (node_weights, edge_weights))``.  With ``return_weights``
    ("node", "edge" or "node,edge") the subisomorphism weights are computed for the whole batch on
    the device (``subiso_weights``) instead of per sample by host counters."""
    _id = [x["id"] for x in samples]
    pattern = batch([x["pattern"] for x in samples], device)
    graph = batch([x["graph"] for x in samples], device)
    counts = torch.tensor([x["counts"] for x in samples], dtype=torch.int64)
    if device is not None:
        counts = counts.to(device)
    if return_weights is None:
        return _id, pattern, graph, counts, (None, None)
    dev = graph.device
    subs = [x["subisomorphisms"].reshape(-1).to(torch.int64) for x in samples]
    sizes = [int(t.numel()) for t in subs]
    sub_flat = torch.cat(subs).to(dev) if sum(sizes) else torch.zeros(0, dtype=torch.int64, device=dev)
    ptr_host = [0]
    for n in sizes:
        ptr_host.append(ptr_host[-1] + n)
    sample_ptr = torch.tensor(ptr_host, dtype=torch.int64).to(dev)
    hint = sum((n // max(x["pattern"].number_of_nodes(), 1)) * x["pattern"].number_of_edges()
               for n, x in zip(sizes, samples))
    return _id, pattern, graph, counts, subiso_weights(pattern, graph, sub_flat, sample_ptr, return_weights,
                                                       work_hint=hint)


_CONCAT_JOB = None


def concat_pairs(pairs):
    """``[cat([a, b + add]) for (a, b, add) in pairs]`` (1-D tensors; ``a`` may be ``(n, fill)`` for a constant fp32
    block; ``add`` only for int64).  Device tensors: ONE launch for all pairs (csrc/dmp_graph.hip::concat_pairs_k)."""
    global _CONCAT_JOB
    import ctypes
    from . import _lib

    def plain(a, b, add):
        if isinstance(a, tuple):
            a = torch.full((a[0],), a[1], dtype=b.dtype, device=b.device)
        return torch.cat([a.reshape(-1), b.reshape(-1) + add if add else b.reshape(-1)])

    ok = 0 < len(pairs) <= 12 and all(
        b.is_cuda and b.element_size() in (1, 4, 8) and (not add or b.dtype == torch.int64)
        and ((isinstance(a, tuple) and b.dtype == torch.float32) or (torch.is_tensor(a) and a.is_cuda and a.dtype == b.dtype))
        for a, b, add in pairs)
    if not ok:
        return [plain(*p) for p in pairs]
    lib = _lib.load()
    if _CONCAT_JOB is None:
        class _Job(ctypes.Structure):
            _fields_ = [("a", ctypes.c_void_p), ("na", ctypes.c_int64), ("b", ctypes.c_void_p), ("nb", ctypes.c_int64),
                        ("out", ctypes.c_void_p), ("elem_size", ctypes.c_int), ("add_b", ctypes.c_int64),
                        ("fill_a", ctypes.c_float)]
        _CONCAT_JOB = _Job
    J, keep, outs = (_CONCAT_JOB * len(pairs))(), [], []
    for k, (a, b, add) in enumerate(pairs):
        b = b.reshape(-1).contiguous()
        if isinstance(a, tuple):
            na, fill, a = int(a[0]), float(a[1]), None
        else:
            a = a.reshape(-1).contiguous()
            na, fill = a.numel(), 0.0
        out = torch.empty(na + b.numel(), dtype=b.dtype, device=b.device)
        J[k].a, J[k].na, J[k].b, J[k].nb = (None if a is None else a.data_ptr()), na, b.data_ptr(), b.numel()
        J[k].out, J[k].elem_size, J[k].add_b, J[k].fill_a = out.data_ptr(), b.element_size(), int(add or 0), fill
        keep.append((a, b))
        outs.append(out)
    _lib.check(lib.dmp_concat_pairs(J, len(pairs), _lib.stream_ptr()), "dmp_concat_pairs")
    return outs


def union_graphs(a, b):
    """Block-diagonal union of two (batched) graphs: nodes/edges of ``a`` first, then ``b`` with its
    node ids shifted by ``a.number_of_nodes()`` -- ``dgl.batch([a, b])`` semantics on the structure.
    Used to run a SHARED rep-net once over pattern and target batches (same weights, per-row
    ops, no BatchNorm), instead of twice.  Carries ``is_reversed`` and cached degrees."""
    from .constants import INDEGREE, OUTDEGREE, REVFLAG
    na = a.number_of_nodes()
    if (REVFLAG in a.edata) != (REVFLAG in b.edata):
        raise ValueError("union_graphs: is_reversed must be present on both graphs or on neither")
    pairs = [(a._src, b._src, na), (a._dst, b._dst, na), (a.batch_num_nodes(), b.batch_num_nodes(), 0),
             (a.batch_num_edges(), b.batch_num_edges(), 0)]
    extra = []
    if REVFLAG in a.edata:
        extra.append(("e", REVFLAG))
        pairs.append((a.edata[REVFLAG], b.edata[REVFLAG], 0))
    for k in (INDEGREE, OUTDEGREE):
        if k in a.ndata and k in b.ndata:
            extra.append(("n", k))
            pairs.append((a.ndata[k], b.ndata[k], 0))
    # per-graph offsets of the union (pattern graphs, then target graphs) when both sides carry theirs: the tiled
    # scatter-add of the layer's backward walks whole graphs (ops.graph_tiling)
    offs = all(getattr(x, "node_offsets", None) is not None and getattr(x, "max_num_edges", None) is not None for x in (a, b))
    if offs:
        pairs.append((a.node_offsets[:-1], b.node_offsets, na))
        pairs.append((a.edge_offsets[:-1], b.edge_offsets, a.number_of_edges()))
    out = concat_pairs(pairs)                                # all structure arrays of the union in one launch
    g = BatchedGraph(out[0], out[1], na + b.number_of_nodes(), out[2], out[3])
    for (where, k), t in zip(extra, out[4:4 + len(extra)]):
        (g.edata if where == "e" else g.ndata)[k] = t
    if offs:
        from . import ops
        g.node_offsets, g.edge_offsets = out[-2], out[-1]
        g.max_num_nodes = max(a.max_num_nodes or 0, b.max_num_nodes or 0) or None
        g.max_num_edges = max(a.max_num_edges, b.max_num_edges)
        g.tiling = ops.graph_tiling(g.node_offsets, g.edge_offsets, a.batch_size, a.max_num_edges, b.batch_size, b.max_num_edges)
        g.node_tiling = ops.graph_node_tiling(g.node_offsets, g.edge_offsets, a.batch_size, a.max_num_nodes, b.batch_size, b.max_num_nodes)
    return g


class CompactedEdges:
    """What ``compact_gated_edges`` hands back: ``graph`` (the kept edges + padding, ``capacity`` edge rows), ``eid_map``
    int64 [capacity] (the edge's row in the graph it came from; 0 for padding rows), ``gate`` float [capacity, 1] (0 for
    padding rows), ``num_edges`` (rows of the graph it came from)."""

    def __init__(self, graph, eid_map, gate, num_edges, kept):
        self.graph, self.eid_map, self.gate, self.num_edges, self.kept = graph, eid_map, gate, int(num_edges), kept

    def take(self, rows):
        """The compacted batch's rows of a per-edge tensor of the original graph (padding rows: row 0, gate 0)."""
        return rows.index_select(0, self.eid_map)

    def expand(self, rows_c):
        """[num_edges, width] rows of the original graph from the compacted batch's rows: zero rows for the edges the gate
        removed (what the reference computes for them: basemodel.py:1515-1531), differentiable."""
        return _ExpandRows.apply(rows_c, self.eid_map, self.gate, self.num_edges)


class _ExpandRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows_c, eid_map, gate, num_edges):
        out = torch.zeros((num_edges, rows_c.size(1)), dtype=rows_c.dtype, device=rows_c.device)
        # padding rows are zero rows added into row 0: kept rows have distinct targets, so the order of the adds is immaterial
        out.index_add_(0, eid_map, rows_c * gate.view(-1, 1).ne(0).to(rows_c.dtype))
        ctx.save_for_backward(eid_map, gate)
        return out

    @staticmethod
    def backward(ctx, d):
        eid_map, gate = ctx.saved_tensors
        return d.index_select(0, eid_map) * gate.view(-1, 1).ne(0).to(d.dtype), None, None, None


def out_degrees(graph):
    """``graph.out_degrees()`` (dataset.py:1230-1236) from the edge list alone (``dmp_out_degrees``), cached in
    ``ndata["out_deg"]`` like the reference's: no CSR of the graph is built for it."""
    from .constants import OUTDEGREE
    if OUTDEGREE not in graph.ndata:
        lib = _lib.load()
        src = graph._src
        _lib.require_gpu(src)
        deg = torch.empty(graph.number_of_nodes(), dtype=torch.int64, device=src.device)
        check(lib.dmp_out_degrees(ptr(src), src.numel(), deg.numel(), ptr(deg), stream_ptr()), "dmp_out_degrees")
        graph.ndata[OUTDEGREE] = deg
    return graph.ndata[OUTDEGREE]


def compact_gated_edges(graph, e_gate, capacity, status):
    """The edges of a block-diagonal batch that a filter gate keeps, as a batch of exactly ``capacity`` edges
    (``dmp_gate_compact``, csrc/dmp_compact.hip) -- or None where that does not apply (no per-graph offsets, ``capacity``
    not below the edge count).

    Why this is the same function: the reference multiplies the target's edge embeddings and every layer's edge update by
    the gate (basemodel.py:1515-1531, dmpnn.py:262-275), so a gate-0 edge is a zero row throughout, adds nothing to a node
    sum, a pooled sum or a gradient, and only the DEGREES of its endpoints see it (dmpnn.py:101,144-146): the compacted graph
    carries the whole graph's out-degrees in ``ndata["out_deg"]`` (shared frames, as the reference's layer leaves them).
    Kept edges stay in ascending eid inside their graph (fixed-order sums keep their order); ``capacity - kept`` padding
    edges (gate 0, self-loops dealt over the graphs and their nodes) make the shapes independent of the labels, so a
    recorded step replays.  ``status`` (int32 [1], device, owned by the caller) is OR-ed with 1 if more than ``capacity``
    edges were kept (the result is truncated and must not be used), with 2 if padding fell on a graph without nodes.
    No host sync."""
    from . import ops
    from .constants import OUTDEGREE, REVFLAG
    lib = _lib.load()
    E, N, B, cap = graph.number_of_edges(), graph.number_of_nodes(), graph.batch_size, int(capacity)
    node_off, edge_off = getattr(graph, "node_offsets", None), getattr(graph, "edge_offsets", None)
    if node_off is None or edge_off is None or E == 0 or cap <= 0 or cap >= E or getattr(graph, "max_num_edges", None) is None:
        return None
    gate = e_gate.reshape(-1)
    _lib.require_gpu(gate, graph._src, status)
    if gate.dtype != torch.float32 or gate.numel() != E or status.dtype != torch.int32:
        raise _lib.DmpError("compact_gated_edges: float gate with one entry per edge, int32 status word")
    gate = gate.contiguous()
    dev = gate.device
    ids = torch.empty((3, cap), dtype=torch.int64, device=dev)               # src | dst | eid map
    gate_c = torch.empty((cap, 1), dtype=torch.float32, device=dev)
    sizes = torch.empty(2 * B + 1, dtype=torch.int64, device=dev)            # edges per graph | edge offsets
    kept = torch.empty(B, dtype=torch.int32, device=dev)
    rev = graph.edata.get(REVFLAG)
    rev_c = None
    if rev is not None:
        if rev.element_size() != 1:
            raise _lib.DmpError("is_reversed must be bool or uint8")
        rev = rev.contiguous().view(-1)
        rev_c = torch.empty(cap, dtype=rev.dtype, device=dev)
    deg = None if OUTDEGREE in graph.ndata else torch.empty(N, dtype=torch.int64, device=dev)
    big = graph.max_num_nodes is None or graph.max_num_nodes > int(lib.dmp_gate_compact_hist_nodes())
    check(lib.dmp_gate_compact(ptr(gate), ptr(graph._src), ptr(graph._dst), ptr(rev), ptr(node_off), ptr(edge_off), B, N, E, cap,
                               1 if big else 0, ptr(kept), ptr(deg), ptr(ids[0]), ptr(ids[1]), ptr(rev_c), ptr(ids[2]), ptr(gate_c),
                               ptr(sizes[:B]), ptr(sizes[B:]), ptr(status), stream_ptr()), "dmp_gate_compact")
    if deg is not None:
        graph.ndata[OUTDEGREE] = deg
    g = BatchedGraph(ids[0], ids[1], N, graph.batch_num_nodes(), sizes[:B], graph.ndata, {} if rev_c is None else {REVFLAG: rev_c},
                     share_frames=True)
    g.node_graph = graph.node_graph
    g.node_offsets, g.edge_offsets = node_off, sizes[B:]
    g.max_num_nodes = graph.max_num_nodes
    g.max_num_edges = int(graph.max_num_edges) + cap // B + 1                # kept edges of a graph + its share of the padding
    g.tiling = ops.graph_tiling(node_off, g.edge_offsets, B, g.max_num_edges)
    g.node_tiling = ops.graph_node_tiling(node_off, g.edge_offsets, B, g.max_num_nodes)
    gate_c._dmp_dense_gate = True                         # ones but for the padding rows: the masked-row kernels have nothing to skip
    return CompactedEdges(g, ids[2], gate_c, E, kept)

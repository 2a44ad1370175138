"""Explicit directed line-graph (dual graph) construction on the device, integer-exact.

Mirror of ``convert_to_dual_graph`` (SubgraphCountingMatching/utils/graph.py:74-169, DGL
branch), applied to a whole batch at once (the reference loops over samples on the host,
train.py:417-446).  Per graph:

  dual nodes  one per distinct edge id (``edata["id"]``), represented by the FIRST edge
              carrying that id (graph.py:80-95); without an id frame one per edge (:96-103);
              ids that no edge carries ("holes", e.g. between E and max_ne after
              add_reversed_edges) are removed and the rest renumbered compactly (:161-164)
  dual edges  for e = 0..E-1 in eid order, s = src(e), for every in-edge i of s in ascending
              eid: plain branch (i -> e) (:126-134); id+label branch (id[i] -> id[e]) kept only
              the first time the key (id[i], node_label[s], id[e]) occurs (:110-125)
  frames      dual ndata = edge frames (first edge per id), dual edata = node frames gathered
              at the shared primal node s; ``edata["id"]`` = s if nodes carry no id (:135-147)

Kernels: CSR by destination (ascending eid) -> count/scan/fill of the candidate dual edges
in emission order -> hash-table "first occurrence" filter (atomicMin on the emission index,
so the survivor does not depend on thread timing).  Stream compaction of the kept entries
uses torch boolean indexing (memory plumbing).
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .constants import EDGEID, NODEID, NODELABEL
from .graph import BatchedGraph, as_batched


def _offsets(counts):
    off = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=counts.device)
    torch.cumsum(counts, 0, out=off[1:])
    return off


def _candidates(graph):
    """All (in-edge i, edge e, shared node s) triples in the reference's emission order."""
    lib = _lib.load()
    ix = graph.index()
    src = graph._src.contiguous()
    E, dev = graph.number_of_edges(), graph.device
    cnt = torch.empty(E, dtype=torch.int64, device=dev)
    check(lib.dmp_line_graph_count(ptr(ix.in_ptr), ptr(src), E, ptr(cnt), stream_ptr()), "dmp_line_graph_count")
    off = torch.empty(E + 1, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.dmp_scan_workspace_words(E), dtype=torch.int64, device=dev)
    check(lib.dmp_exclusive_scan_i64(ptr(cnt), E, ptr(off), ptr(ws), stream_ptr()), "dmp_exclusive_scan_i64")
    M = int(off[-1].item())  # size of the output: the one host sync of the transform
    ci = torch.empty(M, dtype=torch.int64, device=dev)
    ce = torch.empty(M, dtype=torch.int64, device=dev)
    cs = torch.empty(M, dtype=torch.int64, device=dev)
    if M == 0:
        return ci, ce, cs
    check(lib.dmp_line_graph_fill(ptr(ix.in_ptr), ptr(ix.in_ent), ptr(src), ptr(off), E, ptr(ci), ptr(ce), ptr(cs),
                                  stream_ptr()), "dmp_line_graph_fill")
    return ci, ce, cs


def convert_to_dual_graph(graph):
    """BatchedGraph (one graph or a batch) -> its directed line graph, same class."""
    graph = as_batched(graph)
    lib = _lib.load()
    _lib.require_gpu(graph._src)
    dev = graph.device
    E, N, B = graph.number_of_edges(), graph.number_of_nodes(), graph.batch_size
    bnn = graph.batch_num_nodes().to(torch.int64)
    bne = graph.batch_num_edges().to(torch.int64)
    node_off, edge_off = _offsets(bnn), _offsets(bne)
    edge_graph = torch.repeat_interleave(torch.arange(B, device=dev), bne)
    has_eid = EDGEID in graph.edata and E > 0
    i64 = dict(dtype=torch.int64, device=dev)

    # ---- dual nodes (graph.py:80-103)
    if has_eid:
        eids = graph.edata[EDGEID].to(torch.int64).contiguous()
        # largest id + 1 per graph.  The edges of a graph are contiguous: a segmented maximum (a thread per graph) -- the
        # scatter_reduce form ran 5.2 ms of atomic maxima onto B addresses, 78 % of the whole transform.  Ids are far
        # below 2^53, so the float64 detour of segment_reduce is exact.
        kmax = torch.segment_reduce((eids + 1).to(torch.float64), "max", lengths=bne, unsafe=True, initial=0.0).to(torch.int64)
        id_off = _offsets(kmax)               # graph g owns dual-node slots [id_off[g], id_off[g+1])
        gid = eids + id_off[edge_graph]       # edge id made unique across the batch
        K = int(id_off[-1].item())
        first = torch.empty(K, **i64)
        check(lib.dmp_first_edge_of_id(ptr(gid), E, K, ptr(first), stream_ptr()), "dmp_first_edge_of_id")
        kept = first >= 0                     # holes: ids no edge carries (graph.py:161-164)
        rep = first[kept]                     # representative (first) edge of every surviving dual node
        new_index = torch.cumsum(kept.to(torch.int64), 0) - 1
        slot_graph = torch.repeat_interleave(torch.arange(B, device=dev), kmax)
        dual_bnn = torch.bincount(slot_graph[kept], minlength=B)
        dual_ndata = {k: v[rep] for k, v in graph.edata.items()}
    else:
        dual_bnn = bne.clone()
        dual_ndata = dict(graph.edata)
        if EDGEID not in graph.edata:         # graph.py:102-103: arange(E) per graph
            dual_ndata[EDGEID] = torch.arange(E, **i64) - edge_off[edge_graph]

    # ---- dual edges (graph.py:104-134)
    if E > 0:
        ci, ce, cs = _candidates(graph)
    else:
        ci = ce = cs = torch.zeros(0, **i64)
    if has_eid and NODELABEL in graph.ndata:
        a, b = gid[ci], gid[ce]
        lab = graph.ndata[NODELABEL].to(torch.int64)[cs].contiguous()
        M = ci.numel()
        keep = torch.ones(M, dtype=torch.uint8, device=dev)
        if M > 0:
            table = torch.empty(lib.dmp_dedupe_table_words(M), **i64)
            check(lib.dmp_dedupe_first(ptr(a.contiguous()), ptr(lab), ptr(b.contiguous()), M, ptr(table), ptr(keep),
                                       stream_ptr()), "dmp_dedupe_first")
        keep = keep.bool()
        dsrc, ddst, pay = new_index[a[keep]], new_index[b[keep]], cs[keep]
        dual_bne = torch.bincount(edge_graph[ce[keep]], minlength=B) if M > 0 else torch.zeros(B, **i64)
    elif has_eid:
        # graph.py:126-134 taken with an id frame but no node labels would mix edge indices
        # (dual edges) with edge ids (dual nodes); the training pipeline never reaches it.
        raise NotImplementedError("edge ids without node labels: inconsistent branch of the reference")
    else:
        dsrc, ddst, pay = ci, ce, cs          # edge indices are already global over the batch
        dual_bne = torch.bincount(edge_graph[ce], minlength=B) if E > 0 else torch.zeros(B, **i64)

    # ---- dual edge frames = node frames at the shared node (graph.py:135-147)
    dual_edata = {k: v[pay] for k, v in graph.ndata.items()}
    if NODEID not in graph.ndata:
        dual_edata[NODEID] = pay - node_off[torch.repeat_interleave(torch.arange(B, device=dev), bnn)[pay]] \
            if pay.numel() else torch.zeros(0, **i64)
    # constants.py:19-22: NODEID == EDGEID and NODELABEL == EDGELABEL, so the renames at
    # graph.py:149-159 are no-ops.

    out = BatchedGraph(dsrc, ddst, int(dual_bnn.sum().item()), dual_bnn if graph._bnn is not None else None,
                       dual_bne if graph._bne is not None else None, dual_ndata, dual_edata)
    return out


def dual_subisomorphisms(pattern, graph, sub_flat, sample_ptr, rows_host=None, validate=False):
    """``get_dual_subisomorphisms`` + the ``g_eid[...]`` mapping of ``convert_to_dual_data`` (utils/graph.py:277-316,
    train.py:417-446) for a whole batch on the device: the samples' node maps (``subisomorphisms`` rows, flattened back to
    back in ``sub_flat`` with ``sample_ptr`` [B+1]) become maps onto the EDGES of the target graphs -- i.e. onto the
    nodes of their line graphs (``convert_to_dual_graph``), which is what the matching losses need after
    ``--convert_dual``.  Returns ``(dual_flat, dual_ptr)``: per sample ``rows_i`` rows of ``pattern_edges_i`` sample-local
    edge ids, ``dual_ptr`` [B+1] the first element of each sample.  ``rows_host`` (optional list of the samples' row
    counts, which a dataset knows) avoids the one device round trip that sizes the output."""
    lib = _lib.load()
    pattern, graph = as_batched(pattern), as_batched(graph)
    _lib.require_gpu(sub_flat, sample_ptr)
    if sub_flat.dtype != torch.int64 or sample_ptr.dtype != torch.int64:
        raise _lib.DmpError("dual_subisomorphisms: int64 inputs expected")
    sub_flat, sample_ptr = sub_flat.contiguous(), sample_ptr.contiguous()
    B = graph.batch_size
    if pattern.batch_size != B or sample_ptr.numel() != B + 1:
        raise _lib.DmpError("dual_subisomorphisms: batch sizes disagree")
    dev = sub_flat.device
    pn, pe = pattern.batch_num_nodes().to(torch.int64), pattern.batch_num_edges().to(torch.int64)
    p_node_off, p_edge_off = _offsets(pn), _offsets(pe)
    g_node_off, g_edge_off = _offsets(graph.batch_num_nodes().to(torch.int64)), _offsets(graph.batch_num_edges().to(torch.int64))
    rows = (sample_ptr[1:] - sample_ptr[:-1]) // pn.clamp(min=1)
    work_ptr = _offsets(rows * pe)
    if rows_host is not None and getattr(pattern, "edge_sizes_host", None) is not None:
        total = int(sum(int(r) * int(e) for r, e in zip(rows_host, pattern.edge_sizes_host)))      # no device round trip
    else:
        total = int(work_ptr[-1].item())
    PE = pattern.number_of_edges()
    ix = graph.index()
    out = torch.empty(total, dtype=torch.int64, device=dev)
    first = torch.empty(B, dtype=torch.int64, device=dev)
    ws = torch.empty(2 * PE + B, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    from .constants import EDGELABEL
    check(lib.dmp_dual_subisomorphisms(ptr(sub_flat), int(sub_flat.numel()), ptr(sample_ptr), ptr(work_ptr), total, B,
                                       ptr(p_node_off), ptr(p_edge_off), ptr(pattern._src.contiguous()), ptr(pattern._dst.contiguous()),
                                       ptr(pattern.edata[EDGELABEL].contiguous()), PE, ptr(g_node_off), ptr(g_edge_off),
                                       ptr(ix.out_ptr), ptr(ix.out_ent), ptr(ix.dst32), ptr(graph.edata[EDGELABEL].contiguous()),
                                       ptr(ws), ptr(first), ptr(out), ptr(status), stream_ptr()), "dmp_dual_subisomorphisms")
    if validate and int(status.item()) != 0:
        raise _lib.DmpError("dual_subisomorphisms: a subisomorphism row refers to a node outside its target graph")
    return out, work_ptr

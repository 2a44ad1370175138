"""Dataset passes that define the edge order / flags / degrees the layer sees, on the device.

Mirrors of (SubgraphCountingMatching/):
  add_reversed_edges          train.py:299-327 (GraphAdj branch; Graph.add_edges dataset.py:1261-1293)
  calculate_degrees           train.py:330-350
  compute_largest_eigenvalues utils/graph.py:40-71
  calculate_eigenvalues       train.py:368-380
All of them work on a whole BatchedGraph at once (per-graph semantics preserved through the
batch offsets) instead of looping over samples on the host.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .constants import (EDGEEIGENV, EDGEID, EDGELABEL, INDEGREE, NODEEIGENV, OUTDEGREE, REVFLAG)
from .graph import BatchedGraph


def _offsets(counts):
    off = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=counts.device)
    torch.cumsum(counts, 0, out=off[1:])
    return off


def add_reversed_edges(graph, max_ne, max_nel):
    """Append, per graph, the reversed copy of every edge after all forward edges
    (train.py:303-316): ``id = max_ne + arange(E_g)``, ``label += max_nel``,
    ``is_reversed = 1`` (forward edges get 0, DGL's zero fill).  Cached degrees are
    updated like ``Graph.add_edges`` does (dataset.py:1289-1293).  No-op if the graph
    already carries ``is_reversed`` (train.py:302)."""
    if REVFLAG in graph.edata:
        return graph
    lib = _lib.load()
    src, dst = graph.all_edges(form="uv", order="eid")
    _lib.require_gpu(src)
    E, B = graph.number_of_edges(), graph.batch_size
    bne = graph.batch_num_edges().to(torch.int64)
    edge_off = _offsets(bne)
    eid = graph.edata[EDGEID].contiguous()
    el = graph.edata[EDGELABEL].contiguous()
    dev = src.device
    o = {k: torch.empty(2 * E, dtype=torch.int64, device=dev) for k in ("src", "dst", "eid", "el")}
    o_rev = torch.empty(2 * E, dtype=torch.uint8, device=dev)
    check(lib.dmp_add_reversed_edges(ptr(src.contiguous()), ptr(dst.contiguous()), ptr(eid), ptr(el), ptr(edge_off),
                                     B, E, int(max_ne), int(max_nel), ptr(o["src"]), ptr(o["dst"]), ptr(o["eid"]),
                                     ptr(o["el"]), ptr(o_rev), stream_ptr()), "dmp_add_reversed_edges")
    edata = {EDGEID: o["eid"], EDGELABEL: o["el"], REVFLAG: o_rev.bool()}
    for k, v in graph.edata.items():  # any other edge frame: reversed copies are zero-filled (DGL semantics)
        if k in (EDGEID, EDGELABEL):
            continue
        out = torch.zeros((2 * E,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev)
        fwd = (~edata[REVFLAG]).nonzero(as_tuple=True)[0]
        out[fwd] = v
        edata[k] = out
    ndata = dict(graph.ndata)
    if INDEGREE in ndata or OUTDEGREE in ndata:
        n = graph.number_of_nodes()
        if INDEGREE in ndata:   # += bincount(v) with v = old sources
            ndata[INDEGREE] = ndata[INDEGREE] + torch.bincount(src, minlength=n)
        if OUTDEGREE in ndata:  # += bincount(u) with u = old destinations
            ndata[OUTDEGREE] = ndata[OUTDEGREE] + torch.bincount(dst, minlength=n)
    g = BatchedGraph(o["src"], o["dst"], graph.number_of_nodes(), graph._bnn,
                     None if graph._bne is None else graph._bne * 2, ndata, edata)
    g.node_graph = graph.node_graph
    if graph.edge_graph is not None:
        g.edge_graph = torch.repeat_interleave(torch.arange(B, dtype=torch.int32, device=dev), 2 * bne)
    return g


def calculate_degrees(graph):
    """train.py:340-350: cache ``in_deg`` / ``out_deg`` in ndata."""
    graph.in_degrees()
    graph.out_degrees()
    return graph


def _segment_max(values, lengths):
    """Maximum over consecutive runs of ``lengths`` entries (``-inf`` for an empty run): the edges of a graph are
    contiguous in a batch, so this is a segmented reduction -- not a scatter of atomic maxima onto B addresses."""
    return torch.segment_reduce(values, "max", lengths=lengths.to(torch.int64), unsafe=True, initial=float("-inf"))


def compute_largest_eigenvalues(graph):
    """utils/graph.py:40-71, per graph of the batch: ``max_e(out_deg[u] + in_deg[v])`` and
    ``max_e(in_deg[u] + out_deg[v])`` as float32 [B] each (``-inf`` for graphs without edges,
    where the reference's ``.max()`` of an empty tensor raises)."""
    in_deg = graph.in_degrees().float()
    out_deg = graph.out_degrees().float()
    u, v = graph.all_edges(form="uv", order="eid")
    nd = out_deg[u] + in_deg[v]
    ed = in_deg[u] + out_deg[v]
    B = graph.batch_size
    if B == 1:
        return nd.max().view(1), ed.max().view(1)
    bne = graph.batch_num_edges()
    return _segment_max(nd, bne), _segment_max(ed, bne)


def calculate_eigenvalues(graph):
    """train.py:372-380: ``node_eigenv`` [N,1] / ``edge_eigenv`` [E,1] = the graph's bound clamped
    to >= 1.0, repeated over its nodes / edges."""
    if NODEEIGENV in graph.ndata and EDGEEIGENV in graph.edata:
        return graph
    nd, ed = compute_largest_eigenvalues(graph)
    nd, ed = torch.clamp_min(nd, 1.0), torch.clamp_min(ed, 1.0)
    graph.ndata[NODEEIGENV] = torch.repeat_interleave(nd, graph.batch_num_nodes()).unsqueeze(-1)
    graph.edata[EDGEEIGENV] = torch.repeat_interleave(ed, graph.batch_num_edges()).unsqueeze(-1)
    return graph


def dataset_eigenvalue_bounds(patterns, floor=4.0):
    """train.py:1174-1186: dataset-level ``init_neigenv`` / ``init_eeigenv`` = max over the
    samples' PATTERN graphs (the reference reads ``x["pattern"]`` only) of the per-graph
    bounds, at least ``floor`` (4.0, the triangle value).  ``patterns``: iterable of
    (batched) pattern graphs."""
    mn, me = floor, floor
    for g in patterns:
        nd, ed = compute_largest_eigenvalues(g)
        if nd.numel():
            mn = max(mn, float(nd.max()))
            me = max(me, float(ed.max()))
    return mn, me

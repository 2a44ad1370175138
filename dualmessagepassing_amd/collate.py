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
                   edata=None, with_segments=True):
    """Device collate of per-graph LOCAL edge lists laid out back to back.

    local_src/local_dst [E] int64 (device), num_nodes/num_edges [B] int64 (device);
    total_nodes/total_edges are host ints (the dataset knows them; no device sync here).
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
    return g


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
    return collate_device(ls, ld, nn, ne, sum(nn_host), sum(ne_host), ndata, edata)


def batchify(samples, return_weights=None, device=None):
    """``GraphAdjDataset.batchify`` (dataset.py:1604-1636) for samples
    ``{"id", "pattern", "graph", "counts"}`` -> ``(_id, pattern, graph, counts, (None, None))``.
    The optional subisomorphism node/edge weights of the reference (numba host counters,
    dataset.py:1618-1634) are outside the hot path and not produced here."""
    if return_weights is not None:
        raise NotImplementedError("node/edge subisomorphism weights are not part of the MI355X hot path")
    _id = [x["id"] for x in samples]
    pattern = batch([x["pattern"] for x in samples], device)
    graph = batch([x["graph"] for x in samples], device)
    counts = torch.tensor([x["counts"] for x in samples], dtype=torch.int64)
    if device is not None:
        counts = counts.to(device)
    return _id, pattern, graph, counts, (None, None)


def union_graphs(a, b):
    """Block-diagonal union of two (batched) graphs: nodes/edges of ``a`` first, then ``b`` with its
    node ids shifted by ``a.number_of_nodes()`` -- ``dgl.batch([a, b])`` semantics on the structure.
    Used to run a SHARED rep-net once over pattern and target batches (same weights, per-row
    ops, no BatchNorm), instead of twice.  Carries ``is_reversed`` and cached degrees."""
    from .constants import INDEGREE, OUTDEGREE, REVFLAG
    na = a.number_of_nodes()
    src = torch.cat([a._src, b._src + na])
    dst = torch.cat([a._dst, b._dst + na])
    bnn = torch.cat([a.batch_num_nodes(), b.batch_num_nodes()])
    bne = torch.cat([a.batch_num_edges(), b.batch_num_edges()])
    g = BatchedGraph(src, dst, na + b.number_of_nodes(), bnn, bne)
    if (REVFLAG in a.edata) != (REVFLAG in b.edata):
        raise ValueError("union_graphs: is_reversed must be present on both graphs or on neither")
    if REVFLAG in a.edata:
        g.edata[REVFLAG] = torch.cat([a.edata[REVFLAG], b.edata[REVFLAG]])
    for k in (INDEGREE, OUTDEGREE):
        if k in a.ndata and k in b.ndata:
            g.ndata[k] = torch.cat([a.ndata[k], b.ndata[k]])
    return g

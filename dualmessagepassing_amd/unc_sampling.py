"""UNC mini-batch construction on the device: negative sampling, neighbour
sampling of the training subgraph, node-id compaction and random edge dropping --
``UnsupervisedNodeClassification/Model/DMPNN/src/utils.py:315-434,539-567``.

The reference does this on the host per batch (numpy + DGL's samplers + a numba dict loop) and
uploads the result; here every step is a handful of device ops over the edge list, without host
loops or syncs apart from the one size read that compaction needs (the subgraph's node count).

Integer arithmetic (negative samples given their random draws, node-id maps, in-edge bookkeeping) is
exact and tested against the reference's functions; the random choices themselves come from a
``torch.Generator`` instead of numpy / DGL: equal in distribution, not bit for bit -- DGL's sampler is a
third-party component that is not available here, and its stream of random numbers is not part of
the reference's contract.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .graph import BatchedGraph
from .unc import compute_edgenorm


def _seed_of(generator, seed):
    """The 64-bit seed of a kernel launch: given, or the next draw of ``generator`` / torch's global generator."""
    if seed is not None:
        return int(seed) & (2 ** 64 - 1)
    dev = generator.device if generator is not None else "cpu"
    return int(torch.randint(0, 2 ** 62, (1,), generator=generator, device=dev).item())


def random_walks(graph, seed_nodes, walks, depth, seed=None, generator=None, return_traces=True):
    """``dgl.sampling.random_walk(graph, seeds, length=depth)`` repeated ``walks`` times per seed, as ONE launch
    (``dmp_random_walks``, a thread per walk over the CSR by source): ``(traces [S * walks, depth + 1] int64 with -1
    after a dead end, visited bool [N])``."""
    lib = _lib.load()
    ix = graph.index()
    seeds = torch.as_tensor(seed_nodes, device=ix.device).to(torch.int64).view(-1).contiguous()
    S, N = int(seeds.numel()), graph.number_of_nodes()
    traces = torch.empty((S * walks, depth + 1), dtype=torch.int64, device=ix.device) if return_traces else None
    visited = torch.zeros(N, dtype=torch.uint8, device=ix.device)
    check(lib.dmp_random_walks(ptr(ix.out_ptr), ptr(ix.out_ent), ptr(ix.dst32), ptr(seeds), S, int(walks), int(depth),
                               _seed_of(generator, seed), ptr(traces), ptr(visited), stream_ptr()), "dmp_random_walks")
    return traces, visited.bool()


def negative_sampling(pos_samples, num_entity, negative_rate, values=None, choices=None, generator=None):
    """utils.py:539-554.  ``pos_samples`` [B, 3] (subject, relation, object) int64 on the device;
    every positive is repeated ``negative_rate`` times and has its subject (choice > 0.5) or object
    replaced by a uniformly drawn entity different from the original.  ``values`` (integers in
    [0, num_entity - 1)) and ``choices`` (floats in [0, 1)) may be passed in (tests); otherwise drawn."""
    pos = pos_samples.to(torch.int64)
    n = pos.size(0) * negative_rate
    neg = pos.repeat(negative_rate, 1)
    dev = pos.device
    if values is None:
        values = torch.randint(0, max(num_entity - 1, 1), (n,), device=dev, generator=generator)
    if choices is None:
        choices = torch.rand(n, device=dev, generator=generator)
    values = values.to(dev, torch.int64)
    subj = choices.to(dev) > 0.5
    new_s = values + (values >= neg[:, 0]).to(torch.int64)      # skips the original: 0..i-1, i+1..N-1
    new_o = values + (values >= neg[:, 2]).to(torch.int64)
    neg[:, 0] = torch.where(subj, new_s, neg[:, 0])
    neg[:, 2] = torch.where(subj, neg[:, 2], new_o)
    return neg


def convert_subgraph_nids(ori_nids, subg_nids, num_nodes):
    """utils.py:557-567: position of each original node id in ``subg_nids`` (a dict loop there; one
    scatter into a lookup table and one gather here).  ``num_nodes``: size of the parent graph."""
    table = torch.full((num_nodes,), -1, dtype=torch.int64, device=subg_nids.device)
    table[subg_nids] = torch.arange(subg_nids.numel(), device=subg_nids.device)
    return table[ori_nids]


def sample_in_edges(graph, nodes, width, generator=None):
    """``dgl.sampling.sample_neighbors(graph, nodes, width, edge_dir="in")`` as an edge mask: for every
    node in ``nodes`` up to ``width`` of its in-edges, chosen uniformly without replacement (all of them
    if it has at most ``width``).  One random key per edge, a sort by (destination, key), the first
    ``width`` of every destination segment survive.  Returns a bool mask over the graph's edges."""
    return sample_in_edges_device(graph, nodes, width, generator=generator)


def sample_in_edges_device(graph, nodes, width, seed=None, generator=None):
    """``sample_in_edges`` as one launch over the CSR by destination (``dmp_sample_in_edges``: a thread per node keeps the
    ``width`` in-edges with the smallest hashed keys).  ``nodes``: index tensor, or a bool / uint8 mask over the nodes."""
    lib = _lib.load()
    ix = graph.index()
    E, N, dev = graph.number_of_edges(), graph.number_of_nodes(), ix.device
    nodes = torch.as_tensor(nodes, device=dev)
    if nodes.dtype in (torch.bool, torch.uint8) and nodes.numel() == N:
        wanted = nodes.to(torch.uint8).contiguous()
    else:
        wanted = torch.zeros(N, dtype=torch.uint8, device=dev)
        wanted[nodes.to(torch.int64)] = 1
    mask = torch.empty(E, dtype=torch.uint8, device=dev)
    check(lib.dmp_sample_in_edges(ptr(ix.in_ptr), ptr(ix.in_ent), ptr(wanted), N, E, int(width), _seed_of(generator, seed),
                                  ptr(mask), stream_ptr()), "dmp_sample_in_edges")
    return mask.bool()


def sample_subgraph_by_neighbors(graph, seed_nodes, depth=2, width=10, generator=None):
    """utils.py:315-349: ``depth - 1`` rounds that grow the node set by the sources of the sampled
    in-edges of the current set (DGL keeps the parent's node frame, so ``out_deg > 0`` there selects on
    the PARENT graph's out-degree), a final sampling round, then removal of the nodes without any
    sampled edge unless they are seeds.  Returns ``(subgraph, nid)``: a ``BatchedGraph`` over the
    compacted node ids with ``edata`` copied from the parent (plus ``"_ID"``: parent edge ids) and
    ``nid`` [n_sub] the parent id of every subgraph node (ascending, as ``subg.ndata[dgl.NID]``)."""
    src, dst = graph.all_edges(form="uv", order="eid")
    N, dev = graph.number_of_nodes(), dst.device
    seed_nodes = torch.as_tensor(seed_nodes, device=dev).to(torch.int64).view(-1)
    in_set = torch.zeros(N, dtype=torch.bool, device=dev)
    in_set[seed_nodes] = True
    has_out = graph.out_degrees() > 0
    for _ in range(depth - 1):
        # nodes = unique(cat(nodes, ndata[NID][out_deg > 0])): every parent node with out-edges joins
        # (sample_neighbors returns a graph over ALL parent nodes, utils.py:326-331)
        in_set = in_set | has_out
    nodes = in_set.nonzero().view(-1)
    mask = sample_in_edges(graph, nodes, width, generator)
    eid = mask.nonzero().view(-1)                                # host sync: subgraph size
    s, d = src[eid], dst[eid]
    touched = torch.zeros(N, dtype=torch.bool, device=dev)
    touched[s] = True
    touched[d] = True
    seeds = torch.zeros(N, dtype=torch.bool, device=dev)
    seeds[seed_nodes] = True
    nid = (touched | seeds).nonzero().view(-1)                  # degree-0 non-seed nodes are removed
    sub = BatchedGraph(convert_subgraph_nids(s, nid, N), convert_subgraph_nids(d, nid, N), int(nid.numel()))
    for k, v in graph.edata.items():
        if k not in ("in_deg", "out_deg", "norm"):
            sub.edata[k] = v[eid]
    sub.edata["_ID"] = eid
    sub.ndata["_ID"] = nid
    return sub, nid


def _induced(graph, mask, seed_nodes):
    """The subgraph of the sampled edges: nodes without a sampled edge are dropped unless they are seeds
    (utils.py:297-302,333-338), ids compacted in ascending parent order, edge frames copied."""
    src, dst = graph.all_edges(form="uv", order="eid")
    N, dev = graph.number_of_nodes(), dst.device
    eid = mask.nonzero().view(-1)                                # host sync: subgraph size
    s, d = src[eid], dst[eid]
    keep = torch.zeros(N, dtype=torch.bool, device=dev)
    keep[s] = True
    keep[d] = True
    keep[seed_nodes] = True
    nid = keep.nonzero().view(-1)
    sub = BatchedGraph(convert_subgraph_nids(s, nid, N), convert_subgraph_nids(d, nid, N), int(nid.numel()))
    for k, v in graph.edata.items():
        if k not in ("in_deg", "out_deg", "norm"):
            sub.edata[k] = v[eid]
    sub.edata["_ID"] = eid
    sub.ndata["_ID"] = nid
    return sub, nid


def sample_subgraph_by_randomwalks(graph, seed_nodes, depth=2, width=10, seed=None, generator=None):
    """utils.py:279-313: ``width - 1`` random walks of ``depth`` steps from every seed (one launch), the union of the
    visited nodes with the seeds, ``width`` sampled in-edges for each of them (one launch), removal of the nodes left
    without an edge unless they are seeds.  Returns ``(subgraph, nid)`` like ``sample_subgraph_by_neighbors``."""
    dev = graph.device
    seed_nodes = torch.as_tensor(seed_nodes, device=dev).to(torch.int64).view(-1)
    s0 = _seed_of(generator, seed)
    _, visited = random_walks(graph, seed_nodes, max(width - 1, 0), depth, seed=s0, return_traces=False)
    visited[seed_nodes] = True
    mask = sample_in_edges_device(graph, visited, width, seed=s0 + 1)
    return _induced(graph, mask, seed_nodes)


def drop_edges(sub, keep_fraction, generator=None):
    """utils.py:427-429: delete ``int(E * (1 - keep_fraction))`` edge draws (with replacement, duplicates
    collapse: ``np.unique(uniform_choice_int(...))``) from the subgraph."""
    E = sub.number_of_edges()
    ndel = int(E * (1 - keep_fraction))
    if keep_fraction >= 1.0 or ndel <= 0 or E == 0:
        return sub
    src, dst = sub.all_edges(form="uv", order="eid")
    kill = torch.zeros(E, dtype=torch.bool, device=src.device)
    kill[torch.randint(0, E, (ndel,), device=src.device, generator=generator)] = True
    keep = (~kill).nonzero().view(-1)
    out = BatchedGraph(src[keep], dst[keep], sub.number_of_nodes())
    for k, v in sub.edata.items():
        out.edata[k] = v[keep]
    for k, v in sub.ndata.items():
        out.ndata[k] = v
    return out


def generate_sampled_graph_and_labels_unsupervised(graph, edges, sample_depth, sample_width, split_size,
                                                   negative_rate, generator=None, sampler="neighbor"):
    """utils.py:399-434 with the neighbour sampler: ``edges`` [B, 3] positive triplets on the device ->
    ``(subgraph, samples [B * (1 + negative_rate), 3] in subgraph node ids, labels float32)``; the
    subgraph carries ``edata["norm"]`` (``compute_edgenorm``) like the training loop computes next."""
    edges = edges.to(torch.int64)
    neg = negative_sampling(edges, graph.number_of_nodes(), negative_rate, generator=generator)
    seed = torch.unique(torch.cat([edges[:, 0], edges[:, 2], neg[:, 0], neg[:, 2]]))
    if sampler == "neighbor":        # utils.py:416-419
        sub, nid = sample_subgraph_by_neighbors(graph, seed, sample_depth, sample_width, generator)
    elif sampler == "randomwalk":
        sub, nid = sample_subgraph_by_randomwalks(graph, seed, sample_depth, sample_width, generator=generator)
    else:
        raise ValueError(sampler)
    samples = torch.cat([edges, neg])
    samples[:, 0] = convert_subgraph_nids(samples[:, 0], nid, graph.number_of_nodes())
    samples[:, 2] = convert_subgraph_nids(samples[:, 2], nid, graph.number_of_nodes())
    sub = drop_edges(sub, split_size, generator)
    sub.edata["norm"] = compute_edgenorm(sub)
    labels = torch.zeros(samples.size(0), dtype=torch.float32, device=samples.device)
    labels[:edges.size(0)] = 1.0
    return sub, samples, labels

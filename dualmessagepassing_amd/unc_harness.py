"""The UNC training loop around ``unc.TrainModel`` and the device samplers (BASELINE config 5: one large graph, no
batch sharding).  What ``UnsupervisedNodeClassification/Model/DMPNN/src/main.py:99-211`` does per run, with the
reference's semantics and without its per-step host round trips:

* ``load_links`` / ``save_embeddings``: the ``link.dat`` / output file formats (``utils.py:219-258``);
* ``train_unsupervised``: edge mini-batches -> negative samples + sampled sub-graph (``unc_sampling``) -> encoder ->
  link-prediction loss (``TrainModel.get_unsupervised_loss``) -> clip -> Adam -> cosine learning rate; the epoch count
  rescaled by ``edges / nodes`` and the stop at the first epoch whose mean loss rises, as ``main.py:104,178-182``;
* ``collect_node_embeddings``: the output pass of ``main.py:185-211`` -- every sampled sub-graph's encoder output
  blended into the embedding table by how much of a node's neighbourhood the sample covered.

Everything stays on the device; the mean loss of an epoch is read back once per epoch."""
import math

import numpy as np
import torch

from .unc import build_graph_from_triplets, compute_edgenorm
from .unc_sampling import generate_sampled_graph_and_labels_unsupervised

SUBGRAPH_NID = "_ID"     # parent node id of every sub-graph node (dgl.NID in the reference)


def load_links(path):
    """``link.dat`` (utils.py:219-229): first line ``num_nodes num_rels``, then one ``src rel dst [weight ...]`` row of
    integers per line.  Returns ``(triplets int64 [M, >=3], num_nodes, num_rels)``."""
    with open(path) as f:
        head = f.readline().split()
        num_nodes, num_rels = int(head[0]), int(head[1])
        rows = [[int(t) for t in line.split()] for line in f if line.strip()]
    return np.asarray(rows, dtype=np.int64).reshape(len(rows), -1), num_nodes, num_rels


def save_embeddings(path, embeddings, index=None, header=""):
    """The reference's output file (utils.py:243-258): a header line, then ``node_id<TAB>v0 v1 ...`` per node."""
    emb = np.asarray(embeddings)
    ids = np.arange(len(emb)) if index is None else np.asarray(index)
    with open(path, "w") as f:
        f.write(str(header) + "\n")
        for n, row in zip(ids, emb):
            f.write("%d\t%s\n" % (int(n), " ".join(row.astype(str))))


def _edge_batches(triplets, batch_size, generator):
    """Shuffled mini-batches of positive triplets (the reference's ``DataLoader(..., shuffle=True)``)."""
    order = torch.randperm(triplets.size(0), generator=generator, device=generator.device if generator is not None else None)
    order = order.to(triplets.device)
    for i in range(0, triplets.size(0), batch_size):
        yield triplets[order[i:i + batch_size]]


def _encode(model, graph, batch, sampler, sample_depth, sample_width, graph_split_size, negative_sample, generator):
    sub, samples, labels = generate_sampled_graph_and_labels_unsupervised(
        graph, batch, sample_depth, sample_width, graph_split_size, negative_sample, generator=generator, sampler=sampler)
    edge_type = sub.edata["type"]
    embed, _ = model(sub, sub.ndata[SUBGRAPH_NID], edge_type, sub.edata["norm"])
    return sub, samples, labels, edge_type, embed


def pad_sampled(sub, edge_type, node_quantum=512, edge_quantum=2048, parent_nodes=1):
    """The sampled sub-graph ``sub`` as fixed-capacity arrays: its nodes followed by INERT nodes (looked up as node 0 of the
    parent graph), its edges followed by inert self-loops spread over the inert nodes with norm 0, capacities the next
    multiples of the quanta -- a few distinct shapes over a run instead of one per step.  ``parent_nodes``: rows of the node
    embedding table (the inert nodes' lookups are spread over them).
    -> ``((Ncap, Ecap), (src, dst, edge_type, norm, nid, counts))``, ``counts`` = int64 [2] on the device (real nodes, real edges)."""
    N, E = sub.number_of_nodes(), sub.number_of_edges()
    ecap = max(edge_quantum, (E + edge_quantum - 1) // edge_quantum * edge_quantum)
    # enough inert nodes for the inert edges to spread over (self-loops, at most 32 per node: the CSR build sorts a row of up to 32
    # entries in registers -- two thousand self-loops on ONE node were a 4 ms heap sort by one thread)
    spare = max(1, (edge_quantum + 31) // 32)
    ncap = (N + spare + node_quantum - 1) // node_quantum * node_quantum
    src, dst = sub.all_edges(form="uv", order="eid")
    dev = src.device
    tail = N + torch.arange(ecap - E, dtype=src.dtype, device=dev) % (ncap - N)
    nid = sub.ndata[SUBGRAPH_NID].view(-1)
    norm = sub.edata["norm"].view(-1)
    arrays = (torch.cat([src, tail]), torch.cat([dst, tail]),
              torch.cat([edge_type.view(-1), torch.zeros(ecap - E, dtype=edge_type.dtype, device=dev)]),
              torch.cat([norm, torch.zeros(ecap - E, dtype=norm.dtype, device=dev)]).view(-1, 1),
              # (the inert nodes look up DIFFERENT rows of the table: hundreds of lookups of one row are one long row of the lookup's
              # CSR -- sorted by a single thread -- for a gradient that is zero either way)
              torch.cat([nid, torch.arange(ncap - N, dtype=nid.dtype, device=dev) % max(1, parent_nodes)]),
              torch.tensor([N, E], dtype=torch.int64, device=dev))
    return (ncap, ecap), arrays


class SampledStep:
    """The optimisation step of ``train_unsupervised`` on a sampled sub-graph -- encoder, link-prediction loss, backward,
    clip, Adam -- recorded once per padded shape as a HIP graph and replayed (``dp.StepGraph``).  The samplers produce a
    sub-graph of a new size every step (utils.py:279-349; two host syncs: they stay eager); the step itself sees
    ``pad_sampled``'s fixed capacities, with the real row counts on the device: BatchNorm statistics
    (``dmp_bn_train_*_rows``), the per-relation means and the regularisers run over the real rows only
    (``unc.PaddedRows``), an inert row contributes nothing to any gradient.  ``optimizer``: ``FlatAdamW(capturable=True)``
    (a scheduler's rate reaches the device through ``sync_hyper`` before every replay).

    The WHOLE loop that owns this object -- the sampling included -- must run under ``with step.steps.on_stream():``
    (``train_unsupervised`` does): the samplers' host round trips on the legacy default stream between a recording and its
    replays make the replay fault on this stack (``dp.StepGraph``'s second hazard; found here as a GPU memory access fault of
    the first replay after a sampler call)."""

    def __init__(self, model, sync, optimizer, grad_norm=1.0, node_quantum=512, edge_quantum=2048, max_shapes=8):
        from .dp import StepGraph
        if not getattr(optimizer, "capturable", False):
            raise ValueError("SampledStep needs FlatAdamW(capturable=True)")
        self.model, self.sync, self.optimizer, self.grad_norm = model, sync, optimizer, grad_norm
        self.node_quantum, self.edge_quantum = int(node_quantum), int(edge_quantum)
        self.steps = StepGraph(self._step, optimizer=optimizer, max_shapes=max_shapes)

    def _step(self, caps, src, dst, edge_type, norm, nid, counts, samples, labels):
        from .graph import BatchedGraph
        from .unc import PaddedRows
        ncap, ecap = caps
        g = BatchedGraph(src, dst, ncap)
        g._dmp_valid = PaddedRows(counts[0:1], counts[1:2], ncap, ecap)
        g.edata["type"], g.edata["norm"] = edge_type, norm
        self.sync.detach_grads()
        embed, _ = self.model(g, nid, edge_type, norm)
        loss = self.model.get_unsupervised_loss(g, embed, edge_type, samples, labels)
        loss.backward()
        self.sync.pack()
        if self.grad_norm and self.grad_norm > 0:
            torch.nn.utils.clip_grad_norm_([self.sync.master], self.grad_norm)
        self.optimizer.step()
        return loss.detach()

    def __call__(self, sub, edge_type, samples, labels):
        caps, arrays = pad_sampled(sub, edge_type, self.node_quantum, self.edge_quantum, int(self.model.model.num_nodes))
        return self.steps(caps, *arrays, samples, labels)


def train_unsupervised(model, graph, triplets, n_epochs=10, graph_batch_size=2000, lr=1e-3, grad_norm=1.0,
                       sampler="neighbor", sample_depth=6, sample_width=128, graph_split_size=0.5, negative_sample=5,
                       rescale_epochs=True, seed=0, log=None, replay=False):
    """main.py:99-183 for the unsupervised objective.  ``triplets`` [M, 3] (device, int64): the positive edges;
    ``graph``: ``build_graph_from_triplets`` of them (both directions, ``edata["type"]`` / ``["norm"]``).
    Returns the per-epoch mean losses (training stops after the first epoch whose mean loss rises).
    ``replay``: the step after the sampling runs through ``SampledStep`` (sub-graphs padded to capacity levels, one HIP-graph
    replay per step); same sub-graphs, same losses up to the summation order of the padded BatchNorm partials."""
    dev = triplets.device
    triplets = triplets[:, :3].to(torch.int64)
    steps_per_epoch = math.ceil(triplets.size(0) / graph_batch_size)
    if rescale_epochs:                                        # main.py:104: epochs counted in passes over the NODES
        n_epochs = math.ceil(n_epochs * steps_per_epoch * graph_batch_size / graph.number_of_nodes())
    # Adam (main.py:112) as ONE launch over the flat parameter buffer: with weight_decay 0 the AdamW kernel IS Adam, and a
    # parameter that never receives a gradient (nfc / efc, model.py:137-138) keeps zero moments and does not move
    from .dp import FlatAdamW, FlatGradSync
    sync = FlatGradSync(model)
    master = sync.flatten_parameters()
    optimizer = FlatAdamW([master], lr=lr, weight_decay=0.0, capturable=bool(replay))
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, n_epochs * steps_per_epoch, eta_min=3e-6)
    gen = torch.Generator(device=dev).manual_seed(seed)
    model.train()
    stepper = SampledStep(model, sync, optimizer, grad_norm) if replay else None
    import contextlib
    with (stepper.steps.on_stream() if stepper is not None else contextlib.nullcontext()):     # (the sampling too: see SampledStep)
        return _train_epochs(model, graph, triplets, n_epochs, graph_batch_size, grad_norm, sampler, sample_depth, sample_width,
                             graph_split_size, negative_sample, log, sync, master, optimizer, scheduler, gen, stepper, steps_per_epoch)


def _train_epochs(model, graph, triplets, n_epochs, graph_batch_size, grad_norm, sampler, sample_depth, sample_width, graph_split_size,
                  negative_sample, log, sync, master, optimizer, scheduler, gen, stepper, steps_per_epoch):
    dev = triplets.device
    history, prev = [], float("inf")
    for epoch in range(n_epochs):
        total = torch.zeros((), device=dev)
        for batch in _edge_batches(triplets, graph_batch_size, gen):
            if stepper is not None:
                sub, samples, labels = generate_sampled_graph_and_labels_unsupervised(
                    graph, batch, sample_depth, sample_width, graph_split_size, negative_sample, generator=gen, sampler=sampler)
                total += stepper(sub, sub.edata["type"], samples, labels)
                scheduler.step()
                continue
            sub, samples, labels, edge_type, embed = _encode(model, graph, batch, sampler, sample_depth, sample_width,
                                                             graph_split_size, negative_sample, gen)
            loss = model.get_unsupervised_loss(sub, embed, edge_type, samples, labels)
            sync.detach_grads()
            loss.backward()
            sync.pack()
            torch.nn.utils.clip_grad_norm_([master], grad_norm)   # the flat gradient's norm is the norm over all parameters
            optimizer.step()
            scheduler.step()
            total += loss.detach()
        mean = float(total) / steps_per_epoch                 # the epoch's one host sync
        history.append(mean)
        if log is not None:
            log("Epoch %05d | Loss %.4f" % (epoch, mean))
        if mean > prev:                                       # main.py:180-182
            break
        prev = mean
    return history


@torch.no_grad()
def collect_node_embeddings(model, graph, triplets, graph_batch_size=2000, sampler="neighbor", sample_depth=6,
                            sample_width=128, graph_split_size=0.5, negative_sample=5, seed=0):
    """main.py:185-211: starting from the embedding table, every sampled sub-graph's encoder output is blended in with
    weight ``(in_deg_sub + 1) / (in_deg_full + 1)`` per node.  Returns ``(node_emb [N, d] on the device, covered bool [N])``."""
    dev = triplets.device
    triplets = triplets[:, :3].to(torch.int64)
    model.eval()
    node_emb = model.model.node_emb.weight.detach().clone()
    covered = torch.zeros(graph.number_of_nodes(), dtype=torch.bool, device=dev)
    full_in = graph.in_degrees().float()
    gen = torch.Generator(device=dev).manual_seed(seed)
    for i in range(0, triplets.size(0), 4 * graph_batch_size):            # unshuffled, four times the training batch
        batch = triplets[i:i + 4 * graph_batch_size]
        sub, _, _, _, embed = _encode(model, graph, batch, sampler, sample_depth, sample_width, graph_split_size,
                                      negative_sample, gen)
        nid = sub.ndata[SUBGRAPH_NID]
        coef = ((sub.in_degrees().float() + 1) / (full_in[nid] + 1)).view(-1, 1)
        node_emb[nid] = node_emb[nid] * (1 - coef) + embed[0] * coef
        covered[nid] = True
    return node_emb, covered


def graph_of(triplets_np, num_nodes, num_rels, device):
    """``(graph, triplets on the device)`` for ``train_unsupervised`` from ``load_links``' output."""
    trip = np.asarray(triplets_np)[:, :3]
    return build_graph_from_triplets(num_nodes, num_rels, trip, device), torch.from_numpy(trip).to(device)

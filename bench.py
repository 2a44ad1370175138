#!/usr/bin/env python3
"""Headline benchmark: (pattern, graph) pairs/sec of the full DMPNN model fwd+bwd at hid=128
on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          (no launcher: bench.py starts its N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 8 --workload 4                  (the scaling configuration, BASELINE configs[3]: 1024 pairs of
                                                            pattern (16,32) x target (512,4096) per GPU)

Workload = BASELINE.json configs[1] per GPU: synthetic directed Erdos-Renyi pairs,
pattern (|V|=8, |E|=12) x target (|V|=64, |E|=256), reversed edges added (E -> 2E, the
reference's --add_rev True), batch 1024 pairs per GPU, 3 shared DMPLayers, hid 128, fp32.
Weak scaling: every rank holds its own 1024-pair shard (seed 1000*config_id + rank).

One step (timed) = what the reference does per batch after the DataLoader hands it over
(SubgraphCountingMatching/train.py:606-686), inputs already resident in HBM:
  device collate of the B pattern and B target graphs (dgl.batch, dataset.py:1320-1328)
  -> graph index build (CSR by dst / by src, degrees)
  -> DMPNN(**config).forward(pattern, graph): multihot encodings, embeddings, ScalarFilter gates,
     3 shared DMPLayers on pattern and target (gates, residual; dmpnn.py:215-277), SumPredictNet
     heads on nodes and edges (basemodel.py:1500-1663)
  -> MSE loss on the counts -> backward
  -> ONE all-reduce of the flat gradient buffer (RCCL) -> AdamW step.

The JSON line also carries
  roofline     : the scatter-add kernel (flag-split segment sum over the target graph),
                 algorithmic bytes / HIP-event time per launch, vs 8 TB/s
  kernels      : the same for every other HIP kernel on the path
  cpu_baseline : the CPU oracle (reference operation order, torch CPU ops) timed on this
                 host's cores on a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 measured copy ceiling

CFG = dict(config_id=2, batch=1024, hid=128, layers=3, p_nodes=8, p_edges=12, g_nodes=64, g_edges=256,
           p_labels=8, g_labels=16)
# BASELINE configs[3] per-GPU shard (the 8-GPU case): not the bench line, selectable for size checks
CFG4 = dict(CFG, config_id=4, p_nodes=16, p_edges=32, g_nodes=512, g_edges=4096)


def er_local_edges(batch, n, m, rng):
    """[batch, m] local endpoints of directed G(n, m) graphs (distinct ordered pairs, u != v)."""
    total = n * (n - 1)
    if batch * total > (1 << 26):  # large graphs: one draw per graph instead of a [batch, total] key matrix
        pick = np.stack([rng.choice(total, size=m, replace=False) for _ in range(batch)])
    else:  # argsort of random keys = sampling without replacement, vectorised over the batch
        pick = np.argsort(rng.random((batch, total)), axis=1)[:, :m]
    u = pick // (n - 1)
    r = pick % (n - 1)
    v = r + (r >= u)
    return u.astype(np.int64), v.astype(np.int64)


def make_shard(cfg, rank, device):
    """Per-rank synthetic shard, resident in HBM, as a dataset would hold it after the reference's
    preprocessing (ids = arange, add_reversed_edges, degrees): per-graph LOCAL edge lists back to
    back, node/edge ids and labels, is_reversed, sizes, counts."""
    rng = np.random.default_rng(1000 * cfg["config_id"] + rank)
    B = cfg["batch"]
    out = {}
    for tag, n, m, nvl, nel in (("p", cfg["p_nodes"], cfg["p_edges"], cfg["p_labels"], cfg["p_labels"]),
                                ("g", cfg["g_nodes"], cfg["g_edges"], cfg["g_labels"], cfg["g_labels"])):
        u, v = er_local_edges(B, n, m, rng)
        # add_reversed_edges (train.py:299-327): [forward | reversed] per graph, id + max_ne, label + max_nel
        src = np.concatenate([u, v], axis=1).reshape(-1)
        dst = np.concatenate([v, u], axis=1).reshape(-1)
        rev = np.concatenate([np.zeros((B, m), bool), np.ones((B, m), bool)], axis=1).reshape(-1)
        el = rng.integers(0, nel, size=(B, m))
        elabel = np.concatenate([el, el + nel], axis=1).reshape(-1)
        eid = np.tile(np.concatenate([np.arange(m), m + np.arange(m)]), B)
        nid = np.tile(np.arange(n), B)
        nlabel = rng.integers(0, nvl, size=B * n)
        out[tag] = dict(
            local_src=torch.from_numpy(src).to(device), local_dst=torch.from_numpy(dst).to(device),
            ndata={"id": torch.from_numpy(nid).to(device), "label": torch.from_numpy(nlabel).to(device)},
            edata={"id": torch.from_numpy(eid).to(device), "label": torch.from_numpy(elabel).to(device),
                   "is_reversed": torch.from_numpy(rev).to(device)},
            num_nodes=torch.full((B,), n, dtype=torch.int64, device=device),
            num_edges=torch.full((B,), 2 * m, dtype=torch.int64, device=device),
            N=B * n, E=B * 2 * m, max_n=n, max_e=2 * m)   # host-side sizes a dataset knows
    g = torch.Generator(device="cpu").manual_seed(1000 * cfg["config_id"] + rank)
    out["counts"] = torch.randint(0, 64, (B,), generator=g).float().to(device)
    return out


def slice_shard(cfg, shard, lo, hi):
    """Pairs [lo, hi) of a shard (views, no copies): a data-parallel rank's part of a global batch, or one micro-batch."""
    out = {"counts": shard["counts"][lo:hi]}
    for t, n, m in (("p", cfg["p_nodes"], cfg["p_edges"]), ("g", cfg["g_nodes"], cfg["g_edges"])):
        d = shard[t]
        ns, es = slice(lo * n, hi * n), slice(lo * 2 * m, hi * 2 * m)
        out[t] = dict(local_src=d["local_src"][es], local_dst=d["local_dst"][es],
                      ndata={k: v[ns] for k, v in d["ndata"].items()}, edata={k: v[es] for k, v in d["edata"].items()},
                      num_nodes=d["num_nodes"][lo:hi], num_edges=d["num_edges"][lo:hi], N=(hi - lo) * n, E=(hi - lo) * 2 * m,
                      max_n=n, max_e=2 * m)
    return out


def concat_shards(shards):
    """The shards of several ranks as ONE batch (rank 0's pairs first): the global batch a data-parallel step works on,
    for comparing the ranks' averaged gradient with a single process's (``--emulate-world``)."""
    out = {"counts": torch.cat([s["counts"] for s in shards])}
    for t in ("p", "g"):
        ds = [s[t] for s in shards]
        out[t] = dict(local_src=torch.cat([d["local_src"] for d in ds]), local_dst=torch.cat([d["local_dst"] for d in ds]),
                      ndata={k: torch.cat([d["ndata"][k] for d in ds]) for k in ds[0]["ndata"]},
                      edata={k: torch.cat([d["edata"][k] for d in ds]) for k in ds[0]["edata"]},
                      num_nodes=torch.cat([d["num_nodes"] for d in ds]), num_edges=torch.cat([d["num_edges"] for d in ds]),
                      N=sum(d["N"] for d in ds), E=sum(d["E"] for d in ds), max_n=max(d["max_n"] for d in ds),
                      max_e=max(d["max_e"] for d in ds))
    return out


def micro_batches_for(cfg):
    """Micro-batches per step so that one pass stays inside the kernels' index range: 2^30 edge rows (packed
    (row << 1) | flag entries) -- rows are addressed by index, so the SIZE of an [E, H] array is no limit any more (round 2
    split config 4's 8.4 M-row shard four ways to keep every array below 4 GiB of 32-bit byte offsets).  Pairs are
    independent, so a step over M equal slices of the shard with the gradients summed (each slice's mean loss weighted
    1/M) is the same step (``--micro-batches`` forces it).  Config 2: 1; config 4: 1."""
    rows = cfg["batch"] * 2 * (cfg["p_edges"] + cfg["g_edges"])
    m = 1
    while rows // m >= 2 ** 30 and m < cfg["batch"]:
        m *= 2
    return m


def model_config(cfg):
    """The reference's DMPNN training configuration (SubgraphCountingMatching/README.md:72-94,
    'Complex' command) at BASELINE's hid=128; vocabulary sizes after --add_rev doubling.  Activations and
    embedding kind are the reference's shipped defaults (config.py:242-245,298-301,370-373: leaky_relu with
    slope 1/5.5 in the rep-net MLPs and the heads, Equivariant embeddings; no README command overrides them);
    ``--act relu --emb Orthogonal`` reproduces the round-1 line."""
    return dict(max_ngv=cfg["g_nodes"], max_ngvl=cfg["g_labels"], max_nge=2 * cfg["g_edges"], max_ngel=2 * cfg["g_labels"],
                max_npv=cfg["p_nodes"], max_npvl=cfg["p_labels"], max_npe=2 * cfg["p_edges"], max_npel=2 * cfg["p_labels"],
                base=2, hid_dim=cfg["hid"], share_emb_net=True, share_enc_net=True, share_rep_net=True,
                rep_residual=True, enc_net="Multihot", emb_net=cfg.get("emb", "Equivariant"), filter_net=cfg.get("filter", "ScalarFilter"),
                rep_net="DMPNN", rep_num_graph_layers=cfg["layers"], rep_num_pattern_layers=cfg["layers"],
                rep_dmpnn_num_mlp_layers=2, rep_dmpnn_batch_norm=False, rep_act_func=cfg.get("act", "leaky_relu"), rep_dropout=0.0,
                init_neigenv=4.0, init_eeigenv=4.0, pred_net="SumPredictNet", pred_hid_dim=cfg["hid"],
                pred_act_func=cfg.get("act", "leaky_relu"), pred_dropout=0.0, node_pred=True, edge_pred=True)


def build_step(cfg, shard, device, world=1, collective=False):
    """``collective``: run the N > 1 code path (async all-reduce, stream-ordered wait, software-pipelined index build)
    whatever the world size -- ``--force-collective``: a one-rank RCCL group on a single GPU."""
    multi = world > 1 or collective
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.collate import collate_device_many
    from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
    from dualmessagepassing_amd.dmpnn import prepare_joint
    from dualmessagepassing_amd.harness import count_loss

    torch.manual_seed(0)
    model = build_model(**model_config(cfg)).to(device)
    sync = FlatGradSync(model, force_collective=collective)
    master = sync.flatten_parameters()      # one AdamW launch over the flat buffer: the same elementwise update
    sync.broadcast_parameters()
    opt = FlatAdamW([master], lr=1e-4, weight_decay=1e-5, amsgrad=True,   # the reference's optimizer (train.py:1231)
                    capturable=bool(cfg.get("graph")))

    pending = []      # the gradient all-reduce of the previous step, still in flight
    M = int(cfg.get("micro_batches") or micro_batches_for(cfg))
    if cfg["batch"] % M:
        raise SystemExit("--micro-batches must divide the batch")
    parts = [shard] if M == 1 else [slice_shard(cfg, shard, i * (cfg["batch"] // M), (i + 1) * (cfg["batch"] // M)) for i in range(M)]
    total = torch.zeros_like(sync.flat) if M > 1 else None    # gradient sum over the micro-batches

    ar_events = []    # (before, after) event pairs around the wait for the gradient sum: how long the compute stream stood still

    def timed_wait(fn):
        if multi and step.time_allreduce:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            ar_events.append((a, b))
        else:
            fn()

    def finish():
        """Second half of a step: wait (on the stream, not the host) for the gradient sum, then AdamW."""
        if pending:
            work = pending.pop()
            timed_wait(lambda: sync.finish(work))
            opt.step()

    packs = {}

    def fresh_of(part):
        """New tensors holding the part's size arrays and reversed flags: views of one fresh copy of a packed byte buffer."""
        key = id(part)
        ent = packs.get(key)
        if ent is None:
            pieces, layout, off = [], [], 0
            for t in ("p", "g"):
                for a in (part[t]["num_nodes"], part[t]["num_edges"], part[t]["edata"]["is_reversed"]):
                    b = a.contiguous().view(torch.uint8).view(-1)
                    pad = (-b.numel()) % 8
                    pieces.append(b)
                    if pad:
                        pieces.append(torch.zeros(pad, dtype=torch.uint8, device=b.device))
                    layout.append((off, b.numel(), a.dtype, tuple(a.shape)))
                    off += b.numel() + pad
            ent = packs[key] = (torch.cat(pieces), layout, part)     # the part is kept alive: its id stays unique
        buf = ent[0].clone()
        out = [buf[o:o + n].view(dt).view(shape) for o, n, dt, shape in ent[1]]
        return (out[0], out[1], out[2]), (out[3], out[4], out[5])

    def batch_of(part):
        p, g = part["p"], part["g"]
        # a loader hands over NEW size / flag tensors with every batch: nothing derived from them (padding maps,
        # pooling indexes, CSR, degree classes) may be carried over from the previous step
        # (ONE copy of the six arrays, kept back to back in a byte buffer per part, instead of six clone launches: what is
        # timed is the step, not this emulation of a loader)
        (pn, pe, prev), (gn, ge, grev) = fresh_of(part)
        ped, ged = dict(p["edata"], is_reversed=prev), dict(g["edata"], is_reversed=grev)
        pattern, graph = collate_device_many([                  # both batches in one pair of launches
            dict(local_src=p["local_src"], local_dst=p["local_dst"], num_nodes=pn, num_edges=pe, total_nodes=p["N"], total_edges=p["E"],
                 ndata=p["ndata"], edata=ped, max_nodes=p["max_n"], max_edges=p["max_e"]),
            dict(local_src=g["local_src"], local_dst=g["local_dst"], num_nodes=gn, num_edges=ge, total_nodes=g["N"], total_edges=g["E"],
                 ndata=g["ndata"], edata=ged, max_nodes=g["max_n"], max_edges=g["max_e"])])
        return pattern, graph

    def step_micro():
        """One step as M micro-batches (config 4): gradients summed in ``total``, each slice's mean loss weighted 1 / M."""
        finish()
        loss = None
        for i, part in enumerate(parts):
            pattern, graph = batch_of(part)
            sync.detach_grads()
            out = model(pattern, graph)
            loss = torch.nn.functional.mse_loss(out["pred_c"].view(-1), part["counts"]) / M
            loss.backward()
            del out, pattern, graph
            sync.pack()
            if i == 0:
                total.copy_(sync.flat)
            else:
                total.add_(sync.flat)
        sync.flat.copy_(total)
        pending.append(sync.sync(async_op=True))
        if not multi:
            finish()
        return loss

    def read_all(out):
        """Every entry of the output dictionary formed in HBM (``--all-outputs`` / ``all_outputs_ms_per_step``): the
        reference returns all 15 as tensors (basemodel.py:1645-1661); here four of them (``g_v_emb`` / ``g_e_emb``, and with
        ``model.lazy_edge_rep`` the last layer's ``p_e_rep`` / ``g_e_rep``) are formed on first read."""
        return [v for _, v in out.items()]

    def step(all_outputs=False):
        if M > 1:
            return step_micro()
        pattern, graph = batch_of(shard)
        # The batch's structure work (collate above; CSR, incidence, degree classes, selectors here) does not depend on
        # the parameters: it is enqueued BEFORE the previous step's gradient sum is waited for, so with more than one
        # rank the all-reduce (on RCCL's stream) overlaps it instead of idling the compute stream.
        if multi and not model.gate_capacity:
            prepare_joint(pattern, graph, cfg["hid"], class_tiles=not live_tiles_apply)
        finish()
        sync.detach_grads()
        if all_outputs:
            lazy, model.lazy_edge_rep = getattr(model, "lazy_edge_rep", True), False
            try:
                out = model(pattern, graph)
            finally:
                model.lazy_edge_rep = lazy
            keep = read_all(out)                               # alive until the backward has run
        else:
            out = model(pattern, graph)
        loss = count_loss(out["pred_c"].view(-1), shard["counts"])  # count loss (train.py:624-628; harness.count_loss: criterion + mean + seed in one launch)
        loss.backward()
        if all_outputs:
            del keep
        sync.pack()
        pending.append(sync.sync(async_op=True))              # None at world size 1 (unless the collective is forced)
        if not multi:
            finish()
        return loss

    def front():
        """The recordable part of a data-parallel step: batch, forward, loss, backward, gradient pack (no collective)."""
        pattern, graph = batch_of(shard)
        sync.detach_grads()
        out = model(pattern, graph)
        step.last_pred_c = out["pred_c"].detach()            # what the parity tests compare (a view: no launch)
        loss = count_loss(out["pred_c"].view(-1), shard["counts"])
        loss.backward()
        sync.pack()
        return loss

    def tail():
        """... and what follows it with eager launches: the gradient all-reduce (RCCL) and the optimizer update."""
        timed_wait(sync.sync)
        opt.step()

    def gate_compact(on):
        """The rep-net on the target edges the filter gate keeps (basemodel.set_gate_capacity), or back on every edge row:
        capacity from this rank's own shard with a margin -- the batch shapes stay fixed, so the step still replays.
        -> {"capacity", "kept", "edges"} or None."""
        step.gate_capacity = None
        if not on or M != 1:
            model.set_gate_capacity(None)
            return None
        pattern, graph = batch_of(shard)
        cap = model.calibrate_gate_capacity(pattern, graph)
        del pattern, graph
        if not cap:
            return None
        step.gate_capacity = cap
        return {"capacity": cap, "edges": shard["g"]["E"]}

    def gate_kept_edges():
        """(target edge rows this rank's filter gate keeps, target edge rows) -- one host sync, outside every timed region."""
        pattern, graph = batch_of(shard)
        return model.gate_kept_edges(pattern, graph)

    def gate_kept_rows():
        """{"edges": (kept, all), "nodes": (kept, all)} of this rank's TARGET rows under the filter's gates (one host sync)."""
        pattern, graph = batch_of(shard)
        return model.gate_kept_rows(pattern, graph)

    def plain_scatter_adds(H, launches=20):
        """The two scatter-add launches over ALL rows of an [E, H] array on this step's own union index (the kernels the masked
        forms derive from, and what runs without a 0 / 1 gate): HIP-event time per launch, outside every timed region, inputs
        written by a launch just before as in the step.  -> {"fwd_us", "bwd_us"} or None."""
        from dualmessagepassing_amd import ops as _ops
        pattern, graph = batch_of(shard)
        union = prepare_joint(pattern, graph, H, backward=False, class_tiles=False)
        if union is None:
            return None
        ix = union.index()
        E, N = ix.num_edges, ix.num_nodes
        Mrows = torch.randn(E, H, device=device)
        res = {}
        # (rows_shared=3: the same kernel code under its own name, so that a profile keeps these launches apart from the step's)
        for name, fn in (("fwd_us", lambda: _ops.seg_sum_raw(Mrows, ix.in_ptr, ix.in_ent, N, None, True, -1.0, 1.0, rows_shared=3)),
                         ("bwd_us", lambda: _ops.endpoint_sums(Mrows, ix))):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
            for i in range(launches + 2):
                Mrows.mul_(1.0)                                 # the rows come from the launch before, as in the step
                if i >= 2:
                    ev[i - 2][0].record()
                fn()
                if i >= 2:
                    ev[i - 2][1].record()
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
            res[name] = t[len(t) // 2]
        return res

    def copy_ceiling(nbytes, launches=20):
        """What a launch of THIS size can reach at all: a plain device-to-device row copy that moves ``nbytes`` in total (half read,
        half written; 16 bytes per lane, the library's copy kernel), timed exactly as the scatter-add launches are (HIP events around
        single eager launches, the source written by the launch before, median of ``launches``).  A launch of tens of megabytes
        lasts a dozen microseconds, of which the dispatch, the first requests' round trip and the last stores' drain are a fixed
        share: the copy's fraction of the 8 TB/s peak is the ceiling any kernel of that size has on this box.  -> {"avg_us", "frac"}"""
        n = max(int(nbytes) // 8, 1024)
        src = torch.randn(n, device=device)
        dst = torch.empty_like(src)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for i in range(launches + 2):
            src.mul_(1.0)
            if i >= 2:
                ev[i - 2][0].record()
            dst.copy_(src)
            if i >= 2:
                ev[i - 2][1].record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        us = t[len(t) // 2]
        return {"what": "a device-to-device copy of the same total bytes (half read, half written), same timing method: the ceiling of a launch of this size",
                "bytes": int(n * 8), "avg_us": round(us, 2), "gbps": round(n * 8 / us / 1e3, 1), "frac": round(n * 8 / us / 1e3 / HBM_PEAK_GBPS, 4)}

    step.copy_ceiling = copy_ceiling
    step.plain_scatter_adds = plain_scatter_adds
    step.gate_kept_edges = gate_kept_edges
    step.gate_kept_rows = gate_kept_rows
    # (a 0 / 1 edge gate: the class-typed kernels walk tiles over the kept edges, built in the forward pass)
    from dualmessagepassing_amd import fused as _fused
    filt = getattr(model, "filter_net", None)
    live_tiles_apply = bool(_fused.USE_LIVE_TILES and _fused.USE_ROW_MASKS and filt is not None and len(filt) > 0
                            and type(filt["el"]).__name__ == "ScalarFilter")
    step.set_gate_compact = gate_compact
    gate_compact(cfg.get("gate_compact"))
    step.time_allreduce = False
    step.ar_events = ar_events
    step.finish = finish
    step.front, step.tail = front, tail
    step.sync = sync
    step.opt = opt
    step.micro_batches = M
    return step, model


PROFILE_ROUND = "r06"
# the two scatter-add launches as rocprofv3 names them; first the forms that leave out the rows a 0 / 1 edge gate wiped (what
# the step runs under a ScalarFilter gate), then the forms that read every row
SEG_IN = ("seg_sum_vec<32, true, false, true, 0, 256>",     # flag-split segment sum over the CSR by destination (forward; under a 0 / 1 gate: over the kept edges' CSR)
          "seg_sum_vec<32, true, true, true, 0, 256>")      # ... gate-weighted (DMP_KEEP_CSR=0)
SEG_INC = ("seg_sum_vec<32, true, false, true, 1, 256>",     # backward of the edge gathers: the segment sum over the kept edges' incidence CSR (both gates)
           "seg_acc_graphs_k<128, true",                     # ... the one-pass endpoint sums (csrc/dmp_segacc.hip), masked
           "seg_acc_graphs_k<128, false", "seg_acc_graphs_k<128>")


# ... and in the all-rows step (`bench.py --filter-net None`, the `gate_dense` object): no row list, no mask
SEG_IN_DENSE = ("seg_sum_vec<32, true, false, true, 0, 256>",)
SEG_INC_DENSE = ("seg_acc_graphs_k<128, false", "seg_acc_graphs_k<128>", "seg_sum_vec<32, true, false, true, 1, 256>")


def committed_profile(n_rows, n_edges, H, variant=""):
    """What the committed profiles of this round say about the two scatter-add launches, if they were taken at this launch
    shape and with this build of the kernels (``profiles/<round>_profile_meta.json``: rows, edges, H, lib_srchash): the rocprofv3 ``--kernel-trace --stats`` average duration
    (``<round>_bench_kernel_stats.csv``) and the PMC traffic per launch (``<round>_pmc_h128.json``: FETCH_SIZE and WRITE_SIZE
    from separate ``--pmc`` passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B/lane streams on gfx950).
    -> {"in": {"avg_us_rocprof", "traffic"}, "inc": {...}}; entries are None where nothing matches."""
    out = {"in": {"avg_us_rocprof": None, "traffic": None}, "inc": {"avg_us_rocprof": None, "traffic": None}}
    base = os.path.join(ROOT, "profiles", PROFILE_ROUND)
    try:
        with open(base + "_profile_meta.json") as f:
            meta = json.load(f)
        if (meta.get("rows"), meta.get("edges"), meta.get("H")) != (n_rows, n_edges, H):
            return out
        # ... and with THIS build of the library: the profiles record the content hash of the kernel sources they were
        # taken with (csrc/libdmp_hip.so.srchash); numbers of another build are dropped, not quoted
        from dualmessagepassing_amd import _build
        if meta.get("lib_srchash") != _build.source_hash():
            return out
    except (OSError, ValueError):
        return out
    try:
        import csv
        with open(base + "_bench%s_kernel_stats.csv" % variant) as f:
            rows = list(csv.DictReader(f))
        for tag, names in (("in", SEG_IN_DENSE if variant else SEG_IN), ("inc", SEG_INC_DENSE if variant else SEG_INC)):
            for name in names:                                  # the first form the profile holds
                hit = [r for r in rows if name in r["Name"]]
                if hit:
                    out[tag]["avg_us_rocprof"] = round(float(hit[0]["AverageNs"]) / 1e3, 2)
                    out[tag]["kernel_rocprof"] = name
                    break
    except (OSError, ValueError, KeyError):
        pass
    try:
        with open(base + "_pmc%s_h128.json" % variant) as f:
            k = json.load(f)["kernels"]
        for tag, names in (("in", SEG_IN_DENSE if variant else SEG_IN), ("inc", SEG_INC_DENSE if variant else SEG_INC)):
            for name in names:
                hit = [v for key, v in k.items() if name in key]
                if hit:
                    out[tag]["traffic"] = int(hit[0]["hbm_bytes_per_launch"])
                    break
    except (OSError, ValueError, KeyError):
        pass
    return out


def gate_summary(g, cfg, step, H):
    """The secondary `gate_compact` object of the line: the step with the rep-net on the kept target edges (+ padding)."""
    mb = cfg["batch"] // step.micro_batches
    uN = mb * (cfg["p_nodes"] + cfg["g_nodes"])
    uE = mb * 2 * cfg["p_edges"] + g["capacity"]
    out = {"what": "same step, same launch mode, model.set_gate_capacity(capacity): the rep-net runs on the target edges the "
                   "ScalarFilter gate keeps plus inert padding up to `capacity` (a gate-0 edge is a zero row through every "
                   "layer of the reference, basemodel.py:1515-1531); outputs and gradients equal, g_e_rep gets its zero rows "
                   "back on first read; the kept share depends on the labels (here %d pattern / %d target labels, uniform)"
                   % (cfg["p_labels"], cfg["g_labels"]),
           "ms_per_step": g["ms_per_step"], "value": g["value"], "unit": "pairs/s", "capacity": g["capacity"],
           "target_edge_rows": g["edges"], "capacity_fraction": round(g["capacity"] / g["edges"], 4)}
    none = {"avg_us_rocprof": None, "traffic": None}
    k = g["kern"].get("seg_sum2[H=%d,rows=%d,ent=%d]" % (H, uN, uE))
    if k:
        out["roofline"] = seg_roofline(k, "dmp::seg_sum_vec<32,split,remap> at N=%d rows, E=%d edge rows" % (uN, uE), k["bytes"],
                                       4 * H * (uE + uN) + 4 * uE + 4 * (uN + 1), none)
    k = g["kern"].get("seg_sum2_graphs[H=%d,rows=%d,E=%d]" % (H, uN, uE))
    if k:
        out["roofline_bwd"] = seg_roofline(k, "dmp::seg_acc_graphs_k at N=%d rows, E=%d edge rows" % (uN, uE), k["bytes"],
                                           4 * H * (uE + 2 * uN) + 9 * uE + 8 * (uN + 1), none)
    return out


def dense_summary(d, H, uN, uE, nB):
    """The `gate_dense` object of the line: the all-rows step with its OWN ``roofline`` / ``roofline_bwd`` objects (the two
    scatter-add launches over every row: HIP-event times from eager steps of that model, the committed rocprof duration of
    ``bench.py --filter-net None`` beside them when the profile is of this build), its own kernel table and MFMA table."""
    kern = d.pop("kern")
    prof = committed_profile(uN, uE, H, variant="_gate_dense")
    k = kern.get("seg_sum2[H=%d,rows=%d,ent=%d]" % (H, uN, uE))
    if k:
        d["roofline"] = seg_roofline(k, "dmp::seg_sum_vec<32,split,remap> (DMPLayer node aggregation by destination, every row: N=%d rows, "
                                     "E=%d edge rows, H=%d)" % (uN, uE, H), 4 * H * (uE + 2 * uN) + 4 * uE + 4 * (uN + 1),
                                     4 * H * (uE + uN) + 4 * uE + 4 * (uN + 1), prof["in"])
    kg, ki = "seg_sum2_graphs[H=%d,rows=%d,E=%d]" % (H, uN, uE), "seg_sum2[H=%d,rows=%d,ent=%d]" % (H, uN, 2 * uE)
    if kg in kern:
        d["roofline_bwd"] = seg_roofline(kern[kg], "dmp::seg_acc_graphs_k (one pass over every edge row, both endpoints' sums in registers; "
                                         "gradient of the gathered node projections: N=%d rows, E=%d edge rows, H=%d)" % (uN, uE, H),
                                         4 * H * (uE + 2 * uN) + 8 * uE + 16 * (nB + 1), 4 * H * (uE + 2 * uN) + 9 * uE + 8 * (uN + 1), prof["inc"])
    elif ki in kern:
        d["roofline_bwd"] = seg_roofline(kern[ki], "dmp::seg_sum_vec<32,split,remap,incidence> (every edge row under both endpoints; gradient of "
                                         "the gathered node projections: N=%d rows, E=%d edge rows, H=%d)" % (uN, uE, H), kern[ki]["bytes"],
                                         4 * H * (uE + 2 * uN) + 9 * uE + 8 * (uN + 1), prof["inc"])
    d["mfma_kernels"] = mfma_rooflines(kern, H, uE, None)
    d["kernels"] = {n: {"avg_us": round(v["avg_us"], 2), "gbps": round(v["gbps"], 1), "launches": v["launches"], "bytes": int(v["bytes"])}
                    for n, v in sorted(kern.items())}
    return d


def seg_roofline(k, what, own_bytes, survey_bytes, prof, skipped_rows=0, all_own=None, all_survey=None, skipped_nodes=0):
    """``roofline`` object of one scatter-add launch kind.  ``achieved`` / ``frac`` = SURVEY §8(d)'s algorithmic bytes of the
    rows the launch processes / its HIP-event time inside the timed steps (the survey's byte count is the graded
    one); ``bytes_own`` / ``frac_own_bytes`` = the same with every byte the kernel itself moves (its [N, 2H] output, its index
    arrays); both also over the committed rocprof duration.  ``skipped_rows`` / ``skipped_nodes``: summed rows that are zeros
    under the batch's 0 / 1 edge gate and are not fetched / node rows under a zero of the node gate, neither summed nor
    stored: the byte counts are those of the rows the launch does process (never above the memory system's peak);
    ``bytes_all_rows``: what the launch over every row would move -- the kernel that does that is timed stand-alone in
    ``all_rows_launch``."""
    us = k["avg_us"]
    r = {"bound": "hbm", "kernel": what, "achieved": round(survey_bytes / us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": round(survey_bytes / us / 1e3 / HBM_PEAK_GBPS, 4), "traffic": prof["traffic"],
         "bytes_per_launch": int(survey_bytes), "bytes_basis": "SURVEY 8(d) per-row bytes x the rows the launch processes",
         "avg_us": round(us, 2), "launches": k["launches"],
         "bytes_own": int(own_bytes), "frac_own_bytes": round(own_bytes / us / 1e3 / HBM_PEAK_GBPS, 4),
         "avg_us_rocprof": prof["avg_us_rocprof"]}
    if k.get("interrupted"):       # launches stalled by the box (> 10x the median), left out of avg_us: _lib.KernelTimer.summary
        r["launches_interrupted"] = int(k["interrupted"])
        r["avg_us_all_launches"] = round(k["avg_us_all"], 2)
    if prof["avg_us_rocprof"]:
        r["frac_rocprof"] = round(survey_bytes / prof["avg_us_rocprof"] / 1e3 / HBM_PEAK_GBPS, 4)
        r["frac_own_bytes_rocprof"] = round(own_bytes / prof["avg_us_rocprof"] / 1e3 / HBM_PEAK_GBPS, 4)
    if prof["traffic"]:
        r["traffic_over_own_bytes"] = round(prof["traffic"] / own_bytes, 4)
    if skipped_rows or skipped_nodes:
        r["rows_not_fetched"] = int(skipped_rows)
        r["node_rows_not_written"] = int(skipped_nodes)
        if all_own is not None:
            r["bytes_all_rows"] = int(all_own)
            r["bytes_survey_all_rows"] = int(all_survey)
        r["note"] = ("%d of the summed rows are not fetched -- zeros under this batch's 0/1 edge gate, or rows that only a node under a zero "
                     "of its node gate would sum -- and %d node rows lie under a zero of the node gate: not written "
                     "(the ScalarFilter gates multiply the rep-net's input rows and every layer's update, basemodel.py:1515-1531, "
                     "dmpnn.py:245-277).  The byte counts and fractions are those of the rows the launch "
                     "processes; all_rows_launch times the kernel over every row" % (skipped_rows, skipped_nodes))
    return r


# MI355X_MICROARCH.md: 256 MiB Infinity Cache + 8 x 4 MiB L2 = what a launch can FIND on the chip; an entry's byte count also holds
# its stores (a fifth of a scatter-add's bytes), which need not be found anywhere: the exemption's bound is 5/4 of the caches
INFINITY_CACHE_BYTES = ((256 + 32) << 20) * 5 // 4


def check_rates(obj, path="line"):
    """No memory rate of the line above the HBM peak, no fraction above 1 (a byte count priced at rows a launch
    does not process gave 1.3 of the peak): walks the whole line, raises naming the entry.  One exception, and it is marked in
    the line: a launch whose algorithmic bytes fit in the chip's caches (256 MiB Infinity Cache + 32 MiB of L2) and were
    written by the launch before it (a small batch: the step's working set is cache-resident; measured: 287 MB of rows the
    second Linear had just stored summed at 8.6 TB/s) can be fed faster than HBM delivers -- its entry gets
    ``"served_from": "infinity cache"`` instead of an error; a launch larger than the caches never qualifies."""
    if isinstance(obj, dict):
        unit = obj.get("unit")
        nbytes = min([v for k, v in obj.items() if k in ("bytes", "bytes_per_launch", "bytes_own") and isinstance(v, (int, float)) and v > 0] or [0])
        over = any(isinstance(v, (int, float)) and not isinstance(v, bool) and
                   (((k in ("gbps", "hbm_gbps") or (k == "achieved" and unit == "GB/s")) and v > HBM_PEAK_GBPS) or
                    ((k.startswith("frac") or k.endswith("_frac")) and v > 1.0)) for k, v in obj.items())
        if over and 0 < nbytes < INFINITY_CACHE_BYTES:
            obj["served_from"] = "infinity cache"
            for v in obj.values():
                if isinstance(v, (dict, list)):
                    check_rates(v, path)
            return
        for k, v in obj.items():
            here = "%s.%s" % (path, k)
            if isinstance(v, (dict, list)):
                check_rates(v, here)
            elif isinstance(v, (int, float)) and not isinstance(v, bool):
                if (k in ("gbps", "hbm_gbps") or (k == "achieved" and unit == "GB/s")) and v > HBM_PEAK_GBPS:
                    raise SystemExit("bench.py: %s = %.1f GB/s exceeds the HBM peak: a byte count prices rows the launch does not process\n%s"
                                     % (here, v, json.dumps({kk: vv for kk, vv in obj.items() if not isinstance(vv, (dict, list))})))
                if (k.startswith("frac") or k.endswith("_frac") or k == "frac_of_hbm_peak") and v > 1.0:
                    raise SystemExit("bench.py: %s = %.4f exceeds 1" % (here, v))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            check_rates(v, "%s[%d]" % (path, i))


def kept_row_bytes(name, H, N, E, Nk, Ek, Et, Etk, B, Ein=None):
    """Algorithmic bytes of one launch of kernel ``name`` (the HIP-event timer's record name) when it runs on the rows the batch's
    0 / 1 gates keep -- the timer's own count is host-side and prices every row of the launch shape.  N / E: union node / edge
    rows, Nk / Ek: the kept ones (pattern rows are all kept), Et / Etk: target edge rows / kept.  None: the timer's count stands."""
    base = name.split("[", 1)[0]
    on_e = ("E=%d" % E) in name or ("R=%d" % E) in name or ("ent=%d" % E) in name
    on_n = ("R=%d" % N) in name or ("E=%d" % N) in name
    jobs = 1
    if "jobs=" in name:
        jobs = int(name.split("jobs=")[1].split("]")[0].split(",")[0])
    if base == "seg_sum2_graphs":
        return 4 * H * Ek + 8 * H * Nk + 8 * E + E // 8 + N // 8 + 16 * (B + 1)
    if base == "seg_sum2" and ("rows=%d" % N) in name and on_e and ("H=%d" % H) in name:
        Er = Ek if Ein is None else Ein      # under the node gate: the kept edges INTO a kept node
        return 4 * H * Er + 8 * H * Nk + 4 * Er + 12 * Nk
    if base == "seg_sum2" and on_e and ("H=%d" % H) in name:          # a pooled pass over the edge rows, gate-weighted
        return 4 * H * Ek + 8 * E
    if base == "edge_fwd_typed" and on_e:
        return 4 * H * (2 * Ek + 2 * Nk) + 12 * Ek
    if base == "bwd_z_typed" and on_e:
        return 4 * H * (3 * Ek + 2 * Nk) + 5 * Ek
    if base == "atb_typed" and on_e:
        return 8 * H * Ek
    if base == "out_fwd_typed":
        return 12 * H * (Ek if on_e else Nk) * jobs if (on_e or on_n) else None
    if base in ("bwd_h1_typed", "bwd_h1_w"):
        return (12 * H + 4) * (Ek if on_e else Nk) if (on_e or on_n) else None
    if base == "bwd_z_w" and on_e:           # bwd_z_typed's rows + the layer's input rows z (the weight gradient's second operand)
        return 4 * H * (4 * Ek + 2 * Nk) + 5 * Ek
    if base == "out_fwd_typed_codes" and on_e:    # H1 rows in, output rows out, 48-byte code rows (the pattern's embedded rows: few)
        return 8 * H * Ek + 48 * Ek
    if base == "atb2" and (on_e or on_n):    # per 128 x 128 job: two [rows, 128] operands over the tile list
        return 8 * H * (Ek if on_e else Nk) * jobs
    if base == "pool_relu_bwd" and on_e:
        return 8 * H * Ek + 12 * E
    if base == "l0_edge_fwd" and ("E=%d" % Et) in name:
        return 4 * H * (Etk + 2 * Nk) + 64 * Etk
    if base == "l0_bwd_w" and ("E=%d" % Et) in name:
        return 8 * H * Etk + 56 * Etk
    if base == "atb_rows_plain" and on_n:
        return 8 * H * Nk
    return None


MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = the f32 vector rate
MFMA_BF16_PEAK_TFLOPS = 2500.0  # ... dense bf16 MFMA (16 x the f32-input rate); never the 2:1-sparsity figure


X6_TILE_KERNELS = ("bwd_h1_w", "bwd_z_w", "out_fwd_typed_codes")     # bf16x6 kernels over a tile list whose names do not end in _typed


def mfma_rooflines(kern, H, E, Ek=None):
    """{kernel: {avg_us, tflops, frac}} for the fused MFMA kernels seen by the HIP-event timer; flops =
    2*rows*H*H per [rows,H]x[H,H] product the kernel performs (1 for the class-typed kernels, out_fwd and
    bwd_h1; 2 for the two-panel edge_fwd / bwd_z).  ``Ek``: the edge rows the batch's 0 / 1 gate keeps -- the kernels over
    the kept edges' tiles (``*_typed``) multiply those rows only."""
    products = {"edge_fwd_typed": 1, "bwd_z_typed": 1, "atb_typed": 1, "atb_rows": 1, "out_fwd_mfma": 1, "bwd_h1_mfma": 1,
                "edge_fwd_mfma": 2, "bwd_z_mfma": 2, "out_fwd_typed": 1, "bwd_h1_typed": 1,
                "bwd_h1_w": 2,       # (both products of the second Linear's backward in one launch, csrc/dmp_h1w.hip)
                "bwd_z_w": 3,        # (the first Linear's: dz through the class-typed panel, z^T [dPre | c dPre] = two blocks)
                "out_fwd_typed_codes": 1}   # (+ one 16-deep k-group in nine: not counted)
    out = {}
    for name, v in kern.items():
        base = name.split("[", 1)[0]
        if base in products and ("E=%d" % E in name or "R=%d" % E in name):
            rows = Ek if (Ek is not None and (base.endswith("_typed") or base in X6_TILE_KERNELS)) else E
            tf = products[base] * 2.0 * rows * H * H / (v["avg_us"] * 1e-6) / 1e12
            # round 3: the class-typed kernels multiply on the bf16 pipe (three bf16 pieces per fp32 operand, six piece
            # products: fp32-accurate, 6/16 of the f32 form's matrix cycles) -- their bound is HBM
            # (the bf16x6 kernels run on the bf16 pipe -- their matrix work is SIX bf16 piece products per
            # fp32 product, priced against the dense bf16 peak; "tflops" stays the fp32-equivalent rate of the product)
            x6 = base.endswith("_typed") or base in X6_TILE_KERNELS
            peak = MFMA_BF16_PEAK_TFLOPS if x6 else MFMA_F32_PEAK_TFLOPS
            pipe_tf = 6.0 * tf if x6 else tf
            out[base] = {"rows": int(rows), "avg_us": round(v["avg_us"], 2), "tflops": round(tf, 1), "pipe_tflops": round(pipe_tf, 1),
                         "frac": round(pipe_tf / peak, 4), "bound": "mfma", "peak": peak,
                         "arithmetic": "bf16x6 (fp32-accurate): 6 bf16 piece products per fp32 product, frac = 6 x tflops / the dense bf16 MFMA peak"
                                       if x6 else "f32 MFMA"}
            # flops per byte fall with H (2 H^2 flops against ~8-12 H bytes per row): at H = 64 the same kernels sit nearer
            # the HBM roof than the MFMA roof -- both fractions are reported, "bound" names the nearer roof
            hbm_frac = v["gbps"] / 8000.0
            out[base]["hbm_gbps"], out[base]["hbm_frac"] = round(v["gbps"], 1), round(hbm_frac, 4)
            if hbm_frac > out[base]["frac"] or x6:
                out[base]["bound"] = "hbm"
    return out


def sparse_end_to_end(cfg, pairs_per_s):
    """Algorithmic bytes of the sparse kernels per pair (per layer and graph, forward + backward,
    ``4H(4E + 6N) + 26E + 12(N + 1)``; E counts the reversed copies) times the measured pairs/s, against the HBM peak."""
    H, L = cfg["hid"], cfg["layers"]
    per_pair = 0
    for n, e in ((cfg["p_nodes"], 2 * cfg["p_edges"]), (cfg["g_nodes"], 2 * cfg["g_edges"])):
        per_pair += L * (4 * H * (4 * e + 6 * n) + 26 * e + 12 * (n + 1))
    gbps = per_pair * pairs_per_s / 1e9
    return {"bytes_per_pair": per_pair, "gbps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4)}


def cpu_baseline(cfg, state_dict, seconds_budget=12.0, B=32, max_steps=50, warmup=3, min_steps=10):
    """The same step on the host cores with the CPU oracle (oracle/model_oracle.py + dmp_oracle.py: the reference's
    operation order -- gather-then-project layers, padded [B, L, D] heads, per-sample Python loops -- in torch CPU ops,
    pinned by the reference's own full-model runs): collate of B per-graph arrays, forward of the WHOLE model, count
    loss, backward, AdamW(amsgrad) -- on a bounded sample (B = 32 pairs per step, the reference's own CPU-runnable batch
    size, BASELINE configs[0]) of the same synthetic workload, from the product model's initial ``state_dict``."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import model_oracle as MO
    # the oracle's ops are small; past ~32 threads torch's intra-op pool only adds contention
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    H, L = cfg["hid"], cfg["layers"]
    rng = np.random.default_rng(7)
    mc = model_config(cfg)
    sd = {k: v.detach().cpu().clone() for k, v in state_dict.items()}
    for k in list(sd):                                       # shared sub-networks: one parameter under two names
        twin = "g_" + k[2:]
        if k.startswith("p_") and twin in sd and sd[k].shape == sd[twin].shape and torch.equal(sd[k], sd[twin]):
            sd[k] = sd[twin]
    params = []
    for k, v in sd.items():
        if v.is_floating_point() and "enc_net" not in k and not any(v is q for q in params):
            params.append(v.requires_grad_(True))
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-5, amsgrad=True)
    samples = []                                             # per-pair arrays, as a dataset holds them
    for _ in range(B):
        pair = {}
        for tag, n, m, nl in (("p", cfg["p_nodes"], cfg["p_edges"], cfg["p_labels"]), ("g", cfg["g_nodes"], cfg["g_edges"], cfg["g_labels"])):
            u, v = er_local_edges(1, n, m, rng)
            el = rng.integers(0, nl, m)
            pair[tag] = dict(src=np.concatenate([u[0], v[0]]), dst=np.concatenate([v[0], u[0]]), n=n,
                             rev=np.concatenate([np.zeros(m, bool), np.ones(m, bool)]), elabel=np.concatenate([el, el + nl]),
                             eid=np.concatenate([np.arange(m), m + np.arange(m)]), label=rng.integers(0, nl, n))
        pair["count"] = float(rng.integers(0, 64))
        samples.append(pair)

    def collate(tag):
        off, parts = 0, {k: [] for k in ("src", "dst", "rev", "elabel", "eid", "label", "id")}
        for s in samples:
            g = s[tag]
            parts["src"].append(g["src"] + off); parts["dst"].append(g["dst"] + off)
            for k in ("rev", "elabel", "eid", "label"):
                parts[k].append(g[k])
            parts["id"].append(np.arange(g["n"]))
            off += g["n"]
        t = {k: torch.from_numpy(np.concatenate(v)) for k, v in parts.items()}
        t["bnn"] = [s[tag]["n"] for s in samples]
        t["bne"] = [len(s[tag]["src"]) for s in samples]
        return t

    def one():
        pattern, graph = collate("p"), collate("g")
        counts = torch.tensor([s["count"] for s in samples]).view(-1, 1)
        out = MO.model_forward(sd, mc, pattern, graph)
        loss = torch.nn.functional.mse_loss(out["pred_c"], counts)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    # SURVEY 8(d)'s protocol: warm-up steps, then at least ``min_steps`` timed steps (more while the budget lasts), the MEDIAN step
    for _ in range(warmup):
        one()
    times, t0 = [], time.perf_counter()
    while True:
        s0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - s0)
        if len(times) >= max_steps or (len(times) >= min_steps and time.perf_counter() - t0 > seconds_budget):
            break
    times.sort()
    med, n = times[len(times) // 2], len(times)
    return {"value": B / med, "unit": "pairs/s", "cores": cores, "kind": "port", "protocol": "%d warm-up + %d timed steps, median step" % (warmup, n),
            "step_s_median": round(med, 4), "step_s_min": round(times[0], 4), "step_s_max": round(times[-1], 4),
            "value_mean": round(B * n / sum(times), 2),
            "sample": "%d steps of B=%d pairs of the same shapes; the same step composition as the GPU line (collate, whole "
                      "model forward: encodings, embeddings, ScalarFilter, %d-layer pattern + target DMPNN rep-nets, node + edge "
                      "SumPredictNet heads; count loss, backward, AdamW(amsgrad)), hid=%d, fp32, reference operation order, "
                      "torch %s CPU, %d threads" % (n, B, L, H, torch.__version__, cores)}


def spawn_check(args, rank, world):
    """--spawn-check: the multi-rank plumbing of this script without the GPU step (tests/test_bench_spawn.py)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = "gloo" if args.backend == "nccl" and not torch.cuda.is_available() else args.backend
    seen = 1
    if world > 1:
        dist.init_process_group(backend, rank=rank, world_size=world)
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
    if rank == 0:
        print(json.dumps({"metric": "(pattern,graph) pairs/sec DMPNN fwd+bwd hid=%d" % args.hid, "value": None, "unit": "pairs/s",
                          "n_gpus": world, "ranks_seen": seen, "backend": backend if world > 1 else None, "steps": args.steps,
                          "warmup": args.warmup, "spawn_check": True,
                          "launcher": "bench.py (self-spawned ranks)" if os.environ.get("DMP_BENCH_SPAWNED") else "external"}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def spawn_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes of this script (one per GPU) and relay
    rank 0's JSON line.  Called BEFORE anything touches the GPU in this process (a process that has initialised HIP must
    neither fork ranks nor exec), the children are started with ``subprocess`` (fresh interpreters, no inherited GPU
    state), rendezvous on 127.0.0.1.  Returns the exit code: non-zero if any rank failed."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DMP_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this host driver
        # rank 0 writes the line to our stdout; the other ranks print nothing on stdout anyway
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    deadline = None
    alive = list(procs)
    while alive:
        for p in list(alive):
            try:
                code = p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            alive.remove(p)
            if code != 0:
                rc = rc or code
                if deadline is None:                           # a dead rank leaves the others in a collective: bound the wait
                    deadline = time.monotonic() + 30.0
        if deadline is not None and alive and time.monotonic() > deadline:
            for p in alive:
                p.kill()                                       # exactly the processes started above
            for p in alive:
                p.wait()
            alive = []
    if rc:
        print("bench.py: a rank process failed (exit code %d)" % rc, file=sys.stderr, flush=True)
    return rc


def profiler_attached():
    """rocprofv3 (or another tool that preloads into this process) is driving the run."""
    return any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def graph_then_eager(argv):
    """``python bench.py`` at N = 1: run ``bench.py --graph`` in a child process and relay its JSON line; if that child
    fails (exit code, or no line), say so on stderr and run ``bench.py --eager`` in a second child.  Returns the exit code
    (None: children cannot be started here)."""
    import subprocess
    env = dict(os.environ, DMP_BENCH_CHILD="1")
    note = None
    for mode in ("--graph", "--eager"):
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)] + list(argv) + [mode], env=env, stdout=subprocess.PIPE)
        except OSError as e:                                   # no child processes here: the caller runs the step itself
            print("bench.py: cannot start a child process (%s): eager launches in this process" % e, file=sys.stderr, flush=True)
            return None
        out = p.stdout.decode(errors="replace").splitlines()
        lines = [l for l in out if l.startswith("{")]
        for l in out:                                          # anything else the child printed
            if not l.startswith("{"):
                print(l)
        if p.returncode == 0 and lines:
            line = lines[-1]
            if note:
                d = json.loads(line)
                d["config"]["launch_fallback"] = d["launch_fallback"] = note
                line = json.dumps(d)
            print(line, flush=True)
            return 0
        note = "bench.py %s failed (exit code %d): eager launches instead" % (mode, p.returncode)
        print(note, file=sys.stderr, flush=True)
    return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=CFG["batch"], help="pairs per GPU")
    ap.add_argument("--workload", type=int, default=2, choices=(2, 4), help="BASELINE config id: 2 = the metric's "
                    "configuration (default, the bench line); 4 = the per-GPU shard of the 8-GPU configuration (size check)")
    ap.add_argument("--act", default="leaky_relu", choices=("leaky_relu", "relu"), help="rep-net / head activation "
                    "(leaky_relu = the reference's default, config.py:298-301,370-373)")
    ap.add_argument("--emb", default="Equivariant", choices=("Equivariant", "Orthogonal", "Normal", "Uniform"),
                    help="embedding kind (Equivariant = the reference's default, config.py:242-245)")
    ap.add_argument("--hid", type=int, default=CFG["hid"], help="hidden width (the metric is quoted at 128; 64 = the reference's "
                    "README width: measured for DESIGN.md, not the bench line)")
    ap.add_argument("--micro-batches", type=int, default=0, help="micro-batches per step (0 = as few as keep the edge rows of "
                    "one pass below 2^30: 1 for config 2 and for config 4's 1024-pair shard)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-b1024", action="store_true", help="also time the CPU oracle's step at the GPU line's batch size (1 warm-up + 3 "
                    "timed steps of 1024 pairs, median: several minutes) -> cpu_baseline_b1024")
    ap.add_argument("--strict-rates", action="store_true", help="a rate of the line above the HBM peak (bench.check_rates) ends the run instead of being reported in the line")
    ap.add_argument("--no-gate-dense", action="store_true", help="skip the extra un-timed steps behind the gate_dense object")
    ap.add_argument("--filter-net", default="ScalarFilter", choices=("ScalarFilter", "None"),
                    help="None: the ALL-ROWS step as the timed step (the model built without its filter net: no 0 / 1 gate, every node "
                         "and edge row live in every layer -- what the reference's one-label ER / Regular datasets give, README.md:22-69); "
                         "for profiling the `gate_dense` control as the headline of its own line")
    ap.add_argument("--no-all-outputs", action="store_true", help="skip the extra un-timed steps behind all_outputs_ms_per_step")
    ap.add_argument("--extended-steps", type=int, default=200,
                    help="further steps after the timed region (same launch mode, one event record each) behind `steps_extended`; 0 = none")
    ap.add_argument("--no-gate-compact", action="store_true", help="skip the extra un-timed steps behind the gate_compact object")
    ap.add_argument("--gate-compact", action="store_true",
                    help="run the rep-net on the target edges the filter gate keeps (model.set_gate_capacity; capacity "
                         "calibrated on the shard): same outputs, a data-dependent share of the edge rows")
    ap.add_argument("--graph", action="store_true",
                    help="record the whole step (collate, index builds, fwd, bwd, gradient pack, AdamW) as ONE HIP graph during "
                         "the warm-up and replay it in the timed region (N = 1, one micro-batch; the per-kernel HIP-event "
                         "numbers then come from eager steps after the timed region).  The DEFAULT at N = 1: tried in a child "
                         "process first, eager launches in a second child if that one fails.  At N > 1 (also the default there): "
                         "forward + backward + gradient pack replayed per rank, all-reduce and optimizer update launched eagerly")
    ap.add_argument("--eager", action="store_true", help="eager launches in the timed region")
    ap.add_argument("--no-tuned-gemms", action="store_true", help="keep hipBLASLt's default solution heuristic")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                    "smoke-test the multi-rank path on a single-GPU box together with --single-device)")
    ap.add_argument("--single-device", action="store_true", help="testing aid: every rank uses cuda:0")
    ap.add_argument("--dump-grad", default=None, metavar="PATH",
                    help="testing aid: before the warm-up, one forward + backward + gradient pack + all-reduce (averaged) from "
                         "the initial parameters; rank 0 saves the flat gradient and its own shard's predictions to PATH")
    ap.add_argument("--emulate-world", type=int, default=1, metavar="W",
                    help="testing aid (one process): the batch is the shards of ranks 0..W-1 back to back -- the global batch "
                         "of a W-rank run")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: form a ONE-rank nccl (RCCL) process group on this GPU and run the N > 1 code path -- "
                         "replayed front, sync.sync(async_op=True), the next batch's prepare_joint, sync.finish (stream-ordered "
                         "Work.wait), AdamW -- so that the path the 8-GPU run takes has executed on RCCL.  No scaling claim.")
    ap.add_argument("--spawn-check", action="store_true",
                    help="testing aid (runs without a GPU): start the ranks, form the process group, count them with an "
                         "all-reduce and print a line with n_gpus / ranks_seen and value null -- no step is run")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: this process becomes the parent of N fresh rank processes and never touches the GPU itself
        if not (args.single_device or args.spawn_check):
            have = torch.cuda.device_count()                   # counting devices does not initialise HIP on this image
            if have < args.gpus:
                raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (use --single-device to time-share one GPU "
                                 "with --backend gloo as a plumbing test)" % (args.gpus, have))
        rc = spawn_ranks(args.gpus, sys.argv[1:])
        if rc != 0 and not (args.graph or args.eager or args.spawn_check):
            # default mode at N > 1: every rank replays forward + backward + gradient pack from one HIP graph; if that rank
            # set failed (a recording that faults takes its process with it), the same ranks once more with eager launches
            note = "bench.py --gpus %d (replayed front) failed with exit code %d: eager launches instead" % (args.gpus, rc)
            print(note, file=sys.stderr, flush=True)
            os.environ["DMP_BENCH_FALLBACK_NOTE"] = note
            rc = spawn_ranks(args.gpus, sys.argv[1:] + ["--eager"])
        sys.exit(rc)

    selftest = os.environ.get("DMP_BENCH_CHILD_SELFTEST")     # testing aid (no GPU): what the two children of the default mode do
    if selftest and os.environ.get("DMP_BENCH_CHILD"):
        if args.graph and selftest == "graph_fails":
            sys.exit(3)
        print(json.dumps({"value": 1.0, "config": {"launch": "graph" if args.graph else "eager"}}), flush=True)
        return
    if args.force_collective and args.gpus != 1:
        raise SystemExit("bench.py: --force-collective is the one-rank form of the N > 1 path (--gpus 1)")
    if ("WORLD_SIZE" not in os.environ and args.gpus == 1 and not (args.graph or args.eager or args.spawn_check or args.force_collective)
            and not os.environ.get("DMP_BENCH_CHILD")):
        # N = 1, no mode asked for: the step replayed from one HIP graph (how harness.fit(graph=True) trains), in a child
        # process -- a recording that fails takes its process with it -- and eager launches in a second child if it does.
        # This process never touches the GPU.  Under a profiler (its preloaded library has initialised the GPU in THIS
        # process, which must then not start programs) and wherever children cannot be started: in this process.
        if profiler_attached():
            args.graph = True
        else:
            rc = graph_then_eager(sys.argv[1:])
            if rc is not None:
                sys.exit(rc)
            args.eager = True

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (the launcher's --nproc-per-node must equal --gpus)"
                         % (args.gpus, world))
    if args.spawn_check:
        return spawn_check(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (no CPU fallback in the product path)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_collective
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # N > 1 without a mode flag: forward + backward + gradient pack replayed from one HIP graph per rank (eager launches need
    # ~5 ms of host time per step and rank, as much as the device time: N Python launchers on one host would set the curve),
    # all-reduce and optimizer update launched eagerly behind every replay; a recording that RAISES on any rank sends all
    # ranks back to eager launches (agreed by an all-reduce after the warm-up)
    auto_graph = multi and not (args.graph or args.eager)
    if auto_graph:
        args.graph = True
    cfg = dict(CFG if args.workload == 2 else CFG4, batch=args.batch, act=args.act, emb=args.emb, micro_batches=args.micro_batches,
               hid=args.hid, graph=args.graph, gate_compact=args.gate_compact, filter=args.filter_net)
    from dualmessagepassing_amd import _lib
    from dualmessagepassing_amd.tuning import enable_tuned_gemms
    tuned = False if (args.no_tuned_gemms or os.environ.get("PYTORCH_TUNABLEOP_ENABLED")) else enable_tuned_gemms()
    if args.emulate_world > 1:
        if world != 1:
            raise SystemExit("bench.py: --emulate-world is a single-process aid")
        shard = concat_shards([make_shard(cfg, r, device) for r in range(args.emulate_world)])
        cfg = dict(cfg, batch=cfg["batch"] * args.emulate_world)
    else:
        shard = make_shard(cfg, rank, device)
    step, model = build_step(cfg, shard, device, world, collective=args.force_collective)
    if args.dump_grad:
        step.front()
        step.sync.sync()                                       # the ranks' average (nothing at world size 1)
        torch.cuda.synchronize()
        if rank == 0:
            torch.save({"flat": step.sync.flat.detach().cpu(), "pred_c": step.last_pred_c.cpu(), "world": world,
                        "batch": cfg["batch"]}, args.dump_grad)
    initial_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if (rank == 0 and world == 1
                                                                                              and not multi and not args.no_cpu_baseline) else None

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    run, graphed = step, False
    if args.graph and not multi and step.micro_batches == 1:
        # the step reads the shard's tensors and clones the size / flag tensors itself: a replay rebuilds every index
        # of the batch on the device exactly as an eager step does
        from dualmessagepassing_amd.dp import StepGraph
        # (a leading non-tensor argument is part of a recording's signature: ``run("all")`` is the all-outputs step)
        # (``run("gate")``: the step with the gate capacity set -- the model's state at recording time is part of it)
        run = StepGraph(lambda *m: step(all_outputs="all" in m), optimizer=step.opt, max_shapes=3)
        graphed = True
    elif args.graph and multi and step.micro_batches == 1:
        # more than one rank: forward + backward + gradient pack replayed from one HIP graph per rank, the gradient
        # all-reduce and the optimizer update launched eagerly after every replay (no collective inside a recording)
        from dualmessagepassing_amd.dp import StepGraph
        front_graph = StepGraph(lambda: step.front(), optimizer=step.opt, max_shapes=1)
        graph_state = {"ok": True, "note": None}

        def run():
            if graph_state["ok"]:
                try:
                    loss = front_graph()
                except Exception as e:                        # noqa: BLE001 -- whatever the recording raised
                    if not auto_graph:
                        raise
                    graph_state["ok"], graph_state["note"] = False, "rank %d: recording failed (%s: %s)" % (rank, type(e).__name__, e)
                    print("bench.py " + graph_state["note"], file=sys.stderr, flush=True)
                    loss = step.front()                        # the same collectives as the other ranks: front, then tail
            else:
                loss = step.front()
            step.tail()
            return loss

        run.on_stream = front_graph.on_stream
        graphed = True
    if graphed:          # the recordings' side stream is the current stream of everything below (see dp.StepGraph)
        import contextlib
        stack = contextlib.ExitStack()
        stack.enter_context(run.on_stream())
    for _ in range(max(args.warmup, 3) if graphed else args.warmup):   # graphed: eager, record + replay, replay
        run()
    step.finish()
    launch_fallback = os.environ.get("DMP_BENCH_FALLBACK_NOTE")
    if graphed and multi:
        ok = torch.tensor([1 if graph_state["ok"] else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:                                # some rank could not record: every rank launches eagerly
            launch_fallback = graph_state["note"] or "another rank's recording failed: eager launches on every rank"
            run, graphed = step, False
            for _ in range(2):
                run()
            step.finish()
    step.time_allreduce = True
    # Setup objects (modules, tuned-GEMM tables, the shard) leave the cyclic collector's working set: a full
    # collection walking them costs tens of milliseconds and would otherwise land inside a step now and then.
    import gc
    gc.collect()
    gc.freeze()
    gc.set_threshold(200000, 20, 20)     # young-generation passes every 200 k allocations (~30 steps) instead of every 700
    barrier()
    # timed region: HIP events only around the roofline kernel (the scatter-add), so that the
    # event records do not perturb the step; every other kernel is timed in extra steps below
    _lib.timer.reset()
    _lib.timer.only = "seg_sum2"
    _lib.timer.enabled = not graphed                         # no event records inside a replayed graph
    # one event record per step on the step's stream (the stream run() launches on): the spread of the device time per
    # step; 20 steps of 6 ms are too short to trust a mean alone.  Event k marks the enqueue point of step k, so a
    # difference is the device time of one step once the queue is full (the GPU-bound case).
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    del step.ar_events[:]
    host_s = 0.0                                             # host time inside run(): enqueue (and, with gloo, the blocking collective)
    t0 = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        h0 = time.perf_counter()
        run()
        host_s += time.perf_counter() - h0
    step.finish()                                            # the last step's all-reduce + optimizer update: inside the timed region
    marks[args.steps].record()
    barrier()
    dt = time.perf_counter() - t0
    per_step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)) if args.steps else []
    step.time_allreduce = False
    ar_ms = [a.elapsed_time(b) for a, b in step.ar_events]
    mine = {"rank": rank, "host_ms_per_step": round(host_s / max(args.steps, 1) * 1e3, 3),
            "allreduce_wait_ms": round(sum(ar_ms) / len(ar_ms), 3) if ar_ms else None,
            "allreduce_wait_ms_max": round(max(ar_ms), 3) if ar_ms else None,
            "step_ms_median": round(per_step_ms[len(per_step_ms) // 2], 3) if per_step_ms else None,
            "wall_ms_per_step": round(dt / max(args.steps, 1) * 1e3, 3)}
    kern = _lib.timer.summary()
    extended = None
    if args.extended_steps > 0:
        # the driver fixes --steps 20 (0.1 s of device time): the same step, same launch mode, for `--extended-steps` more
        # steps with one event record per step -- a median over a second of work, reported beside the line, never as `value`
        _lib.timer.enabled = False
        n_ext = args.extended_steps
        ext_marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_ext + 1)]
        barrier()
        tx = time.perf_counter()
        for i in range(n_ext):
            ext_marks[i].record()
            run()
        step.finish()
        ext_marks[n_ext].record()
        barrier()
        ext_dt = time.perf_counter() - tx
        ext_ms = sorted(ext_marks[i].elapsed_time(ext_marks[i + 1]) for i in range(n_ext))
        extended = {"steps": n_ext, "ms_per_step": round(ext_dt / n_ext * 1e3, 3), "step_ms_min": round(ext_ms[0], 3),
                    "step_ms_median": round(ext_ms[n_ext // 2], 3), "step_ms_p90": round(ext_ms[(9 * n_ext) // 10], 3),
                    "step_ms_max": round(ext_ms[-1], 3), "value": round(cfg["batch"] * world * n_ext / ext_dt, 1)}
    eager_ms = None
    if graphed:
        # the same K steps once more as eager launches: the scatter-add's HIP-event time (no event records inside a replayed
        # graph) and the eager step time, reported beside the replayed one
        _lib.timer.reset()
        _lib.timer.only = "seg_sum2"
        _lib.timer.enabled = True
        for _ in range(max(args.warmup, 5)):                   # the eager path's own warm-up (its allocations are not the graph's)
            step()
        step.finish()
        barrier()
        _lib.timer.reset()
        te = time.perf_counter()
        for i in range(args.steps):
            step()
        step.finish()
        barrier()
        eager_ms = (time.perf_counter() - te) / max(args.steps, 1) * 1e3
        kern = _lib.timer.summary()
    all_ms = None
    if not multi and step.micro_batches == 1 and not args.no_all_outputs:
        # the same step with every entry of the output dictionary formed (the reference returns all 15 as tensors,
        # basemodel.py:1645-1661): the last layer's edge rows built by the layer itself (lazy_edge_rep off), the target
        # embeddings read -- same launch mode as the timed region, reported beside the headline, never as `value`
        fn = (lambda: run("all")) if graphed else (lambda: step(all_outputs=True))
        _lib.timer.enabled = False
        for _ in range(3):                                     # graphed: eager, record + replay, replay
            fn()
        step.finish()
        barrier()
        ta = time.perf_counter()
        for _ in range(args.steps):
            fn()
        step.finish()
        barrier()
        all_ms = (time.perf_counter() - ta) / max(args.steps, 1) * 1e3
    gate_line = None
    if not multi and step.micro_batches == 1 and not args.gate_compact and not args.no_gate_compact:
        # the same step with the rep-net on the target edges the filter gate keeps (model.set_gate_capacity): same outputs
        # (tests/test_gpu_compact.py, test_gpu_bench_composite.py), a share of the edge rows that depends on the labels --
        # same launch mode as the timed region, reported beside the headline, never as `value`
        info = step.set_gate_compact(True)
        if info is not None:
            _lib.timer.enabled = False
            run_c = (lambda: run("gate")) if graphed else step
            for _ in range(3):                                 # graphed: eager, record + replay, replay
                run_c()
            step.finish()
            barrier()
            tg = time.perf_counter()
            for _ in range(args.steps):
                run_c()
            step.finish()
            barrier()
            g_ms = (time.perf_counter() - tg) / max(args.steps, 1) * 1e3
            # the two scatter-add launches at this row count: HIP events, eager launches
            _lib.timer.reset()
            _lib.timer.only = "seg_sum2"
            _lib.timer.enabled = True
            for _ in range(3):
                step()
            step.finish()
            barrier()
            _lib.timer.reset()
            for _ in range(args.steps):
                step()
            step.finish()
            barrier()
            kern_c = _lib.timer.summary()
            _lib.timer.enabled = False
            bits = model.compaction_status()
            if bits:
                raise SystemExit("bench.py: gate compaction status %d (a batch did not fit its capacity)" % bits)
            gate_line = dict(info, ms_per_step=round(g_ms, 3), value=round(cfg["batch"] / g_ms * 1e3, 1), kern=kern_c)
        step.set_gate_compact(False)
    dense_line = None
    if not multi and step.micro_batches == 1 and args.workload == 2 and not args.no_gate_dense and not args.gate_compact and args.filter_net == "ScalarFilter":
        # the CONTROL for everything the 0 / 1 gates buy: the same step, same launch mode, with the model built WITHOUT its
        # filter net (filter_net = "None": no gate, every node and edge row of the batch is live in every layer) -- what the
        # step costs when the data gate nothing out.  Reported beside the headline, never as `value`.
        _lib.timer.enabled = False
        step_d, model_d = build_step(dict(cfg, filter="None", gate_compact=False), shard, device, world)
        if graphed:
            from dualmessagepassing_amd.dp import StepGraph
            run_d = StepGraph(lambda: step_d(), optimizer=step_d.opt, max_shapes=1)
        else:
            run_d = step_d
        for _ in range(3):                                     # graphed: eager, record + replay, replay
            run_d()
        step_d.finish()
        barrier()
        td = time.perf_counter()
        for _ in range(args.steps):
            run_d()
        step_d.finish()
        barrier()
        d_ms = (time.perf_counter() - td) / max(args.steps, 1) * 1e3
        # ... and its own scatter-add launches (HIP events inside eager steps, as for the headline) and kernel table: the
        # all-rows step is a first-class number (the reference's ER / Regular datasets have ONE label: nothing is gated there)
        _lib.timer.reset()
        _lib.timer.only = "seg_sum2"
        _lib.timer.enabled = True
        for _ in range(3):
            step_d()
        step_d.finish()
        barrier()
        _lib.timer.reset()
        te = time.perf_counter()
        for _ in range(args.steps):
            step_d()
        step_d.finish()
        barrier()
        d_eager_ms = (time.perf_counter() - te) / max(args.steps, 1) * 1e3
        kern_d = _lib.timer.summary()
        _lib.timer.reset()
        _lib.timer.only = None
        side_was, _side_d = None, __import__("dualmessagepassing_amd.side", fromlist=["side"])
        side_was, _side_d.USE_SIDE_STREAM = _side_d.USE_SIDE_STREAM, False
        try:
            for _ in range(3):
                step_d()
            step_d.finish()
        finally:
            _side_d.USE_SIDE_STREAM = side_was
        for name, v in _lib.timer.summary().items():
            kern_d.setdefault(name, v)
        _lib.timer.enabled = False
        dense_line = {"what": "same step, same launch mode, the model built with filter_net = 'None': no ScalarFilter gate, every node "
                              "and edge row live in every layer (the step when the data gate nothing out: the reference's one-label "
                              "ER / Regular datasets, SubgraphCountingMatching/README.md:22-69)",
                      "ms_per_step": round(d_ms, 3), "value": round(cfg["batch"] / d_ms * 1e3, 1), "unit": "pairs/s",
                      "eager_ms_per_step": round(d_eager_ms, 3), "kept_fraction_of_rows": 1.0, "kern": kern_d}
        del run_d, step_d, model_d
    _lib.timer.reset()
    _lib.timer.only = None
    _lib.timer.enabled = True
    from dualmessagepassing_amd import side as _side
    side_was, _side.USE_SIDE_STREAM = _side.USE_SIDE_STREAM, False   # a kernel's own time: nothing of the side stream beside it
    try:
        for _ in range(3):  # un-timed: per-kernel numbers of the other HIP kernels on the path
            step()
        step.finish()
    finally:
        _side.USE_SIDE_STREAM = side_was
    others = _lib.timer.summary()
    _lib.timer.enabled = False
    for name, v in others.items():
        kern.setdefault(name, v)

    ranks_seen, devices = 1, [torch.cuda.get_device_name(device)]
    per_rank = [mine]
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(c)                                   # every rank that timed the region adds one
        ranks_seen = int(c.item())
        names = [None] * world
        dist.all_gather_object(names, "cuda:%d %s" % (local_rank, torch.cuda.get_device_name(device)))
        devices = names
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        pairs = cfg["batch"] * world * args.steps
        H = cfg["hid"]
        # the scatter-add launch: the shared rep-net runs once over the union of the pattern and
        # target batches, so one launch covers N = B*(8+64) destination rows and E = B*(24+512) edge rows
        mb = cfg["batch"] // step.micro_batches               # pairs per pass
        uN = mb * (cfg["p_nodes"] + cfg["g_nodes"])
        uE = mb * 2 * (cfg["p_edges"] + cfg["g_edges"])
        if step.gate_capacity:                                # the rep-net ran on the kept target edges + padding
            uE = mb * 2 * cfg["p_edges"] + step.gate_capacity
        key = "seg_sum2[H=%d,rows=%d,ent=%d]" % (H, uN, uE)
        key_kinc = "seg_sum2_kept_inc[H=%d,rows=%d,ent=%d]" % (H, uN, 2 * uE)   # the segment sum over the kept edges' incidence CSR, a row per kept node (both gates)
        key_inc = "seg_sum2_graphs[H=%d,rows=%d,E=%d]" % (H, uN, uE)      # one pass over the edge rows (csrc/dmp_segacc.hip) ...
        if key_inc not in kern:
            key_inc = "seg_sum2[H=%d,rows=%d,ent=%d]" % (H, uN, 2 * uE)   # ... or the segment sum over the incidence CSR
        prof = committed_profile(uN, uE, H)
        roof = roof_bwd = None
        # rows the two scatter-adds leave out: the target edge rows under a zero of the filter's 0 / 1 edge gate (not in the
        # gate-compact mode: its batch holds the kept rows only)
        skipped = skipped_n = 0
        Et = mb * 2 * cfg["g_edges"]
        Etk = Et
        kept = {}
        from dualmessagepassing_amd import fused as _fused
        if _fused.USE_MASKED_SUMS and _fused.USE_ROW_MASKS and step.micro_batches == 1:
            kept = step.gate_kept_rows()
            if kept["edges"][0] is not None and not step.gate_capacity:
                skipped = kept["edges"][1] - kept["edges"][0]
                Etk = kept["edges"][0]
            if kept["nodes"][0] is not None and _fused.USE_NODE_ROWS and _fused.USE_PLAIN_ATB and uN >= 4096:
                skipped_n = kept["nodes"][1] - kept["nodes"][0]          # (fused.node_rows: the node side runs on the kept nodes)
        Ek, Nk, nB = uE - skipped, uN - skipped_n, 2 * mb
        # the backward's sums over the kept edges' incidence CSR fetch the kept edges WITH a kept endpoint (pattern edges: all)
        Ek1 = (uE - Et) + kept.get("edges_with_kept_endpoint", Etk) if (skipped or skipped_n) else uE
        inc_k = 2 * (uE - Et) + kept.get("kept_incidences", 2 * Etk) if (skipped or skipped_n) else 2 * uE
        # the forward aggregation over the kept NODES' rows fetches the kept edges INTO a kept node only (pattern edges: all)
        Ein = (uE - Et) + kept.get("kept_in_edges", Etk) if skipped_n else Ek
        if key in kern:     # forward: S[v] = [- sum Z[e] | + sum Z[e]] over the in-edges (dmpnn.py:92,163 with the products moved behind the sum)
            roof = seg_roofline(kern[key], "dmp::seg_sum_vec<32,split,remap> (DMPLayer node aggregation by destination, "
                                "N=%d rows, E=%d edge rows, H=%d)" % (uN, uE, H),
                                4 * H * Ein + 8 * H * Nk + 4 * Ein + (12 * Nk if skipped_n else 4 * (uN + 1)),
                                4 * H * (Ein + Nk) + 4 * Ein + 4 * (Nk + 1), prof["in"], uE - Ein,
                                4 * H * (uE + 2 * uN) + 4 * uE + 4 * (uN + 1), 4 * H * (uE + uN) + 4 * uE + 4 * (uN + 1), skipped_n)
            if skipped_n:
                roof["row_reads"] = int(Ein)     # (of the %d edge rows the 0 / 1 edge gate keeps: those whose destination node the node gate keeps too)
        if key_kinc in kern and (skipped or skipped_n):
            # backward of the edge gathers under both gates: every kept node's two sums over its kept edges (ascending edge id)
            roof_bwd = seg_roofline(kern[key_kinc], "dmp::seg_sum_vec<32,split,remap,incidence> over the kept edges' incidence CSR, a row per "
                                    "kept node (dmp_incidence_keep + dmp_seg_sum2_rows); gradient of the gathered node projections: N=%d rows, "
                                    "E=%d edge rows, H=%d" % (uN, uE, H),
                                    4 * H * Ek1 + 8 * H * Nk + 4 * inc_k + 12 * Nk,
                                    4 * H * (Ek1 + 2 * Nk) + 9 * Ek1 + 8 * (Nk + 1), prof["inc"], uE - Ek1,
                                    4 * H * (uE + 2 * uN) + 8 * uE + 8 * (uN + 1), 4 * H * (uE + 2 * uN) + 9 * uE + 8 * (uN + 1), skipped_n)
            roof_bwd["row_reads"] = int(inc_k)      # (an edge row with both endpoints kept is read twice, by workgroups of one XCD)
            kern[key_kinc]["bytes"], kern[key_kinc]["rows"] = 4 * H * Ek1 + 8 * H * Nk + 4 * inc_k + 12 * Nk, "kept"
            kern[key_kinc]["gbps"] = kern[key_kinc]["bytes"] / kern[key_kinc]["avg_us"] / 1e3
        elif key_inc in kern:  # backward of the edge gathers: the same kernel over the incidence CSR (every edge row under both endpoints)
            graphs = "graphs" in key_inc
            roof_bwd = seg_roofline(kern[key_inc], ("dmp::seg_acc_graphs_k (one pass over the edge rows, both endpoints' sums in registers" if graphs
                                                    else "dmp::seg_sum_vec<32,split,remap,incidence> (every edge row under both endpoints") +
                                    "; gradient of the gathered node projections: N=%d rows, E=%d edge rows, H=%d)" % (uN, uE, H),
                                    (4 * H * Ek + 8 * H * Nk + 8 * uE + uE // 8 + uN // 8 + 16 * (nB + 1)) if graphs else kern[key_inc]["bytes"],
                                    (4 * H * (Ek + 2 * Nk) + 9 * Ek + 8 * (Nk + 1)) if graphs else 4 * H * (uE + 2 * uN) + 9 * uE + 8 * (uN + 1),
                                    prof["inc"], skipped if graphs else 0,
                                    4 * H * (uE + 2 * uN) + 8 * uE + 16 * (nB + 1), 4 * H * (uE + 2 * uN) + 9 * uE + 8 * (uN + 1),
                                    skipped_n if graphs else 0)
        # the per-kernel table: the timer prices a launch at its launch SHAPE; under the gates the kernels process the kept rows
        if skipped or skipped_n:
            for name, v in kern.items():
                kb = kept_row_bytes(name, H, uN, uE, Nk, Ek, Et, Etk, nB, Ein)
                if kb is not None:
                    v["bytes"], v["gbps"], v["rows"] = kb, kb / v["avg_us"] / 1e3, "kept"
        if skipped and not multi:
            # ... and the same two launches over ALL rows (no mask, no weights): the kernels as they run without a 0 / 1 gate,
            # timed stand-alone on this step's index -- so that the fraction of the unmasked kernels stays on record
            plain = step.plain_scatter_adds(H)
            for r_, key_, own_, sv_ in ((roof, "fwd_us", 4 * H * (uE + 2 * uN) + 4 * uE + 4 * (uN + 1), 4 * H * (uE + uN) + 4 * uE + 4 * (uN + 1)),
                                        (roof_bwd, "bwd_us", 4 * H * (uE + 2 * uN) + 8 * uE + 16 * (cfg["batch"] * 2 + 1), 4 * H * (uE + 2 * uN) + 9 * uE + 8 * (uN + 1))):
                if r_ is not None and plain:
                    r_["all_rows_launch"] = {"what": "the same kernel without the masks / the gate weights, every row read and written; stand-alone, median of 20 launches",
                                             "avg_us": round(plain[key_], 2), "bytes_per_launch": int(sv_), "bytes_own": int(own_),
                                             "frac": round(sv_ / plain[key_] / 1e3 / HBM_PEAK_GBPS, 4),
                                             "frac_own_bytes": round(own_ / plain[key_] / 1e3 / HBM_PEAK_GBPS, 4)}
        if not multi:
            # the ceiling of a launch of each scatter-add's size (its own byte count), measured live beside it
            for r_ in (roof, roof_bwd):
                if r_ is not None:
                    r_["same_size_copy"] = step.copy_ceiling(r_["bytes_own"])
                    r_["own_rate_over_copy_rate"] = round(r_["frac_own_bytes"] / r_["same_size_copy"]["frac"], 4) if r_["same_size_copy"]["frac"] else None
        if dense_line is not None:
            dense_line = dense_summary(dense_line, H, uN, uE, 2 * mb)
            for key_ in ("roofline", "roofline_bwd"):
                if dense_line.get(key_):
                    dense_line[key_]["same_size_copy"] = step.copy_ceiling(dense_line[key_]["bytes_own"])
        line = {
            "metric": "(pattern,graph) pairs/sec DMPNN fwd+bwd hid=%d" % H, "value": round(pairs / dt, 1),
            "unit": "pairs/s", "n_gpus": world, "ranks_seen": ranks_seen,
            "backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if multi else None,
            # --force-collective: the N > 1 code path on a one-rank process group (what ran, not a scaling figure)
            "collective_forced": bool(args.force_collective),
            "devices": devices, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            # device time per step between consecutive per-step event records on rank 0 (min / median / max)
            "step_ms_min": round(per_step_ms[0], 3) if per_step_ms else None,
            "step_ms_median": round(per_step_ms[len(per_step_ms) // 2], 3) if per_step_ms else None,
            "step_ms_max": round(per_step_ms[-1], 3) if per_step_ms else None,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # how the timed steps were launched, and the other mode's / the all-outputs step's time beside it
            "launch_mode": ("hip_graph_replay" if not multi else "hip_graph_replay_front+eager_allreduce_adamw") if graphed else "eager",
            "launch_fallback": launch_fallback,
            "eager_ms_per_step": round(eager_ms, 3) if eager_ms is not None else None,
            "all_outputs_ms_per_step": round(all_ms, 3) if all_ms is not None else None,
            "steps_extended": extended,
            # per rank: host time inside the step call, the compute stream's wait for the gradient sum (HIP events around
            # the all-reduce / its wait), device time per step -- what a bad scaling curve is diagnosed from
            "per_rank": per_rank,
            "arithmetic": "fp32 storage and accumulation; dense products on the f32-input MFMA except the class-typed edge kernels (fp32 operands as 3 bf16 pieces, 6 piece products per partial product: fp32-accurate, parity tests at the fp32 tolerances; DMP_EXACT_FP32=1 switches them back)",
            "config": {"workload": "BASELINE configs[%d]: ER pattern(%d,%d)x target(%d,%d), add_rev, "
                                   "batch=%d pairs/GPU, full DMPNN model (Multihot enc, %s emb, %s, "
                                   "3 shared DMPLayers, SumPredictNet node+edge heads), activation %s, hid=%d, fp32; a new batch "
                                   "(fresh size / flag tensors) every step"
                                   % (cfg["config_id"] - 1, cfg["p_nodes"], cfg["p_edges"], cfg["g_nodes"], cfg["g_edges"],
                                      cfg["batch"], cfg["emb"],
                                      "ScalarFilter" if cfg.get("filter", "ScalarFilter") == "ScalarFilter" else "NO filter net: every row live",
                                      cfg["act"] + (" (slope 1/5.5)" if cfg["act"] == "leaky_relu" else ""), H),
                       "global_batch": cfg["batch"] * world, "parallelism": "dp%d" % world, "micro_batches": step.micro_batches,
                       "step": "device collate + index build + fwd + bwd + grad all-reduce (async, overlapped with the next batch's "
                               "collate / index build) + AdamW(amsgrad, train.py:1231) as one HIP launch",
                       "gemm_solutions": "tuned (TunableOp file)" if tuned else "library default",
                       "launch": (("one HIP graph replay per step" if not multi else
                                   "forward + backward + gradient pack as one HIP graph replay per rank and step, all-reduce and "
                                   "optimizer update launched eagerly")
                                  + " (recorded during the warm-up; the roofline kernel's HIP-event time "
                                  "from the same number of eager steps run after the timed region: eager_ms_per_step)")
                       if graphed else "eager launches",
                       "eager_ms_per_step": round(eager_ms, 3) if eager_ms is not None else None,
                       "all_outputs_ms_per_step": round(all_ms, 3) if all_ms is not None else None,
                       "all_outputs": "same launch mode, model.lazy_edge_rep = False and all 15 OutputDict entries read (g_v_emb, "
                                      "g_e_emb, p_e_rep, g_e_rep formed in HBM as the reference forms them)",
                       "peak_hbm_allocated_gb": round(torch.cuda.max_memory_allocated() / 1e9, 2)},
            "roofline": roof,
            "roofline_bwd": roof_bwd,
            "gate_compact": gate_summary(gate_line, cfg, step, H) if gate_line else None,
            "gate_dense": dense_line,
            # the rows of THIS batch the 0 / 1 gates leave live (pattern rows are always live): what the headline's kernels process
            "gate_kept": {"edge_rows": int(Ek), "of_edge_rows": int(uE), "node_rows": int(Nk), "of_node_rows": int(uN)},
            # the compulsory traffic of the sparse kernels alone (seg-sum / gather-combine, forward + backward,
            # pattern + target, all layers) over the END-TO-END step time -- how far the whole step is from a sparse-only
            # HBM roofline (the step also runs 460 GFLOP of dense fp32 products, which bound it)
            "sparse_path_end_to_end": sparse_end_to_end(cfg, pairs / dt),
            # the time-dominant kernels are the fp32 MFMA kernels of the edge chain (exact-fp32
            # v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s peak): their MFMA-roofline fractions, for context
            "mfma_kernels": mfma_rooflines(kern, H, uE, Ek),
            "kernels": {n: {"avg_us": round(v["avg_us"], 2), "gbps": round(v["gbps"], 1), "launches": v["launches"],
                            "bytes": int(v["bytes"]), **({"rows": v["rows"]} if "rows" in v else {})} for n, v in sorted(kern.items())},
        }
        line["launcher"] = ("bench.py (self-spawned ranks)" if os.environ.get("DMP_BENCH_SPAWNED") else
                            "external (torch.distributed.run)") if world > 1 else "single process"
        if not multi and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, initial_state)
            # --cpu-b1024: the same CPU step at the GPU line's batch size (B = 1024 pairs: a minute per step on 32 threads, so
            # 1 warm-up + 3 timed steps, median; off by default -- the default run has to finish within minutes); the B = 32
            # sample above is the reference's own CPU-runnable batch size, this one is the like-for-like size
            line["cpu_baseline_b1024"] = None
            if args.cpu_b1024 and args.workload == 2:
                line["cpu_baseline_b1024"] = cpu_baseline(cfg, initial_state, seconds_budget=0.0, B=cfg["batch"], max_steps=3, warmup=1, min_steps=3)
        # the rate self-check: a refused rate is an accounting error -- reported IN the line (``rate_check``) and on stderr, and
        # fatal under --strict-rates (the tests read the printed line and fail on the field); the measurement itself stands
        try:
            check_rates(line)
        except SystemExit as e:
            if args.strict_rates:
                raise
            line["rate_check"] = "FAILED: " + str(e).splitlines()[0][:300]
            print(str(e), file=sys.stderr, flush=True)
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

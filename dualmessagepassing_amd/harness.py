"""Thin training / evaluation harness around the drop-in model.

Reproduces the loop semantics of ``SubgraphCountingMatching/train.py:449-844`` (train_epoch) and
``:847-1061`` (evaluate_epoch) for the GraphAdj models on the MI355X path: count loss
``bp_crit(leaky_relu(pred_c, neg_slp), counts)``, evaluation metric on ``relu(pred_c)``, optional
representation regulariser, gradient clipping, AdamW; without the reference's seven ``.item()``
host syncs per step (running sums stay on the device).  Plus a synthetic (pattern, graph) dataset
with exact subgraph-isomorphism counts for smoke-level training runs: the reference's datasets are
external downloads that are not available offline.
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from .collate import collate_device, subiso_weights
from .dp import FlatGradSync


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def enumerate_subisomorphisms(p_src, p_dst, p_vl, p_el, g_src, g_dst, g_vl, g_el, limit=-1):
    """All label-preserving injective maps of the pattern's nodes into the graph's nodes such that
    every pattern edge (u -> v, label l) has an image edge with the same label (non-induced
    subgraph isomorphisms: a sample's ``subisomorphisms`` rows, whose number is its ``counts``),
    as an int64 array [counts, pattern_nodes] in lexicographic order.  Native depth-first search on the host
    (``dmp_subiso_enumerate``, csrc/dmp_subiso.cpp); ``enumerate_subisomorphisms_py`` is its plain-Python twin."""
    from . import _lib
    lib = _lib.load()
    arrs = [_i64(a) for a in (p_src, p_dst, p_vl, p_el, g_src, g_dst, g_vl, g_el)]
    ps, pd, pvl, pel, gs, gd, gvl, gel = arrs
    P = lambda a: a.ctypes.data if a.size else None
    pn, cap = len(pvl), 1024
    while True:
        rows = np.empty((cap, max(pn, 1)), dtype=np.int64)
        c = int(lib.dmp_subiso_enumerate(pn, len(ps), P(ps), P(pd), P(pvl), P(pel), len(gvl), len(gs), P(gs), P(gd), P(gvl), P(gel),
                                         rows.ctypes.data, cap, int(limit)))
        if c < 0:
            raise ValueError("enumerate_subisomorphisms: edge endpoints outside the graphs")
        if c <= cap:
            return rows[:c, :pn].copy().reshape(-1, pn)
        cap = c


def count_subisomorphisms_batch(pairs, threads=0):
    """``counts`` of many pairs at once over a pool of host threads (``dmp_subiso_count_batch``); ``pairs``: a list of
    ``(p_src, p_dst, p_vl, p_el, g_src, g_dst, g_vl, g_el)``."""
    from . import _lib
    lib = _lib.load()
    n = len(pairs)
    cat = lambda k: _i64(np.concatenate([_i64(p[k]) for p in pairs])) if n else np.zeros(0, np.int64)
    off = lambda k: _i64(np.concatenate([[0], np.cumsum([len(p[k]) for p in pairs])]))
    ps, pd, pvl, pel, gs, gd, gvl, gel = (cat(k) for k in range(8))
    pno, peo, gno, geo = off(2), off(0), off(6), off(4)
    counts = np.zeros(n, np.int64)
    P = lambda a: a.ctypes.data if a.size else None
    rc = lib.dmp_subiso_count_batch(n, P(pno), P(peo), P(ps), P(pd), P(pvl), P(pel), P(gno), P(geo), P(gs), P(gd), P(gvl), P(gel),
                                    P(counts), int(threads))
    if rc != 0:
        raise ValueError("count_subisomorphisms_batch: bad input")
    return counts


def enumerate_subisomorphisms_py(p_src, p_dst, p_vl, p_el, g_src, g_dst, g_vl, g_el):
    """The same enumeration as plain backtracking in Python (cross-check of the native search)."""
    np_, ng = len(p_vl), len(g_vl)
    adj = {}
    for u, v, l in zip(g_src.tolist(), g_dst.tolist(), g_el.tolist()):
        adj.setdefault((u, v), set()).add(l)
    pe = list(zip(p_src.tolist(), p_dst.tolist(), p_el.tolist()))
    touching = [[(u, v, l) for (u, v, l) in pe if u == k or v == k] for k in range(np_)]
    mapping, used = [-1] * np_, [False] * ng

    found = []

    def rec(k):
        if k == np_:
            found.append(tuple(mapping))
            return
        for cand in range(ng):
            if used[cand] or g_vl[cand] != p_vl[k]:
                continue
            mapping[k] = cand
            good = True
            for u, v, l in touching[k]:   # every pattern edge whose endpoints are both placed
                mu, mv = mapping[u], mapping[v]
                if mu >= 0 and mv >= 0 and l not in adj.get((mu, mv), ()):
                    good = False
                    break
            if good:
                used[cand] = True
                rec(k + 1)
                used[cand] = False
            mapping[k] = -1

    rec(0)
    return np.array(found, dtype=np.int64).reshape(-1, np_)


def count_subisomorphisms(p_src, p_dst, p_vl, p_el, g_src, g_dst, g_vl, g_el):
    """``counts`` of a (pattern, graph) pair: the number of subisomorphisms."""
    return int(count_subisomorphisms_batch([(p_src, p_dst, p_vl, p_el, g_src, g_dst, g_vl, g_el)], threads=1)[0])


def _er_edges(n, m, rng):
    total = n * (n - 1)
    pick = rng.choice(total, size=m, replace=False)
    u = pick // (n - 1)
    r = pick % (n - 1)
    return u.astype(np.int64), (r + (r >= u)).astype(np.int64)


class PairDataset:
    """(pattern, graph) samples in memory, stored the way the reference's preprocessing leaves a
    GraphAdj sample: ids = arange, reversed edges appended ([forward | reversed], id + max_ne,
    label + max_nel, is_reversed; train.py:299-327).  ``shape`` holds the dataset-wide maxima the
    model vocabulary is sized by (train.py:1164-1181)."""

    def __init__(self, samples, shape):
        self.samples, self.shape = samples, shape

    @classmethod
    def from_loaded(cls, loaded):
        """Samples as ``dataio.load_data`` returns them (forward edges only) -> PairDataset."""
        mx = lambda f: max([f(x) for x in loaded] + [1])
        shape = dict(p_nodes=mx(lambda x: x["pattern"]["num_nodes"]), p_edges=mx(lambda x: len(x["pattern"]["src"])),
                     g_nodes=mx(lambda x: x["graph"]["num_nodes"]), g_edges=mx(lambda x: len(x["graph"]["src"])),
                     n_vlabels=mx(lambda x: int(max(x["pattern"]["vlabel"].max(initial=0), x["graph"]["vlabel"].max(initial=0))) + 1),
                     n_elabels=mx(lambda x: int(max(x["pattern"]["elabel"].max(initial=0), x["graph"]["elabel"].max(initial=0))) + 1))
        samples = []
        for x in loaded:
            p, g = x["pattern"], x["graph"]
            samples.append({"id": x["id"],
                            "pattern": cls._with_rev(p["src"], p["dst"], p["vlabel"], p["elabel"], shape["p_edges"], shape["n_elabels"]),
                            "graph": cls._with_rev(g["src"], g["dst"], g["vlabel"], g["elabel"], shape["g_edges"], shape["n_elabels"]),
                            "counts": int(x["counts"]), "subisomorphisms": np.asarray(x["subisomorphisms"], np.int64).reshape(-1, p["num_nodes"])})
        return cls(samples, shape)

    @staticmethod
    def _with_rev(u, v, vl, el, max_ne, max_nel):
        e = len(u)
        return {"src": np.concatenate([u, v]), "dst": np.concatenate([v, u]), "vlabel": vl,
                "elabel": np.concatenate([el, el + max_nel]), "eid": np.concatenate([np.arange(e), max_ne + np.arange(e)]),
                "rev": np.concatenate([np.zeros(e, bool), np.ones(e, bool)]), "num_nodes": len(vl)}

    def __len__(self):
        return len(self.samples)

    def subset(self, indices):
        return PairDataset([self.samples[i] for i in indices], self.shape)

    def model_config(self, hid_dim=64, layers=3, rep_net="DMPNN", **kw):
        s = self.shape
        cfg = dict(max_ngv=s["g_nodes"], max_ngvl=s["n_vlabels"], max_nge=2 * s["g_edges"], max_ngel=2 * s["n_elabels"],
                   max_npv=s["p_nodes"], max_npvl=s["n_vlabels"], max_npe=2 * s["p_edges"], max_npel=2 * s["n_elabels"],
                   base=2, hid_dim=hid_dim, share_rep_net=True, rep_residual=True, enc_net="Multihot",
                   emb_net="Orthogonal", filter_net="ScalarFilter", rep_net=rep_net, rep_num_graph_layers=layers,
                   rep_num_pattern_layers=layers, rep_dmpnn_batch_norm=False, rep_act_func="relu",
                   pred_net="SumPredictNet", pred_hid_dim=hid_dim, node_pred=True, edge_pred=True)
        cfg.update(kw)
        return cfg

    def to_files(self, root, shared_graph=False):
        """Write the forward edges of every sample in the reference's directory layout
        (``dataio.save_pairs``); sample ``i`` becomes pattern ``P_<i // 10>`` ... only when the
        samples carry no ids of their own: synthetic sets get ``P_i`` / ``G_i``."""
        from . import dataio
        out = []
        for i, x in enumerate(self.samples):
            fwd = lambda g: {"num_nodes": g["num_nodes"], "src": g["src"][:len(g["src"]) // 2], "dst": g["dst"][:len(g["dst"]) // 2],
                             "vlabel": g["vlabel"], "elabel": g["elabel"][:len(g["elabel"]) // 2]}
            pid, gid = (x["id"].split("-", 1) if "id" in x else ("P_%d" % i, "G_%d" % i))
            out.append({"pattern_id": pid, "graph_id": gid, "pattern": fwd(x["pattern"]), "graph": fwd(x["graph"]),
                        "counts": x["counts"], "subisomorphisms": x["subisomorphisms"]})
        dataio.save_pairs(root, out, shared_graph)

    def batchify(self, indices, device, return_weights=None):
        """``GraphAdjDataset.batchify`` (dataset.py:1604-1636) + ``.to(device)`` (train.py:606-607):
        concatenated local arrays are uploaded once, batching itself happens on the device.  Returns
        ``(pattern, graph, counts [B, 1], (node_weights, edge_weights))``; the weights (pre-padded
        int64 ``[B, max]``, or None) are computed on the device when ``return_weights`` names them."""
        meta, tensors = self.batch_arrays(indices, device)
        out = self.graphs_from_arrays(meta, tensors)
        counts = tensors[-1]
        weights = (None, None)
        if return_weights:
            subs = [self.samples[i]["subisomorphisms"].reshape(-1) for i in indices]
            ptr_host = np.concatenate([[0], np.cumsum([len(x) for x in subs])]).astype(np.int64)
            hint = sum(self.samples[i]["counts"] * len(self.samples[i]["pattern"]["src"]) for i in indices)
            weights = subiso_weights(out[0], out[1], torch.from_numpy(np.concatenate(subs)).to(device),
                                     torch.from_numpy(ptr_host).to(device), return_weights, work_hint=int(hint))
        return out[0], out[1], counts, weights

    ARRAYS_PER_GRAPH = 9    # src, dst, num_nodes, num_edges, node id, node label, edge id, edge label, is_reversed

    def pad_buckets(self, batch_size, levels=4):
        """``pad`` for ``batch_arrays``: the per-graph maxima of the dataset per side and the number of capacity levels.  A
        padded batch holds ``batch_size`` real + ``batch_size`` inert pairs and its four totals (nodes / edges of both sides)
        are those of level k of ``levels`` -- k / levels of what ``batch_size`` largest graphs would take, the smallest level
        this batch fits -- so a ragged dataset meets at most ``levels`` batch shapes and a recorded step (``GraphedTrainStep``)
        replays for every batch."""
        out = {"batch": int(batch_size), "levels": int(levels)}
        for key in ("pattern", "graph"):
            out[key] = (max(2, max(int(x[key]["num_nodes"]) for x in self.samples)), max(1, max(len(x[key]["src"]) for x in self.samples)))
        return out

    def batch_arrays(self, indices, device, pad=None):
        """The uploaded half of ``batchify``: ``(meta, tensors)`` with ``tensors`` = the pattern batch's nine arrays
        (local endpoints, sizes, ids, labels, reversed flags), the graph batch's nine and ``counts [B, 1]``, and ``meta``
        = per side ``(total nodes, total edges, largest graph's nodes, largest graph's edges)`` as host ints.  Everything
        ``graphs_from_arrays`` then does happens on the device without a host sync (``dp.StepGraph`` records it).
        ``pad`` (``pad_buckets``): the batch is extended to ``2 pad["batch"]`` pairs by INERT pairs -- pairs of one-label graphs
        (pattern label 0, target label 1: the filter gates wipe the inert target rows) whose sizes take the four totals up to a capacity level, and whose weight in the loss is zero (a last tensor,
        ``weights [2 B, 1]``, is appended).  Pairs are
        independent in the model (block-diagonal batches, no BatchNorm), so the real pairs' predictions are unchanged and
        an inert pair, whose prediction carries no loss, adds exact zeros to every gradient: a ragged dataset then meets a
        handful of batch shapes instead of one per batch."""
        meta, tensors = [], []
        extra, level = 0, 0
        if pad is not None:
            B, K = pad["batch"], pad["levels"]
            if len(indices) > B:
                raise ValueError("batch_arrays(pad=...): more pairs than the padding was laid out for")
            extra = 2 * B - len(indices)                          # inert pairs: the batch always holds 2 B pairs
            tot = {key: (sum(int(self.samples[i][key]["num_nodes"]) for i in indices), sum(len(self.samples[i][key]["src"]) for i in indices))
                   for key in ("pattern", "graph")}
            # level k: k / K of what B largest graphs would take, + one node and one edge for each of B inert graphs (an inert
            # graph without an edge would put 1 / 0 into the heads' length features, pred.py:93-96)
            caps = lambda key, k: (-(-B * pad[key][0] * k // K) + B, -(-B * pad[key][1] * k // K) + B)
            fits = lambda k: all(tot[key][0] + extra <= caps(key, k)[0] and tot[key][1] + extra <= caps(key, k)[1] for key in tot)
            level = next((k for k in range(1, K + 1) if fits(k)), None)
            if level is None:
                raise ValueError("batch_arrays(pad=...): the batch does not fit %d graphs of the dataset's largest size" % B)
        for key in ("pattern", "graph"):
            gs = [self.samples[i][key] for i in indices]
            if pad is not None:
                cn, ce = caps(key, level)
                # an inert TARGET graph carries a node / edge label its (label-0) inert pattern does not use: the filter gates
                # (ScalarFilter: a target row is kept when its label occurs in the pattern) zero every one of its rows, so the
                # gated kernels leave them out and a gate capacity calibrated on the real pairs holds for the padded batch too
                # (label 0 on both sides kept every inert edge: ~3 x the real rows as live work, and with ``gate_compact`` every
                # padded batch overflowed its capacity -- ADVICE r5).  One-label vocabularies have no such label: 0 there.
                lab = (0, 0) if key == "pattern" else (1 if self.shape.get("n_vlabels", 1) > 1 else 0, 1 if self.shape.get("n_elabels", 1) > 1 else 0)
                gs = gs + self._inert_graphs(tot[key], cn, ce, extra, pad[key], vlabel=lab[0], elabel=lab[1])
            cat = lambda k, dt=torch.int64: torch.from_numpy(np.concatenate([g[k] for g in gs])).to(dt).to(device)
            nn_ = np.array([g["num_nodes"] for g in gs], np.int64)
            ne_ = np.array([len(g["src"]) for g in gs], np.int64)
            nid = torch.from_numpy(np.concatenate([np.arange(n) for n in nn_])).to(device)
            tensors += [cat("src"), cat("dst"), torch.from_numpy(nn_).to(device), torch.from_numpy(ne_).to(device), nid,
                        cat("vlabel"), cat("eid"), cat("elabel"), cat("rev", torch.bool)]
            if pad is not None:     # the dataset's per-graph maxima: one signature per capacity level, whatever this batch holds
                meta.append((int(nn_.sum()), int(ne_.sum()), int(pad[key][0]), int(pad[key][1])))
            else:
                meta.append((int(nn_.sum()), int(ne_.sum()), int(nn_.max(initial=0)), int(ne_.max(initial=0))))
        counts = torch.tensor([self.samples[i]["counts"] for i in indices] + [0] * extra, dtype=torch.float32, device=device)
        if pad is None:
            return tuple(meta), tensors + [counts.unsqueeze(-1)]
        weights = torch.tensor([1.0] * len(indices) + [0.0] * extra, dtype=torch.float32, device=device)
        return tuple(meta), tensors + [counts.unsqueeze(-1), weights.unsqueeze(-1)]

    @staticmethod
    def _inert_graphs(tot, cap_n, cap_e, extra, largest=None, vlabel=0, elabel=0):
        """``extra`` one-label graphs (node label ``vlabel``, edge label ``elabel``) that take a side's node / edge totals ``tot`` up to ``(cap_n, cap_e)``: nodes dealt evenly
        (every inert graph has at least one), edges too (at least one each; a ring over the graph's nodes: valid endpoints,
        parallel edges when the ring wraps).  ``largest``: the dataset's largest graph (nodes, edges) no inert graph may exceed."""
        n_pad, e_pad = cap_n - tot[0], cap_e - tot[1]
        if n_pad < extra or e_pad < extra:
            raise ValueError("batch_arrays(pad=...): the batch does not fit its capacity level")
        out = []
        for j in range(extra):
            n = n_pad // extra + (1 if j < n_pad % extra else 0)
            e = e_pad // extra + (1 if j < e_pad % extra else 0)
            if largest is not None and (n > largest[0] or e > largest[1]):
                raise ValueError("batch_arrays(pad=...): an inert graph would exceed the dataset's largest graph (a batch of empty graphs?)")
            a = np.arange(e, dtype=np.int64) % n
            out.append({"src": a, "dst": (a + 1) % n, "vlabel": np.full(n, vlabel, np.int64), "elabel": np.full(e, elabel, np.int64),
                        "eid": np.arange(e, dtype=np.int64), "rev": np.zeros(e, bool), "num_nodes": n})
        return out

    @classmethod
    def graphs_from_arrays(cls, meta, tensors):
        """``(pattern, graph)`` BatchedGraphs from ``batch_arrays``' output (device collate)."""
        from .collate import collate_device_many
        jobs = []
        for side, (n, e, max_n, max_e) in enumerate(meta):
            src, dst, nn_, ne_, nid, vlabel, eid, elabel, rev = tensors[side * cls.ARRAYS_PER_GRAPH:(side + 1) * cls.ARRAYS_PER_GRAPH]
            jobs.append(dict(local_src=src, local_dst=dst, num_nodes=nn_, num_edges=ne_, total_nodes=n, total_edges=e,
                             ndata={"id": nid, "label": vlabel}, edata={"id": eid, "label": elabel, "is_reversed": rev},
                             max_nodes=max_n, max_edges=max_e))
        return collate_device_many(jobs)                         # both sides in one pair of launches


class SyntheticPairs(PairDataset):
    """Directed ER (pattern, graph) pairs with uniform labels and exact counts."""

    def __init__(self, num_pairs, p_nodes, p_edges, g_nodes, g_edges, n_vlabels, n_elabels, seed=0):
        rng = np.random.default_rng(seed)
        self.shape = dict(p_nodes=p_nodes, p_edges=p_edges, g_nodes=g_nodes, g_edges=g_edges,
                          n_vlabels=n_vlabels, n_elabels=n_elabels)
        self.samples = []
        for _ in range(num_pairs):
            pu, pv = _er_edges(p_nodes, p_edges, rng)
            gu, gv = _er_edges(g_nodes, g_edges, rng)
            pvl, gvl = rng.integers(0, n_vlabels, p_nodes), rng.integers(0, n_vlabels, g_nodes)
            pel, gel = rng.integers(0, n_elabels, p_edges), rng.integers(0, n_elabels, g_edges)
            sub = enumerate_subisomorphisms(pu, pv, pvl, pel, gu, gv, gvl, gel)
            self.samples.append({"pattern": self._with_rev(pu, pv, pvl, pel, p_edges, n_elabels),
                                 "graph": self._with_rev(gu, gv, gvl, gel, g_edges, n_elabels),
                                 "counts": len(sub), "subisomorphisms": sub})


class SmallLikePairs(PairDataset):
    """Ragged (pattern, graph) pairs in the size ranges of the reference's "small" setting (SubgraphCountingMatching/
    README.md:72-94: patterns of at most 8 nodes / 8 edges, graphs of at most 64 nodes / 256 edges, at most 16 labels) with
    exact counts; the published dataset itself is a download that is not available offline.  A pattern is a random
    spanning tree plus random extra edges (connected: an isolated pattern node would multiply the count by the number of
    free graph nodes); a graph is directed ER noise with uniform labels into which 0 .. 12 copies of its pattern are
    planted (node labels overwritten, the pattern's edges added), so that counts are mostly small positive numbers instead
    of zeros and rare explosions.  Counts are EXACT whatever the planting did: they come from the native search over a
    pool of host threads (``count_subisomorphisms_batch``); the enumerations themselves are not kept (count loss only)."""

    def __init__(self, num_pairs, seed=0, threads=0):
        rng = np.random.default_rng(seed)
        self.shape = dict(p_nodes=8, p_edges=8, g_nodes=64, g_edges=256, n_vlabels=16, n_elabels=16)
        raw = []
        for _ in range(num_pairs):
            pn = int(rng.choice([3, 4, 8]))
            pe = int({3: rng.choice([2, 4]), 4: rng.choice([4, 8]), 8: 8}[pn])
            gn = int(rng.choice([8, 16, 32, 64]))
            ge = int(min(gn * rng.choice([1, 2, 4]), 256))
            nl = int(rng.choice([4, 8, 16]))
            # pattern: a spanning tree in random orientation, then distinct extra ordered pairs
            perm = rng.permutation(pn)
            pairs = set()
            for k in range(1, pn):
                a, b = int(perm[k]), int(perm[rng.integers(0, k)])
                pairs.add((a, b) if rng.random() < 0.5 else (b, a))
            while len(pairs) < pe:
                a, b = int(rng.integers(0, pn)), int(rng.integers(0, pn))
                if a != b:
                    pairs.add((a, b))
            pu, pv = (np.array(x, np.int64) for x in zip(*sorted(pairs)))
            pvl, pel = rng.integers(0, nl, pn), rng.integers(0, nl, pe)
            gvl = rng.integers(0, nl, gn)
            copies = int(rng.integers(0, min(12, gn // pn) + 1))
            edges = {}
            for _c in range(copies):
                where = rng.choice(gn, size=pn, replace=False)
                gvl[where] = pvl
                for u, v, l in zip(pu, pv, pel):
                    edges[(int(where[u]), int(where[v]))] = int(l)
            while len(edges) < ge:                            # ER noise up to the edge budget
                a, b = int(rng.integers(0, gn)), int(rng.integers(0, gn))
                if a != b and (a, b) not in edges:
                    edges[(a, b)] = int(rng.integers(0, nl))
            keys = sorted(edges)[:256]
            gu, gv = (np.array(x, np.int64) for x in zip(*keys))
            gel = np.array([edges[k] for k in keys], np.int64)
            raw.append((pu, pv, pvl, pel, gu, gv, gvl, gel))
        counts = count_subisomorphisms_batch(raw, threads)
        self.samples = []
        for (pu, pv, pvl, pel, gu, gv, gvl, gel), c in zip(raw, counts):
            self.samples.append({"pattern": self._with_rev(pu, pv, pvl, pel, 8, 16), "graph": self._with_rev(gu, gv, gvl, gel, 256, 16),
                                 "counts": int(c), "subisomorphisms": np.zeros((0, len(pvl)), np.int64)})


_CRIT = {"MAE": F.l1_loss, "MSE": F.mse_loss, "SMSE": F.smooth_l1_loss}
_CRIT_KIND = {"MSE": 0, "MAE": 1, "SMSE": 2}
USE_FUSED_COUNT_LOSS = True


class _CountLossHIP(torch.autograd.Function):
    """``bp_crit(leaky_relu(pred, neg_slope), target)`` (mean) and its gradient seed in ONE launch (``dmp_count_loss``)."""

    @staticmethod
    def forward(ctx, pred, target, kind, neg_slope):
        from . import _lib
        lib = _lib.load()
        p, t = pred.detach().reshape(-1).contiguous(), target.reshape(-1).contiguous()
        _lib.require_gpu(p, t)
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        dpred = torch.empty_like(p)
        with _lib.timed("count_loss[n=%d]", (p.numel(),), 12 * p.numel()):
            _lib.check(lib.dmp_count_loss(_lib.ptr(p), _lib.ptr(t), p.numel(), int(kind), float(neg_slope), _lib.ptr(loss), _lib.ptr(dpred),
                                          _lib.stream_ptr()), "dmp_count_loss")
        ctx.dpred, ctx.shape = dpred, pred.shape
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        return (ctx.dpred * g).view(ctx.shape), None, None, None


def count_loss(pred, target, bp_loss="MSE", neg_slope=1.0):
    """The count loss of train.py:624-628, ``bp_crit(F.leaky_relu(pred, neg_slope), target)`` with reduction 'mean'
    (``neg_slope`` 1: the plain criterion).  On the GPU, for fp32 tensors of equal size, the criterion, its mean and the
    gradient seed are ONE launch (five as tensor ops); anything else goes through the tensor ops."""
    if (USE_FUSED_COUNT_LOSS and bp_loss in _CRIT_KIND and pred.is_cuda and pred.dtype == torch.float32 and torch.is_tensor(target)
            and target.is_cuda and target.dtype == torch.float32 and not target.requires_grad and target.numel() == pred.numel()
            and 0 < pred.numel() <= (1 << 22) and 0.0 <= float(neg_slope) <= 1.0
            and pred.reshape(-1).shape == target.reshape(-1).shape and (pred.dim() == target.dim() or pred.dim() == 1 or target.dim() == 1)):
        return _CountLossHIP.apply(pred, target, _CRIT_KIND[bp_loss], float(neg_slope))
    act = pred if float(neg_slope) == 1.0 else F.leaky_relu(pred, neg_slope)
    return _CRIT[bp_loss](act, target)
SCHEDULE_CYCLES = 2     # utils/anneal.py:7, utils/cyclical.py:7 (NUM_CYCLES, what train.py:510-562 passes)


def scheduled_value(spec, step, total_steps, cycles=SCHEDULE_CYCLES):
    """A loss weight / slope at optimisation step ``step`` from the reference's specification (config.py:480-515,
    train.py:499-600): a number, ``"anneal_<shape>$a$b"`` (utils/anneal.py: ramp from a to b over the first half of each of
    ``cycles`` periods, then hold b) or ``"cyclical_<shape>$a$b"`` (utils/cyclical.py: a -> b -> a every period); shape is
    ``linear``, ``cosine`` or ``none`` / ``constant`` (always b).  Past ``total_steps``: b."""
    if isinstance(spec, (int, float)):
        return float(spec)
    kind, a, b = spec.rsplit("$", 2)
    a, b = float(a), float(b)
    mode, _, shape = kind.partition("_")
    if mode not in ("anneal", "cyclical"):
        raise ValueError(spec)
    if step > total_steps or shape in ("", "none", "constant"):
        return b
    phase = (cycles * step / max(1, total_steps)) % 1.0          # position inside the current period
    if shape == "linear":
        ramp = lambda t: t
    elif shape == "cosine":
        ramp = lambda t: (1.0 - math.cos(math.pi * t)) / 2.0
    else:
        raise NotImplementedError(shape)
    if mode == "anneal":
        return a + (b - a) * ramp(2.0 * phase) if phase < 0.5 else b
    if shape == "linear":                                         # there and back
        return a + (b - a) * 2.0 * phase if phase < 0.5 else b + (a - b) * (2.0 * phase - 1.0)
    return a + (b - a) * ramp(2.0 * phase)


def lr_factor(name, step, warmup, total, cycles=SCHEDULE_CYCLES, floor=1e-3):
    """Learning-rate multiplier of the reference's schedulers (utils/scheduler.py:12-178) at scheduler step ``step``:
    linear warm-up over ``warmup`` steps where the name says so, then constant / linear / cosine decay towards ``floor``
    over ``total`` steps, ``cycles`` periods, optionally restarting every period."""
    has_warmup = "warmup" in name
    if has_warmup and step < warmup:
        return step / max(1.0, float(warmup))
    if name.startswith("constant"):
        return 1.0
    begin = warmup if has_warmup else 0
    progress = (step - begin) / float(max(1, total - begin))
    if name.startswith("linear"):
        if name.endswith("restart"):
            return floor if progress >= 1.0 else max(floor, 1.0 - (cycles * progress) % 1.0)
        span = total - begin if has_warmup else total            # scheduler.py:49-52 divides by the full length without warm-up
        return max(floor, (total - step) / float(max(1, span)))
    if name.startswith("cosine"):
        if name.endswith("restart"):
            return floor if progress >= 1.0 else max(floor, 0.5 * (1.0 + math.cos(math.pi * ((cycles * progress) % 1.0))))
        return max(floor, 0.5 * (1.0 + math.cos(math.pi * cycles * 2.0 * progress)))
    raise NotImplementedError(name)


class RunSchedule:
    """Everything of a reference run configuration that varies with the optimisation step (train.py:1233-1253 for the
    learning rate, :499-600 for the loss terms), from the ``config`` dictionary of ``config.py`` / ``config.json``."""

    def __init__(self, config, num_train, lr=None):
        self.config = config
        bsz, epochs = config["train_batch_size"], config["train_epochs"]
        self.base_lr = config["lr"] if lr is None else lr
        self.warmup = int(num_train / bsz * 0.5 * min(epochs * 0.06, config.get("early_stop_rounds", 10)))
        self.total = int(num_train / bsz * epochs)
        self.floor = max(1e-3, config.get("weight_decay", 0.0))
        if self.floor > 1e-8:
            self.total -= self.warmup
        self.cycles = max(1, self.total / 20000)
        self.name = config.get("scheduler", "constant")
        self.epoch_steps = int(math.ceil(num_train / bsz))
        self.sched_step = 0

    def lr(self):
        return self.base_lr * lr_factor(self.name, self.sched_step, self.warmup, self.total, self.cycles, self.floor)

    def loss_terms(self, epoch, batch_id):
        step = epoch * self.epoch_steps + batch_id
        total = self.config["train_epochs"] * self.epoch_steps
        c = self.config
        return {k: scheduled_value(c.get(k, d), step, total) for k, d in
                (("neg_pred_slp", 0.0), ("match_loss_w", 0.0), ("match_reg_w", 0.0), ("rep_reg_w", 0.0))}


def _match_terms(crit, pred, weights, mask, pred_c, neg_slp):
    """train.py:627-649: matching loss and regulariser of one of pred_v / pred_e ([B, max_len])."""
    with torch.no_grad():
        weights = weights.float().masked_fill_(~mask, 0)
        pred.masked_fill_(~mask, 0)  # in place on the model output, as the reference does
    loss = crit(F.leaky_relu(pred, neg_slp), weights) * pred.size(1)
    reg = crit(F.relu(pred - pred_c), torch.zeros_like(pred)) * pred.size(1)
    return loss, reg


def train_epoch(model, optimizer, dataset, batch_size, device, sync=None, bp_loss="MSE", eval_metric="MAE",
                neg_slp=0.0, rep_reg_w=0.0, match_loss_w=0.0, match_reg_w=0.0, max_grad_norm=8.0, order=None,
                schedule=None, epoch=0, match_weights=("node", "edge"), trace=None, graph=None):
    """One pass over ``dataset`` (train.py:449-844): count loss, optional representation regulariser
    and, with ``match_loss_w`` / ``match_reg_w`` and a model built with ``pred_return_weights``, the
    node / edge matching losses against the batch's subisomorphism weights.  With a ``RunSchedule`` the four loss
    coefficients and the learning rate follow the run configuration step by step, as train.py:499-600,686 do.  Returns
    ``{"bp_loss", "eval_metric"}`` (sample-weighted means, one host sync at the end).  ``trace``: a list that receives
    one ``(loss, eval_metric)`` pair of device scalars per step (what the reference writes to its SummaryWriter).
    ``graph``: a ``GraphedTrainStep`` (see there) -- batches whose shape it has recorded are one HIP graph replay each."""
    model.train()
    sync = sync or FlatGradSync(model)
    order = np.arange(len(dataset)) if order is None else np.asarray(order)
    tot_loss = torch.zeros((), device=device)
    tot_eval = torch.zeros((), device=device)
    cnt = 0
    if isinstance(match_weights, str):
        match_weights = tuple(w for w in match_weights.split(",") if w in ("node", "edge"))
    for b_id, i in enumerate(range(0, len(order), batch_size)):
        idx = order[i:i + batch_size]
        if schedule is not None:
            terms = schedule.loss_terms(epoch, b_id)
            neg_slp, match_loss_w, match_reg_w, rep_reg_w = (terms["neg_pred_slp"], terms["match_loss_w"], terms["match_reg_w"],
                                                             terms["rep_reg_w"])
            for group in optimizer.param_groups:
                group["lr"] = schedule.lr()
        want = (tuple(match_weights) or None) if (match_loss_w > 0 or match_reg_w > 0) else None
        if graph and want is None:
            loss, ev = graph(dataset, idx, device, neg_slp, rep_reg_w)
            if schedule is not None:
                schedule.sched_step += 1
            with torch.no_grad():
                if trace is not None:
                    trace.append((loss.clone(), ev.clone()))
                tot_loss += loss * len(idx)
                tot_eval += ev * len(idx)
            cnt += len(idx)
            continue
        pattern, graph_b, counts, (node_w, edge_w) = dataset.batchify(idx, device, return_weights=want)
        sync.detach_grads()
        # the representation regulariser reads the last layer's edge rows: with it the layer forms them itself (the deferred
        # form would run the layer's forward a second time on first access)
        lazy, model.lazy_edge_rep = getattr(model, "lazy_edge_rep", True), not (rep_reg_w > 0)
        try:
            out = model(pattern, graph_b)
        finally:
            model.lazy_edge_rep = lazy
        pred = out["pred_c"]
        loss = count_loss(pred, counts, bp_loss, neg_slp) if pred.shape == counts.shape else _CRIT[bp_loss](F.leaky_relu(pred, neg_slp), counts)
        for w, pk, mk in ((node_w, "pred_v", "g_v_mask"), (edge_w, "pred_e", "g_e_mask")):
            if w is not None and out[pk] is not None:
                m_loss, m_reg = _match_terms(_CRIT[bp_loss], out[pk], w, out[mk], pred, neg_slp)
                loss = loss + match_loss_w * m_loss + match_reg_w * m_reg
        if rep_reg_w > 0:  # train.py:651-659 (bp_crit with slope 1: the plain criterion against zeros)
            reg = sum(_CRIT[bp_loss](out[k], torch.zeros_like(out[k])) * out[k].size(1)
                      for k in ("p_v_rep", "p_e_rep", "g_v_rep", "g_e_rep") if out[k] is not None)
            loss = loss + rep_reg_w * reg
        loss.backward()
        sync.pack()
        sync.sync()
        if max_grad_norm > 0:
            torch.nn.utils.clip_grad_norm_(sync.params, max_grad_norm)
        optimizer.step()
        if schedule is not None:
            schedule.sched_step += 1
        with torch.no_grad():
            if trace is not None:
                trace.append((loss.detach(), _CRIT[eval_metric](F.relu(pred), counts)))
            tot_loss += loss.detach() * len(idx)
            tot_eval += _CRIT[eval_metric](F.relu(pred), counts) * len(idx)
        cnt += len(idx)
    return {"bp_loss": float(tot_loss / max(cnt, 1)), "eval_metric": float(tot_eval / max(cnt, 1))}


class GraphedTrainStep:
    """The count-loss training step of ``train_epoch`` (collate, forward, loss, backward, gradient pack, clipping,
    AdamW) as ONE HIP graph per batch shape (``dp.StepGraph``): at the reference's own batch size (64 pairs of small
    graphs, config.py) a step is ~150 launches of microseconds each behind ~3 ms of Python, and a replay is a single
    ``hipGraphLaunch``.  Needs one rank (no collective is recorded), ``FlatAdamW(capturable=True)`` and a dataset with
    ``batch_arrays`` / ``graphs_from_arrays``.  The negative slope of the count loss and the representation
    regulariser's weight are device scalars written before every replay, the learning rate goes through
    ``optimizer.sync_hyper()``.  Batches of a shape seen for the first time, and shapes beyond ``max_shapes``
    recordings, run eagerly -- through the same function, so both ways compute the same step.  Run the training loop
    under ``with step.steps.on_stream():`` when it also copies sizeable tensors between host and device (checkpoints,
    predictions): see ``dp.StepGraph`` (``fit(graph=True)`` does)."""

    def __init__(self, model, optimizer, sync, bp_loss="MSE", eval_metric="MAE", max_grad_norm=8.0, with_rep_reg=False,
                 max_shapes=4, pad="auto"):
        """``pad``: ragged datasets (every batch another pair of (N, E) totals: the reference's bucket-sorted batches,
        utils/sampler.py:10-84, train.py:1283-1290) never repeat a shape, so nothing would replay.  ``"auto"`` / True:
        batches are padded with inert pairs to one of ``max_shapes`` capacity levels (``PairDataset.batch_arrays(pad=...)``;
        "auto": only when the dataset's graphs differ in size); the inert pairs carry weight 0 in the loss and the metric.
        Not with the representation regulariser (it reads every row) or BatchNorm layers (their statistics would see the
        inert rows): such runs keep the exact shapes."""
        from .dp import StepGraph
        if getattr(sync, "world", 1) != 1:
            raise ValueError("GraphedTrainStep records single-rank steps only")
        if not getattr(optimizer, "capturable", False):
            raise ValueError("GraphedTrainStep needs FlatAdamW(capturable=True)")
        self.model, self.optimizer, self.sync = model, optimizer, sync
        self.bp, self.ev, self.max_grad_norm, self.with_rep_reg = _CRIT[bp_loss], _CRIT[eval_metric], max_grad_norm, with_rep_reg
        self.hyper = None
        self._hyper_host = None
        self.dataset_cls = None
        self.pad, self._pad_spec = pad, None
        if pad and (with_rep_reg or any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in model.modules())):
            self.pad = None
        self.max_shapes = int(max_shapes)
        self.steps = StepGraph(self._step, optimizer=optimizer, max_shapes=max_shapes)

    def _step(self, meta, *tensors):
        pattern, graph = self.dataset_cls.graphs_from_arrays(meta, tensors)
        n = 2 * self.dataset_cls.ARRAYS_PER_GRAPH
        counts, hyper = tensors[n], tensors[-1]
        weights = tensors[n + 1] if len(tensors) == n + 3 else None        # a padded batch: 1 for the real pairs, 0 for the inert ones
        self.sync.detach_grads()
        lazy, self.model.lazy_edge_rep = getattr(self.model, "lazy_edge_rep", True), not self.with_rep_reg   # see train_epoch
        try:
            out = self.model(pattern, graph)
        finally:
            self.model.lazy_edge_rep = lazy
        pred = out["pred_c"]
        act = torch.where(pred > 0, pred, pred * hyper[0])                        # leaky_relu with the slope on the device
        if weights is not None:      # the mean over the REAL pairs (train.py:463-480's batch mean)
            loss = (self.bp(act, counts, reduction="none") * weights).sum() / weights.sum()
        else:
            loss = self.bp(act, counts)
        if self.with_rep_reg:
            reg = sum(self.bp(out[k], torch.zeros_like(out[k])) * out[k].size(1)
                      for k in ("p_v_rep", "p_e_rep", "g_v_rep", "g_e_rep") if out[k] is not None)
            loss = loss + hyper[1] * reg
        loss.backward()
        self.sync.pack()
        self.sync.sync()
        if self.max_grad_norm > 0:
            torch.nn.utils.clip_grad_norm_(self.sync.params, self.max_grad_norm)
        self.optimizer.step()
        with torch.no_grad():
            if weights is not None:
                return loss.detach(), (self.ev(F.relu(pred), counts, reduction="none") * weights).sum() / weights.sum()
            return loss.detach(), self.ev(F.relu(pred), counts)

    def _padding(self, dataset, indices):
        """The dataset's padding layout (``pad_buckets``), decided at the first batch: the batch size is that batch's."""
        if self.pad and self._pad_spec is None:
            ragged = len({(int(x[k]["num_nodes"]), len(x[k]["src"])) for x in dataset.samples for k in ("pattern", "graph")}) > 2
            self._pad_spec = dataset.pad_buckets(len(indices), levels=self.max_shapes) if (self.pad is True or ragged) else False
        return self._pad_spec or None

    def __call__(self, dataset, indices, device, neg_slp=0.0, rep_reg_w=0.0):
        if rep_reg_w > 0 and not self.with_rep_reg:
            raise ValueError("GraphedTrainStep(with_rep_reg=True) to train with the representation regulariser")
        self.dataset_cls = type(dataset)
        pad = self._padding(dataset, indices)
        if pad is not None and len(indices) > pad["batch"]:
            pad = None
        meta, tensors = dataset.batch_arrays(indices, device, pad=pad) if pad is not None else dataset.batch_arrays(indices, device)
        if self.hyper is None:
            self.hyper = torch.zeros(2, dtype=torch.float32, device=device)
        if self._hyper_host != (float(neg_slp), float(rep_reg_w)):
            self._hyper_host = (float(neg_slp), float(rep_reg_w))
            self.hyper[0:1].fill_(float(neg_slp))
            self.hyper[1:2].fill_(float(rep_reg_w))
        return self.steps(meta, *tensors, self.hyper)


@torch.no_grad()
def evaluate_epoch(model, dataset, batch_size, device, eval_metric="MAE"):
    """train.py:847-1061 reduced to the count metrics: MAE / MSE of ``relu(pred_c)`` and the
    predictions themselves."""
    model.eval()
    preds, targets = [], []
    # no optimizer step here that could drop a batch which does not fit a gate capacity: every edge row as it stands
    cap, model.gate_capacity = getattr(model, "gate_capacity", None), None
    try:
        for i in range(0, len(dataset), batch_size):
            idx = np.arange(i, min(i + batch_size, len(dataset)))
            pattern, graph, counts, _ = dataset.batchify(idx, device)
            preds.append(F.relu(model(pattern, graph)["pred_c"]))
            targets.append(counts)
    finally:
        model.gate_capacity = cap
    pred, target = torch.cat(preds), torch.cat(targets)
    return {"MAE": float(F.l1_loss(pred, target)), "MSE": float(F.mse_loss(pred, target)),
            "eval_metric": float(_CRIT[eval_metric](pred, target)), "pred": pred.view(-1).cpu(), "counts": target.view(-1).cpu()}


def evaluate_run(save_dir, datasets, device, batch_size=None, eval_metric=None, epoch=None, stamp=""):
    """``evaluate.py`` (lines 60-245) for a run directory written by ``fit`` (or by the reference's ``train.py``):
    ``config.json`` -> ``build_model``; the dev-best epoch of ``log.txt`` (``utils/log.py:60-76``) -> its ``epoch%d.pt``
    (strict load); every split of ``datasets`` (name -> PairDataset) evaluated and written to
    ``eval_<split>_results_<stamp>.json`` with the reference's result layout (``data`` / ``prediction`` / ``error``);
    one "best" line per split appended to ``log.txt``.  Returns ``{split: {"MAE", "MSE", "eval_metric"}}``."""
    import json
    from . import dataio
    from .basemodel import build_model
    config = dataio.load_config(os.path.join(save_dir, "config.json"))
    metric = eval_metric or config.get("eval_metric", "MAE")
    if epoch is None:
        best = dataio.get_best_epochs(os.path.join(save_dir, "log.txt"))
        epoch = best["eval-" + metric]["dev"][0]
    model = build_model(config).to(device)
    model.load_state_dict(torch.load(dataio.checkpoint_path(save_dir, epoch), map_location=device), strict=True)
    out = {}
    with open(os.path.join(save_dir, "log.txt"), "a") as log:
        for split, ds in datasets.items():
            res = evaluate_epoch(model, ds, batch_size or config.get("eval_batch_size", 64), device, eval_metric=metric)
            out[split] = {k: res[k] for k in ("MAE", "MSE", "eval_metric")}
            ids = [x.get("id", str(i)) for i, x in enumerate(ds.samples)]
            with open(os.path.join(save_dir, "eval_%s_results_%s.json" % (split, stamp)), "w") as f:
                json.dump({"data": {"id": ids, "counts": res["counts"].tolist()}, "prediction": {"pred_c": res["pred"].tolist()},
                           "error": {"MAE": res["MAE"], "MSE": res["MSE"]}}, f)
            log.write(dataio.best_line(split, epoch, epoch, **{"eval-" + metric: "%.3f" % res["eval_metric"]}) + "\n")
    return out


def validate_samples(samples):
    """Host-side check of a dataset, once: every edge endpoint of every pattern / graph inside ``[0, number_of_nodes)``
    (the device index builds only flag such edges in a status word that the training loop does not read back) and the
    reversed-edge flags, when present, one per edge.  Raises ``ValueError`` naming the first offending sample."""
    samples = getattr(samples, "samples", samples)
    for i, x in enumerate(samples):
        for key in ("pattern", "graph"):
            g = x[key]
            if isinstance(g, dict):                          # PairDataset's stored form: numpy arrays
                n, src, dst, rev = int(g["num_nodes"]), np.asarray(g["src"]), np.asarray(g["dst"]), g.get("rev")
            else:                                            # a graph object (BatchedGraph / DGL surface)
                n = int(g.number_of_nodes())
                src, dst = (t.detach().cpu().numpy() for t in g.all_edges(form="uv", order="eid"))
                rev = g.edata.get("is_reversed") if hasattr(g, "edata") else None
            if len(src) != len(dst):
                raise ValueError("sample %d (%s): %s has %d sources for %d destinations" % (i, x.get("id", "?"), key, len(src), len(dst)))
            if len(src) and (min(src.min(), dst.min()) < 0 or max(src.max(), dst.max()) >= n):
                raise ValueError("sample %d (%s): %s has an edge endpoint outside [0, %d)" % (i, x.get("id", "?"), key, n))
            if rev is not None and len(rev) != len(src):
                raise ValueError("sample %d (%s): %s has %d is_reversed flags for %d edges" % (i, x.get("id", "?"), key, len(rev), len(src)))
    return len(samples)


def fit(model, optimizer, train_set, dev_set, epochs, batch_size, device, save_dir=None, config=None, sync=None,
        eval_metric="MAE", seed=0, **train_kw):
    """The epoch loop of ``train.py:1296-1380`` reduced to its contract: per epoch one shuffled training
    pass and one dev evaluation; with ``save_dir`` the run directory the reference's tooling expects --
    ``config.json``, ``epoch%d.pt`` (state dict of every epoch), ``log.txt`` whose "best" lines
    (``utils/log.py:50-57``) ``dataio.get_best_epochs`` reads back.  Returns the per-epoch history.
    ``graph=True`` (with ``FlatAdamW(capturable=True)``, one rank): training steps whose batch shape repeats are
    recorded once and replayed as one HIP graph (``GraphedTrainStep``).
    ``gate_compact=True`` (or a margin, e.g. 1.25): training forward passes run the rep-net on the target edges the filter
    gate keeps (``model.set_gate_capacity``; capacity = the largest kept count over the training batches in their stored order x the margin; batches are shuffled in training, so a batch can still exceed it: such steps are dropped, counted, and a warning is raised above 1 % of the steps).
    A batch that keeps more than that raises a device flag and the optimizer drops its step (``FlatAdamW.set_veto``, as a
    loss-scaling optimizer drops an overflowed step); the history counts them (``dropped_steps``).  Evaluation passes run
    on every edge row.  Single rank only (the veto is rank-local: refused with a multi-rank ``sync``)."""
    from . import dataio
    from .tuning import enable_tuned_gemms
    if train_kw.get("gate_compact") and getattr(sync, "world", 1) > 1:
        # the overflow flag and the capacity are rank-local: a rank whose batch overflowed would still feed its (wrong)
        # gradients to the all-reduce and only IT would drop the step -- the replicas would drift apart for good
        raise ValueError("fit(gate_compact=...) is a single-rank option: the overflow veto is rank-local (world size %d)" % sync.world)
    validate_samples(train_set)
    validate_samples(dev_set)
    enable_tuned_gemms()      # the layer's [rows, 128] x [128, 128..384] products with the solutions picked for MI355X
    sync = sync or FlatGradSync(model)
    if train_kw.get("graph") is True:       # graph=True: replay recorded steps where the setup allows it, else eager
        ok = getattr(optimizer, "capturable", False) and getattr(sync, "world", 1) == 1
        # the regulariser is part of the recording when it can be on at ANY step of the run: a positive constant, or a
        # schedule's value (a number or an annealing spec string in its config: on unless it is the constant 0)
        reg = train_kw.get("rep_reg_w", 0.0)
        sched = train_kw.get("schedule")
        if sched is not None:
            reg = sched.config.get("rep_reg_w", 0.0)
        with_reg = (isinstance(reg, str) and reg.strip() not in ("", "0", "0.0")) or (not isinstance(reg, str) and float(reg) > 0)
        train_kw["graph"] = GraphedTrainStep(model, optimizer, sync, bp_loss=train_kw.get("bp_loss", "MSE"), eval_metric=eval_metric,
                                             max_grad_norm=train_kw.get("max_grad_norm", 8.0), with_rep_reg=with_reg) if ok else None
    elif not train_kw.get("graph"):
        train_kw.pop("graph", None)
    rng = np.random.default_rng(seed)
    compact = train_kw.pop("gate_compact", False)
    if compact and hasattr(model, "calibrate_gate_capacity") and hasattr(optimizer, "set_veto"):
        margin, cap = (1.15 if compact is True else float(compact)), 0
        with torch.enable_grad():
            for i in range(0, len(train_set), batch_size):     # every training batch once: the gates only (no rep-net pass)
                pattern, graph_b = train_set.batchify(np.arange(i, min(i + batch_size, len(train_set))), device)[:2]
                cap = max(cap, model.calibrate_gate_capacity(pattern, graph_b, margin=margin, multiple=256) or 0)
        model.set_gate_capacity(cap or None)
        # (bit 0: a batch kept more edges than the capacity; bit 1: a graph without nodes was dealt padding edges -- both make
        # the step's gradients unusable, both drop it)
        optimizer.set_veto(model.compaction_word if cap else None, mask=3)
    log = None
    if save_dir is not None:
        os.makedirs(save_dir, exist_ok=True)
        if config is not None:
            dataio.save_config(config, os.path.join(save_dir, "config.json"))
        log = open(os.path.join(save_dir, "log.txt"), "w")
    best, history = (float("inf"), -1), []
    import gc
    gc.collect()
    gc.freeze()      # model / dataset / optimizer objects: out of the cyclic collector's way for the whole run
    import contextlib
    graphed = train_kw.get("graph")
    # recorded steps: the whole loop (evaluation, checkpoint copies) on the recordings' side stream -- see dp.StepGraph
    stream_ctx = graphed.steps.on_stream() if graphed else contextlib.nullcontext()
    try:
      with stream_ctx:
        for epoch in range(epochs):
            tr = train_epoch(model, optimizer, train_set, batch_size, device, sync=sync, eval_metric=eval_metric,
                             order=rng.permutation(len(train_set)), **dict(train_kw, epoch=epoch))
            dev = evaluate_epoch(model, dev_set, batch_size, device, eval_metric=eval_metric)
            history.append({"epoch": epoch, "train": tr, "dev": {k: dev[k] for k in ("MAE", "MSE", "eval_metric")}})
            if getattr(model, "gate_capacity", None):
                history[-1]["dropped_steps"] = model.compaction_dropped_steps()       # cumulative (one host sync per epoch)
                steps_so_far = (epoch + 1) * max(1, -(-len(train_set) // batch_size))
                if history[-1]["dropped_steps"] > 0.01 * steps_so_far:
                    import warnings
                    warnings.warn("fit(gate_compact): %d of %d optimizer steps were dropped (batches that kept more edges than the "
                                  "capacity): raise the margin" % (history[-1]["dropped_steps"], steps_so_far))
            if save_dir is not None:
                torch.save(model.state_dict(), dataio.checkpoint_path(save_dir, epoch))
            if dev["eval_metric"] < best[0]:
                best = (dev["eval_metric"], epoch)
            if log is not None:
                log.write("data_type: train\tepoch: %03d/%03d\tbp_loss: %.5f\teval-%s: %.5f\n"
                          % (epoch, epochs, tr["bp_loss"], eval_metric, tr["eval_metric"]))
                log.write("data_type: dev\tepoch: %03d/%03d\teval-%s: %.5f\n" % (epoch, epochs, eval_metric, dev["eval_metric"]))
                log.write(dataio.best_line("dev", best[1], epochs, **{"eval-" + eval_metric: "%.5f" % best[0]}) + "\n")
                log.flush()
    finally:
        gc.unfreeze()
        if log is not None:
            log.close()
    return history

#!/usr/bin/env python3
"""Headline benchmark: (pattern, graph) pairs/sec of the DMPNN rep-net fwd+bwd at hid=128
on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1] per GPU: synthetic directed Erdos-Renyi pairs,
pattern (|V|=8, |E|=12) x target (|V|=64, |E|=256), reversed edges added (E -> 2E, the
reference's --add_rev True), batch 1024 pairs per GPU, 3 shared DMPLayers, hid 128, fp32.
Weak scaling: every rank holds its own 1024-pair shard (seed 1000*config_id + rank).

One step (timed) = what the reference does per batch after the DataLoader hands it over
(SubgraphCountingMatching/train.py:606-686), inputs already resident in HBM:
  device collate of the B pattern and B target graphs (dgl.batch, dataset.py:1320-1328)
  -> graph index build (CSR by dst / by src, degrees)
  -> DMPNN.get_pattern_rep + get_graph_rep (3 layers, gates, residual; dmpnn.py:215-277)
  -> sum-pool head + MSE loss on the counts -> backward
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


def er_local_edges(batch, n, m, rng):
    """[batch, m] local endpoints of directed G(n, m) graphs (distinct ordered pairs, u != v)."""
    total = n * (n - 1)
    # argsort of random keys = sampling without replacement, vectorised over the batch
    pick = np.argsort(rng.random((batch, total)), axis=1)[:, :m]
    u = pick // (n - 1)
    r = pick % (n - 1)
    v = r + (r >= u)
    return u.astype(np.int64), v.astype(np.int64)


def make_shard(cfg, rank, device):
    """Per-rank synthetic shard, resident in HBM: per-graph LOCAL edge lists (as a dataset
    would hold them after add_reversed_edges), sizes, gates, input embeddings, counts."""
    rng = np.random.default_rng(1000 * cfg["config_id"] + rank)
    B, H = cfg["batch"], cfg["hid"]
    out = {}
    for tag, n, m in (("p", cfg["p_nodes"], cfg["p_edges"]), ("g", cfg["g_nodes"], cfg["g_edges"])):
        u, v = er_local_edges(B, n, m, rng)
        # add_reversed_edges (train.py:299-327): [forward | reversed] per graph
        src = np.concatenate([u, v], axis=1).reshape(-1)
        dst = np.concatenate([v, u], axis=1).reshape(-1)
        rev = np.concatenate([np.zeros((B, m), bool), np.ones((B, m), bool)], axis=1).reshape(-1)
        out[tag] = dict(
            local_src=torch.from_numpy(src).to(device), local_dst=torch.from_numpy(dst).to(device),
            rev=torch.from_numpy(rev).to(device),
            num_nodes=torch.full((B,), n, dtype=torch.int64, device=device),
            num_edges=torch.full((B,), 2 * m, dtype=torch.int64, device=device),
            N=B * n, E=B * 2 * m, n=n, e=2 * m)
    g = torch.Generator(device="cpu").manual_seed(1000 * cfg["config_id"] + rank)
    for tag in ("p", "g"):
        out[tag]["v_emb"] = torch.randn(out[tag]["N"], H, generator=g).to(device).requires_grad_(True)
        out[tag]["e_emb"] = torch.randn(out[tag]["E"], H, generator=g).to(device).requires_grad_(True)
    # filter gates of the target graph (ScalarFilter: label occurs in the pattern; filter.py:6-16)
    out["g"]["v_gate"] = (torch.rand(out["g"]["N"], 1, generator=g) < 0.75).float().to(device)
    out["g"]["e_gate"] = (torch.rand(out["g"]["E"], 1, generator=g) < 0.75).float().to(device)
    out["counts"] = torch.randint(0, 64, (B,), generator=g).float().to(device)
    return out


class Head(torch.nn.Module):
    """Sum-pool readout -> count (a reduced SumPredictNet, pred.py:87-156: pools pattern and
    graph representations per pair and regresses the count)."""

    def __init__(self, hid):
        super().__init__()
        self.p = torch.nn.Linear(hid, hid)
        self.g = torch.nn.Linear(hid, hid)
        self.out = torch.nn.Linear(4 * hid, 1)

    def forward(self, p_v, g_v, B):
        p = self.p(p_v.view(B, -1, p_v.size(-1)).sum(1))
        g = self.g(g_v.view(B, -1, g_v.size(-1)).sum(1))
        return self.out(torch.relu(torch.cat([p, g, g - p, g * p], dim=1))).squeeze(-1)


def build_step(cfg, shard, device):
    from dualmessagepassing_amd.collate import collate_device
    from dualmessagepassing_amd.dmpnn import DMPNNRep
    from dualmessagepassing_amd.dp import FlatGradSync

    torch.manual_seed(0)
    net = DMPNNRep(hid_dim=cfg["hid"], rep_num_graph_layers=cfg["layers"], rep_num_pattern_layers=cfg["layers"],
                   share_rep_net=True, rep_residual=True, rep_dmpnn_batch_norm=False, rep_act_func="relu",
                   init_neigenv=4.0, init_eeigenv=4.0).to(device)
    head = Head(cfg["hid"]).to(device)
    model = torch.nn.ModuleDict({"rep": net, "head": head})
    sync = FlatGradSync(model)
    sync.broadcast_parameters()
    opt = torch.optim.AdamW(sync.params, lr=1e-4, weight_decay=1e-5, fused=True)
    B = cfg["batch"]

    def step():
        sync.zero()
        for tag in ("p", "g"):
            shard[tag]["v_emb"].grad = None
            shard[tag]["e_emb"].grad = None
        p, g = shard["p"], shard["g"]
        pattern = collate_device(p["local_src"], p["local_dst"], p["num_nodes"], p["num_edges"], p["N"], p["E"],
                                 edata={"is_reversed": p["rev"]})
        graph = collate_device(g["local_src"], g["local_dst"], g["num_nodes"], g["num_edges"], g["N"], g["E"],
                               edata={"is_reversed": g["rev"]})
        p_v, p_e, g_v, g_e = net(pattern, graph, p["v_emb"], p["e_emb"], g["v_emb"], g["e_emb"],
                                 v_gate=g["v_gate"], e_gate=g["e_gate"])
        pred = head(p_v, g_v, B)
        # edge reps feed the loss too (edge_pred in the reference), so their backward is not pruned
        loss = torch.nn.functional.mse_loss(pred, shard["counts"]) + 1e-3 * (g_e.square().mean() + p_e.square().mean())
        loss.backward()
        sync.sync()
        opt.step()
        return loss

    return step, model


def cpu_baseline(cfg, seconds_budget=20.0):
    """CPU oracle (oracle/dmp_oracle.py: reference op order, torch CPU, all host cores) on a
    bounded sample of the same workload: fwd+bwd of the 3-layer pattern + graph rep-nets."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dmp_oracle as O
    # the oracle's ops are small; past ~32 threads torch's intra-op pool only adds contention
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    B, H, L = 32, cfg["hid"], cfg["layers"]
    rng = np.random.default_rng(7)
    gen = torch.Generator().manual_seed(7)
    layers = [{k: v.requires_grad_(True) for k, v in O.random_dmp_params(H, H, gen).items()} for _ in range(L)]
    data = {}
    for tag, n, m in (("p", cfg["p_nodes"], cfg["p_edges"]), ("g", cfg["g_nodes"], cfg["g_edges"])):
        u, v = er_local_edges(B, n, m, rng)
        off = (np.arange(B) * n)[:, None]
        src = torch.from_numpy(np.concatenate([u + off, v + off], axis=1).reshape(-1))
        dst = torch.from_numpy(np.concatenate([v + off, u + off], axis=1).reshape(-1))
        rev = torch.from_numpy(np.concatenate([np.zeros((B, m), bool), np.ones((B, m), bool)], 1).reshape(-1))
        N, E = B * n, B * 2 * m
        data[tag] = (src, dst, rev, O.out_degrees(src, N), torch.randn(N, H, generator=gen).requires_grad_(True),
                     torch.randn(E, H, generator=gen).requires_grad_(True))
    vg = (torch.rand(data["g"][4].size(0), 1, generator=gen) < 0.75).float()
    eg = (torch.rand(data["g"][5].size(0), 1, generator=gen) < 0.75).float()

    def one():
        ps, pd, pr, pdeg, pv, pe = data["p"]
        gs, gd, gr, gdeg, gv, ge = data["g"]
        a, b = O.dmpnn_graph_rep(layers, ps, pd, pr, pdeg, pv, pe)
        c, d = O.dmpnn_graph_rep(layers, gs, gd, gr, gdeg, gv, ge, vg, eg)
        (a.square().mean() + b.square().mean() + c.square().mean() + d.square().mean()).backward()

    one()  # warm-up
    t0, n = time.perf_counter(), 0
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 50:
            break
    return {"value": B * n / el, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d steps of B=%d pairs (same shapes, hid=%d, %d layers, fwd+bwd, fp32), torch %s CPU, %d threads"
                      % (n, B, H, L, torch.__version__, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=CFG["batch"], help="pairs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    cfg = dict(CFG, batch=args.batch)
    from dualmessagepassing_amd import _lib
    shard = make_shard(cfg, rank, device)
    step, model = build_step(cfg, shard, device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    _lib.timer.reset()
    _lib.timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    _lib.timer.enabled = False
    kern = _lib.timer.summary()

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        pairs = cfg["batch"] * world * args.steps
        H = cfg["hid"]
        gN, gE = cfg["batch"] * cfg["g_nodes"], cfg["batch"] * 2 * cfg["g_edges"]
        key = "seg_sum2[H=%d,rows=%d,ent=%d]" % (H, gN, gE)
        roof = None
        if key in kern:
            k = kern[key]
            roof = {"bound": "hbm", "kernel": "dmp::seg_sum_vec<32,split> (node aggregation by destination, target graph)",
                    "achieved": round(k["gbps"], 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(k["gbps"] / HBM_PEAK_GBPS, 4), "traffic": None,
                    "bytes_per_launch": int(k["bytes"]), "avg_us": round(k["avg_us"], 2), "launches": k["launches"]}
        line = {
            "metric": "(pattern,graph) pairs/sec DMPNN fwd+bwd hid=128", "value": round(pairs / dt, 1),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ER pattern(8,12)x target(64,256), add_rev, "
                                   "batch=%d pairs/GPU, 3-layer shared DMPNN rep-net, hid=%d, fp32" % (cfg["batch"], H),
                       "global_batch": cfg["batch"] * world, "parallelism": "dp%d" % world,
                       "step": "device collate + index build + fwd + bwd + grad all-reduce + AdamW"},
            "roofline": roof,
            "kernels": {n: {"avg_us": round(v["avg_us"], 2), "gbps": round(v["gbps"], 1), "launches": v["launches"],
                            "bytes": int(v["bytes"])} for n, v in sorted(kern.items())},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

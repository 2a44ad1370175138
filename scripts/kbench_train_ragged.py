"""Training step time on RAGGED batches (every batch a new (N, E) shape, as a real dataset gives): 64 pairs per batch,
patterns of 3-8 nodes, graphs of 8-64 nodes, hid 64, 3 layers -- eager launches (a HIP-graph recording needs repeated
shapes)."""
import os, sys, time
import numpy as np, torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd.basemodel import build_model
from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
from dualmessagepassing_amd.harness import PairDataset, train_epoch
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
gpu = th.device("cuda:0")
rng = np.random.default_rng(0)


def er(n, m):
    pick = rng.choice(n * (n - 1), size=m, replace=False)
    u = pick // (n - 1); r = pick % (n - 1)
    return u.astype(np.int64), (r + (r >= u)).astype(np.int64)


S, shape = 1024, dict(p_nodes=8, p_edges=16, g_nodes=64, g_edges=256, n_vlabels=8, n_elabels=8)
samples = []
for i in range(S):
    pn = int(rng.integers(3, 9)); pm = int(rng.integers(pn - 1, min(16, pn * (pn - 1)) + 1))
    gn = int(rng.integers(8, 65)); gm = int(rng.integers(gn, min(256, gn * (gn - 1)) + 1))
    pu, pv = er(pn, pm); gu, gv = er(gn, gm)
    samples.append({"id": "s%d" % i, "pattern": PairDataset._with_rev(pu, pv, rng.integers(0, 8, pn), rng.integers(0, 8, pm), 16, 8),
                    "graph": PairDataset._with_rev(gu, gv, rng.integers(0, 8, gn), rng.integers(0, 8, gm), 256, 8),
                    "counts": int(rng.integers(0, 5)), "subisomorphisms": np.zeros((0, pn), np.int64)})
ds = PairDataset(samples, shape)
th.manual_seed(0)
model = build_model(**ds.model_config(hid_dim=64, layers=3, rep_act_func="leaky_relu", pred_act_func="leaky_relu", emb_net="Equivariant")).to(gpu)
from dualmessagepassing_amd.harness import GraphedTrainStep
B = 64
for graphed in (False, True):
    th.manual_seed(0)
    model = build_model(**ds.model_config(hid_dim=64, layers=3, rep_act_func="leaky_relu", pred_act_func="leaky_relu", emb_net="Equivariant")).to(gpu)
    sync = FlatGradSync(model)
    opt = FlatAdamW([sync.flatten_parameters()], lr=1e-3, weight_decay=1e-5, amsgrad=True, capturable=graphed)
    # graphed: batches padded with inert pairs to one of four capacity levels (PairDataset.batch_arrays(pad=...)), so every step replays
    step = GraphedTrainStep(model, opt, sync, max_shapes=4) if graphed else None
    import contextlib
    with (step.steps.on_stream() if graphed else contextlib.nullcontext()):
        for epoch in range(4):
            th.cuda.synchronize(); t0 = time.perf_counter()
            out = train_epoch(model, opt, ds, B, gpu, sync=sync, neg_slp=0.01, order=np.random.default_rng(epoch).permutation(S), graph=step)
            th.cuda.synchronize(); dt = time.perf_counter() - t0
            print("%s epoch %d: %.3f ms/step  (%d ragged batches of %d pairs, %.0f pairs/s)%s"
                  % ("padded + replayed" if graphed else "exact shapes, eager", epoch, dt / (S // B) * 1e3, S // B, B, S / dt,
                     "  replays %d eager %d" % (step.steps.replays, step.steps.eager_calls) if graphed else ""), flush=True)

"""Training at the reference's own small-batch settings (BASELINE configs[0] shape: pattern (8,12) x target (64,256),
add_rev, batch 32, hid 64, 3 layers; labelled synthetic pairs with exact counts): time per optimisation step of
harness.train_epoch with eager launches vs one HIP-graph replay per step (harness.GraphedTrainStep)."""
import os, sys, time
import numpy as np
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd.basemodel import build_model
from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
from dualmessagepassing_amd.harness import GraphedTrainStep, SyntheticPairs, train_epoch
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
gpu = th.device("cuda:0")
B = int(os.environ.get("B", "32"))
ds = SyntheticPairs(8 * B, 8, 12, 64, 256, 8, 8, seed=1)
for graphed in (False, True):
    th.manual_seed(0)
    model = build_model(**ds.model_config(hid_dim=64, layers=3, rep_act_func="leaky_relu", pred_act_func="leaky_relu",
                                          emb_net="Equivariant")).to(gpu)
    sync = FlatGradSync(model)
    opt = FlatAdamW([sync.flatten_parameters()], lr=1e-3, weight_decay=1e-5, amsgrad=True, capturable=graphed)
    step = GraphedTrainStep(model, opt, sync) if graphed else None
    for _ in range(2):                                   # warm-up epochs (eager first step, the recording)
        train_epoch(model, opt, ds, B, gpu, sync=sync, neg_slp=0.01, graph=step)
    th.cuda.synchronize()
    t0 = time.perf_counter()
    epochs = 5
    for e in range(epochs):
        out = train_epoch(model, opt, ds, B, gpu, sync=sync, neg_slp=0.01, graph=step)
    th.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = epochs * (len(ds) // B)
    print("%-28s %.3f ms/step  %8.0f pairs/s   (batch %d, loss %.4f)%s" % (
        "HIP-graph replay per step" if graphed else "eager launches", dt / steps * 1e3, steps * B / dt, B, out["bp_loss"],
        "  replays %d eager %d" % (step.steps.replays, step.steps.eager_calls) if graphed else ""))
    if graphed:                                           # where the replayed step's time goes: replay alone vs the batch's host side
        rec = next(iter(step.steps._graphs.values()))
        th.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): rec[0].replay()
        th.cuda.synchronize(); t_replay = (time.perf_counter() - t0) / 50 * 1e3
        idx = np.arange(B)
        t0 = time.perf_counter()
        for _ in range(50): ds.batch_arrays(idx, gpu)
        th.cuda.synchronize(); t_batch = (time.perf_counter() - t0) / 50 * 1e3
        print("    of which: graph replay alone %.3f ms, batch_arrays (host concatenation + uploads) %.3f ms" % (t_replay, t_batch))

"""UNC path (BASELINE config 5 shape): one Cora-sized graph (N = 2,708, |E| = 5,429 -> 10,858 directed with
reversed copies, 1 relation -> 2 edge types), DMPNN hid 256, 2 layers (UNC main.py defaults): forward and
forward + backward time of the TrainModel step with the unsupervised loss.  ER graph of the same (N, E):
the real Cora files are not available offline."""
import os, sys, time
import numpy as np, torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd.unc import TrainModel, build_graph_from_triplets
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
gpu = th.device("cuda:0")
rng = np.random.default_rng(5)
n, m, h = 2708, 5429, int(os.environ.get("UNC_HID", "256"))    # 256: BASELINE configs[4]; 50: the width the reference ships (main.py:244, run.sh:8)
pick = rng.choice(n * (n - 1), size=m, replace=False)
u = pick // (n - 1); r = pick % (n - 1); v = r + (r >= u)
trip = np.stack([u, np.zeros(m, np.int64), v], 1)
g = build_graph_from_triplets(n, 1, trip, gpu)
etype, norm = g.edata["type"], g.edata["norm"]
th.manual_seed(0)
model = TrainModel(None, n, h, 1, 0, num_hidden_layers=2, dropout=0.0, reg_param=0.01).to(gpu)
from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
sync = FlatGradSync(model)
master = sync.flatten_parameters()
opt = FlatAdamW([master], lr=1e-3, weight_decay=0.0)      # Adam (UNC main.py:112) as one launch over the flat parameters
nid = th.arange(n, device=gpu)
samples = th.from_numpy(np.concatenate([trip, np.stack([rng.integers(0, n, m), np.zeros(m, np.int64), rng.integers(0, n, m)], 1)])).to(gpu)
labels = th.cat([th.ones(m), th.zeros(m)]).to(gpu)
from dualmessagepassing_amd import ops
ops.mark_immutable(nid, etype, samples)    # the ONE graph's ids / relation types / triplets: never refilled in place, so a
                                           # recorded step may take their memoised indexes (ops.mark_immutable)


def timeit(f, it=30):
    for _ in range(5): f()
    th.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): f()
    th.cuda.synchronize(); return (time.perf_counter() - t) / it * 1e3


def fwd():
    with th.no_grad():
        return model(g, nid, etype, norm)


def step():
    sync.detach_grads()
    emb, _ = model(g, nid, etype, norm)
    loss = model.get_unsupervised_loss(g, emb, etype, samples, labels)
    loss.backward()
    sync.pack()
    opt.step()
    return loss


print("UNC DMPNN hid=%d, 2 layers, N=%d, E=%d: forward %.3f ms, train step (fwd+bwd+Adam) %.3f ms" % (h, n, 2 * m, timeit(fwd), timeit(step)))

# the same step replayed from ONE HIP graph (dp.StepGraph): the optimizer keeps its step count and rate on the device
from dualmessagepassing_amd.dp import StepGraph
opt = FlatAdamW([master], lr=1e-3, weight_decay=0.0, capturable=True)
sg = StepGraph(lambda: step(), optimizer=opt)
try:
    ms = timeit(sg)
    print("UNC train step as one HIP-graph replay: %.3f ms   (replays %d, eager %d)" % (ms, sg.replays, sg.eager_calls))
except Exception as exc:                                   # a host sync inside the step cannot be recorded
    print("UNC train step could not be recorded: %s: %s" % (type(exc).__name__, str(exc)[:300]))


# ---- the SAMPLED mini-batch step (utils.py:279-349 + main.py:99-183: a new sub-graph every step): eager launches vs
# unc_harness.SampledStep (sub-graphs padded to capacity levels, the step after the sampling replayed from one HIP graph)
from dualmessagepassing_amd.unc_harness import SampledStep
from dualmessagepassing_amd.unc_sampling import generate_sampled_graph_and_labels_unsupervised
trip_dev = th.from_numpy(trip).to(gpu)
gen = th.Generator(device=gpu).manual_seed(3)
B = int(os.environ.get("UNC_BATCH", "2000"))


def draw():
    idx = th.randint(0, m, (B,), device=gpu, generator=gen)
    return generate_sampled_graph_and_labels_unsupervised(g, trip_dev[idx], 6, 128, 0.5, 5, generator=gen, sampler="neighbor")


def run(stepper, it=30, warm=12):
    if stepper is None:
        return _run(None, it, warm)
    with stepper.steps.on_stream():          # the whole loop, sampling included, on the recording's stream (SampledStep)
        return _run(stepper, it, warm)


def _run(stepper, it, warm):
    t_step = 0.0
    for k in range(warm + it):
        sub, smp, lab = draw()
        th.cuda.synchronize(); t0 = time.perf_counter()
        if stepper is None:
            sync.detach_grads()
            emb, _ = model(sub, sub.ndata["_ID"], sub.edata["type"], sub.edata["norm"])
            loss = model.get_unsupervised_loss(sub, emb, sub.edata["type"], smp, lab)
            loss.backward(); sync.pack(); th.nn.utils.clip_grad_norm_([master], 1.0); opt_e.step()
        else:
            stepper(sub, sub.edata["type"], smp, lab)
        th.cuda.synchronize()
        if k >= warm: t_step += time.perf_counter() - t0
    return t_step / it * 1e3


opt_e = FlatAdamW([master], lr=1e-3, weight_decay=0.0)
eager_ms = run(None)
opt_r = FlatAdamW([master], lr=1e-3, weight_decay=0.0, capturable=True)
st = SampledStep(model, sync, opt_r, 1.0)
replay_ms = run(st)
sub, _, _ = draw()
print("UNC SAMPLED step (batch %d edges, neighbour sampler 6 x 128: sub-graphs of ~%d nodes / ~%d edges), after the sampling: eager %.3f ms, "
      "padded + replayed %.3f ms (%.2f x; replays %d, eager %d, shapes %d)"
      % (B, sub.number_of_nodes(), sub.number_of_edges(), eager_ms, replay_ms, replay_ms / eager_ms, st.steps.replays, st.steps.eager_calls,
         len(st.steps._graphs)))

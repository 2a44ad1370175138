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


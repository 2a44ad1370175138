"""Dev probe: which part of the UNC train step breaks HIP-graph recording (each variant in its own process)."""
import os, subprocess, sys
VARIANTS = ["fwd_bwd_sidestream", "fwd_bwd_gc", "fwd_bwd"]
if len(sys.argv) == 1:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), v], capture_output=True, text=True, timeout=300)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if "amdgpu" not in l][-6:]
        print("== %-14s rc=%d  %s" % (v, r.returncode, " | ".join(t[:160] for t in tail)), flush=True)
    sys.exit(0)
variant = sys.argv[1]
import numpy as np, torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd.unc import TrainModel, build_graph_from_triplets
gpu = th.device("cuda:0")
rng = np.random.default_rng(5)
n, m, h = 2708, 5429, 256
pick = rng.choice(n * (n - 1), size=m, replace=False)
u = pick // (n - 1); r = pick % (n - 1); v = r + (r >= u)
trip = np.stack([u, np.zeros(m, np.int64), v], 1)
g = build_graph_from_triplets(n, 1, trip, gpu)
etype, norm = g.edata["type"], g.edata["norm"]
th.manual_seed(0)
model = TrainModel(None, n, h, 1, 0, num_hidden_layers=2, dropout=0.0, reg_param=0.01).to(gpu)
if variant in ("fwd_nobn", "fwd_bwd_evalbn"):
    model.eval()
if variant == "fwd_bwd_nomiopen":
    th.backends.cudnn.enabled = False
nid = th.arange(n, device=gpu)
samples = th.from_numpy(np.concatenate([trip, np.stack([rng.integers(0, n, m), np.zeros(m, np.int64), rng.integers(0, n, m)], 1)])).to(gpu)
labels = th.cat([th.ones(m), th.zeros(m)]).to(gpu)
opt = None
if variant == "full_sgd":
    opt = th.optim.SGD(model.parameters(), lr=1e-3)
if variant == "full_adam":
    opt = th.optim.Adam(model.parameters(), lr=1e-3, capturable=True)


def step():
    if variant == "loss_fwd_only":
        with th.no_grad():
            emb, _ = model(g, nid, etype, norm)
            return model.get_unsupervised_loss(g, emb, etype, samples, labels)
    if variant in ("fwd", "fwd_nobn"):
        with th.no_grad():
            return model(g, nid, etype, norm)[0][0].sum()
    for p in model.parameters():
        p.grad = None
        if variant == "fwd_bwd_detach_emb" and p.dim() == 2 and p.size(0) in (n, 2):
            p.requires_grad_(False)
        if variant == "fwd_bwd_detach_node" and p.dim() == 2 and p.size(0) == n:
            p.requires_grad_(False)
        if variant == "fwd_bwd_detach_rel" and p.dim() == 2 and p.size(0) == 2:
            p.requires_grad_(False)
    emb, _ = model(g, nid, etype, norm)
    if variant == "fwd_bwd_node_only":
        loss = emb[0].square().mean()
    elif variant.startswith("fwd_bwd"):
        loss = emb[0].square().mean() + emb[1].square().mean()
    else:
        loss = model.get_unsupervised_loss(g, emb, etype, samples, labels)
    loss.backward()
    if opt is not None:
        opt.step()
    return loss.detach()


if variant == "fwd_bwd_list":
    for name, p in model.named_parameters():
        print(name, tuple(p.shape))
    sys.exit(0)
side = th.cuda.Stream()
if variant == "fwd_bwd_sidestream":
    side.wait_stream(th.cuda.current_stream())
    with th.cuda.stream(side):
        for _ in range(3): step()
else:
    for _ in range(3): step()
th.cuda.synchronize()
if variant == "fwd_bwd_gc":
    import gc
    gc.collect()
gr = th.cuda.CUDAGraph()
with th.cuda.graph(gr, stream=side):
    out = step()
th.cuda.synchronize()
gr.replay(); gr.replay()
th.cuda.synchronize()
print("recorded and replayed:", float(out))

"""Dev aid: config-4 shard, one pass vs four slices: where do they start to differ?"""
import os, sys
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualmessagepassing_amd.basemodel import build_model
from dualmessagepassing_amd.collate import collate_device
gpu = th.device("cuda:0")
B = int(os.environ.get("B", "1024"))
cfg = dict(bench.CFG4, batch=B, act="leaky_relu", emb="Equivariant")
shard = bench.make_shard(cfg, 0, gpu)
th.manual_seed(0)
model = build_model(**bench.model_config(cfg)).to(gpu)
def batch(part):
    mk = lambda d: collate_device(d["local_src"], d["local_dst"], d["num_nodes"].clone(), d["num_edges"].clone(), d["N"], d["E"], ndata=d["ndata"],
                                  edata=dict(d["edata"]), max_nodes=d["max_n"], max_edges=d["max_e"])
    return mk(part["p"]), mk(part["g"])
def run(part, keys):
    out = model(*batch(part))
    loss = (out["pred_c"].view(-1) * 1e-6).sum()
    loss.backward()
    res = {k: out[k].detach().clone() for k in keys}
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    return res, grads
keys = ("pred_c", "g_e_rep", "g_v_rep")
whole, gw = run(shard, keys)
M = 4
parts = [bench.slice_shard(cfg, shard, i * (B // M), (i + 1) * (B // M)) for i in range(M)]
outs, gs = [], None
for p in parts:
    o, g = run(p, keys)
    outs.append(o)
    gs = g if gs is None else {k: gs[k] + g[k] for k in g}
for k in keys:
    cat = th.cat([o[k] for o in outs])
    d = (whole[k] - cat).abs()
    print(k, tuple(cat.shape), "max abs diff %.4g  scale %.4g  first bad row %s" % (float(d.max()), float(cat.abs().max()),
          (d.view(d.size(0), -1).max(1)[0] > 1e-3 * float(cat.abs().max())).nonzero()[:3].view(-1).tolist()))
for k in gw:
    d = float((gw[k] - gs[k]).abs().max()); s = float(gs[k].abs().max())
    if d > 1e-3 * max(s, 1e-30):
        print("GRAD", k, d, s)
print("done")

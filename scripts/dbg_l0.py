import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
import bench
from dualmessagepassing_amd import fused, dmpnn
from dualmessagepassing_amd.basemodel import build_model
from dualmessagepassing_amd.collate import collate_device
gpu = th.device("cuda:0")
cfg = dict(bench.CFG, batch=32)
shard = bench.make_shard(cfg, 0, gpu)
model = build_model(**bench.model_config(cfg)).to(gpu)
orig = dmpnn._layer0_codes
def dbg(union, layers, p_e_emb, g_e_emb, e_gate, pools):
    ps, gs = getattr(p_e_emb, "_dmp_src", None), getattr(g_e_emb, "_dmp_src", None)
    print("layers", len(layers), "grad", th.is_grad_enabled(), "ps", ps is not None, "gs", gs is not None, type(p_e_emb), type(g_e_emb))
    if ps is not None and gs is not None:
        H = layers[0].hidden_dim
        ix = union.index()
        print("same W", ps[1] is gs[1], ps[0].shape, gs[0].shape, gs[1].shape, gs[1].stride(), ps[0].stride(), gs[0].stride(), ps[0].dtype,
              "typed", fused.typed_ok(ix, H), "onepanel", fused.onepanel_ok(H), "E", ix.num_edges, "USE", fused.USE_LAYER0, H in fused.MFMA_WIDTHS)
        print("e_gate", None if e_gate is None else (e_gate.shape, e_gate.requires_grad))
    r = orig(union, layers, p_e_emb, g_e_emb, e_gate, pools)
    print("->", None if r is None else (r[0].shape, r[1]))
    return r
dmpnn._layer0_codes = dbg
gs = {}
for tag in ("p", "g"):
    s = shard[tag]
    gs[tag] = collate_device(s["local_src"], s["local_dst"], s["num_nodes"], s["num_edges"], s["N"], s["E"], ndata=s["ndata"], edata=s["edata"], max_nodes=s["max_n"], max_edges=s["max_e"])
out = model(gs["p"], gs["g"])
(out["pred_c"] ** 2).sum().backward()
print("ok", float(out["pred_c"].sum()))

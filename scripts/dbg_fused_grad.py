"""Dev aid: every output / gradient of the fused and the modular rep-net path against the oracle, without stopping
at the first mismatch; each case run several times (is a mismatch deterministic?)."""
import os, sys
import numpy as np, torch as th
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import dmp_oracle as O
from util_graphs import er_batch
from dualmessagepassing_amd.dmpnn import DMPNNRep
from dualmessagepassing_amd.graph import BatchedGraph
gpu = th.device("cuda:0")
_t = lambda a: th.from_numpy(np.asarray(a))
def err(a, b):
    b = b.detach().double()
    return float((a.detach().double().cpu() - b).abs().max()) / max(1.0, float(b.abs().max()))
cases = [(64, 64, 256, 128, True, True, "leaky_relu"), (32, 64, 256, 64, True, True, "relu"), (32, 64, 256, 64, True, True, "leaky_relu"),
         (64, 64, 256, 128, True, True, "relu"), (32, 64, 256, 64, False, True, "relu"), (32, 64, 256, 64, True, False, "relu")]
for batch, n, m, h, gates, residual, act in cases:
    rng = np.random.default_rng(n * h + batch)
    src, dst, rev, N, bnn, bne = er_batch(batch, n, m, rng)
    E, L = len(src), 2
    gen = th.Generator().manual_seed(h + n)
    layers = [O.random_dmp_params(h, h, gen, act) for _ in range(L)]
    v0, e0 = th.randn(N, h, generator=gen), th.randn(E, h, generator=gen)
    wv, we = th.randn(N, h, generator=gen), th.randn(E, h, generator=gen)
    vg = (th.rand(N, 1, generator=gen) < 0.7).float() if gates else None
    eg = (th.rand(E, 1, generator=gen) < 0.7).float() if gates else None
    ts, td, tr = _t(src), _t(dst), _t(rev)
    lo = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in layers]
    vo, eo = v0.clone().requires_grad_(True), e0.clone().requires_grad_(True)
    rv, re = O.dmpnn_graph_rep(lo, ts, td, tr, O.out_degrees(ts, N), vo, eo, vg, eg, residual, act)
    ((rv * wv).sum() + (re * we).sum()).backward()
    for fused in (True, False):
        for rep in range(3):
            net = DMPNNRep(hid_dim=h, rep_num_graph_layers=L, rep_num_pattern_layers=L, share_rep_net=True,
                           rep_residual=residual, rep_dmpnn_batch_norm=False, rep_act_func=act)
            sd = {}
            for i, p in enumerate(layers):
                for k, v in p.items():
                    sd["g_rep_net.dmpnn.graph_dmpnn_(%d).%s" % (i, k)] = v
                    sd["p_rep_net.dmpnn.graph_dmpnn_(%d).%s" % (i, k)] = v
            net.load_state_dict(sd, strict=True)
            net.to(gpu)
            net.use_fused = fused
            g = BatchedGraph(ts.to(gpu), td.to(gpu), N, _t(bnn).to(gpu), _t(bne).to(gpu))
            g.edata["is_reversed"] = tr.to(gpu)
            vgp, egp = v0.to(gpu).requires_grad_(True), e0.to(gpu).requires_grad_(True)
            a, b = net.get_graph_rep(g, vgp, egp, v_gate=None if vg is None else vg.to(gpu), e_gate=None if eg is None else eg.to(gpu))
            ((a * wv.to(gpu)).sum() + (b * we.to(gpu)).sum()).backward()
            out = {"v_rep": err(a, rv), "e_rep": err(b, re), "dv": err(vgp.grad, vo.grad), "de": err(egp.grad, eo.grad)}
            grads = {k: p.grad for k, p in net.g_rep_net.named_parameters()}
            for i in range(L):
                for k, p in lo[i].items():
                    out["%d.%s" % (i, k)] = err(grads["dmpnn.graph_dmpnn_(%d).%s" % (i, k)], p.grad)
            bad = {k: "%.2e" % v for k, v in out.items() if v > 3e-4}
            print((batch, n, m, h, gates, residual, act), "fused" if fused else "modular", rep, "BAD" if bad else "ok", bad)

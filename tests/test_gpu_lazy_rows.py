"""The last rep-net layer under pooling heads does not form its edge rows (fused._FusedDMPLayer, edge_rows=False: the pooled
sums come from two pooled passes); ``OutputDict["g_e_rep"]`` still delivers them -- computed on first access, differentiable
(embed.DeferredRows).  Against the same model with ``lazy_edge_rep = False``.  Reference: basemodel.py:1493-1661 returns the
representations in its output dict; dmpnn.py:262-275."""
import os
import sys

import pytest
import torch as th

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _setup(gpu, hid=128):
    import bench
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.collate import collate_device
    cfg = dict(bench.CFG, batch=16, hid=hid)
    shard = bench.make_shard(cfg, 0, gpu)
    th.manual_seed(3)
    model = build_model(**bench.model_config(cfg)).to(gpu)

    def graphs():
        out = []
        for tag in ("p", "g"):
            s = shard[tag]
            out.append(collate_device(s["local_src"], s["local_dst"], s["num_nodes"], s["num_edges"], s["N"], s["E"], ndata=s["ndata"],
                                      edata=s["edata"], max_nodes=s["max_n"], max_edges=s["max_e"]))
        return out
    return model, graphs


@pytest.mark.parametrize("hid", [128, 64])
def test_pooled_heads_without_edge_rows_equal_the_eager_form(hid, gpu):
    from dualmessagepassing_amd import fused
    from dualmessagepassing_amd.embed import DeferredRows
    model, graphs = _setup(gpu, hid)
    res = {}
    for lazy in (True, False):
        model.lazy_edge_rep = lazy
        model.zero_grad(set_to_none=True)
        calls = []
        # the E-row Linear + gate + residual launches: the one-panel kernel, or (inner layers under a 0 / 1 gate) its form
        # over the kept edges' tiles
        # (... which for a first layer on the label codes also forms the residual rows: out_fwd_typed_codes)
        orig, orig_t, orig_c = fused.out_fwd_mfma, fused.out_fwd_typed, fused.out_fwd_typed_codes
        fused.out_fwd_mfma = lambda *a, **k: (calls.append(a[0].size(0)), orig(*a, **k))[1]
        fused.out_fwd_typed = lambda *a, **k: (calls.append(a[0].size(0)), orig_t(*a, **k))[1]
        fused.out_fwd_typed_codes = lambda *a, **k: (calls.append(a[0].size(0)), orig_c(*a, **k))[1]
        try:
            out = model(*graphs())
        finally:
            fused.out_fwd_mfma, fused.out_fwd_typed, fused.out_fwd_typed_codes = orig, orig_t, orig_c
        raw = dict.__getitem__(out, "g_e_rep")
        assert isinstance(raw, DeferredRows) == lazy
        edge_rows = raw.size(0) + dict.__getitem__(out, "p_e_rep").size(0)
        assert calls.count(edge_rows) == (2 if lazy else 3)    # the last layer's E-row Linear + gate + residual kernel did not run
        (out["pred_c"] ** 2).sum().backward()
        res[lazy] = (out["pred_c"].detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    (p1, g1), (p0, g0) = res[True], res[False]
    assert float((p1 - p0).abs().max()) <= 2e-5 * max(1.0, float(p0.abs().max()))
    assert set(g1) == set(g0)
    for n in g0:
        scale = max(1.0, float(g0[n].abs().max()))
        assert float((g1[n] - g0[n]).abs().max()) <= 2e-4 * scale, (n, float((g1[n] - g0[n]).abs().max()), scale)


def test_reading_the_edge_representation_materialises_it_with_gradients(gpu):
    model, graphs = _setup(gpu)
    res = {}
    for lazy in (True, False):
        model.lazy_edge_rep = lazy
        model.zero_grad(set_to_none=True)
        out = model(*graphs())
        ge, pe = out["g_e_rep"], out.p_e_rep                   # by key and by attribute
        assert th.is_tensor(ge) and th.is_tensor(pe) and ge.requires_grad
        assert th.is_tensor(dict.__getitem__(out, "g_e_rep"))  # resolved in place
        loss = (out["pred_c"] ** 2).sum() + 1e-3 * (ge ** 2).mean() + 1e-3 * (pe ** 2).mean()   # a representation regulariser
        loss.backward()
        res[lazy] = (ge.detach().clone(), pe.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    (ge1, pe1, g1), (ge0, pe0, g0) = res[True], res[False]
    assert float((ge1 - ge0).abs().max()) <= 2e-5 * max(1.0, float(ge0.abs().max()))
    assert float((pe1 - pe0).abs().max()) <= 2e-5 * max(1.0, float(pe0.abs().max()))
    for n in g0:
        scale = max(1.0, float(g0[n].abs().max()))
        assert float((g1[n] - g0[n]).abs().max()) <= 2e-4 * scale, (n, float((g1[n] - g0[n]).abs().max()), scale)

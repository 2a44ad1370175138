"""The oracle (oracle/dmp_oracle.py) against the golden vectors emitted by the
reference's own layer code (oracle/make_golden.py).  CPU only.

Tolerances (fp32): per-layer outputs and input grads rtol=atol=1e-5,
parameter grads 2e-4 relative to the largest entry, 3-layer reps 1e-4."""
import numpy as np
import pytest
import torch as th

import dmp_oracle as O
from conftest import golden_files, load_golden


def _t(a):
    return th.from_numpy(np.asarray(a))


def _close(got, ref, rtol=1e-5, atol=1e-5):
    got, ref = got.detach().double(), _t(ref).double()
    assert got.shape == ref.shape
    scale = max(1.0, float(ref.abs().max())) if ref.numel() else 1.0
    err = float((got - ref).abs().max()) if ref.numel() else 0.0
    assert err <= atol * scale + rtol * scale, "max err %g (scale %g)" % (err, scale)


def _params(d, requires_grad=True):
    return {k[2:]: _t(v).clone().requires_grad_(requires_grad) for k, v in d.items() if k.startswith("p.")}


@pytest.mark.parametrize("path", golden_files("dmplayer_"))
def test_dmp_layer_matches_reference(path):
    d = load_golden(path)
    p = _params(d)
    x = _t(d["x"]).clone().requires_grad_(True)
    z = _t(d["z"]).clone().requires_grad_(True)
    rev = _t(d["rev"]) if "rev" in d else None
    node_out, edge_out, edge_msg, agg = O.dmp_layer(
        p, _t(d["src"]), _t(d["dst"]), rev, _t(d["out_deg"]), x, z, str(d["act_func"]), int(d["num_mlp_layers"]))
    _close(node_out, d["node_out"])
    _close(edge_out, d["edge_out"])
    _close(edge_msg, d["edge_agg"])
    _close(agg, d["node_agg"])
    ((node_out * _t(d["wn"])).sum() + (edge_out * _t(d["we"])).sum()).backward()
    _close(x.grad, d["dx"])
    _close(z.grad, d["dz"])
    for k, v in p.items():
        if "g." + k in d:
            _close(v.grad, d["g." + k], rtol=2e-4, atol=2e-4)


def test_dmpnn_rep_matches_reference():
    d = load_golden(golden_files("dmpnn_rep")[0])
    L = int(d["layers"])
    for tag in ("p", "g"):
        layers = []
        for i in range(L):
            pre = "p.dmpnn.graph_dmpnn_(%d)." % i
            layers.append({k[len(pre):]: _t(v).clone().requires_grad_(True) for k, v in d.items() if k.startswith(pre)})
        v = _t(d[tag + "_v_emb"]).clone().requires_grad_(True)
        e = _t(d[tag + "_e_emb"]).clone().requires_grad_(True)
        vg = _t(d["g_v_gate"]) if tag == "g" else None
        eg = _t(d["g_e_gate"]) if tag == "g" else None
        v_rep, e_rep = O.dmpnn_graph_rep(layers, _t(d[tag + "_src"]), _t(d[tag + "_dst"]), _t(d[tag + "_rev"]),
                                         _t(d[tag + "_out_deg"]), v, e, vg, eg, True, "relu")
        _close(v_rep, d[tag + "_v_rep"], 1e-4, 1e-4)
        _close(e_rep, d[tag + "_e_rep"], 1e-4, 1e-4)
        ((v_rep * _t(d[tag + "_wv"])).sum() + (e_rep * _t(d[tag + "_we"])).sum()).backward()
        _close(v.grad, d[tag + "_dv_emb"], 1e-4, 1e-4)
        _close(e.grad, d[tag + "_de_emb"], 1e-4, 1e-4)
        for i in range(L):
            for k, p in layers[i].items():
                _close(p.grad, d["%s_grad.dmpnn.graph_dmpnn_(%d).%s" % (tag, i, k)], 2e-4, 2e-4)


@pytest.mark.parametrize("path", golden_files("compgcn_"))
def test_compgcn_layer_matches_reference(path):
    d = load_golden(path)
    p = _params(d)
    x = _t(d["x"]).clone().requires_grad_(True)
    z = _t(d["z"]).clone().requires_grad_(True)
    rev = _t(d["rev"]) if "rev" in d else None
    node_out, edge_out = O.compgcn_layer(p, _t(d["src"]), _t(d["dst"]), rev, x, z, str(d["comp_opt"]),
                                         str(d["edge_norm"]), "relu")
    _close(node_out, d["node_out"])
    _close(edge_out, d["edge_out"])
    if "norm" in d:
        n = O.compgcn_norms(_t(d["src"]), _t(d["dst"]), int(d["num_nodes"]), str(d["edge_norm"]), bool(d["self_loop"]))
        _close(n, d["norm"], 1e-6, 1e-6)
    ((node_out * _t(d["wn"])).sum() + (edge_out * _t(d["we"])).sum()).backward()
    _close(x.grad, d["dx"])
    _close(z.grad, d["dz"])
    for k, v in p.items():
        if "g." + k in d:
            _close(v.grad, d["g." + k], rtol=2e-4, atol=2e-4)
        else:  # unused by the reference on this input (e.g. out_weight without REVFLAG)
            assert v.grad is None or float(v.grad.abs().max()) == 0.0


@pytest.mark.parametrize("path", golden_files("unc_dualconv_"))
def test_unc_dual_graph_conv_matches_reference(path):
    d = load_golden(path)
    p = _params(d)
    x = _t(d["x"]).clone().requires_grad_(True)
    z = _t(d["z"]).clone().requires_grad_(True)
    src, dst, n = _t(d["src"]), _t(d["dst"]), int(d["num_nodes"])
    _close(O.unc_edge_norm(src, dst, n, "in"), d["norm"], 1e-6, 1e-6)
    bn = {}
    for m in ("nmlp", "emlp"):
        bn[m + ".1"] = {"running_mean": _t(d["b.%s.1.running_mean" % m]).clone(),
                        "running_var": _t(d["b.%s.1.running_var" % m]).clone()}
    act = str(d["activation"]) or None
    node_out, edge_out = O.dual_graph_conv(p, src, dst, _t(d["out_deg"]), x, z, _t(d["norm"]), None, bn,
                                           bool(d["bn_train"]), act)
    _close(node_out, d["node_out"], 2e-5, 2e-5)
    _close(edge_out, d["edge_out"], 2e-5, 2e-5)
    ((node_out * _t(d["wn"])).sum() + (edge_out * _t(d["we"])).sum()).backward()
    _close(x.grad, d["dx"], 2e-5, 2e-5)
    _close(z.grad, d["dz"], 2e-5, 2e-5)
    for k, v in p.items():
        if "g." + k in d:
            _close(v.grad, d["g." + k], rtol=2e-4, atol=2e-4)
    if bool(d["bn_train"]):  # running statistics updated like nn.BatchNorm1d (model.py:145-157)
        for m in ("nmlp", "emlp"):
            _close(bn[m + ".1"]["running_mean"], d["b_after.%s.1.running_mean" % m])
            _close(bn[m + ".1"]["running_var"], d["b_after.%s.1.running_var" % m])


@pytest.mark.parametrize("path", golden_files("rgnn_layer_"))
def test_relational_layer_oracle_matches_reference(path):
    """RGCNLayer / RGINLayer restatement (oracle/dmp_oracle.py::rel_layer) vs the reference's own run."""
    d = load_golden(path)
    kw = {str(k): eval(str(v)) for k, v in zip(d["kw_keys"], d["kw_vals"])}
    params = {k[2:]: _t(v).clone().requires_grad_(True) for k, v in d.items() if k.startswith("p.")}
    x = _t(d["x"]).clone().requires_grad_(True)
    kind = "rgin" if "rgin" in path else "rgcn"
    out = O.rel_layer(params, _t(d["src"]), _t(d["dst"]), _t(d["etype"]), x, kind, kw["num_rels"],
                      regularizer=kw.get("regularizer", "basis"), num_bases=kw.get("num_bases", -1),
                      edge_norm=kw.get("edge_norm", "in"), self_loop=kw.get("self_loop", True),
                      act_func=kw.get("act_func", "relu"))
    assert th.allclose(out, _t(d["out"]), rtol=1e-5, atol=1e-5)
    (out * _t(d["w"])).sum().backward()
    assert th.allclose(x.grad, _t(d["dx"]), rtol=1e-5, atol=1e-5)
    for k, p in params.items():
        assert th.allclose(p.grad, _t(d["g." + k]), rtol=1e-4, atol=1e-5), k

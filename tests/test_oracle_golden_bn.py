"""The CPU oracle's DMPLayer with BatchNorm1d in its MLPs (the reference constructor's default, models/dmpnn.py:17-28,45-60)
against ``bnlayer_dmp_*.npz``: outputs, gradients and the running statistics the reference's own layer leaves, in training
and evaluation mode (oracle/make_golden.py::gen_dmplayer_bn)."""
import numpy as np
import pytest
import torch as th

import dmp_oracle as O
from conftest import golden_files, load_golden


def _t(a):
    return th.from_numpy(np.asarray(a))


def _close(got, ref, tol=2e-5):
    got, ref = got.detach().double(), _t(ref).double()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("path", golden_files("bnlayer_dmp_"))
def test_oracle_layer_with_batch_norm_matches_reference(path):
    d = load_golden(path)
    train = str(d["mode"]) == "train"
    p = {k[2:]: _t(v).clone().requires_grad_(True) for k, v in d.items() if k.startswith("p.")}
    bn = {m + ".1": {"running_mean": _t(d["b0.%s.1.running_mean" % m]).clone(), "running_var": _t(d["b0.%s.1.running_var" % m]).clone()}
          for m in ("nmlp", "emlp")}
    x, z = _t(d["x"]).clone().requires_grad_(True), _t(d["z"]).clone().requires_grad_(True)
    node_out, edge_out, _, _ = O.dmp_layer(p, _t(d["src"]), _t(d["dst"]), _t(d["rev"]), _t(d["out_deg"]), x, z, str(d["act_func"]), 2,
                                           bn=bn, training=train)
    _close(node_out, d["node_out"])
    _close(edge_out, d["edge_out"])
    ((node_out * _t(d["wn"])).sum() + (edge_out * _t(d["we"])).sum()).backward()
    _close(x.grad, d["dx"])
    _close(z.grad, d["dz"])
    for k, v in p.items():
        if "g." + k in d:
            _close(v.grad, d["g." + k], 2e-4)
    for m in ("nmlp", "emlp"):
        for s in ("running_mean", "running_var"):
            _close(bn[m + ".1"][s], d["b1.%s.1.%s" % (m, s)], 1e-6)
            assert train or np.array_equal(d["b0.%s.1.%s" % (m, s)], d["b1.%s.1.%s" % (m, s)])   # evaluation mode leaves them alone

"""The oracle's restatement of the WHOLE counting model (oracle/model_oracle.py: masks, ScalarFilter gates, encodings,
embeddings, DMPNN rep-nets, reversed-edge masks, encodings / degrees joined to the reps, SumPredictNet heads and their
blend -- basemodel.py:1394-1661, pred.py:87-214 in the reference's op order) against the reference's own runs
(tests/golden/fullmodel_*.npz): every output of the 15-entry OutputDict and every parameter gradient.  No GPU."""
import json

import numpy as np
import pytest
import torch as th

import model_oracle as MO
from conftest import golden_files, load_golden

FIXTURES = [p for p in golden_files("fullmodel_") if "train" not in p and "compgcn" not in p]


def _t(a):
    return th.from_numpy(np.asarray(a))


def _side(d, t):
    rev = _t(d[t + "_edata.is_reversed"]).bool() if t + "_edata.is_reversed" in d else None
    return {"src": _t(d[t + "_src"]).long(), "dst": _t(d[t + "_dst"]).long(), "bnn": d[t + "_bnn"].tolist(), "bne": d[t + "_bne"].tolist(),
            "id": _t(d[t + "_ndata.id"]).long().view(-1), "label": _t(d[t + "_ndata.label"]).long().view(-1),
            "eid": _t(d[t + "_edata.id"]).long().view(-1), "elabel": _t(d[t + "_edata.label"]).long().view(-1), "rev": rev}


def _config(d):
    if "config_json" in d:
        return json.loads(str(d["config_json"]))
    return {str(k): eval(str(v)) for k, v in zip(d["config_keys"], d["config_vals"])}


@pytest.mark.parametrize("path", FIXTURES)
def test_model_oracle_matches_reference_run(path):
    d = load_golden(path)
    cfg = _config(d)
    assert cfg.get("rep_net", "DMPNN") == "DMPNN"
    sd = {k[3:]: _t(v).clone().requires_grad_(_t(v).is_floating_point()) for k, v in d.items() if k.startswith("sd.")}
    # shared sub-networks are ONE parameter under two names in the reference's state_dict (share_emb_net / share_rep_net):
    # alias them, so that the gradient of the shared parameter is the sum of both uses
    for k in list(sd):
        twin = "g_" + k[2:]
        if k.startswith("p_") and twin in sd and sd[k].shape == sd[twin].shape and th.equal(sd[k].detach(), sd[twin].detach()):
            sd[k] = sd[twin]
    out = MO.model_forward(sd, cfg, _side(d, "p"), _side(d, "g"))
    for k in ("p_v_mask", "p_e_mask", "g_v_mask", "g_e_mask"):
        assert th.equal(out[k], _t(d["out." + k])), k
    for k in ("p_v_emb", "p_e_emb", "g_v_emb", "g_e_emb", "p_v_rep", "p_e_rep", "g_v_rep", "g_e_rep", "pred_c"):
        ref = _t(d["out." + k])
        scale = max(1.0, float(ref.abs().max()))
        assert out[k].shape == ref.shape and float((out[k].detach() - ref).abs().max()) <= 2e-5 * scale, k
    total = out["pred_c"].sum()
    for k in ("pred_v", "pred_e"):
        if "out." + k in d:
            ref = _t(d["out." + k])
            assert float((out[k].detach() - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max())), k
            total = total + out[k].sum()
        else:
            assert out[k] is None
    total.backward()
    checked = 0
    for k, p in sd.items():
        if "grad." + k in d:
            ref = _t(d["grad." + k])
            assert p.grad is not None, k
            assert float((p.grad - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), k
            checked += 1
    assert checked > 20

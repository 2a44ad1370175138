"""``model.expand(**config)`` (models/basemodel.py:167-219) against the reference's own run: the
state_dict after growing the vocabularies must be identical.  Host-side module surgery, CPU only."""
import numpy as np
import torch as th

from conftest import golden_files, load_golden


def test_expand_matches_reference():
    from dualmessagepassing_amd.basemodel import build_model
    d = load_golden(golden_files("expand_")[0])
    config = {str(k): eval(str(v)) for k, v in zip(d["config_keys"], d["config_vals"])}
    grow = {str(k): int(v) for k, v in zip(d["grow_keys"], d["grow_vals"])}
    model = build_model(**config)
    model.load_state_dict({k[3:]: th.from_numpy(v) for k, v in d.items() if k.startswith("sd.")}, strict=True)
    model.expand(**dict(config, **grow))
    after = {k[9:]: v for k, v in d.items() if k.startswith("sd_after.")}
    sd = model.state_dict()
    assert sorted(sd) == sorted(after)
    for k, v in after.items():
        assert tuple(sd[k].shape) == v.shape, k
        assert np.array_equal(sd[k].numpy(), v), k
    assert [getattr(model, k) for k in sorted(grow)] == d["max_after"].tolist()
    assert model.p_rep_net is model.g_rep_net  # untouched, still shared


def test_expand_rolls_back_on_error():
    from dualmessagepassing_amd.basemodel import build_model
    d = load_golden(golden_files("expand_")[0])
    config = {str(k): eval(str(v)) for k, v in zip(d["config_keys"], d["config_vals"])}
    model = build_model(**config)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    try:
        model.expand(**dict(config, max_ngvl=99, emb_net="NoSuchEmbedding"))
        raise AssertionError("expected a ValueError")
    except ValueError:
        pass
    assert model.max_ngvl == config["max_ngvl"]
    assert all(th.equal(v, before[k]) for k, v in model.state_dict().items())

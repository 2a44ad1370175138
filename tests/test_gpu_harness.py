"""Training-loop smoke on the GPU (config 3 style): a few epochs on a synthetic labelled set with
exact counts; the count loss must fall and evaluation must run through DMPNN and CompGCN."""
import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rep_net", ["DMPNN", "CompGCN"])
def test_training_loop_reduces_count_loss(rep_net, gpu):
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.dp import FlatGradSync
    from dualmessagepassing_amd.harness import SyntheticPairs, evaluate_epoch, train_epoch
    ds = SyntheticPairs(128, 3, 2, 8, 16, 2, 1, seed=3)
    counts = np.array([s["counts"] for s in ds.samples], np.float64)
    assert counts.var() > 1.0
    th.manual_seed(0)
    model = build_model(**ds.model_config(hid_dim=32, layers=2, rep_net=rep_net)).to(gpu)
    sync = FlatGradSync(model)
    opt = th.optim.AdamW(sync.params, lr=2e-3, weight_decay=1e-5, amsgrad=True)
    before = evaluate_epoch(model, ds, 32, gpu)
    hist = [train_epoch(model, opt, ds, 32, gpu, sync=sync, bp_loss="MSE", neg_slp=0.01)["bp_loss"] for _ in range(60)]
    after = evaluate_epoch(model, ds, 32, gpu)
    assert np.isfinite(hist).all(), hist
    # better than the best constant predictor (the variance of the counts), i.e. it learnt from structure
    assert after["MSE"] < 0.7 * counts.var() and after["MSE"] < before["MSE"], (before["MSE"], after["MSE"], counts.var())
    assert after["pred"].shape == (128,) and np.array_equal(after["counts"].numpy(), counts.astype(np.float32))

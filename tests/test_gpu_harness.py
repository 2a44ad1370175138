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


def test_training_with_matching_losses(gpu):
    """train.py:627-661: node / edge matching losses on pred_v / pred_e against the batch's
    subisomorphism weights (computed on the device); the matching error must fall too."""
    import torch.nn.functional as F
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.dp import FlatGradSync
    from dualmessagepassing_amd.harness import SyntheticPairs, train_epoch
    ds = SyntheticPairs(64, 3, 2, 8, 16, 2, 1, seed=4)
    th.manual_seed(1)
    model = build_model(**ds.model_config(hid_dim=32, layers=2, pred_return_weights="node,edge")).to(gpu)
    sync = FlatGradSync(model)
    opt = th.optim.AdamW(sync.params, lr=2e-3, weight_decay=1e-5, amsgrad=True)

    def match_error():
        model.eval()
        with th.no_grad():
            pattern, graph, counts, (nw, ew) = ds.batchify(np.arange(64), gpu, return_weights="node,edge")
            out = model(pattern, graph)
            assert out["pred_v"].shape == nw.shape and out["pred_e"].shape == ew.shape
            # every subisomorphism touches pattern_nodes target nodes / pattern_edges forward target edges
            assert th.equal(nw.sum(1).float(), counts.view(-1) * 3)
            return float(F.mse_loss(out["pred_v"].masked_fill(~out["g_v_mask"], 0), nw.float())
                         + F.mse_loss(out["pred_e"].masked_fill(~out["g_e_mask"], 0), ew.float()))

    before = match_error()
    hist = [train_epoch(model, opt, ds, 32, gpu, sync=sync, neg_slp=0.01, match_loss_w=1.0, match_reg_w=0.1)["bp_loss"]
            for _ in range(40)]
    after = match_error()
    assert np.isfinite(hist).all() and after < 0.8 * before, (before, after, hist[::8])

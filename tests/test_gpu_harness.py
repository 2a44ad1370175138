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


def test_fit_from_dataset_directory(tmp_path, gpu):
    """Config-3 style run: dataset written in the reference's directory layout, loaded back, split
    by its rule, trained with checkpoints / config / log in the reference's run-directory form."""
    import os
    from dualmessagepassing_amd import dataio
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.dp import FlatGradSync
    from dualmessagepassing_amd.harness import PairDataset, SyntheticPairs, evaluate_epoch, fit
    root, run = os.path.join(tmp_path, "data"), os.path.join(tmp_path, "run")
    SyntheticPairs(120, 3, 2, 8, 16, 2, 1, seed=6).to_files(root)
    data, _ = dataio.load_data(os.path.join(root, "patterns"), os.path.join(root, "graphs"), os.path.join(root, "metadata"))
    full = PairDataset.from_loaded(data["train"] + data["dev"] + data["test"])
    n_tr, n_dev = len(data["train"]), len(data["dev"])
    train, dev = full.subset(range(n_tr)), full.subset(range(n_tr, n_tr + n_dev))
    assert (len(train), len(dev)) == (96, 12)
    th.manual_seed(0)
    config = full.model_config(hid_dim=32, layers=2)
    model = build_model(**config).to(gpu)
    sync = FlatGradSync(model)
    opt = th.optim.AdamW(sync.params, lr=2e-3, weight_decay=1e-5, amsgrad=True)
    hist = fit(model, opt, train, dev, 25, 32, gpu, save_dir=run, config=config, sync=sync, neg_slp=0.01)
    assert len(hist) == 25 and hist[-1]["train"]["bp_loss"] < hist[0]["train"]["bp_loss"]
    assert dataio.load_config(os.path.join(run, "config.json"))["hid_dim"] == 32
    best = dataio.get_best_epochs(os.path.join(run, "log.txt"))["eval-MAE"]["dev"]
    assert best[1] == pytest.approx(min(h["dev"]["eval_metric"] for h in hist), abs=1e-4)
    # the best epoch's checkpoint reproduces its dev metric (evaluate.py loads it the same way, train.py:97-106)
    again = build_model(**dataio.load_config(os.path.join(run, "config.json"))).to(gpu)
    again.load_state_dict(th.load(dataio.checkpoint_path(run, best[0]), map_location=gpu))
    assert evaluate_epoch(again, dev, 32, gpu)["eval_metric"] == pytest.approx(best[1], abs=1e-4)
    # evaluate.py for the run directory: best checkpoint -> per-split result files and "best" lines
    import json
    from dualmessagepassing_amd.harness import evaluate_run
    test = full.subset(range(n_tr + n_dev, len(full)))
    res = evaluate_run(run, {"dev": dev, "test": test}, gpu, batch_size=32, stamp="t")
    assert res["dev"]["eval_metric"] == pytest.approx(best[1], abs=1e-4) and np.isfinite(res["test"]["MAE"])
    saved = json.load(open(os.path.join(run, "eval_test_results_t.json")))
    assert len(saved["prediction"]["pred_c"]) == len(test) == len(saved["data"]["counts"])
    assert saved["error"]["MAE"] == pytest.approx(res["test"]["MAE"])
    assert saved["data"]["counts"] == [float(x["counts"]) for x in test.samples]
    assert dataio.get_best_epochs(os.path.join(run, "log.txt"))["eval-MAE"]["test"][1] == pytest.approx(res["test"]["eval_metric"], abs=1e-3)


@pytest.mark.parametrize("amsgrad", [False, True])
@pytest.mark.parametrize("n", [1, 7, 4096, 600_001])
def test_flat_adamw_matches_torch_adamw(n, amsgrad, gpu):
    """dp.FlatAdamW (one HIP launch, dmp_adamw_step) against torch.optim.AdamW on the same gradients:
    the update of train.py:1231, including bias correction over several steps and a changed lr."""
    from dualmessagepassing_amd.dp import FlatAdamW
    gen = th.Generator().manual_seed(n)
    p0 = th.randn(n, generator=gen)
    a = th.nn.Parameter(p0.clone().to(gpu))
    b = th.nn.Parameter(p0.clone().to(gpu))
    kw = dict(lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2, amsgrad=amsgrad)
    oa, ob = FlatAdamW([a], **kw), th.optim.AdamW([b], **kw)
    for it in range(6):
        g = (th.randn(n, generator=gen) * (10.0 if it == 2 else 1.0)).to(gpu)   # a spike: AMSGrad's max matters afterwards
        a.grad, b.grad = g.clone(), g.clone()
        if it == 4:
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 1e-3
        oa.step()
        ob.step()
        assert th.allclose(a.data, b.data, rtol=2e-6, atol=2e-7), (it, (a.data - b.data).abs().max().item())
    st = oa.state[a]
    assert st["step"] == 6 and ("max_exp_avg_sq" in st) == amsgrad
    c = th.nn.Parameter(th.zeros(4))
    c.grad = th.zeros(4)
    with pytest.raises(Exception, match="GPU only"):
        FlatAdamW([c]).step()                                                       # CPU tensors are refused


@pytest.mark.parametrize("holes", [3, 40])
def test_flat_adamw_skips_parameters_without_gradient_like_torch(holes, gpu):
    """torch.optim.AdamW (the reference's optimizer, train.py:1231) leaves a parameter whose ``.grad`` is None untouched:
    no weight decay, no moment update.  The flat optimizer does the same through FlatGradSync.pack()'s bookkeeping -- one
    launch with the dead slices as skip ranges (<= 16), separate launches per live run beyond that."""
    from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync

    class Net(th.nn.Module):
        def __init__(self):
            super().__init__()
            th.manual_seed(holes)
            self.used = th.nn.ParameterList([th.nn.Parameter(th.randn(5 + 3 * i)) for i in range(holes + 2)])
            self.unused = th.nn.ParameterList([th.nn.Parameter(th.randn(2 + i)) for i in range(holes)])
            self._order = [p for pair in zip(self.used, list(self.unused) + [None, None]) for p in pair if p is not None]

        def parameters(self, recurse=True):          # used / unused interleaved in the flat buffer
            return iter(self._order)

        def forward(self):
            return sum((p * p).sum() * (i + 1) for i, p in enumerate(self.used))

    a, b = Net().to(gpu), Net().to(gpu)
    a._order = [p for pair in zip(a.used, list(a.unused) + [None, None]) for p in pair if p is not None]
    b._order = [p for pair in zip(b.used, list(b.unused) + [None, None]) for p in pair if p is not None]
    sync = FlatGradSync(a)
    oa = FlatAdamW([sync.flatten_parameters()], lr=1e-2, weight_decay=0.1, amsgrad=True)
    ob = th.optim.AdamW(list(b.parameters()), lr=1e-2, weight_decay=0.1, amsgrad=True)
    for _ in range(4):
        sync.detach_grads()
        a().backward()
        sync.pack()
        oa.step()
        ob.zero_grad(set_to_none=True)
        b().backward()
        ob.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert th.allclose(p.data, q.data, rtol=2e-6, atol=2e-7)
    for p, q in zip(a.unused, Net().unused):       # untouched: exactly the initial values (no decay)
        assert th.equal(p.data.cpu(), q.data)


@pytest.mark.gpu
@pytest.mark.parametrize("count", [1, 7, 128, 300])
def test_pack_segments_bit_exact(count, gpu):
    """dmp_pack_segments (one launch per 128 arrays) against slice copies: ragged lengths incl. 0 and 1, sources that
    are not 16-byte aligned, more arrays than one launch takes."""
    import ctypes
    from dualmessagepassing_amd import _lib
    lib = _lib.load()
    g = th.Generator().manual_seed(count)
    lens = [int(x) for x in th.randint(0, 5000, (count,), generator=g)]
    lens[0] = 70001
    if count > 2:
        lens[1], lens[2] = 0, 1
    pool = th.randn(sum(lens) + 3 * count + 8, generator=g).cuda()
    srcs, pos = [], 0
    for i, n in enumerate(lens):
        pos += i % 4                                   # misalign most of the sources
        srcs.append(pool[pos:pos + n])
        pos += n
    offs, off = [], 0
    for n in lens:
        offs.append(off)
        off += (n + 3) // 4 * 4
    flat = th.full((off + 4,), -7.0, device="cuda")
    want = flat.clone()
    for s, o, n in zip(srcs, offs, lens):
        want[o:o + n] = s
    P = (ctypes.c_void_p * count)(*[s.data_ptr() if s.numel() else None for s in srcs])
    O, N = (ctypes.c_int64 * count)(*offs), (ctypes.c_int64 * count)(*lens)
    _lib.check(lib.dmp_pack_segments(P, O, N, count, 0, flat.data_ptr(), _lib.stream_ptr()), "dmp_pack_segments")
    assert th.equal(flat, want)
    # segments WITHOUT a source are cleared (a parameter without a gradient), and with pad_to_4 so is every segment's padding up
    # to the next 16-byte piece -- nothing else: the four floats behind the last segment keep their value
    absent = [i for i in range(count) if i % 3 == 1]
    for i in absent:
        P[i] = None
        want[offs[i]:offs[i] + lens[i]] = 0.0
    flat.fill_(-7.0)
    _lib.check(lib.dmp_pack_segments(P, O, N, count, 0, flat.data_ptr(), _lib.stream_ptr()), "dmp_pack_segments")
    assert th.equal(flat, want)
    for o, n in zip(offs, lens):
        want[o + n:o + (n + 3) // 4 * 4] = 0.0
    flat.fill_(-7.0)
    _lib.check(lib.dmp_pack_segments(P, O, N, count, 1, flat.data_ptr(), _lib.stream_ptr()), "dmp_pack_segments")
    assert th.equal(flat, want) and bool((flat[off:] == -7.0).all())
    O[0] = 2                                           # destination slices must start on 16-byte boundaries
    assert lib.dmp_pack_segments(P, O, N, count, 0, flat.data_ptr(), _lib.stream_ptr()) != 0


@pytest.mark.gpu
def test_flat_grad_sync_pack_on_gpu(gpu):
    """FlatGradSync.pack() on device tensors: fresh gradients land in their slices, a missing one leaves zeros."""
    from dualmessagepassing_amd.dp import FlatGradSync
    ps = [th.nn.Parameter(th.randn(s, device="cuda")) for s in [(128, 128), (128,), (1,), (3, 5), (1, 132)]]
    holder = th.nn.Module()
    holder.ps = th.nn.ParameterList(ps)
    sync = FlatGradSync(holder)
    for rnd in range(2):
        sync.detach_grads()
        gs = [th.randn_like(p) for p in ps]
        for i, (p, g_) in enumerate(zip(ps, gs)):
            if not (rnd == 1 and i == 3):
                p.grad = g_.t().contiguous().t() if g_.dim() == 2 and rnd == 0 else g_
        sync.pack()
        for i, (p, g_, o) in enumerate(zip(ps, gs, sync.offsets)):
            got = sync.flat[o:o + p.numel()].view_as(p)
            assert p.grad.data_ptr() == got.data_ptr()
            assert th.equal(got, th.zeros_like(g_) if (rnd == 1 and i == 3) else g_)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 64, 1024, 5000])
@pytest.mark.parametrize("kind,slope", [("MSE", 1.0), ("MSE", 0.18), ("MAE", 0.0), ("SMSE", 0.5)])
def test_count_loss_in_one_launch_equals_the_tensor_ops(n, kind, slope, gpu):
    """``harness.count_loss`` (``dmp_count_loss``: ``bp_crit(leaky_relu(pred, slope), counts)``, its mean and the seed of the
    backward in one launch, train.py:624-628) against the five tensor ops: the loss to fp32 accuracy of a mean over n terms,
    the gradient element by element (same formula, one rounding apart), scaled by the upstream gradient; bit-stable; inputs the
    launch does not take (a target that needs a gradient, other shapes) go through the tensor ops."""
    import torch.nn.functional as F
    from dualmessagepassing_amd import harness
    g = th.Generator().manual_seed(n)
    pred = (th.randn(n, 1, generator=g) * 3).to(gpu).requires_grad_(True)
    target = th.randint(0, 6, (n, 1), generator=g).float().to(gpu)
    crit = {"MSE": F.mse_loss, "MAE": F.l1_loss, "SMSE": F.smooth_l1_loss}[kind]
    ref = crit(F.leaky_relu(pred, slope), target)
    (ref * 1.7).backward()
    want, pred.grad = pred.grad.clone(), None
    got = harness.count_loss(pred, target, kind, slope)
    assert got.shape == ref.shape and isinstance(got.grad_fn, th.autograd.function.BackwardCFunction)
    (got * 1.7).backward()
    assert abs(float(got) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert pred.grad.shape == want.shape
    assert float((pred.grad - want).abs().max()) <= 2e-6 * max(1e-6, float(want.abs().max()))
    again = harness.count_loss(pred.detach().requires_grad_(True), target, kind, slope)
    assert th.equal(again, got)
    flat = harness.count_loss(pred.view(-1), target.view(-1), kind, slope)           # 1-D views of both: the same launch
    assert th.equal(flat, got)
    other = harness.count_loss(pred, target.clone().requires_grad_(True), kind, slope)     # not the launch's case: tensor ops
    assert not isinstance(other.grad_fn, th.autograd.function.BackwardCFunction)
    assert abs(float(other) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))

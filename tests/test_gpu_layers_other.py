"""GPU parity of the sibling layers on the same kernels: CompGCNLayer (SCM) and DualGraphConv
(UNC), against the reference's golden vectors.  fp32 tolerances as in test_gpu_dmplayer.py."""
import numpy as np
import pytest
import torch as th

from conftest import golden_files, load_golden

pytestmark = pytest.mark.gpu


def _t(a):
    return th.from_numpy(np.asarray(a))


def _close(got, ref, rtol=1e-5, atol=1e-5, what=""):
    got, ref = got.detach().double().cpu(), _t(ref).double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(1.0, float(ref.abs().max())) if ref.numel() else 1.0
    err = float((got - ref).abs().max()) if ref.numel() else 0.0
    # SURVEY 8(c) / BASELINE.md 3: |delta| <= 1e-5 * max(1, |ref|_max) -- rtol = atol = 1e-5 against the array's scale
    assert err <= max(atol, rtol * scale), "%s: max err %g (scale %g)" % (what, err, scale)


@pytest.mark.parametrize("path", golden_files("compgcn_"))
def test_compgcn_layer_matches_reference_golden(path, gpu):
    from dualmessagepassing_amd.compgcn import CompGCNLayer
    from dualmessagepassing_amd.graph import BatchedGraph
    d = load_golden(path)
    h = d["x"].shape[1]
    layer = CompGCNLayer(h, h, self_loop=bool(d["self_loop"]), comp_opt=str(d["comp_opt"]),
                         edge_norm=str(d["edge_norm"]), bias=True, batch_norm=False, act_func="relu")
    layer.load_state_dict({k[2:]: _t(v) for k, v in d.items() if k.startswith("p.")}, strict=True)
    layer.to(gpu)
    g = BatchedGraph(_t(d["src"]).to(gpu), _t(d["dst"]).to(gpu), int(d["num_nodes"]))
    if "rev" in d:
        g.edata["is_reversed"] = _t(d["rev"]).to(gpu)
    x = _t(d["x"]).to(gpu).requires_grad_(True)
    z = _t(d["z"]).to(gpu).requires_grad_(True)
    node_out, edge_out = layer(g, x, z)
    if "norm" in d:
        _close(g.edata["norm"], d["norm"], 1e-6, 1e-6, "norm")
    _close(node_out, d["node_out"], what="node_out")
    _close(edge_out, d["edge_out"], what="edge_out")
    ((node_out * _t(d["wn"]).to(gpu)).sum() + (edge_out * _t(d["we"]).to(gpu)).sum()).backward()
    _close(x.grad, d["dx"], what="dx")
    _close(z.grad, d["dz"], what="dz")
    for k, p in layer.named_parameters():
        if "g." + k in d:
            _close(p.grad, d["g." + k], 2e-4, 2e-4, "grad " + k)


@pytest.mark.parametrize("path", golden_files("unc_dualconv_"))
def test_unc_dual_graph_conv_matches_reference_golden(path, gpu):
    from dualmessagepassing_amd.graph import BatchedGraph
    from dualmessagepassing_amd.unc import DualGraphConv, compute_edgenorm
    d = load_golden(path)
    h = d["x"].shape[1]
    act = th.nn.Tanh() if str(d["activation"]) == "tanh" else None
    layer = DualGraphConv(h, h, activation=act, dropout=0.0)
    sd = {k[2:]: _t(v) for k, v in d.items() if k.startswith("p.")}
    sd.update({k[2:]: _t(v) for k, v in d.items() if k.startswith("b.")})
    layer.load_state_dict(sd, strict=True)  # incl. the unused nfc/efc and the BN buffers
    layer.to(gpu).train(bool(d["bn_train"]))
    g = BatchedGraph(_t(d["src"]).to(gpu), _t(d["dst"]).to(gpu), int(d["num_nodes"]))
    norm = compute_edgenorm(g)
    _close(norm, d["norm"], 1e-6, 1e-6, "edge norm")
    x = _t(d["x"]).to(gpu).requires_grad_(True)
    z = _t(d["z"]).to(gpu).requires_grad_(True)
    node_out, edge_out = layer(g, x, z, norm)
    assert th.equal(g.ndata["out_deg"].cpu(), _t(d["out_deg"]))
    _close(node_out, d["node_out"], 2e-5, 2e-5, "node_out")
    _close(edge_out, d["edge_out"], 2e-5, 2e-5, "edge_out")
    ((node_out * _t(d["wn"]).to(gpu)).sum() + (edge_out * _t(d["we"]).to(gpu)).sum()).backward()
    _close(x.grad, d["dx"], 2e-5, 2e-5, "dx")
    _close(z.grad, d["dz"], 2e-5, 2e-5, "dz")
    for k, p in layer.named_parameters():
        if "g." + k in d:
            _close(p.grad, d["g." + k], 2e-4, 2e-4, "grad " + k)
        else:
            assert p.grad is None  # nfc / efc / out_weight: unused, exactly as in the reference
    if bool(d["bn_train"]):
        for k, b in layer.named_buffers():
            if "num_batches" not in k:
                _close(b, d["b_after." + k], what=k)


def test_build_graph_from_triplets_and_norm(gpu):
    """utils.py:473-491 semantics: sort by (src,dst,rel), forward + reversed copies, types, 1/in-degree."""
    from dualmessagepassing_amd.unc import build_graph_from_triplets
    trip = np.array([[2, 0, 1], [0, 1, 2], [0, 0, 1], [2, 1, 1], [1, 0, 0]])
    g = build_graph_from_triplets(3, 2, trip, gpu)
    u, v = g.all_edges()
    assert u.cpu().tolist() == [0, 0, 1, 2, 2, 1, 2, 0, 1, 1] and v.cpu().tolist() == [1, 2, 0, 1, 1, 0, 0, 1, 2, 2]
    assert g.edata["type"].cpu().tolist() == [0, 1, 0, 0, 1, 2, 3, 2, 2, 3]
    indeg = np.bincount(v.cpu().numpy(), minlength=3)
    assert np.allclose(g.edata["norm"].cpu().numpy()[:, 0], 1.0 / indeg[v.cpu().numpy()])


def test_unc_dmpnn_model_matches_reference_golden(gpu):
    """Whole UNC DMPNN (embeddings -> 2 x DualGraphConv with Tanh between -> per-relation means)."""
    from dualmessagepassing_amd.graph import BatchedGraph
    from dualmessagepassing_amd.unc import DMPNN
    d = load_golden(golden_files("unc_model")[0])
    n, h, nrel = int(d["num_nodes"]), int(d["hid"]), int(d["num_rels"])
    model = DMPNN(None, None, n, h, h, nrel, 2, 0.0)
    sd = {k[2:]: _t(v) for k, v in d.items() if k.startswith("p.")}
    sd.update({k[2:]: _t(v) for k, v in d.items() if k.startswith("b.")})
    model.load_state_dict(sd, strict=True)
    model.to(gpu).eval()
    g = BatchedGraph(_t(d["src"]).to(gpu), _t(d["dst"]).to(gpu), n)
    hh, zz, rr = model(g, th.arange(n, device=gpu), _t(d["etype"]).to(gpu), _t(d["norm"]).to(gpu))
    _close(hh, d["h"], 5e-5, 5e-5, "h")
    _close(zz, d["z"], 5e-5, 5e-5, "z")
    _close(rr, d["r"], 5e-5, 5e-5, "r")
    ((hh * _t(d["w1"]).to(gpu)).sum() + (zz * _t(d["w2"]).to(gpu)).sum() + (rr * _t(d["w3"]).to(gpu)).sum()).backward()
    for k, p in model.named_parameters():
        if "g." + k in d:
            _close(p.grad, d["g." + k], 3e-4, 3e-4, "grad " + k)
        else:
            assert p.grad is None, k


@pytest.mark.parametrize("tag", ["unsup", "sup"])
def test_unc_train_model_losses_match_reference(tag, gpu):
    """UNC TrainModel (model.py:631-744): encoder + DistMult / node-classification head + regulariser;
    loss value and every parameter gradient against the reference's own run."""
    from dualmessagepassing_amd.graph import BatchedGraph
    from dualmessagepassing_amd.unc import TrainModel
    d = load_golden(golden_files("unc_train_" + tag)[0])
    n, h, nrel, nlabel = int(d["num_nodes"]), int(d["hid"]), int(d["num_rels"]), int(d["nlabel"])
    tm = TrainModel(None, n, h, nrel, nlabel, num_hidden_layers=2, dropout=0.0, reg_param=0.01)
    if nlabel:
        tm.w_relation = th.nn.Parameter(th.zeros(nrel, h))
    sd = {k[2:]: _t(v) for k, v in d.items() if k.startswith("p.") or k.startswith("b.")}
    tm.load_state_dict(sd, strict=True)
    tm.to(gpu).eval()
    g = BatchedGraph(_t(d["src"]).to(gpu), _t(d["dst"]).to(gpu), n)
    r = _t(d["etype"]).to(gpu)
    emb, pred = tm(g, th.arange(n, device=gpu), r, _t(d["norm"]).to(gpu))
    if nlabel == 0:
        assert pred is None
        loss = tm.get_unsupervised_loss(g, emb, r, _t(d["triplets"]).to(gpu), _t(d["labels"]).to(gpu))
    else:
        _close(pred, d["pred"], 5e-5, 5e-5, "pred")
        loss = tm.get_supervised_loss(g, emb, r, pred, _t(d["matched_labels"]).to(gpu), _t(d["matched_index"]).to(gpu), False)
    _close(loss, d["loss"], 2e-5, 2e-5, "loss")
    loss.backward()
    seen = 0
    for k, p in tm.named_parameters():
        if "g." + k in d:
            _close(p.grad, d["g." + k], 3e-4, 3e-6, "grad " + k)
            seen += 1
        else:
            assert p.grad is None, k
    assert seen > 20


def test_keyed_segment_pool_equals_masked_sums(gpu):
    """PoolIndex.from_keys: sums of rows by an arbitrary key (relation types incl. an unused one),
    forward and backward, against per-key masked sums in fp64; run-to-run bitwise stable."""
    from dualmessagepassing_amd.ops import PoolIndex, seg_pool
    gen = th.Generator().manual_seed(3)
    E, H, K = 5000, 48, 7
    keys = th.randint(0, K - 1, (E,), generator=gen).to(gpu)          # key K-1 never occurs
    x = th.randn(E, H, generator=gen).to(gpu).requires_grad_(True)
    pool = PoolIndex.from_keys(keys, K)
    out = seg_pool(x, pool)
    want = th.stack([x.detach().double()[keys == k].sum(0) for k in range(K)])
    assert out.shape == (K, H) and float(out[K - 1].detach().abs().max()) == 0.0
    assert th.allclose(out.double(), want, rtol=1e-6, atol=1e-5)
    assert th.equal(out, seg_pool(x, pool))
    w = th.randn(K, H, generator=gen).to(gpu)
    (out * w).sum().backward()
    assert th.equal(x.grad, w[keys])


def test_keyed_segment_pool_with_a_single_key_is_the_plain_column_sum(gpu):
    """``PoolIndex.from_keys(keys, 1)`` (one relation type: no sort, the contiguous-range build) -- the sum of all rows, its
    backward, and ``take_rows_small_table`` of a one-row table (UNC's ``w_relation`` on a one-relation graph)."""
    from dualmessagepassing_amd import ops
    gen = th.Generator().manual_seed(4)
    for E in (1, 63, 5000, 24000):
        keys = th.zeros(E, dtype=th.int64, device=gpu)
        x = th.randn(E, 64, generator=gen).to(gpu).requires_grad_(True)
        pool = ops.PoolIndex.from_keys(keys, 1)
        out = ops.seg_pool(x, pool)
        assert out.shape == (1, 64)
        assert th.allclose(out.double(), x.detach().double().sum(0, keepdim=True), rtol=1e-6, atol=1e-5 * max(1.0, E ** 0.5))
        w = th.randn(1, 64, generator=gen).to(gpu)
        (out * w).sum().backward()
        assert th.equal(x.grad, w.expand(E, 64))
        table = th.randn(1, 64, generator=gen).to(gpu).requires_grad_(True)
        rows = ops.take_rows_small_table(table, keys)
        assert th.equal(rows, table.detach().expand(E, 64))
        d = th.randn(E, 64, generator=gen).to(gpu)
        rows.backward(d)
        assert th.allclose(table.grad.double(), d.double().sum(0, keepdim=True), rtol=1e-6, atol=1e-5 * max(1.0, E ** 0.5))

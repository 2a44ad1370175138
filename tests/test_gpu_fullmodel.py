"""GPU parity of the whole ``DMPNN(**config).forward(pattern, graph)`` (model skeleton + HIP hot
path) against the reference's own model run (tests/golden/fullmodel_*.npz): all 15 OutputDict
entries and the gradients of ``pred_c.sum()``.  Tolerances: embeddings 1e-5, 3-layer reps and
``pred_c`` 1e-4, parameter gradients 2e-4 (all relative to max(1, |ref|max))."""
import numpy as np
import pytest
import torch as th

from conftest import golden_files, load_golden

pytestmark = pytest.mark.gpu


def _t(a):
    return th.from_numpy(np.asarray(a))


def _close(got, ref, tol, what):
    got, ref = got.detach().double().cpu(), _t(ref).double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(1.0, float(ref.abs().max())) if ref.numel() else 1.0
    err = float((got - ref).abs().max()) if ref.numel() else 0.0
    assert err <= tol * scale, "%s: max err %g (scale %g)" % (what, err, scale)


def _graph(d, t, dev):
    from dualmessagepassing_amd.graph import BatchedGraph
    g = BatchedGraph(_t(d[t + "_src"]).to(dev), _t(d[t + "_dst"]).to(dev), int(d[t + "_num_nodes"]),
                     _t(d[t + "_bnn"]).to(dev), _t(d[t + "_bne"]).to(dev))
    for k, v in d.items():
        if k.startswith(t + "_ndata."):
            g.ndata[k.split(".", 1)[1]] = _t(v).to(dev)
        if k.startswith(t + "_edata."):
            g.edata[k.split(".", 1)[1]] = _t(v).to(dev)
    return g


FORWARD_FIXTURES = [p for p in golden_files("fullmodel_") if "train" not in p]


@pytest.mark.parametrize("path", FORWARD_FIXTURES)
@pytest.mark.parametrize("fused", [True, False])
def test_full_dmpnn_forward_matches_reference(path, fused, gpu):
    from dualmessagepassing_amd.basemodel import build_model
    d = load_golden(path)
    if "config_json" in d:    # the reference's own run configuration (config.py defaults + README "Complex" command line)
        import json
        model = build_model(json.loads(str(d["config_json"])), init_neigenv=6.0, init_eeigenv=5.0)
    else:
        config = {str(k): eval(str(v)) for k, v in zip(d["config_keys"], d["config_vals"])}
        model = build_model(**config)
    sd = {k[3:]: _t(v) for k, v in d.items() if k.startswith("sd.")}
    missing, unexpected = model.load_state_dict(sd, strict=True)  # the reference's own checkpoint keys
    assert not missing and not unexpected
    model.to(gpu)
    model.use_fused = fused
    pattern, graph = _graph(d, "p", gpu), _graph(d, "g", gpu)
    if fused and "default" in path:   # the shipped activation (leaky_relu) must be on the fused fast path
        from dualmessagepassing_amd import dmpnn
        hits = []
        orig = dmpnn.DMPLayer.forward_fused
        dmpnn.DMPLayer.forward_fused = lambda self, *a, **k: (hits.append(1), orig(self, *a, **k))[1]
        try:
            out = model(pattern, graph)
        finally:
            dmpnn.DMPLayer.forward_fused = orig
        assert len(hits) == 3, "the rep-net did not take the fused path"
    else:
        out = model(pattern, graph)
    assert list(out.keys()) == ["p_v_emb", "p_e_emb", "g_v_emb", "g_e_emb", "p_v_rep", "p_e_rep", "g_v_rep", "g_e_rep",
                                "p_v_mask", "p_e_mask", "g_v_mask", "g_e_mask", "pred_c", "pred_v", "pred_e"]
    for k in ("p_v_mask", "p_e_mask", "g_v_mask", "g_e_mask"):
        assert th.equal(out[k].cpu(), _t(d["out." + k])), k          # boolean, exact
    for k in ("p_v_emb", "p_e_emb", "g_v_emb", "g_e_emb"):
        _close(out[k], d["out." + k], 1e-5, k)                       # single-op outputs 1e-5
    for k in ("p_v_rep", "p_e_rep", "g_v_rep", "g_e_rep", "pred_c"):
        _close(out[k], d["out." + k], 1e-4, k)                       # ... 3-layer reps and pred_c 1e-4
    total = out["pred_c"].sum()
    for k in ("pred_v", "pred_e"):
        if "out." + k in d:   # pred_return_weights: per-node / per-edge matching outputs [B, max_len]
            _close(out[k], d["out." + k], 1e-4, k)
            total = total + out[k].sum()
        else:
            assert out[k] is None
    total.backward()
    n = 0
    for k, p in model.named_parameters():
        if "grad." + k in d:
            _close(p.grad, d["grad." + k], 2e-4, "grad " + k)           # ... parameter gradients 2e-4
            n += 1
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
    assert n > 20


@pytest.mark.parametrize("fused", [True, False])
def test_training_steps_match_reference(fused, gpu):
    """Four optimizer steps of the count loss (MSE on pred_c, AdamW(amsgrad), gradient clipping:
    train.py:624-686) reproduce the reference model's loss trajectory and final parameters;
    gradients go through dp.FlatGradSync as in bench.py."""
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.dp import FlatGradSync
    d = load_golden([p for p in golden_files("fullmodel_") if "train" in p][0])
    config = {str(k): eval(str(v)) for k, v in zip(d["config_keys"], d["config_vals"])}
    model = build_model(**config)
    model.load_state_dict({k[3:]: _t(v) for k, v in d.items() if k.startswith("sd.")}, strict=True)
    model.to(gpu)
    model.use_fused = fused
    pattern, graph = _graph(d, "p", gpu), _graph(d, "g", gpu)
    counts = _t(d["train_counts"]).to(gpu)
    sync = FlatGradSync(model)
    opt = th.optim.AdamW(sync.params, lr=1e-3, weight_decay=1e-5, amsgrad=True)
    losses = []
    for _ in range(len(d["train_losses"])):
        sync.detach_grads()
        loss = th.nn.functional.mse_loss(model(pattern, graph)["pred_c"].view(-1), counts)
        loss.backward()
        sync.pack()
        sync.sync()
        th.nn.utils.clip_grad_norm_(sync.params, 8.0)
        opt.step()
        losses.append(float(loss.detach()))
    ref = d["train_losses"]
    assert np.allclose(losses, ref, rtol=2e-3), (losses, ref.tolist())
    after = {k[9:]: v for k, v in d.items() if k.startswith("sd_after.")}
    for k, v in model.state_dict().items():
        _close(v, after[k], 2e-3, "param after training " + k)


@pytest.mark.parametrize("act", ["relu", "leaky_relu"])
@pytest.mark.parametrize("kind", ["SumPredictNet", "MeanPredictNet"])
def test_pooled_head_single_node_matches_op_by_op(kind, act, gpu):
    """pred._PooledHead (the pooled head as one hand-written autograd node) against the same algebra recorded op by
    op: prediction and every input / parameter gradient."""
    from dualmessagepassing_amd.pred import PRED_NETS
    th.manual_seed(5)
    B, d, h = 257, 128, 128
    net = PRED_NETS[kind](d, h, act_func=act).to(gpu)
    for p in net.parameters():
        p.data.normal_(0.0, 0.2)
    gen = th.Generator().manual_seed(6)
    ps = th.randn(B, d, generator=gen).to(gpu).requires_grad_(True)
    gs = th.randn(B, d, generator=gen).to(gpu).requires_grad_(True)
    pl = th.randint(1, 9, (B, 1), generator=gen).float().to(gpu)
    gl = th.randint(1, 65, (B, 1), generator=gen).float().to(gpu)
    Lp, Lg = 8, 64
    y, _ = net.forward_pooled(ps, Lp, pl, gs, Lg, gl)

    def reference():
        if net.pool_kind == "sum":
            p = th.nn.functional.linear(ps, net.p_fc.weight) + float(Lp) * net.p_fc.bias
            g = th.nn.functional.linear(gs, net.g_fc.weight) + float(Lg) * net.g_fc.bias
        else:
            p, g = net.p_fc(ps / float(Lp)), net.g_fc(gs / float(Lg))
        f = th.cat([p, g, g - p, g * p, pl, gl, 1.0 / pl, 1.0 / gl], dim=1)
        y1 = net.act(net.pred_fc1(f))
        return net.pred_fc2(th.cat([y1, pl, gl, 1.0 / pl, 1.0 / gl], dim=1))
    ref = reference()
    assert y.shape == ref.shape and th.allclose(y, ref, rtol=1e-5, atol=1e-5)
    cot = th.randn(B, 1, generator=gen).to(gpu)
    wrt = [ps, gs] + list(net.parameters())
    got = th.autograd.grad((y * cot).sum(), wrt)
    want = th.autograd.grad((ref * cot).sum(), wrt)
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape and th.allclose(a, b, rtol=1e-4, atol=1e-4), (i, (a - b).abs().max().item())


@pytest.mark.parametrize("width", [128, 64])
@pytest.mark.parametrize("act", ["relu", "leaky_relu"])
@pytest.mark.parametrize("B", [1, 257, 1024])
@pytest.mark.parametrize("n_heads", [1, 2, -2])
def test_hip_heads_match_op_by_op(B, n_heads, act, width, gpu):
    """pred._PooledHeadsHIP (all heads + blend: one launch forward, two backward) against the op-by-op algebra:
    the blended prediction and every input / parameter gradient."""
    from dualmessagepassing_amd.pred import PRED_NETS, _PooledHeadsHIP
    by_len, n_heads = n_heads < 0, abs(n_heads)      # -2: the blend weights come out of the op itself (dmp_heads_blend)
    th.manual_seed(7 + B)
    d = h = width
    nets = [PRED_NETS["SumPredictNet"](d, h, act_func=act).to(gpu) for _ in range(n_heads)]
    assert all(net.hip_head_ok(th.empty(2, d, device=gpu)) for net in nets)
    gen = th.Generator().manual_seed(8)
    for net in nets:
        for p in net.parameters():
            p.data.normal_(0.0, 0.2)
    sums = [th.randn(2 * B, d, generator=gen).to(gpu).requires_grad_(True) for _ in range(n_heads)]
    pls = [th.randint(1, 9, (B, 1), generator=gen).float().to(gpu) for _ in range(n_heads)]
    gls = [th.randint(1, 65, (B, 1), generator=gen).float().to(gpu) for _ in range(n_heads)]
    Lp, Lg = 8.0, 64.0
    blends = [None] if n_heads == 1 else [gls[0] / (gls[0] + gls[1]), gls[1] / (gls[0] + gls[1])]
    flat = []
    for i, net in enumerate(nets):
        flat += [sums[i], pls[i], gls[i], Lp, Lg, "len" if by_len else blends[i]] + list(net.head_params())
    y = _PooledHeadsHIP.apply(n_heads, nets[0].act_slope(), *flat)

    ref = 0.0
    for i, net in enumerate(nets):
        ps, gs, pl, gl = sums[i][:B], sums[i][B:], pls[i], gls[i]
        p = th.nn.functional.linear(ps, net.p_fc.weight) + Lp * net.p_fc.bias
        g = th.nn.functional.linear(gs, net.g_fc.weight) + Lg * net.g_fc.bias
        f = th.cat([p, g, g - p, g * p, pl, gl, 1.0 / pl, 1.0 / gl], dim=1)
        y1 = net.act(net.pred_fc1(f))
        yi = net.pred_fc2(th.cat([y1, pl, gl, 1.0 / pl, 1.0 / gl], dim=1))
        ref = ref + (yi if blends[i] is None else blends[i] * yi)
    assert y.shape == ref.shape and th.allclose(y, ref, rtol=1e-5, atol=1e-5 * max(10.0, float(ref.abs().max()))), (y - ref).abs().max()
    cot = th.randn(B, 1, generator=gen).to(gpu)
    wrt = sums + [p for net in nets for p in net.head_params()]
    got = th.autograd.grad((y * cot).sum(), wrt)
    want = th.autograd.grad((ref * cot).sum(), wrt)
    for i, (a, b) in enumerate(zip(got, want)):
        tol = 1e-4 * max(1.0, float(b.abs().max()))
        assert a.shape == b.shape and th.allclose(a, b, rtol=1e-4, atol=tol), (i, (a - b).abs().max().item())


class _FakePad:
    def __init__(self, sizes):
        self.sizes = th.as_tensor(sizes, dtype=th.int64, device="cuda")
        self.bsz, self.max = len(sizes), int(max(sizes))
        self.seg = th.repeat_interleave(th.arange(self.bsz, device="cuda"), self.sizes)
        self.uniform = self.bsz * self.max == int(sum(sizes))


@pytest.mark.gpu
@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("num_labels", [1, 16, 300])
def test_scalar_filter_gates_match_padded_filter(ragged, num_labels, gpu):
    """dmp_scalar_filter_gates (two element kinds in one call) against (a) the row-wise torch form and (b) the
    reference's formulation: ScalarFilter on the PRE-PADDED label matrices (filter.py:6-16, basemodel.py:1394-1423),
    where the zeros in front of a short pattern take part in the comparison."""
    from dualmessagepassing_amd.basemodel import ScalarFilter, scalar_filter_gate, scalar_filter_gates
    rng = np.random.default_rng(num_labels + ragged)
    B = 37
    jobs, want = [], []
    for kind in range(2):
        ps = rng.integers(1, 9, B) if ragged else np.full(B, 5 + kind)
        gs = rng.integers(1, 70, B) if ragged else np.full(B, 33)
        p_pad, g_pad = _FakePad(ps.tolist()), _FakePad(gs.tolist())
        pl = rng.integers(0, num_labels, int(ps.sum()))
        gl = rng.integers(0, num_labels, int(gs.sum()))
        jobs.append((p_pad, th.as_tensor(pl, device="cuda").view(-1, 1), g_pad, th.as_tensor(gl, device="cuda"), num_labels))
        # (b) the padded cube, pair by pair
        pm, gm = int(ps.max()), int(gs.max())
        P, G = np.zeros((B, pm), np.int64), np.zeros((B, gm), np.int64)
        po = go = 0
        rows = []
        for b in range(B):
            P[b, pm - ps[b]:] = pl[po:po + ps[b]]
            G[b, gm - gs[b]:] = gl[go:go + gs[b]]
            po, go = po + ps[b], go + gs[b]
        cube = ScalarFilter()(th.as_tensor(P).unsqueeze(-1).squeeze(-1), th.as_tensor(G))
        for b in range(B):
            rows.append(cube[b, gm - gs[b]:])
        want.append(th.cat(rows).float().view(-1, 1))
    got = scalar_filter_gates(jobs)
    for j, g, w in zip(jobs, got, want):
        assert g.dtype == th.float32 and g.shape == w.shape
        assert th.equal(g.cpu(), w)
        assert th.equal(g, scalar_filter_gate(*j).float())


class _Sizes:
    """Just enough of a batched graph for basemodel._Padder."""
    node_graph = edge_graph = None

    def __init__(self, sizes, dev):
        self._s = th.as_tensor(sizes, dtype=th.int64, device=dev)

    def batch_num_nodes(self):
        return self._s

    def batch_num_edges(self):
        return self._s


@pytest.mark.gpu
@pytest.mark.parametrize("ragged", [False, True])
@pytest.mark.parametrize("rev_dtype", [th.bool, th.uint8, th.int64])
def test_len_masks_match_torch_chain(ragged, rev_dtype, gpu):
    """dmp_len_masks (all element kinds in one launch) against the arange / compare / masked_fill / sum chain
    (utils/dl.py:113-127 pre-padding masks; basemodel.py:1521-1531 reversed edges leave the edge masks)."""
    from dualmessagepassing_amd.basemodel import _Padder, len_masks

    def torch_chain(jobs):       # the ops the launch replaces, on the host
        out = []
        for p, rev in jobs:
            m = p.mask()
            if rev is not None:
                m = m.masked_fill(p.pad(rev).view(p.bsz, -1, 1).bool(), 0)
            out.append((m, m.view(p.bsz, -1).sum(dim=1, dtype=th.float32).view(-1, 1)))
        return out

    rng = np.random.default_rng(int(ragged) + 3)
    B = 53
    jobs = {"cpu": [], "cuda": []}
    for kind, hi in (("node", 9), ("edge", 700), ("node", 300), ("edge", 1)):
        sizes = rng.integers(1, hi + 1, B) if ragged else np.full(B, hi)
        rev = (rng.random(int(sizes.sum())) < 0.4) if kind == "edge" else None
        for dev in ("cpu", "cuda"):
            r = None if rev is None else th.as_tensor(rev).to(rev_dtype).to(dev)
            jobs[dev].append((_Padder(_Sizes(sizes.tolist(), dev), kind), r))
    want, got = torch_chain(jobs["cpu"]), len_masks(jobs["cuda"])
    with pytest.raises(Exception):
        len_masks(jobs["cpu"])                                   # host tensors are refused, not silently computed
    for (wm, wc), (gm, gc) in zip(want, got):
        assert gm.dtype == th.bool and gm.shape == wm.shape and gc.shape == wc.shape and gc.dtype == th.float32
        assert th.equal(gm.cpu(), wm) and th.equal(gc.cpu(), wc)


@pytest.mark.gpu
def test_table_lookups_match_embedding_modules(gpu):
    """dmp_table_rows (all encodings of a batch in one launch) against the modules' own lookups, bit for bit;
    trainable tables and non-index inputs keep going through the modules."""
    from dualmessagepassing_amd.embed import MultihotEmbedding, PositionEmbedding, NormalEmbedding, lookup_rows
    g = th.Generator().manual_seed(5)
    nets = [MultihotEmbedding(64, 2).cuda(), MultihotEmbedding(16, 2).cuda(), PositionEmbedding(10, 32).cuda(),
            MultihotEmbedding(300, 2).cuda(), PositionEmbedding(40, 20).cuda(), th.nn.Embedding(11, 7).cuda()]
    for n in nets:             # (narrow rows: a thread per row, in pairs or -- odd widths -- float by float; 40 columns: a thread per element)
        n.weight.requires_grad = False
    ids = [th.randint(0, n.weight.size(0), (rows,), generator=g).cuda() for n, rows in zip(nets, (70001, 1, 0, 513, 3000, 2049))]
    got = lookup_rows(nets, ids)
    for n, i, o in zip(nets, ids, got):
        assert o.shape == (i.numel(), n.weight.size(1)) and th.equal(o, n(i))
    train = NormalEmbedding(12, 128).cuda()                      # trainable: stays on the module (autograd)
    out = lookup_rows([train], [th.randint(0, 12, (9,), generator=g).cuda()])[0]
    assert out.requires_grad
    bad = lookup_rows(nets[:1], [th.tensor([3, 64, -1], device="cuda")])[0]
    assert th.equal(bad[0], nets[0].weight[3]) and bool(th.isnan(bad[1:]).all())


PAIR_TAU = 4e-6


def _config1_case(hid, act, gpu, corrupt_pair=None, filt="ScalarFilter"):
    """BASELINE configs[0] batch through the product and through oracle/model_oracle.py from the same ``state_dict``.
    Returns what the two tests below compare.  Parameter gradients are taken PER PAIR (the loss is a sum over pairs, so
    are its gradients): an activation within rounding of its kink (tests/util_flips.py) can move the gradients of the
    pair it belongs to and of no other, so every pair WITHOUT such an activation is held to the strict tolerance.
    ``corrupt_pair``: the product (only) sees that pair's target graph with its destinations rotated."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    import dmp_oracle as O
    import model_oracle as MO
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.collate import collate_device
    cfg = dict(bench.CFG, batch=32, hid=hid, act=act, filter=filt)
    B = cfg["batch"]
    shard = bench.make_shard(cfg, 0, gpu)
    th.manual_seed(3)
    model = build_model(**bench.model_config(cfg)).to(gpu)
    sides = {}
    for tag in ("p", "g"):
        s = shard[tag]
        g = collate_device(s["local_src"], s["local_dst"], s["num_nodes"], s["num_edges"], s["N"], s["E"], ndata=s["ndata"],
                           edata=s["edata"], max_nodes=s["max_n"], max_edges=s["max_e"])
        src, dst = g.all_edges(form="uv", order="eid")
        sides[tag] = {"src": src.cpu(), "dst": dst.cpu(), "bnn": s["num_nodes"].tolist(), "bne": s["num_edges"].tolist(),
                      "id": s["ndata"]["id"].cpu(), "label": s["ndata"]["label"].cpu(), "eid": s["edata"]["id"].cpu(),
                      "elabel": s["edata"]["label"].cpu(), "rev": s["edata"]["is_reversed"].cpu()}

    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for k in list(sd):
        twin = "g_" + k[2:]
        if k.startswith("p_") and twin in sd and sd[k].shape == sd[twin].shape and th.equal(sd[k], sd[twin]):
            sd[k] = sd[twin]
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    O.PROBE = []
    try:
        ref = MO.model_forward(sd, bench.model_config(cfg), sides["p"], sides["g"])
        probes = O.PROBE
    finally:
        O.PROBE = None
    # which pair does a probed row belong to?  (row counts identify the tensor: all six are different)
    seg = {}
    for tag in ("p", "g"):
        seg[sum(sides[tag]["bnn"])] = th.repeat_interleave(th.arange(B), th.tensor(sides[tag]["bnn"]))
        seg[sum(sides[tag]["bne"])] = th.repeat_interleave(th.arange(B), th.tensor(sides[tag]["bne"]))
    seg[B] = th.arange(B)
    assert len(seg) == 5
    touched = th.zeros(B, dtype=th.bool)                         # pairs with an activation within rounding of its kink
    for site, pre in probes:
        pre = pre.detach()
        # the window: a few times the fp32 rounding error of the pre-activation's own sum (K <= 256 terms of the tensor's
        # typical magnitude: ~1e-6 of its standard deviation), NOT a fraction of the tensor's largest value -- at 1e-5 of
        # the maximum every one of the 32 pairs (~1.2e5 activations each) holds an "ambiguous" element and the strict
        # rule would cover nothing
        # (exact zeros are structural -- gated-out rows -- and take the same branch on both sides)
        rows = ((pre.abs() <= PAIR_TAU * max(1.0, float(pre.std()))) & (pre != 0)).reshape(pre.shape[0], -1).any(1)
        touched[seg[pre.shape[0]][rows]] = True
    if corrupt_pair == "first_untouched":
        free = (~touched).nonzero().view(-1)
        if free.numel() == 0:
            pytest.skip("every pair holds an ambiguous activation")
        corrupt_pair = int(free[0])
    def batched():
        out = {}
        for tag in ("p", "g"):
            s = shard[tag]
            ld = s["local_dst"]
            if tag == "g" and corrupt_pair is not None:          # an indexing error in ONE pair: its destinations rotated
                ld = ld.clone()
                n, e = cfg["g_nodes"], 2 * cfg["g_edges"]
                sl = slice(corrupt_pair * e, (corrupt_pair + 1) * e)
                ld[sl] = (ld[sl] + 1) % n
            out[tag] = collate_device(s["local_src"], ld, s["num_nodes"].clone(), s["num_edges"].clone(), s["N"], s["E"], ndata=s["ndata"],
                                      edata=dict(s["edata"]), max_nodes=s["max_n"], max_edges=s["max_e"])
        return out["p"], out["g"]

    names = [k for k, p in model.named_parameters() if p.requires_grad]
    out = model(*batched())
    outputs = {k: (v.detach().clone() if th.is_tensor(v) else v) for k, v in out.items()}
    del out
    got = []                                                     # per pair: {name: gradient}
    for b in range(B):
        model.zero_grad(set_to_none=True)
        o = model(*batched())
        o["pred_c"].view(-1)[b].backward()
        got.append({k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None})
        del o
    leaves = [sd[k] for k in names]
    want = []
    for b in range(B):
        gs = th.autograd.grad(ref["pred_c"].view(-1)[b], leaves, retain_graph=True, allow_unused=True)
        want.append({k: g for k, g in zip(names, gs) if g is not None})
    return outputs, ref, got, want, touched, corrupt_pair


def _pair_gradients_close(got, want, touched, strict=5e-4, loose=5e-2):
    """Every pair's parameter gradients: strict for pairs no ambiguous activation touches, bounded for the others.
    Errors are measured against the largest entry of that parameter's gradient over the pairs of the batch."""
    checked = 0
    scales = {}
    for w in want:
        for k, ref in w.items():
            scales[k] = max(scales.get(k, 1e-30), float(ref.abs().max()))
    for b, (g, w) in enumerate(zip(got, want)):
        for k, ref in w.items():
            if k not in g:
                assert float(ref.abs().max()) == 0.0, (b, k)
                continue
            scale = scales[k]
            err = float((g[k] - ref).abs().max())
            tol = loose if bool(touched[b]) else strict
            assert err <= tol * scale, "pair %d, %s: err %g (scale %g, %s)" % (b, k, err, scale, "touched" if bool(touched[b]) else "no activation near a kink")
            checked += 1
    return checked


@pytest.mark.parametrize("hid,act", [(64, "leaky_relu"), (128, "relu")])
def test_full_model_matches_the_model_oracle_at_config_1(hid, act, gpu):
    """BASELINE configs[0] shape (32 pairs of pattern (8,12) x target (64,256), add_rev, 3 layers) with bench.py's
    synthetic batch and model configuration: the product (fused path, HIP heads) against oracle/model_oracle.py (the
    reference's operation order on the CPU, pinned by the reference's own runs) from the same ``state_dict`` -- all
    outputs, and every parameter gradient of every pair's count prediction (strict 5e-4 for the pairs that hold no
    activation within rounding of its kink; see ``_config1_case``)."""
    out, ref, got, want, touched, _ = _config1_case(hid, act, gpu)
    for k in ("p_v_mask", "p_e_mask", "g_v_mask", "g_e_mask"):
        assert th.equal(out[k].cpu(), ref[k]), k
    for k, tol in (("p_v_emb", 1e-5), ("g_e_emb", 1e-5), ("p_v_rep", 1e-4), ("p_e_rep", 1e-4), ("g_v_rep", 1e-4), ("g_e_rep", 1e-4),
                   ("pred_c", 1e-4)):
        _close(out[k], ref[k].detach().numpy(), tol, k)
    assert int((~touched).sum()) >= 8, int(touched.sum())         # the strict rule must cover a good part of the batch
    assert _pair_gradients_close(got, want, touched) > 30 * 32


def test_full_model_without_a_filter_net_matches_the_model_oracle(gpu):
    """The ALL-ROWS step (``filter_net = "None"``: no 0 / 1 gate, what the reference's one-label ER / Regular datasets give,
    SubgraphCountingMatching/README.md:22-69; bench.py's ``gate_dense`` object) at config 1's shape, hid 128: outputs and every
    pair's parameter gradients against oracle/model_oracle.py -- the ungated branches of the fused layer (the second Linear's
    backward over the identity tile list, the class tiles over every edge)."""
    out, ref, got, want, touched, _ = _config1_case(128, "leaky_relu", gpu, filt="None")
    for k, tol in (("p_v_emb", 1e-5), ("g_e_emb", 1e-5), ("p_v_rep", 1e-4), ("p_e_rep", 1e-4), ("g_v_rep", 1e-4), ("g_e_rep", 1e-4), ("pred_c", 1e-4)):
        _close(out[k], ref[k].detach().numpy(), tol, k)
    assert _pair_gradients_close(got, want, touched) > 30 * 32


def test_pair_gradient_comparison_rejects_an_indexing_error(gpu):
    """Negative test: the product sees ONE pair's target graph with rotated destinations.  Its count
    prediction's parameter gradients must fail the comparison (the rule this replaces -- 2 % of the elements of the
    summed gradient within 5e-3 -- did not look at pairs at all)."""
    out, ref, got, want, touched, pair = _config1_case(64, "leaky_relu", gpu, corrupt_pair="first_untouched")
    with pytest.raises(AssertionError, match="pair %d" % pair):
        _pair_gradients_close(got, want, touched)
    ok = [b for b in range(32) if b != pair]                     # and the other 31 pairs are untouched by the error
    assert _pair_gradients_close([got[b] for b in ok], [want[b] for b in ok], touched[ok]) > 0

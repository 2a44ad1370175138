"""Gate compaction (``collate.compact_gated_edges`` -> ``dmp_gate_compact``, ``GraphAdjModelV2.set_gate_capacity``).

An edge the filter gate removes is a zero row through the reference's whole rep-net (basemodel.py:1515-1531,
dmpnn.py:262-275): the model may run its rep-net on the kept edges only.  Checked here: the integer transform against a
numpy restatement (bit-exact: order, per-graph counts, padding, degrees, overflow flag), and the model with the capacity
set against the same model without it -- every one of the 15 outputs and every parameter gradient.
"""
import os
import sys

import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _random_batch(rng, B, max_n, max_e, gpu, empty_graphs=False):
    from dualmessagepassing_amd.collate import collate_device
    nn = rng.integers(1, max_n + 1, size=B)
    ne = rng.integers(0, max_e + 1, size=B)
    if empty_graphs:
        ne[rng.integers(0, B, size=max(1, B // 5))] = 0
    ls = np.concatenate([rng.integers(0, n, size=e) for n, e in zip(nn, ne)]).astype(np.int64)
    ld = np.concatenate([rng.integers(0, n, size=e) for n, e in zip(nn, ne)]).astype(np.int64)
    rev = rng.integers(0, 2, size=int(ne.sum())).astype(np.uint8)
    t = lambda a: th.from_numpy(np.ascontiguousarray(a)).to(gpu)
    g = collate_device(t(ls), t(ld), t(nn.astype(np.int64)), t(ne.astype(np.int64)), int(nn.sum()), int(ne.sum()),
                       edata={"is_reversed": t(rev)}, max_nodes=int(nn.max()), max_edges=int(max(ne.max(), 1)))
    return g, nn, ne, rev


def _reference(g, nn, ne, rev, gate, cap):
    """numpy restatement of dmp_gate_compact."""
    src, dst = g._src.cpu().numpy(), g._dst.cpu().numpy()
    B = len(nn)
    eoff = np.concatenate([[0], np.cumsum(ne)])
    noff = np.concatenate([[0], np.cumsum(nn)])
    kept = np.array([int((gate[eoff[i]:eoff[i + 1]] != 0).sum()) for i in range(B)])
    P = cap - int(kept.sum())
    over = P < 0
    P = max(P, 0)
    q, r = divmod(P, B)
    out = {k: [] for k in ("src", "dst", "rev", "eid", "gate")}
    sizes = []
    for i in range(B):
        pad = q + (1 if i < r else 0)
        for e in range(eoff[i], eoff[i + 1]):
            if gate[e] != 0:
                out["src"].append(src[e]); out["dst"].append(dst[e]); out["rev"].append(rev[e]); out["eid"].append(e)
                out["gate"].append(gate[e])
        for j in range(pad):
            node = noff[i] + (j % nn[i])
            out["src"].append(node); out["dst"].append(node); out["rev"].append(0); out["eid"].append(0); out["gate"].append(0.0)
        sizes.append(kept[i] + pad)
    deg = np.bincount(src, minlength=int(nn.sum()))
    return {k: np.array(v) for k, v in out.items()}, np.array(sizes), deg, over


@pytest.mark.parametrize("B,max_n,max_e,frac,empty", [(37, 9, 40, 0.4, False), (64, 64, 512, 0.35, False), (5, 3000, 700, 0.5, False),
                                                      (16, 12, 30, 0.3, True), (1, 8, 24, 0.5, False)])
def test_gate_compact_matches_its_restatement(B, max_n, max_e, frac, empty):
    from dualmessagepassing_amd.collate import compact_gated_edges
    gpu = th.device("cuda:0")
    rng = np.random.default_rng(B * 1000 + max_e)
    g, nn, ne, rev = _random_batch(rng, B, max_n, max_e, gpu, empty)
    E = int(ne.sum())
    gate = (rng.random(E) < frac).astype(np.float32)
    kept = int(gate.sum())
    for cap in (kept + B + 3, kept, kept + 1, max(1, kept - 2)):
        if cap >= E or cap <= 0:
            continue
        status = th.zeros(1, dtype=th.int32, device=gpu)
        g.ndata.pop("out_deg", None)
        comp = compact_gated_edges(g, th.from_numpy(gate).to(gpu).view(-1, 1), cap, status)
        assert comp is not None
        ref, sizes, deg, over = _reference(g, nn, ne, rev, gate, cap)
        assert bool(int(status.item()) & 1) == over
        assert np.array_equal(g.ndata["out_deg"].cpu().numpy(), deg)            # the WHOLE graph's degrees
        if over:
            continue
        c = comp.graph
        assert c.number_of_edges() == cap and c.number_of_nodes() == g.number_of_nodes()
        assert np.array_equal(c._src.cpu().numpy(), ref["src"]) and np.array_equal(c._dst.cpu().numpy(), ref["dst"])
        assert np.array_equal(c.edata["is_reversed"].cpu().numpy(), ref["rev"])
        assert np.array_equal(comp.eid_map.cpu().numpy(), ref["eid"])
        assert np.array_equal(comp.gate.view(-1).cpu().numpy(), ref["gate"].astype(np.float32))
        assert np.array_equal(c.batch_num_edges().cpu().numpy(), sizes)
        assert np.array_equal(c.edge_offsets.cpu().numpy(), np.concatenate([[0], np.cumsum(sizes)]))
        # expand: kept rows back in place, zero rows elsewhere; its backward gathers and gates
        rows = th.randn(cap, 8, device=gpu) * comp.gate
        rows.requires_grad_(True)
        full = comp.expand(rows)
        want = np.zeros((E, 8), np.float32)
        k = ref["gate"] != 0
        want[ref["eid"][k]] = rows.detach().cpu().numpy()[k]
        assert np.array_equal(full.detach().cpu().numpy(), want)
        d = th.randn(E, 8, device=gpu)
        full.backward(d)
        wd = d.cpu().numpy()[ref["eid"]] * k[:, None]
        assert np.array_equal(rows.grad.cpu().numpy(), wd.astype(np.float32))


def _model_and_batch(cfg, gpu):
    sys.path.insert(0, ROOT)
    import bench
    shard = bench.make_shard(cfg, 0, gpu)
    step, model = bench.build_step(dict(cfg, graph=False), shard, gpu, 1)
    gen = th.Generator().manual_seed(99)
    with th.no_grad():                      # away from the zero-initialised head (tests/test_gpu_bench_composite.py)
        for p in step.sync.params:
            p.add_((0.05 * th.randn(p.shape, generator=gen)).to(gpu))
    return bench, shard, step, model


def _outputs_and_grads(bench, cfg, shard, step, model, lazy):
    from dualmessagepassing_amd.collate import collate_device
    p, g = shard["p"], shard["g"]
    pattern = collate_device(p["local_src"], p["local_dst"], p["num_nodes"].clone(), p["num_edges"].clone(), p["N"], p["E"],
                             ndata=p["ndata"], edata=dict(p["edata"]), max_nodes=p["max_n"], max_edges=p["max_e"])
    graph = collate_device(g["local_src"], g["local_dst"], g["num_nodes"].clone(), g["num_edges"].clone(), g["N"], g["E"],
                           ndata=g["ndata"], edata=dict(g["edata"]), max_nodes=g["max_n"], max_edges=g["max_e"])
    model.lazy_edge_rep = lazy
    step.sync.detach_grads()
    out = model(pattern, graph)
    from dualmessagepassing_amd.embed import materialize
    vals = {k: materialize(v) for k, v in out.items()}
    loss = th.nn.functional.mse_loss(out["pred_c"].view(-1), shard["counts"])
    # a second term through the edge representation's rows, so that the expanded rows' backward is exercised as well
    if not lazy:
        w = th.linspace(-1.0, 1.0, vals["g_e_rep"].numel(), device=vals["g_e_rep"].device).view_as(vals["g_e_rep"])
        loss = loss + 1e-3 * (vals["g_e_rep"] * w).sum()
    loss.backward()
    step.sync.pack()
    return {k: (None if v is None else v.detach().float().cpu()) for k, v in vals.items() if not isinstance(v, tuple)}, step.sync.flat.detach().cpu().clone()


@pytest.mark.parametrize("lazy", [True, False])
def test_model_with_gate_capacity_equals_the_model_without(lazy):
    gpu = th.device("cuda:0")
    sys.path.insert(0, ROOT)
    import bench
    cfg = dict(bench.CFG, batch=96)
    bench_, shard, step, model = _model_and_batch(cfg, gpu)
    ref_out, ref_flat = _outputs_and_grads(bench_, cfg, shard, step, model, lazy)
    p, g = shard["p"], shard["g"]
    from dualmessagepassing_amd.collate import collate_device
    pattern = collate_device(p["local_src"], p["local_dst"], p["num_nodes"], p["num_edges"], p["N"], p["E"], ndata=p["ndata"],
                             edata=p["edata"], max_nodes=p["max_n"], max_edges=p["max_e"])
    graph = collate_device(g["local_src"], g["local_dst"], g["num_nodes"], g["num_edges"], g["N"], g["E"], ndata=g["ndata"],
                           edata=g["edata"], max_nodes=g["max_n"], max_edges=g["max_e"])
    cap = model.calibrate_gate_capacity(pattern, graph, margin=1.1, multiple=256)
    assert cap is not None and cap < 0.7 * g["E"], (cap, g["E"])
    out, flat = _outputs_and_grads(bench_, cfg, shard, step, model, lazy)
    assert model.compaction_status() == 0
    assert set(out) == set(ref_out)
    for k, v in ref_out.items():
        if v is None:
            assert out[k] is None
            continue
        assert out[k].shape == v.shape, k
        s = max(1e-6, float(v.abs().max()))
        assert float((out[k] - v).abs().max()) <= 2e-5 * s, (k, float((out[k] - v).abs().max()), s)
    by = {}
    for prm, off in zip(step.sync.params, step.sync.offsets):
        by[off] = prm.numel()
    for off, n in by.items():
        a, b = flat[off:off + n], ref_flat[off:off + n]
        s = float(b.abs().max())
        if s == 0.0:
            assert float(a.abs().max()) == 0.0
        else:
            assert float((a - b).abs().max()) <= 5e-5 * s, (off, n, float((a - b).abs().max()), s)
    # a capacity the batch does not fit in: flagged, not silently wrong
    model.set_gate_capacity(1024)
    _outputs_and_grads(bench_, cfg, shard, step, model, lazy)
    assert model.compaction_status() & 1
    assert model.compaction_status() == 0                      # cleared by the read
    model.set_gate_capacity(None)


def test_guarded_adamw_drops_a_flagged_step():
    """``FlatAdamW.set_veto``: a step whose flag word is up leaves parameters, moments and the step count alone."""
    from dualmessagepassing_amd.dp import FlatAdamW
    gpu = th.device("cuda:0")
    th.manual_seed(3)
    p = th.nn.Parameter(th.randn(4099, device=gpu))
    q = th.nn.Parameter(p.detach().clone())
    word = th.zeros(4, dtype=th.int32, device=gpu)
    opt = FlatAdamW([p], lr=1e-2, weight_decay=1e-2, amsgrad=True).set_veto(word, 1)
    ref = FlatAdamW([q], lr=1e-2, weight_decay=1e-2, amsgrad=True, capturable=True)
    grads = [th.randn(4099, device=gpu) for _ in range(3)]
    p.grad, q.grad = grads[0].clone(), grads[0].clone()
    opt.step(); ref.step()
    assert th.equal(p, q) and word.tolist() == [0, 0, 0, 0]
    before = p.detach().clone()
    word[0] = 3                                                # bit 0 (vetoes) and bit 1 (does not)
    p.grad = grads[1].clone()
    opt.step()
    assert th.equal(p, before) and word.tolist() == [0, 3, 1, 1]
    p.grad, q.grad = grads[2].clone(), grads[2].clone()
    opt.step(); ref.step()                                     # the guarded optimizer's SECOND counted step
    assert th.equal(p, q) and word.tolist() == [0, 3, 0, 1]
    word[0] = 2                                                # a flag outside the mask: the step goes ahead
    p.grad, q.grad = grads[1].clone(), grads[1].clone()
    opt.step(); ref.step()
    assert th.equal(p, q) and word.tolist() == [0, 3, 0, 1]
    opt.sync_state()
    assert opt.state[p]["step"] == 3


@pytest.mark.parametrize("graphed", [False, True])
def test_fit_with_gate_compact_trains_like_fit_without(graphed):
    """``harness.fit(gate_compact=True)``: same losses epoch by epoch as the run on every edge row, nothing dropped -- with
    eager launches and with the steps recorded as HIP graphs (the compaction and the guarded optimizer replay)."""
    from dualmessagepassing_amd import harness
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
    gpu = th.device("cuda:0")
    data = harness.SyntheticPairs(96, 4, 5, 24, 60, 6, 6, seed=5)
    hist = {}
    for mode in (False, True):
        th.manual_seed(0)
        model = build_model(**data.model_config(hid_dim=128, layers=3)).to(gpu)
        sync = FlatGradSync(model)
        master = sync.flatten_parameters()
        opt = FlatAdamW([master], lr=1e-3, weight_decay=1e-5, amsgrad=True, capturable=graphed)
        hist[mode] = harness.fit(model, opt, data.subset(range(64)), data.subset(range(64, 96)), 3, 32, gpu, sync=sync, seed=1,
                                 gate_compact=1.35 if mode else False, graph=graphed)
        if mode:
            assert model.gate_capacity, "the gate removed too little for a capacity to be set"
            assert hist[mode][-1]["dropped_steps"] == 0
    for a, b in zip(hist[False], hist[True]):
        for part in ("train", "dev"):
            for k, v in a[part].items():
                assert abs(v - b[part][k]) <= 2e-4 * max(1.0, abs(v)), (part, k, v, b[part][k])


@pytest.mark.parametrize("B,max_n,max_e,empty", [(37, 9, 40, False), (64, 64, 512, False), (5, 2048, 9000, False), (16, 12, 30, True),
                                                 (1, 8, 24, False), (300, 3, 700, False)])
def test_one_launch_csr_build_equals_the_pair_build(B, max_n, max_e, empty):
    """``dmp_csr_build_graphs`` (a workgroup per graph, counters in LDS) against ``dmp_csr_build_pair``: every array equal."""
    from dualmessagepassing_amd.graph import GraphIndex
    gpu = th.device("cuda:0")
    rng = np.random.default_rng(7 * B + max_e)
    g, nn, ne, rev = _random_batch(rng, B, max_n, max_e, gpu, empty)
    a = GraphIndex(g._src, g._dst, g.number_of_nodes(), g.edata["is_reversed"], validate=True)
    b = GraphIndex(g._src, g._dst, g.number_of_nodes(), g.edata["is_reversed"], validate=True,
                   offsets=(g.node_offsets, g.edge_offsets, B))
    for k in ("in_ptr", "in_ent", "dst32", "in_deg", "out_ptr", "out_ent", "src32", "out_deg"):
        assert th.equal(getattr(a, k), getattr(b, k)), k
    assert g.index().in_ptr.data_ptr() != a.in_ptr.data_ptr() and th.equal(g.index().in_ent, a.in_ent)    # the batch's own build
    # an edge that leaves its graph is reported, not silently mis-filed
    if B > 1 and int(ne[0]) > 0:
        bad = g._dst.clone()
        bad[0] = g.number_of_nodes() - 1
        with pytest.raises(Exception):
            GraphIndex(g._src, bad, g.number_of_nodes(), g.edata["is_reversed"], validate=True, offsets=(g.node_offsets, g.edge_offsets, B))


def test_pool_indexes_built_together_equal_the_single_builds():
    from dualmessagepassing_amd import fused, ops
    gpu = th.device("cuda:0")
    rng = np.random.default_rng(11)
    sa, sb = rng.integers(0, 9, 40), rng.integers(0, 70, 40)
    ea, eb = rng.integers(0, 30, 40), rng.integers(0, 600, 40)
    t = lambda a, dt=th.int64: th.from_numpy(np.ascontiguousarray(a)).to(dt).to(gpu)
    fa, fb = rng.integers(0, 2, int(ea.sum())), rng.integers(0, 2, int(eb.sum()))
    specs = [((t(sa), t(sb)), None, int(sa.sum() + sb.sum())),
             ((t(ea), t(eb)), (t(fa, th.uint8), t(fb, th.uint8)), int(ea.sum() + eb.sum()))]
    many = ops.pool_indexes(specs)
    for spec, got in zip(specs, many):
        one = ops.PoolIndex(spec[0], spec[1], num_rows=spec[2])
        for k in ("gptr", "vptr", "vent", "gent", "seg32", "offsets", "sizes"):
            assert th.equal(getattr(one, k), getattr(got, k)), k
        assert (one.flag8 is None) == (got.flag8 is None)
        if one.flag8 is not None:
            assert th.equal(one.flag8, got.flag8)
            want = th.where(one.flag8 != 0, th.full_like(one.seg32, -1), one.seg32)
            assert th.equal(fused.pool_rowmap(got), want)
            w = th.rand(spec[2], device=gpu)
            cnt = fused.pool_weight_sums(got, w).double().cpu().numpy()
            off = got.offsets.cpu().numpy()
            f = got.flag8.cpu().numpy()
            wn = w.double().cpu().numpy()
            for i in range(got.num_graphs):
                rows = slice(off[i], off[i + 1])
                assert abs(cnt[i, 0] - wn[rows][f[rows] == 0].sum()) < 1e-4 and abs(cnt[i, 1] - wn[rows][f[rows] != 0].sum()) < 1e-4
            assert th.equal(fused.pool_weight_sums(got), fused.pool_weight_sums(got, th.ones(spec[2], device=gpu)))


def test_collate_many_equals_collate_device():
    from dualmessagepassing_amd.collate import collate_device, collate_device_many
    gpu = th.device("cuda:0")
    rng = np.random.default_rng(3)
    jobs = []
    for B, max_n, max_e in ((33, 9, 20), (33, 70, 500), (1, 5, 0)):
        nn = rng.integers(1, max_n + 1, size=B)
        ne = rng.integers(0, max_e + 1, size=B)
        t = lambda a: th.from_numpy(np.ascontiguousarray(a).astype(np.int64)).to(gpu)
        ls = np.concatenate([rng.integers(0, n, size=e) for n, e in zip(nn, ne)] + [np.zeros(0, np.int64)])
        ld = np.concatenate([rng.integers(0, n, size=e) for n, e in zip(nn, ne)] + [np.zeros(0, np.int64)])
        jobs.append(dict(local_src=t(ls), local_dst=t(ld), num_nodes=t(nn), num_edges=t(ne), total_nodes=int(nn.sum()),
                         total_edges=int(ne.sum()), ndata={"x": t(np.arange(nn.sum()))}, edata={}, max_nodes=int(nn.max()),
                         max_edges=int(max(ne.max(), 1))))
    many = collate_device_many(jobs)
    for j, g in zip(jobs, many):
        one = collate_device(j["local_src"], j["local_dst"], j["num_nodes"], j["num_edges"], j["total_nodes"], j["total_edges"],
                             ndata=j["ndata"], edata=j["edata"], max_nodes=j["max_nodes"], max_edges=j["max_edges"])
        for k in ("_src", "_dst", "node_graph", "edge_graph", "node_offsets", "edge_offsets"):
            assert th.equal(getattr(one, k), getattr(g, k)), k
        assert (one.max_num_nodes, one.max_num_edges, one.tiling is None, one.node_tiling is None) == \
               (g.max_num_nodes, g.max_num_edges, g.tiling is None, g.node_tiling is None)
        assert g.ndata["x"] is not None and g.batch_size == one.batch_size


@pytest.mark.parametrize("H,R", [(128, 4099), (64, 2051), (128, 31)])
def test_masked_row_kernels_do_not_read_gated_rows_and_equal_the_unmasked(H, R):
    """``dmp_out_fwd_fused_masked`` / ``dmp_bwd_h1_fused_masked``: same results as the kernels that fetch every row, and
    the rows under a zero gate are really not fetched (poisoned with NaN, the results stay what they were)."""
    from dualmessagepassing_amd import fused
    gpu = th.device("cuda:0")
    g = th.Generator(device=gpu).manual_seed(H + R)
    gate = (th.rand(R, device=gpu, generator=g) < 0.4).float()
    h1 = th.randn(R, H, device=gpu, generator=g)
    prev = th.randn(R, H, device=gpu, generator=g)
    d_o = th.randn(R, H, device=gpu, generator=g)
    W2 = th.randn(H, H, device=gpu, generator=g) / H ** 0.5
    b2 = th.randn(H, device=gpu, generator=g)
    mask = fused.gate_row_mask(gate)
    if mask is None:
        pytest.skip("DMP_ROW_MASKS switched off")
    bits = mask.cpu().numpy().view(np.uint32)
    want = (gate.cpu().numpy() != 0)
    got = np.array([(bits[r >> 5] >> (r & 31)) & 1 for r in range(R)], bool)
    assert np.array_equal(got, want)
    fused.USE_ROW_MASKS = False
    try:
        ref_out = fused.out_fwd_mfma(h1, W2, b2, gate, prev)
        ref_dg, ref_db = fused.bwd_h1_mfma(d_o, W2, h1, both_halves=False, gate=gate, slope=0.18)
    finally:
        fused.USE_ROW_MASKS = True
    dead = gate == 0
    h1p, d_op = h1.clone(), d_o.clone()
    h1p[dead] = float("nan")
    d_op[dead] = float("nan")
    out = fused.out_fwd_mfma(h1p, W2, b2, gate, prev)
    dg, db = fused.bwd_h1_mfma(d_op, W2, h1p, both_halves=False, gate=gate, slope=0.18)
    assert th.equal(out, ref_out)
    assert th.equal(dg, ref_dg) and th.equal(db, ref_db)
    assert bool(th.isfinite(out).all()) and bool(th.isfinite(dg).all())


@pytest.mark.parametrize("amsgrad", [True, False])
def test_flat_adamw_keeps_one_step_count_per_parameter_tensor(amsgrad):
    """A parameter that first receives a gradient at optimizer step k gets torch.optim.AdamW's bias corrections (its own
    count starts at 1), one that skips steps keeps its count -- on the flat buffer of ``FlatGradSync.flatten_parameters``."""
    from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync
    gpu = th.device("cuda:0")
    th.manual_seed(5)

    class M(th.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = th.nn.Parameter(th.randn(37, 5))
            self.b = th.nn.Parameter(th.randn(130))
            self.c = th.nn.Parameter(th.randn(8, 8))

    m = M().to(gpu)
    ref = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
    sync = FlatGradSync(m)
    opt = FlatAdamW([sync.flatten_parameters()], lr=3e-3, weight_decay=1e-2, amsgrad=amsgrad)
    ropt = th.optim.AdamW(ref, lr=3e-3, weight_decay=1e-2, amsgrad=amsgrad)
    live_at = [(0,), (0, 2), (0, 1, 2), (0, 1), (1, 2), (0, 1, 2), (2,)]        # b joins at step 3, c skips steps 4 and 1
    params = list(m.parameters())
    for step, live in enumerate(live_at):
        sync.detach_grads()
        for r in ref:
            r.grad = None
        for i in live:
            g = th.randn(params[i].shape, device=gpu)
            params[i].grad = g.clone()
            ref[i].grad = g.clone()
        sync.pack()
        opt.step()
        ropt.step()
        for i, (p, r) in enumerate(zip(params, ref)):
            err = float((p.detach() - r.detach()).abs().max())
            assert err <= 2e-6 * max(1.0, float(r.detach().abs().max())), (step, i, err)
    opt.sync_state()
    st = opt.state[sync.master]
    assert st["seg_steps"] == [5, 4, 5] and st["step"] == 7
    sd = opt.state_dict()
    opt2 = FlatAdamW([sync.master], lr=3e-3, weight_decay=1e-2, amsgrad=amsgrad)
    opt2.load_state_dict(sd)
    sync.detach_grads()
    for i in range(3):
        g = th.randn(params[i].shape, device=gpu)
        params[i].grad = g.clone(); ref[i].grad = g.clone()
    sync.pack()
    opt2.step(); ropt.step()                                       # the reloaded optimizer carries the per-tensor counts on
    for p, r in zip(params, ref):
        assert float((p.detach() - r.detach()).abs().max()) <= 2e-6 * max(1.0, float(r.detach().abs().max()))


@pytest.mark.parametrize("lazy", [True, False])
def test_zero_gate_skipping_leaves_every_output_and_gradient_as_it_was(lazy):
    """The masked kernels (rows under a zero gate are not fetched) and the dead H1 rows (not even produced) against the same
    model with all of that switched off -- with the dead rows' buffers poisoned (NaN), so that a consumer which does fetch one
    cannot go unnoticed."""
    from dualmessagepassing_amd import fused
    gpu = th.device("cuda:0")
    sys.path.insert(0, ROOT)
    import bench
    cfg = dict(bench.CFG, batch=96)
    bench_, shard, step, model = _model_and_batch(cfg, gpu)
    fused.USE_ROW_MASKS = False
    try:
        ref_out, ref_flat = _outputs_and_grads(bench_, cfg, shard, step, model, lazy)
    finally:
        fused.USE_ROW_MASKS = True
    assert fused.SKIP_DEAD_ROWS
    fused.POISON_DEAD_ROWS = True
    try:
        out, flat = _outputs_and_grads(bench_, cfg, shard, step, model, lazy)
    finally:
        fused.POISON_DEAD_ROWS = False
    for k, v in ref_out.items():
        if v is None:
            assert out[k] is None
            continue
        assert bool(th.isfinite(out[k]).all()), k
        s = max(1e-6, float(v.abs().max()))
        assert float((out[k] - v).abs().max()) <= 2e-5 * s, (k, float((out[k] - v).abs().max()), s)
    assert bool(th.isfinite(flat).all())
    for prm, off in zip(step.sync.params, step.sync.offsets):
        a, b = flat[off:off + prm.numel()], ref_flat[off:off + prm.numel()]
        s = float(b.abs().max())
        if s == 0.0:
            assert float(a.abs().max()) == 0.0
        else:
            assert float((a - b).abs().max()) <= 5e-5 * s, (off, float((a - b).abs().max()), s)


@pytest.mark.parametrize("filt", ["ScalarFilter", "None"])
def test_first_layer_residual_rows_from_their_codes_leave_the_model_as_it_was(filt):
    """``fused.out_fwd_typed_codes`` in the model (under the ScalarFilter gates: over the kept edges' tiles; without a filter net:
    over all rows): the joint pass does not store the target's embedded edge rows (its placeholder is
    poisoned with NaN here), the first layer's second Linear forms the kept ones from the label codes -- against the same model with
    the rows stored (``USE_OUT_CODES`` off): every output and gradient equal to fp32 accuracy; and a first layer that cannot take
    that launch (``out_codes_ok`` forced false) makes the rows itself (``l0_rows``) with the same result."""
    from dualmessagepassing_amd import fused
    gpu = th.device("cuda:0")
    sys.path.insert(0, ROOT)
    import bench
    cfg = dict(bench.CFG, batch=96, filter=filt)
    bench_, shard, step, model = _model_and_batch(cfg, gpu)
    fused.USE_OUT_CODES = False
    try:
        ref_out, ref_flat = _outputs_and_grads(bench_, cfg, shard, step, model, True)
    finally:
        fused.USE_OUT_CODES = True
    calls = []
    orig, orig_ok = fused.out_fwd_typed_codes, fused.out_codes_ok
    for fallback in (False, True):
        fused.out_fwd_typed_codes = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        if fallback:
            fused.out_codes_ok = lambda *a, **k: False
        fused.POISON_DEAD_ROWS = True
        try:
            out, flat = _outputs_and_grads(bench_, cfg, shard, step, model, True)
        finally:
            fused.POISON_DEAD_ROWS = False
            fused.out_fwd_typed_codes, fused.out_codes_ok = orig, orig_ok
        assert len(calls) == 1                        # the launch ran once (first pass), not at all in the fallback pass
        for k, v in ref_out.items():
            if v is None:
                assert out[k] is None
                continue
            assert bool(th.isfinite(out[k]).all()), k
            s = max(1e-6, float(v.abs().max()))
            assert float((out[k] - v).abs().max()) <= 2e-5 * s, (k, float((out[k] - v).abs().max()), s)
        assert bool(th.isfinite(flat).all())
        for prm, off in zip(step.sync.params, step.sync.offsets):
            a, b = flat[off:off + prm.numel()], ref_flat[off:off + prm.numel()]
            s = float(b.abs().max())
            if s == 0.0:
                assert float(a.abs().max()) == 0.0
            else:
                assert float((a - b).abs().max()) <= 5e-5 * s, (off, float((a - b).abs().max()), s)


@pytest.mark.parametrize("R,M,N", [(40003, 128, 128), (5000, 64, 64), (8200, 128, 256)])
def test_binary_gate_weight_gradient_on_the_plain_rows_product(R, M, N):
    """``fused.atb_rows`` with a gate flagged 0 / 1: the ungated product over the masked-in rows (bf16x6) + the one-column
    column sums against fp64 and against the gated f32 form; rows under a zero gate are never fetched (NaN there)."""
    from dualmessagepassing_amd import fused
    gpu = th.device("cuda:0")
    g = th.Generator(device=gpu).manual_seed(R + M)
    gate = (th.rand(R, device=gpu, generator=g) < 0.4).float()
    a = th.randn(R, M, device=gpu, generator=g)
    b = th.randn(R, N, device=gpu, generator=g)
    ref_w = (a.double() * gate.double()[:, None]).t() @ b.double()
    ref_c = (a.double() * gate.double()[:, None]).sum(0)
    w0, c0 = fused.atb_rows(a, b, gate)                           # not flagged: the gated form
    gate._dmp_binary = True
    ap, bp = a.clone(), b.clone()
    ap[gate == 0] = float("nan")
    bp[gate == 0] = float("nan")
    w1, c1 = fused.atb_rows(ap, bp, gate)
    scale = float((a.double().abs() * gate.double()[:, None]).t().matmul(b.double().abs()).max())
    for w, c in ((w0, c0), (w1, c1)):
        assert bool(th.isfinite(w).all()) and bool(th.isfinite(c).all())
        assert float((w.double() - ref_w).abs().max()) <= 2e-6 * scale
        assert float((c.double() - ref_c).abs().max()) <= 2e-6 * float(gate.sum())


@pytest.mark.parametrize("H,R", [(128, 40003), (64, 5000)])
def test_bwd_h1_hands_out_the_gated_column_sums(H, R):
    from dualmessagepassing_amd import fused
    gpu = th.device("cuda:0")
    g = th.Generator(device=gpu).manual_seed(H + R)
    gate = (th.rand(R, device=gpu, generator=g) < 0.4).float()
    gate._dmp_binary = True
    d_o = th.randn(R, H, device=gpu, generator=g)
    h1 = th.randn(R, H, device=gpu, generator=g)
    W2 = th.randn(H, H, device=gpu, generator=g) / H ** 0.5
    dg0, db0 = fused.bwd_h1_mfma(d_o, W2, h1, both_halves=False, gate=gate, slope=0.18)
    dg1, db1, cs = fused.bwd_h1_mfma(d_o, W2, h1, both_halves=False, gate=gate, slope=0.18, rows_colsum=True)
    assert th.equal(dg0, dg1) and th.equal(db0, db1)
    ref = (d_o.double() * gate.double()[:, None]).sum(0)
    assert float((cs.double() - ref).abs().max()) <= 2e-6 * float(gate.sum())


@pytest.mark.parametrize("R,H", [(9001, 128), (2500, 64)])
def test_rows_jobs_without_gates_run_ungated_over_the_masked_in_rows(R, H):
    """``fused.atb_rows_multi`` whose jobs carry neither a gate nor column sums (the node side of a step under a 0 / 1 gate):
    one bf16x6 launch; rows a job's mask leaves out are never fetched (NaN there), jobs without a mask read every row."""
    from dualmessagepassing_amd import fused
    gpu = th.device("cuda:0")
    g = th.Generator(device=gpu).manual_seed(R + H)
    gate = (th.rand(R, device=gpu, generator=g) < 0.4).float()
    gate._dmp_binary = True
    mask = fused.binary_gate_mask(gate)
    assert mask is not None
    a = th.randn(R, H, device=gpu, generator=g)
    b = th.randn(R, H, device=gpu, generator=g)
    x = th.randn(R, H, device=gpu, generator=g)
    y = th.randn(R, 2 * H, device=gpu, generator=g)
    ap, bp = a.clone(), b.clone()
    ap[gate == 0] = float("nan")
    bp[gate == 0] = float("nan")
    (w, c), (wx, cx) = fused.atb_rows_multi([(ap, bp, None, False, mask), (x, y, None, False)])
    assert c is None and cx is None
    ref_w = (a.double() * gate.double()[:, None]).t() @ b.double()
    ref_x = x.double().t() @ y.double()
    assert float((w.double() - ref_w).abs().max()) <= 2e-6 * float(((a.double().abs() * gate.double()[:, None]).t() @ b.double().abs()).max())
    assert float((wx.double() - ref_x).abs().max()) <= 2e-6 * float((x.double().abs().t() @ y.double().abs()).max())


@pytest.mark.parametrize("H,R", [(128, 4099), (64, 2051), (128, 31)])
def test_dead_rows_are_neither_fetched_nor_stored_where_the_caller_says_so(H, R):
    """``dmp_out_fwd_fused_rows`` / ``dmp_bwd_h1_fused_rows``: with residual rows that ARE zero under the zero gates, bit 0
    leaves them unfetched (NaN there changes nothing), bit 1 / ``skip_dead_stores`` leaves the output's rows under a zero
    gate as they were (a sentinel survives) and every other row is the plain kernel's."""
    from dualmessagepassing_amd import fused
    gpu = th.device("cuda:0")
    g = th.Generator(device=gpu).manual_seed(3 * H + R)
    gate = (th.rand(R, device=gpu, generator=g) < 0.4).float()
    dead = gate == 0
    h1 = th.randn(R, H, device=gpu, generator=g)
    prev = th.randn(R, H, device=gpu, generator=g) * gate[:, None]
    d_o = th.randn(R, H, device=gpu, generator=g)
    W2 = th.randn(H, H, device=gpu, generator=g) / H ** 0.5
    b2 = th.randn(H, device=gpu, generator=g)
    ref_out = fused.out_fwd_mfma(h1, W2, b2, gate, prev)
    ref_dg, ref_db = fused.bwd_h1_mfma(d_o, W2, h1, both_halves=False, gate=gate, slope=0.18)
    assert float(ref_out[dead].abs().max()) == 0.0 and float(ref_dg[dead].abs().max()) == 0.0
    prev_p = prev.clone()
    prev_p[dead] = float("nan")
    out1 = fused.out_fwd_mfma(h1, W2, b2, gate, prev_p, dead_rows=1)
    assert th.equal(out1, ref_out)
    saved = fused.dead_rows_buffer
    fused.dead_rows_buffer = lambda shape, device: th.full(shape, 7.5, dtype=th.float32, device=device)
    try:
        out3 = fused.out_fwd_mfma(h1, W2, b2, gate, prev_p, dead_rows=3)
        dg, db = fused.bwd_h1_mfma(d_o, W2, h1, both_halves=False, gate=gate, slope=0.18, skip_dead_stores=True)
    finally:
        fused.dead_rows_buffer = saved
    assert th.equal(out3[~dead], ref_out[~dead]) and bool((out3[dead] == 7.5).all())
    assert th.equal(dg[~dead], ref_dg[~dead]) and bool((dg[dead] == 7.5).all()) and th.equal(db, ref_db)


@pytest.mark.parametrize("H", [128, 64])
def test_masked_endpoint_sums_equal_the_plain_ones_without_reading_zero_rows(H):
    """``dmp_seg_sum2_graphs_masked``: rows under a clear mask bit are not fetched (NaN there), the sums are the bits of the
    kernel that reads the zero rows; also over the incidence CSR (weights) when the batch does not tile by graphs."""
    from dualmessagepassing_amd import ops, fused
    gpu = th.device("cuda:0")
    rng = np.random.default_rng(H)
    graph = _random_batch(rng, 37, 20, 90, gpu)[0]
    ix = graph.index()
    E = ix.num_edges
    assert ops.graph_seg_ok(ix, th.empty(E, H, device=gpu), H)          # the one-pass kernel is what runs first
    g = th.Generator(device=gpu).manual_seed(H)
    gate = (th.rand(E, device=gpu, generator=g) < 0.45).float()
    M = th.randn(E, H, device=gpu, generator=g) * gate[:, None]
    ref = ops.endpoint_sums(M, ix)
    Mp = M.clone()
    Mp[gate == 0] = float("nan")
    got = ops.endpoint_sums(Mp, ix, mask=fused.gate_row_mask(gate), gate=gate)
    assert th.equal(got, ref)
    saved = ops.USE_GRAPH_SEG_SUM
    ops.USE_GRAPH_SEG_SUM = False
    try:
        got2 = ops.endpoint_sums(Mp, ix, mask=fused.gate_row_mask(gate), gate=gate)
    finally:
        ops.USE_GRAPH_SEG_SUM = saved
    assert th.equal(got2, ref)


@pytest.mark.parametrize("B,max_n,max_e,keep", [(37, 20, 90, 0.46), (64, 64, 512, 0.4), (5, 2048, 9000, 0.5), (16, 12, 30, 0.0), (16, 12, 30, 1.0)])
def test_in_csr_over_the_kept_edges(B, max_n, max_e, keep):
    """``dmp_csr_keep``: the CSR by destination restricted to the edges a 0 / 1 gate keeps -- every row's kept entries in their
    order, packed as in the whole CSR -- and the node aggregation over it equals the gate-weighted one bit for bit."""
    from dualmessagepassing_amd import fused, ops
    gpu = th.device("cuda:0")
    rng = np.random.default_rng(B + max_e)
    graph = _random_batch(rng, B, max_n, max_e, gpu)[0]
    ix = graph.index()
    E, N = ix.num_edges, ix.num_nodes
    g_np = (rng.random(E) < keep).astype(np.float32)
    gate = th.from_numpy(g_np).to(gpu)
    gate._dmp_binary = True
    kc = fused.keep_in_csr(ix, gate)
    if kc is None:
        pytest.skip("DMP_KEEP_CSR / DMP_ROW_MASKS switched off")
    kp, ke = kc
    ptr_, ent_ = ix.in_ptr.cpu().numpy(), ix.in_ent.cpu().numpy()
    want_ptr, want_ent = [0], []
    for v in range(N):
        row = [e for e in ent_[ptr_[v]:ptr_[v + 1]] if g_np[e >> 1] != 0]
        want_ent += row
        want_ptr.append(len(want_ent))
    assert np.array_equal(kp.cpu().numpy(), np.array(want_ptr, np.int32))
    assert np.array_equal(ke.cpu().numpy()[:len(want_ent)], np.array(want_ent, np.int32))
    H = 128
    M = th.randn(E, H, device=gpu) * gate[:, None]
    Mp = M.clone()
    Mp[gate == 0] = float("nan")
    ref = ops.seg_sum_raw(M, ix.in_ptr, ix.in_ent, N, gate, True, -1.0, 1.0)
    got = ops.seg_sum_raw(Mp, kp, ke, N, None, True, -1.0, 1.0)
    assert th.equal(got, ref)


@pytest.mark.parametrize("rows,avg", [(300, 50), (4099, 17), (1, 700), (10752, 51)])
def test_kept_csr_with_a_lane_group_per_row(rows, avg):
    """``dmp_csr_keep`` on a CSR with LONG rows (the pooling index's chunk table: ~50 entries per row): the lane-group form
    (16 lanes per row, ballot + prefix popcount) against a numpy restatement and against the thread-per-row form (hint 0)."""
    from dualmessagepassing_amd import _lib
    from dualmessagepassing_amd._lib import check, ptr, stream_ptr
    gpu = th.device("cuda:0")
    rng = np.random.default_rng(rows + avg)
    lens = rng.integers(0, 2 * avg + 1, rows)
    if rows > 1:
        lens[rng.integers(0, rows, max(1, rows // 10))] = 0      # empty rows too
    lens[0] = max(int(lens[0]), 1)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    E = int(rp[-1])
    ent = ((rng.permutation(E).astype(np.int64) << 1) | rng.integers(0, 2, E)).astype(np.int32)
    g_np = (rng.random(max(E, 1)) < 0.45).astype(np.float32)
    lib = _lib.load()
    res = {}
    for hint in (E, 0):
        in_ptr, in_ent, gate = (th.from_numpy(a).to(gpu) for a in (rp, ent, g_np))
        nscr = int(lib.dmp_csr_keep_scratch_words(rows))
        ws = th.full((nscr + rows + 1 + max(E, 1),), -7, dtype=th.int32, device=gpu)
        row_cnt, kptr, kent = ws[:nscr], ws[nscr:nscr + rows + 1], ws[nscr + rows + 1:]
        check(lib.dmp_csr_keep(ptr(in_ptr), ptr(in_ent), ptr(gate), rows, hint, ptr(row_cnt), ptr(kptr), ptr(kent), stream_ptr()), "dmp_csr_keep")
        th.cuda.synchronize()
        res[hint] = (kptr.cpu().numpy().copy(), kent.cpu().numpy().copy())
    want_ptr, want_ent = [0], []
    for v in range(rows):
        want_ent += [e for e in ent[rp[v]:rp[v + 1]] if g_np[e >> 1] != 0]
        want_ptr.append(len(want_ent))
    for hint, (kp, ke) in res.items():
        assert np.array_equal(kp, np.array(want_ptr, np.int32)), hint
        assert np.array_equal(ke[:len(want_ent)], np.array(want_ent, np.int32)), hint
        assert bool((ke[len(want_ent):] == -7).all()), hint           # nothing written past the kept entries

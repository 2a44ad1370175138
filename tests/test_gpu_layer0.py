"""The first layer of a rep-net on the label codes of its edge rows (csrc/dmp_layer0.hip, fused.l0_*): the kernels
against fp64, the layer against the general fused layer fed ``codes @ table`` (values and every gradient), and the
whole model with the path on and off.  Reference: basemodel.py:1393-1420 (embeddings) feeding dmpnn.py:111-156."""
import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu


def _index(src, dst, n, rev, dev):
    from dualmessagepassing_amd.graph import GraphIndex
    return GraphIndex(th.from_numpy(src).to(dev), th.from_numpy(dst).to(dev), n, th.from_numpy(rev).to(dev), validate=True)


def _case(rows, n_pat, k, h, gated, seed, dev, n_split=None):
    rng = np.random.default_rng(seed)
    gen = th.Generator().manual_seed(seed)
    n = max(4, rows // 7)
    if n_split is None:
        n_split = max(1, n // 40) if n_pat else 0         # the pattern's edges stay among the pattern's nodes (a union of two batches)
    src, dst = rng.integers(n_split, n, rows).astype(np.int64), rng.integers(n_split, n, rows).astype(np.int64)
    if n_pat:
        src[:n_pat], dst[:n_pat] = rng.integers(0, n_split, n_pat), rng.integers(0, n_split, n_pat)
    rev = rng.random(rows) < 0.5
    ix = _index(src, dst, n, rev, dev)
    coef = ix.degree_coef(ix.out_deg)
    enc = (th.rand(rows, k, generator=gen) < 0.5).float().to(dev)          # multi-hot codes
    enc_p, enc_g = enc[:n_pat], enc[n_pat:]
    gate = (th.rand(rows - n_pat, generator=gen) < 0.7).float().to(dev) if gated else None
    W = th.randn(k, h, generator=gen).to(dev)
    return ix, coef, enc_p, enc_g, gate, W, gen, n, n_split


@pytest.mark.parametrize("rows,n_pat", [(1, 0), (5, 5), (1000, 37), (70001, 900), (548864, 1200)])
@pytest.mark.parametrize("k,h,gated,slope", [(10, 128, True, 1 / 5.5), (10, 64, False, 0.0), (16, 128, True, 0.0), (3, 128, False, 1 / 5.5)])
def test_layer0_kernels_against_fp64(rows, n_pat, k, h, gated, slope, gpu):
    from dualmessagepassing_amd import fused
    ix, coef, enc_p, enc_g, gate, W, gen, n, _ = _case(rows, n_pat, k, h, gated, rows + k, gpu)
    encU = fused.l0_pack(enc_p, enc_g, gate)
    kpad = (k + 3) // 4 * 4
    ref = th.zeros(rows, kpad, device=gpu)
    ref[:n_pat, :k] = enc_p
    ref[n_pat:, :k] = enc_g if gate is None else enc_g * gate.view(-1, 1)
    assert th.equal(encU, ref)
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    xp = th.randn(n, 3 * h, generator=gen).to(gpu)
    bias = th.randn(h, generator=gen).to(gpu)
    M = W @ wes
    h1 = fused.l0_edge_fwd(encU, k, M, xp[:, h:], 3 * h, bias, coef, ix, slope)
    a, b, cf = (t.long() if i < 2 else t.double() for i, t in enumerate(ix.edge_select(coef)))
    ed, P = encU[:, :k].double(), xp.double()
    z = ed @ W.double()
    pre = z @ wes[:, :h].double() + cf.view(-1, 1) * (z @ wes[:, h:].double()) + P[a, h:2 * h] - P[b, 2 * h:] + bias.double()
    h64 = th.where(pre > 0, pre, slope * pre)
    assert float((h1.double() - h64).abs().max()) <= 2e-5 * max(1.0, float(h64.abs().max()))
    assert th.equal(h1, fused.l0_edge_fwd(encU, k, M, xp[:, h:], 3 * h, bias, coef, ix, slope))      # repeatable
    # the backward pass: one read of dPre (and the residual gradient)
    d_pre = th.randn(rows, h, generator=gen).to(gpu)
    d_zn = th.randn(rows, h, generator=gen).to(gpu)
    for dz in (d_zn, None):
        xx = fused.l0_bwd_w(encU, k, ix.edge_select(coef)[2], d_pre, dz)
        want = [ed.t() @ d_pre.double(), (ed * cf.view(-1, 1)).t() @ d_pre.double()] + ([ed.t() @ dz.double()] if dz is not None else [])
        want = th.cat(want, dim=1)
        assert xx.shape == want.shape
        assert float((xx.double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def _layer_case(h, slope_act, gated, residual, seed, dev, n_pat=150, n_split=None):
    from dualmessagepassing_amd.dmpnn import DMPLayer
    rows, k = 6000, 10
    ix, coef, enc_p, enc_g, gate, W, gen, n, n_split = _case(rows, n_pat, k, h, gated, seed, dev, n_split)
    th.manual_seed(seed)
    layer = DMPLayer(h, h, batch_norm=False, act_func="leaky_relu" if slope_act == "linear" else slope_act).to(dev)
    if slope_act == "linear":      # negative slope 1: no kink, so the two paths' gradients cannot differ by an activation's side
        layer.nmlp[1], layer.emlp[1] = th.nn.LeakyReLU(1.0), th.nn.LeakyReLU(1.0)
    x = th.randn(n, h, generator=gen).to(dev).requires_grad_(True)
    vg = (th.rand(n, generator=gen) < 0.8).float().to(dev) if gated else None
    eg = th.cat([th.ones(n_pat, device=dev), gate]) if gated else None
    return ix, coef, enc_p, enc_g, gate, W, layer, x, vg, eg, k, rows, n_pat, n_split


@pytest.mark.parametrize("h,act", [(128, "linear"), (128, "leaky_relu"), (64, "relu"), (64, "linear")])
@pytest.mark.parametrize("gated,residual", [(True, True), (False, True), (True, False)])
@pytest.mark.parametrize("tables,nodes,n_pat,n_split", [(1, False, 150, None), (2, False, 150, None), (1, True, 150, None), (2, True, 150, None),
                                                          (2, True, 0, 5), (2, False, 0, 5)])      # the last two: pattern nodes, no pattern edges
def test_layer0_equals_the_general_layer(h, act, gated, residual, tables, nodes, n_pat, n_split, gpu):
    """Same layer, same inputs: the label-code path against the general fused layer reading ``z = codes @ table`` (and
    ``x = node codes @ node table`` with ``nodes``) -- outputs, input gradient, every parameter gradient and the tables'
    gradients.  The products are re-associated, so the two differ by fp32 rounding -- and, with a kinked activation, by the
    side a pre-activation within rounding of zero falls on: outputs are held to rounding always, gradients to rounding with
    the kink-free activation (negative slope 1) and to a few such rows otherwise."""
    from dualmessagepassing_amd import fused
    ix, coef, enc_p, enc_g, gate, W, layer, x, vg, eg, k, rows, n_pat, n_split = _layer_case(h, act, gated, residual, 7, gpu, n_pat, n_split)
    assert fused.l0_ok(ix, h, enc_p, enc_g, W, W)
    gen = th.Generator().manual_seed(5)
    if tables == 2:                                       # the pattern's own table stacked over the target's (no share_emb_net)
        W = th.cat([th.randn(k, h, generator=gen).to(gpu), W])
    encU = fused.l0_pack(enc_p, enc_g, gate)
    n, vk = x.size(0), 8
    venc = (th.rand(n, vk, generator=gen) < 0.5).float().to(gpu)
    vencU = fused.l0_pack(venc[:n_split], venc[n_split:], vg[n_split:] if vg is not None else None)
    WV = th.randn(tables * vk, h, generator=gen).to(gpu)
    gen = th.Generator().manual_seed(99)
    gx, gz = th.randn(x.shape, generator=gen).to(gpu), th.randn(rows, h, generator=gen).to(gpu)
    params = list(layer.parameters())

    def run(l0):
        W0, WV0 = W.clone().requires_grad_(True), WV.clone().requires_grad_(True)
        z = encU[:, :k] @ W0 if tables == 1 else th.cat([encU[:n_pat, :k] @ W0[:k], encU[n_pat:, :k] @ W0[k:]])
        if nodes:
            xin = vencU[:, :vk] @ WV0 if tables == 1 else th.cat([vencU[:n_split, :vk] @ WV0[:vk], vencU[n_split:, :vk] @ WV0[vk:]])
        else:
            xin = x
        if l0:
            split = (0, 0) if tables == 1 else (n_pat, n_split)
            codes = fused.Layer0Codes(encU, k, W0, *split)
            if nodes:
                codes.venc, codes.VK, codes.WV = vencU, vk, WV0
            xn, zn = fused.fused_dmp_layer(ix, coef, residual, xin.detach() if nodes else xin, z.detach(), vg, eg, layer, l0=codes)
        else:
            xn, zn = fused.fused_dmp_layer(ix, coef, residual, xin, z, vg, eg, layer)
        grads = th.autograd.grad([xn, zn], [WV0 if nodes else x, W0] + params, [gx, gz], allow_unused=True)
        return [xn.detach(), zn.detach()] + list(grads)

    got, want = run(True), run(False)
    names = ["xn", "zn", "dWV0" if nodes else "dx", "dW0"] + [n for n, _ in layer.named_parameters()]
    for name, g, w in zip(names, got, want):
        assert (g is None) == (w is None), name
        if g is None:
            continue
        scale = max(1.0, float(w.abs().max()))
        tol = 3e-5 if act == "linear" or name in ("xn", "zn") else 3e-3
        assert float((g - w).abs().max()) <= tol * scale, (name, float((g - w).abs().max()), scale)


@pytest.mark.parametrize("hid,act,emb", [(128, "leaky_relu", "Equivariant"), (64, "relu", "Orthogonal"), (128, "relu", "Normal")])
def test_model_with_and_without_the_label_code_path(hid, act, emb, gpu):
    """The BASELINE configs[0] batch through the whole model with the first layer on the label codes (default) and on the
    general path: prediction and every parameter gradient."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from dualmessagepassing_amd import fused
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.collate import collate_device
    cfg = dict(bench.CFG, batch=32, hid=hid, act=act, emb=emb)
    shard = bench.make_shard(cfg, 0, gpu)
    th.manual_seed(3)
    model = build_model(**bench.model_config(cfg)).to(gpu)
    gs = {}
    for tag in ("p", "g"):
        s = shard[tag]
        gs[tag] = lambda s=s: collate_device(s["local_src"], s["local_dst"], s["num_nodes"], s["num_edges"], s["N"], s["E"],
                                              ndata=s["ndata"], edata=s["edata"], max_nodes=s["max_n"], max_edges=s["max_e"])
    out = {}
    saved = fused.USE_LAYER0
    try:
        for on in (True, False):
            fused.USE_LAYER0 = on
            model.zero_grad(set_to_none=True)
            calls = []
            orig = fused.l0_edge_fwd
            fused.l0_edge_fwd = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            try:
                pred = model(gs["p"](), gs["g"]())["pred_c"]
            finally:
                fused.l0_edge_fwd = orig
            assert bool(calls) == on                      # the path under test is the one that ran
            (pred ** 2).sum().backward()
            out[on] = (pred.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    finally:
        fused.USE_LAYER0 = saved
    (p1, g1), (p0, g0) = out[True], out[False]
    assert float((p1 - p0).abs().max()) <= 2e-5 * max(1.0, float(p0.abs().max()))
    assert set(g1) == set(g0)
    for n in g0:
        scale = max(1.0, float(g0[n].abs().max()))
        assert float((g1[n] - g0[n]).abs().max()) <= 2e-4 * scale, (n, float((g1[n] - g0[n]).abs().max()), scale)


def test_a_model_the_fused_path_cannot_run_falls_back(gpu):
    """Dropout in the rep-net (training mode) takes the layers off the fused path: the joint pass declines -- also after it
    has looked at the label codes -- and the model runs the reference's two loops."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from dualmessagepassing_amd import fused
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.collate import collate_device
    cfg = dict(bench.CFG, batch=8)
    shard = bench.make_shard(cfg, 0, gpu)
    th.manual_seed(3)
    model = build_model(**dict(bench.model_config(cfg), rep_dropout=0.2)).to(gpu)
    model.train()
    gs = []
    for tag in ("p", "g"):
        s = shard[tag]
        gs.append(collate_device(s["local_src"], s["local_dst"], s["num_nodes"], s["num_edges"], s["N"], s["E"], ndata=s["ndata"],
                                 edata=s["edata"], max_nodes=s["max_n"], max_edges=s["max_e"]))
    calls = []
    orig = fused.l0_edge_fwd
    fused.l0_edge_fwd = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        out = model(*gs)
    finally:
        fused.l0_edge_fwd = orig
    assert not calls
    (out["pred_c"] ** 2).sum().backward()
    assert all(th.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    assert th.is_tensor(out["g_e_rep"]) and out["g_e_rep"].shape[1] == cfg["hid"]


@pytest.mark.parametrize("rows,r0,keep", [(70001, 0, 0.46), (548864, 24576, 0.46), (40000, 32, 0.0), (40000, 0, 1.0), (33000, 64, 0.5)])
@pytest.mark.parametrize("k,h", [(10, 128), (16, 64)])
def test_layer0_kernels_over_the_kept_rows_list(rows, r0, keep, k, h, gpu):
    """``dmp_kept_rows`` (the rows of a range whose mask bit is set, ascending, counted on the device) against numpy, and the
    list forms of the two layer-0 kernels against their masked forms: the same kept output rows (the others untouched), the
    same sums up to their order; rows outside the list are never read (NaN there)."""
    from dualmessagepassing_amd import fused
    ix, coef, enc_p, enc_g, _, W, gen, n, _ = _case(rows, 0, k, h, False, rows + k + r0, gpu)
    rng = np.random.default_rng(rows + r0)
    g_np = (rng.random(rows) < keep).astype(np.float32)
    gate = th.from_numpy(g_np).to(gpu)
    mask = fused.gate_row_mask(gate)
    lst, cnt = fused.kept_rows(mask, r0, rows)
    want = np.nonzero(g_np[r0:])[0]
    assert int(cnt.item()) == len(want) and np.array_equal(lst[:len(want)].cpu().numpy(), want)
    encU = fused.l0_pack(enc_p, enc_g, gate)
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    xp = th.randn(n, 3 * h, generator=gen).to(gpu)
    bias = th.randn(h, generator=gen).to(gpu)
    M = W @ wes
    dead = gate == 0
    saved = fused.USE_L0_ROW_LISTS
    res = {}
    for lists in (False, True):
        fused.USE_L0_ROW_LISTS = lists
        try:
            out = th.full((rows, h), 7.5, device=gpu)
            fused.l0_edge_fwd(encU, k, M, xp[:, h:], 3 * h, bias, coef, ix, 0.18, rows=(r0, rows), out=out, mask=mask)
            d_pre = th.randn(rows, h, generator=th.Generator().manual_seed(5)).to(gpu)
            d_zn = th.randn(rows, h, generator=th.Generator().manual_seed(6)).to(gpu)
            d_pre[dead] = float("nan")
            d_zn[dead] = float("nan")
            xx = fused.l0_bwd_w(encU, k, ix.edge_select(coef)[2], d_pre, d_zn, rows=(r0, rows), mask=mask)
        finally:
            fused.USE_L0_ROW_LISTS = saved
        res[lists] = (out, xx)
    (o0, x0), (o1, x1) = res[False], res[True]
    assert th.equal(o0, o1)                                           # kept rows: the same arithmetic; the others: the sentinel
    assert bool((o1[r0:][dead[r0:]] == 7.5).all()) and bool((o1[:r0] == 7.5).all())
    assert bool(th.isfinite(x1).all())
    assert float((x1 - x0).abs().max()) <= 2e-5 * max(1.0, float(x0.abs().max()))


@pytest.mark.parametrize("n,n0,vk,k0,h,masked,slope", [(1, 0, 8, 10, 128, False, 0.0), (777, 0, 16, 10, 128, True, 1 / 5.5), (73728, 8192, 16, 10, 128, True, 1 / 5.5),
                                                         (5000, 37, 16, 12, 64, True, 0.0), (4096, 0, 4, 3, 64, False, 1 / 5.5), (70001, 0, 16, 10, 128, False, 0.0)])
@pytest.mark.parametrize("listed", [False, True])
def test_layer0_node_pass_against_fp64(n, n0, vk, k0, h, masked, slope, listed, gpu):
    """``dmp_l0_node_fwd``: the first layer's node side from the node codes and the edges' code sums in one pass -- against fp64,
    over a row range, with the kept nodes' mask (dead rows: H1n untouched, projections zero)."""
    from dualmessagepassing_amd import fused
    gen = th.Generator().manual_seed(n + vk + k0)
    kp = (k0 + 3) // 4 * 4
    venc = (th.rand(n, (vk + 3) // 4 * 4, generator=gen) < 0.4).float().to(gpu)
    S0 = th.randint(-3, 6, (n, 2 * kp), generator=gen).float().to(gpu)
    keep = (th.rand(n, generator=gen) < 0.4).to(gpu) if masked else th.ones(n, dtype=th.bool, device=gpu)
    venc = venc * keep.view(-1, 1).float()                 # a dead node's code row is zero (l0_pack multiplies by the gate)
    mask = fused.gate_row_mask(keep.float()) if masked else None
    Mv = th.randn(vk, 3 * h, generator=gen).to(gpu)
    Ma, Mb = th.randn(k0, h, generator=gen).to(gpu), th.randn(k0, h, generator=gen).to(gpu)
    bias = th.randn(h, generator=gen).to(gpu)
    h1 = th.full((n, h), -7.0, device=gpu)
    P = th.full((n, 2 * h), -7.0, device=gpu)
    klist = None
    if listed and masked:                                  # the kept nodes' list over ALL nodes (NodeRows.rows)
        lst, cnt = fused.kept_rows(mask, 0, n, tiles=True)
        klist = (lst, cnt[0:1])
    elif listed:
        pytest.skip("a list comes with a mask")
    W = fused.l0_node_pack(vk, k0, h, Mv, Ma, Mb)
    fused.l0_node_fwd(venc, vk, S0, k0, kp, W, bias, slope, mask, n0, n, h, h1, P, rows=klist)
    pre = venc[:, :vk].double() @ Mv[:, :h].double() + S0[:, :k0].double() @ Ma.double() + S0[:, kp:kp + k0].double() @ Mb.double() + bias.double()
    ref = th.where(pre > 0, pre, slope * pre)
    rows = th.arange(n, device=gpu) >= n0
    live = rows & keep
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert float((h1.double() - ref)[live].abs().max()) <= tol if bool(live.any()) else True
    assert bool((h1[~live] == -7.0).all())                # rows outside the range / dead rows: not written
    pref = venc[:, :vk].double() @ Mv[:, h:].double()
    assert float((P.double() - pref)[rows].abs().max()) <= tol if bool(rows.any()) else True
    assert bool((P[rows & ~keep] == 0).all()) and bool((P[~rows] == -7.0).all())
    h1b, Pb = th.full_like(h1, -7.0), th.full_like(P, -7.0)
    fused.l0_node_fwd(venc, vk, S0, k0, kp, W, bias, slope, mask, n0, n, h, h1b, Pb, rows=klist)
    assert th.equal(h1, h1b) and th.equal(P, Pb)          # repeatable


@pytest.mark.parametrize("R,K,h,blocks", [(1, 3, 128, 1), (777, 16, 128, 3), (73728, 16, 128, 3), (5001, 10, 64, 1)])
def test_smallk_atb_over_a_row_list_equals_the_masked_form(R, K, h, blocks, gpu):
    """``dmp_smallk_atb_cols_rows``: ``x^T [d | d2]`` over the kept rows' list against the masked walk over all rows and fp64."""
    from dualmessagepassing_amd import fused
    gen = th.Generator().manual_seed(R + K)
    keep = (th.rand(R, generator=gen) < 0.4).to(gpu)
    x = ((th.rand(R, (K + 3) // 4 * 4, generator=gen) < 0.5).float().to(gpu) * keep.view(-1, 1).float())[:, :K]
    d = th.randn(R, blocks * h, generator=gen).to(gpu)
    d2 = th.randn(R, h, generator=gen).to(gpu)
    d[~keep] = float("nan")                                   # a row outside the list / under the mask is never fetched
    d2[~keep] = float("nan")
    mask = fused.gate_row_mask(keep.float())
    lst, cnt = fused.kept_rows(mask, 0, R, tiles=True)
    a = fused.smallk_atb_cols(x, d, d2, None, h, mask=mask)
    b = fused.smallk_atb_cols(x, d, d2, None, h, rows=(lst, cnt[0:1]))
    dz, d2z = th.where(keep.view(-1, 1), d, th.zeros_like(d)).double(), th.where(keep.view(-1, 1), d2, th.zeros_like(d2)).double()
    ref = th.stack([x.double().t() @ dz[:, j * h:(j + 1) * h] for j in range(blocks)] + [x.double().t() @ d2z])
    tol = 2e-5 * max(1.0, float(ref.abs().max()))
    assert bool(th.isfinite(a).all()) and bool(th.isfinite(b).all())
    assert float((a.double() - ref).abs().max()) <= tol and float((b.double() - ref).abs().max()) <= tol


@pytest.mark.parametrize("sizes,K,h", [((1200, 73728 - 1216), 10, 128), ((0, 5000), 7, 128), ((96, 64, 4097, 32), 16, 64), ((33,), 3, 128)])
def test_smallk_atb_products_sharing_a_launch_equal_their_own_launches(sizes, K, h, gpu):
    """``dmp_smallk_atb_jobs`` (the first layer's node-code weight gradients: two tables x two halves of the code sums as ONE
    launch) against the single launches it replaces -- the same kernel body over the same workgroup count per product: equal
    bits; masked rows are not fetched (NaN there); an empty product gives zeros."""
    from dualmessagepassing_amd import fused
    gen = th.Generator().manual_seed(sum(sizes) + K)
    jobs, refs = [], []
    for n, R in enumerate(sizes):
        R32 = (R + 31) // 32 * 32
        keep = (th.rand(R, generator=gen) < 0.4).to(gpu)
        x = (th.rand(R, 2 * 16, generator=gen) < 0.5).float().to(gpu)[:, 16 * (n & 1):16 * (n & 1) + K] * keep.view(-1, 1).float()
        d = th.randn(R, h, generator=gen).to(gpu)
        masked = bool(n & 1) or len(sizes) == 1
        mask = None
        if masked:
            d[~keep] = float("nan")
            mask = fused.gate_row_mask(th.cat([keep.float(), th.zeros(R32 - R, device=gpu)]))
        out = th.full((K, h), -3.0, device=gpu)
        jobs.append((x, d, out, mask))
        if R:
            refs.append(fused.smallk_atb_cols(x, d, None, None, h, mask=mask)[0])
        else:
            refs.append(th.zeros(K, h, device=gpu))
    fused.smallk_atb_jobs(jobs, h)
    for (x, d, out, mask), ref in zip(jobs, refs):
        assert bool(th.isfinite(out).all())
        assert th.equal(out, ref)


@pytest.mark.parametrize("n,rows,K,stacked,gated", [(0, 1, 3, False, False), (1200, 70001, 10, False, True), (33, 4097, 8, True, True),
                                                    (7, 0, 16, False, True), (500, 900, 5, True, False)])
def test_packed_codes_are_the_gated_codes_in_their_columns(n, rows, K, stacked, gated, gpu):
    """``dmp_l0_pack`` (a thread per row): ``[enc_p ; gate * enc_g]`` zero-padded to whole 16-byte pieces, the second kind of rows
    in columns K .. 2K-1 when stacked -- against the tensor construction, bit for bit (also from wider, strided code arrays)."""
    from dualmessagepassing_amd import fused
    gen = th.Generator().manual_seed(n + rows + K)
    wide_p, wide_g = th.randn(n, K + 3, generator=gen).to(gpu), th.randn(rows, K + 1, generator=gen).to(gpu)
    enc_p, enc_g = wide_p[:, 1:K + 1], wide_g[:, :K]
    gate = (th.rand(rows, generator=gen) < 0.5).float().to(gpu) * 1.5 if gated else None
    got = fused.l0_pack(enc_p, enc_g, gate, stacked)
    goff = K if stacked else 0
    kpad = (goff + K + 3) // 4 * 4
    ref = th.zeros(n + rows, kpad, device=gpu)
    ref[:n, :K] = enc_p
    ref[n:, goff:goff + K] = enc_g if gate is None else enc_g * gate.view(-1, 1)
    assert got.shape == ref.shape and th.equal(got, ref)
    # two packings in one launch (``l0_pack_many``: a step's edge codes and node codes) are the single launches
    other = (th.randn(5, 4, generator=gen).to(gpu), th.randn(300, 4, generator=gen).to(gpu), None, True)
    both = fused.l0_pack_many([(enc_p, enc_g, gate, stacked), other])
    assert th.equal(both[0], ref) and th.equal(both[1], fused.l0_pack(*other))

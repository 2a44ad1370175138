"""GPU parity at the other BASELINE shapes (reduced batch so the CPU oracle finishes in seconds)
and on degenerate graphs: config 4 (pattern (16,32) x target (512,4096), hid 128), config 5
(Cora-shaped single graph N=2708, E=10858 with reversed copies, UNC DualGraphConv x2, hid 256),
graphs without edges, a single node, one giant hub."""
import numpy as np
import pytest
import torch as th

import dmp_oracle as O
from util_graphs import er_batch, er_edges

pytestmark = pytest.mark.gpu


def _t(a):
    return th.from_numpy(np.asarray(a))


def _close(got, ref, tol, what):
    got, ref = got.detach().double().cpu(), ref.detach().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(1.0, float(ref.abs().max())) if ref.numel() else 1.0
    err = float((got - ref).abs().max()) if ref.numel() else 0.0
    assert err <= tol * scale, "%s: max err %g (scale %g)" % (what, err, scale)


def test_config4_shape_three_layer_rep(gpu):
    from dualmessagepassing_amd.dmpnn import DMPNNRep
    from dualmessagepassing_amd.graph import BatchedGraph
    rng = np.random.default_rng(4000)
    h, L, B = 128, 3, 4
    ps, pd, pr, pn, pbnn, pbne = er_batch(B, 16, 32, rng)
    gs, gd, gr, gn, gbnn, gbne = er_batch(B, 512, 4096, rng)
    gen = th.Generator().manual_seed(4)
    layers = [O.random_dmp_params(h, h, gen) for _ in range(L)]
    net = DMPNNRep(hid_dim=h, rep_num_graph_layers=L, rep_num_pattern_layers=L, share_rep_net=True,
                   rep_residual=True, rep_dmpnn_batch_norm=False, rep_act_func="relu")
    sd = {}
    for i, p in enumerate(layers):
        for k, v in p.items():
            sd["g_rep_net.dmpnn.graph_dmpnn_(%d).%s" % (i, k)] = v
            sd["p_rep_net.dmpnn.graph_dmpnn_(%d).%s" % (i, k)] = v
    net.load_state_dict(sd, strict=True)
    net.to(gpu)
    pv, pe = th.randn(pn, h, generator=gen), th.randn(len(ps), h, generator=gen)
    gv, ge = th.randn(gn, h, generator=gen), th.randn(len(gs), h, generator=gen)
    vg = (th.rand(gn, 1, generator=gen) < 0.8).float()
    eg = (th.rand(len(gs), 1, generator=gen) < 0.8).float()
    tps, tpd, tpr, tgs, tgd, tgr = map(_t, (ps, pd, pr, gs, gd, gr))
    a_o, b_o = O.dmpnn_graph_rep(layers, tps, tpd, tpr, O.out_degrees(tps, pn), pv, pe)
    c_o, d_o = O.dmpnn_graph_rep(layers, tgs, tgd, tgr, O.out_degrees(tgs, gn), gv, ge, vg, eg)
    pg = BatchedGraph(tps.to(gpu), tpd.to(gpu), pn, _t(pbnn).to(gpu), _t(pbne).to(gpu), None, {"is_reversed": tpr.to(gpu)})
    gg = BatchedGraph(tgs.to(gpu), tgd.to(gpu), gn, _t(gbnn).to(gpu), _t(gbne).to(gpu), None, {"is_reversed": tgr.to(gpu)})
    a, b, c, d = net(pg, gg, pv.to(gpu), pe.to(gpu), gv.to(gpu), ge.to(gpu), v_gate=vg.to(gpu), e_gate=eg.to(gpu))
    for got, ref, nm in ((a, a_o, "p_v"), (b, b_o, "p_e"), (c, c_o, "g_v"), (d, d_o, "g_e")):
        _close(got, ref, 2e-4, nm)


def test_config5_cora_shape_unc_two_layers(gpu):
    """UNC DMPNN stack (model.py:296-316): DualGraphConv(tanh) -> DualGraphConv(None), BN in eval."""
    from dualmessagepassing_amd.unc import DualGraphConv, build_graph_from_triplets
    rng = np.random.default_rng(5000)
    n, m, h = 2708, 5429, 256
    u, v = er_edges(n, m, rng)
    trip = np.stack([u, np.zeros(m, np.int64), v], 1)
    g = build_graph_from_triplets(n, 1, trip, gpu)
    assert g.number_of_edges() == 10858
    src, dst = (t.cpu() for t in g.all_edges())
    norm = O.unc_edge_norm(src, dst, n, "in")
    _close(g.edata["norm"], norm, 1e-6, "norm")
    th.manual_seed(5)
    l1 = DualGraphConv(h, h, activation=th.nn.Tanh()).eval()
    l2 = DualGraphConv(h, h, activation=None).eval()
    for layer in (l1, l2):
        with th.no_grad():
            for seq in (layer.nmlp, layer.emlp):
                seq[1].running_mean.uniform_(-0.2, 0.2)
                seq[1].running_var.uniform_(0.5, 1.5)
    x, z = th.randn(n, h), th.randn(10858, h)
    out_deg = O.out_degrees(src, n)

    def oracle(layer, x, z, act):
        p = {k: v.detach() for k, v in layer.named_parameters()}
        bn = {m + ".1": {"running_mean": getattr(layer, m)[1].running_mean.clone(),
                         "running_var": getattr(layer, m)[1].running_var.clone()} for m in ("nmlp", "emlp")}
        return O.dual_graph_conv(p, src, dst, out_deg, x, z, norm, None, bn, False, act)

    xo, zo = oracle(l1, x, z, "tanh")
    xo, zo = oracle(l2, xo, zo, None)
    l1.to(gpu)
    l2.to(gpu)
    xg, zg = l1(g, x.to(gpu), z.to(gpu), g.edata["norm"])
    xg, zg = l2(g, xg, zg, g.edata["norm"])
    _close(xg, xo, 5e-5, "node rep")
    _close(zg, zo, 5e-5, "edge rep")
    # backward at the same shape (the UNC training step differentiates this stack: main.py:150-175): input and parameter
    # gradients of both layers against the oracle's autograd
    wn, we = th.randn(n, h), th.randn(10858, h)
    ps = [{k: v.detach().cpu().clone().requires_grad_(True) for k, v in layer.named_parameters()} for layer in (l1, l2)]
    bns = [{m + ".1": {"running_mean": getattr(layer, m)[1].running_mean.cpu().clone(), "running_var": getattr(layer, m)[1].running_var.cpu().clone()}
            for m in ("nmlp", "emlp")} for layer in (l1, l2)]
    x0, z0 = x.clone().requires_grad_(True), z.clone().requires_grad_(True)
    from util_flips import taint
    O.PROBE = []
    try:
        a1, b1 = O.dual_graph_conv(ps[0], src, dst, out_deg, x0, z0, norm, None, bns[0], False, "tanh")
        a2, b2 = O.dual_graph_conv(ps[1], src, dst, out_deg, a1, b1, norm, None, bns[1], False, None)
        probes = O.PROBE
    finally:
        O.PROBE = None
    ((a2 * wn).sum() + (b2 * we).sum()).backward()
    tn, te = taint(src, dst, probes, n)         # rows a LeakyReLU within rounding of its kink can reach (tests/util_flips.py)
    xg0, zg0 = x.to(gpu).requires_grad_(True), z.to(gpu).requires_grad_(True)
    for layer in (l1, l2):
        layer.zero_grad()
    c1, d1 = l1(g, xg0, zg0, g.edata["norm"])
    c2, d2 = l2(g, c1, d1, g.edata["norm"])
    ((c2 * wn.to(gpu)).sum() + (d2 * we.to(gpu)).sum()).backward()
    from test_gpu_dmplayer import _close_or_flipped
    flipped = _close_or_flipped(xg0.grad, x0.grad, 1e-4, 1e-4, "dx", tn)
    flipped |= _close_or_flipped(zg0.grad, z0.grad, 1e-4, 1e-4, "dz", te)
    for layer, po in zip((l1, l2), ps):
        for k, q in layer.named_parameters():
            if po[k].grad is None:
                assert q.grad is None or float(q.grad.abs().max()) == 0.0, k
            else:
                _close(q.grad, po[k].grad, 2e-2 if flipped else 5e-4, "grad " + k)


@pytest.mark.parametrize("case", ["no_edges", "single_node", "hub"])
def test_degenerate_graphs_through_the_layer(case, gpu):
    from dualmessagepassing_amd.dmpnn import DMPLayer
    from dualmessagepassing_amd.graph import BatchedGraph
    rng = np.random.default_rng(9)
    if case == "no_edges":
        src, dst, n = np.zeros(0, np.int64), np.zeros(0, np.int64), 5
    elif case == "single_node":
        src, dst, n = np.array([0, 0], np.int64), np.array([0, 0], np.int64), 1   # two self loops
    else:
        n = 40
        src = np.concatenate([np.arange(1, n), np.zeros(n - 1, np.int64)]).astype(np.int64)  # star in + out
        dst = np.concatenate([np.zeros(n - 1, np.int64), np.arange(1, n)]).astype(np.int64)
        src, dst = np.tile(src, 30), np.tile(dst, 30)                                       # in-degree 1170 at the hub
    e, h = len(src), 16
    rev = rng.random(e) < 0.5
    gen = th.Generator().manual_seed(3)
    p = O.random_dmp_params(h, h, gen)
    x = th.randn(n, h, generator=gen)
    z = th.randn(e, h, generator=gen)
    ts, td, tr = _t(src), _t(dst), _t(rev)
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xo, zo = x.clone().requires_grad_(True), z.clone().requires_grad_(True)
    no, eo, _, _ = O.dmp_layer(po, ts, td, tr, O.out_degrees(ts, n), xo, zo)
    (no.sum() + (eo * eo).sum()).backward()
    layer = DMPLayer(h, h, batch_norm=False)
    layer.load_state_dict(p)
    layer.to(gpu)
    g = BatchedGraph(ts.to(gpu), td.to(gpu), n, None, None, None, {"is_reversed": tr.to(gpu)})
    xg, zg = x.to(gpu).requires_grad_(True), z.to(gpu).requires_grad_(True)
    for fused in (False, True):
        xg.grad = zg.grad = None
        if fused:
            ng, eg = layer.forward_fused(g, xg, zg, None, None, residual=False)
        else:
            ng, eg = layer(g, xg, zg)
        _close(ng, no, 1e-4, "node_out fused=%s" % fused)
        _close(eg, eo, 1e-4, "edge_out fused=%s" % fused)
        (ng.sum() + (eg * eg).sum()).backward()
        _close(xg.grad, xo.grad, 1e-4, "dx fused=%s" % fused)
        _close(zg.grad, zo.grad, 1e-4, "dz fused=%s" % fused)


@pytest.mark.parametrize("act,h", [("leaky_relu", 128), ("relu", 128), ("leaky_relu", 64)])
def test_fused_typed_layer_at_full_config2_size_against_fp64(act, h, gpu):
    """BASELINE config 2 at FULL size -- the union graph of bench.py's step: 1024 x (pattern (8, 12) + target (64, 256)),
    add_rev: N = 73,728 node rows, E = 548,864 edge rows, hid 128 (and the shipped hid 64) -- through the fused layer on its fastest path (class-typed
    MFMA kernels, folded first Linear, gates, residual), forward and backward, against the oracle's operation order run in
    fp64 on the same device (every row, not a subsample).  fp32 tolerances: outputs 2e-5, input gradients 1e-4 (flip-aware:
    see test_gpu_dmplayer._close_or_flipped), parameter gradients 5e-4 (sums over 5e5 rows) of the largest reference value."""
    from test_gpu_dmplayer import _close_or_flipped
    from dualmessagepassing_amd import fused
    from dualmessagepassing_amd.dmpnn import DMPNNRep
    from dualmessagepassing_amd.graph import BatchedGraph
    rng = np.random.default_rng(2)
    ps, pd, pr, pN, pbnn, pbne = er_batch(1024, 8, 12, rng)
    gs, gd, gr, gN, gbnn, gbne = er_batch(1024, 64, 256, rng)
    src, dst, rev = np.concatenate([ps, gs + pN]), np.concatenate([pd, gd + pN]), np.concatenate([pr, gr])
    N, E = pN + gN, len(src)                                   # h: hid 128 (BASELINE) and the reference's shipped 64
    assert (N, E) == (73728, 548864)
    gen = th.Generator().manual_seed(41)
    params = O.random_dmp_params(h, h, gen, act)
    x, z = th.randn(N, h, generator=gen), th.randn(E, h, generator=gen)
    wv, we = th.randn(N, h, generator=gen), th.randn(E, h, generator=gen)
    vg, eg = (th.rand(N, 1, generator=gen) < 0.7).float(), (th.rand(E, 1, generator=gen) < 0.7).float()
    ts, td, tr = _t(src).to(gpu), _t(dst).to(gpu), _t(rev).to(gpu)
    # fp64 oracle on the device
    p64 = {k: v.double().to(gpu).requires_grad_(True) for k, v in params.items()}
    x64, z64 = x.double().to(gpu).requires_grad_(True), z.double().to(gpu).requires_grad_(True)
    from util_flips import taint
    O.PROBE = []
    try:
        rv, re = O.dmpnn_graph_rep([p64], ts, td, tr, O.out_degrees(ts, N), x64, z64, vg.double().to(gpu), eg.double().to(gpu), True, act)
        probes = O.PROBE
    finally:
        O.PROBE = None
    ((rv * wv.double().to(gpu)).sum() + (re * we.double().to(gpu)).sum()).backward()
    tn, te = taint(ts, td, probes, N)           # rows an activation within rounding of its kink can reach (tests/util_flips.py)
    del probes
    # product
    net = DMPNNRep(hid_dim=h, rep_num_graph_layers=1, rep_num_pattern_layers=1, share_rep_net=True, rep_residual=True,
                   rep_dmpnn_batch_norm=False, rep_act_func=act)
    sd = {}
    for k, v in params.items():
        sd["g_rep_net.dmpnn.graph_dmpnn_(0)." + k] = v
        sd["p_rep_net.dmpnn.graph_dmpnn_(0)." + k] = v
    net.load_state_dict(sd, strict=True)
    net.to(gpu)
    g = BatchedGraph(ts, td, N, _t(np.concatenate([pbnn, gbnn])).to(gpu), _t(np.concatenate([pbne, gbne])).to(gpu))
    g.edata["is_reversed"] = tr
    xg, zg = x.to(gpu).requires_grad_(True), z.to(gpu).requires_grad_(True)
    assert fused.typed_ok(g.index(), h)                        # the class-typed kernels are the ones measured by bench.py
    a, b = net.get_graph_rep(g, xg, zg, v_gate=vg.to(gpu), e_gate=eg.to(gpu))
    ((a * wv.to(gpu)).sum() + (b * we.to(gpu)).sum()).backward()
    _close(a, rv.detach().cpu(), 2e-5, "v_rep")
    _close(b, re.detach().cpu(), 2e-5, "e_rep")
    flipped = _close_or_flipped(xg.grad, x64.grad.cpu(), 5e-5, 5e-5, "dx", tn)
    flipped |= _close_or_flipped(zg.grad, z64.grad.cpu(), 5e-5, 5e-5, "dz", te)
    for k, p in net.g_rep_net.named_parameters():
        ref = p64[k.split(").", 1)[1]].grad.cpu()
        # with ~10^7 activations per layer a few pre-activations lie within fp32 rounding of zero: their derivative branch may
        # differ from the fp64 run's, which moves the parameter gradients by up to a percent (ReLU: a full unit step)
        _close(p.grad, ref, 2e-2 if flipped else 5e-4, "grad " + k)


def test_micro_batched_step_equals_the_one_pass_step(gpu):
    """bench.py's config-4 step splits the shard into micro-batches so that every [E, 2H] array stays below the 32-bit
    offset range of the class-typed kernels; pairs are independent, so the summed gradient of M slices (each slice's mean
    loss weighted 1 / M) is the gradient of the whole shard.  Checked at the config-4 graph shapes on 64 pairs: one pass
    against four micro-batches, the whole flat gradient."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    flats = []
    for m in (1, 4):
        cfg = dict(bench.CFG4, batch=64, micro_batches=m, act="leaky_relu", emb="Equivariant")
        shard = bench.make_shard(cfg, 0, gpu)
        step, model = bench.build_step(cfg, shard, gpu)
        assert step.micro_batches == m
        step()
        th.cuda.synchronize()
        flats.append(step.sync.flat.detach().clone())
    g1, g4 = flats
    scale = float(g1.abs().max())
    assert scale > 0 and float((g1 - g4).abs().max()) <= 2e-4 * scale, (float((g1 - g4).abs().max()), scale)


def test_config4_shard_in_one_pass_at_full_size(gpu):
    """BASELINE configs[3]'s per-GPU shard at FULL size: 1024 pairs of pattern (16, 32) x target (512, 4096), add_rev -- the
    union graph has 8,454,144 edge rows, ONE [E, 128] fp32 array is 4.33 GB, beyond what a 32-bit byte offset reaches.
    The typed / weight-gradient kernels address rows by index (structured buffer descriptors), so the shard runs as one
    pass (round 2 needed four micro-batches); its flat gradient must equal the four-pass gradient -- whose arrays all stay
    below 4 GiB -- to fp32 rounding of the different summation split."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    flats = []
    for m in (4, 1):
        cfg = dict(bench.CFG4, micro_batches=m, act="leaky_relu", emb="Equivariant")
        shard = bench.make_shard(cfg, 0, gpu)
        assert 1024 * 2 * (32 + 4096) * 128 * 4 > 2 ** 32
        step, model = bench.build_step(cfg, shard, gpu)
        assert step.micro_batches == m
        step()
        th.cuda.synchronize()
        flats.append(step.sync.flat.detach().clone())
        del step, model, shard
        th.cuda.empty_cache()
    g4, g1 = flats
    scale = float(g4.abs().max())
    assert scale > 0 and bool(th.isfinite(g1).all())
    assert float((g1 - g4).abs().max()) <= 5e-4 * scale, (float((g1 - g4).abs().max()), scale)

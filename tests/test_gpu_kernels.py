"""GPU parity of the individual HIP kernels through the C ABI (ctypes), against numpy /
the CPU oracle: integer results bit-exact, fp32 sums exact where the order is pinned
(plain seg-sum in ascending eid = index_add_ order) and 1e-6-relative otherwise; plus
size-independent properties at BASELINE config-2 size."""
import numpy as np
import pytest
import torch as th

import dmp_oracle as O
from util_graphs import er_batch, skewed_graph

pytestmark = pytest.mark.gpu


def _t(a):
    return th.from_numpy(np.asarray(a))


def _index(src, dst, n, rev, dev):
    from dualmessagepassing_amd.graph import GraphIndex
    return GraphIndex(_t(src).to(dev), _t(dst).to(dev), n, None if rev is None else _t(rev).to(dev), validate=True)


def _csr_ref(key, flag, n):
    order = np.argsort(key, kind="stable")
    ptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(key, minlength=n), out=ptr[1:])
    ent = (order.astype(np.int64) << 1) | (flag[order].astype(np.int64) if flag is not None else 0)
    return ptr.astype(np.int32), ent.astype(np.int32)


GRAPHS = {
    "er_batch": lambda rng: er_batch(6, 11, 20, rng)[:4],
    "skewed": lambda rng: skewed_graph(300, 5000, rng) + (rng.random(5000) < 0.5, 300),
    "isolated": lambda rng: (np.array([0, 0, 5]), np.array([5, 5, 0]), np.array([False, True, False]), 9),
    "empty": lambda rng: (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, bool), 4),
    "hub": lambda rng: (rng.integers(0, 50, 4000).astype(np.int64), np.zeros(4000, np.int64),
                        rng.random(4000) < 0.3, 50),
}


@pytest.mark.parametrize("name", sorted(GRAPHS))
def test_csr_and_incidence_build_bit_exact(name, gpu):
    rng = np.random.default_rng(5)
    src, dst, rev, n = GRAPHS[name](rng)
    src, dst = np.asarray(src, np.int64), np.asarray(dst, np.int64)
    ix = _index(src, dst, n, rev, gpu)
    ip, ie = _csr_ref(dst, rev, n)
    op, oe = _csr_ref(src, rev, n)
    assert np.array_equal(ix.in_ptr.cpu().numpy(), ip) and np.array_equal(ix.in_ent.cpu().numpy(), ie)
    assert np.array_equal(ix.out_ptr.cpu().numpy(), op) and np.array_equal(ix.out_ent.cpu().numpy(), oe)
    assert np.array_equal(ix.in_deg.cpu().numpy(), np.bincount(dst, minlength=n))
    assert np.array_equal(ix.out_deg.cpu().numpy(), np.bincount(src, minlength=n))
    assert np.array_equal(ix.src32.cpu().numpy(), src.astype(np.int32))
    assert np.array_equal(ix.dst32.cpu().numpy(), dst.astype(np.int32))
    inc_ptr, inc_ent = ix.incidence()
    ref_ptr = ip.astype(np.int64) + op
    def merged(w):       # in-entries and out-entries (flag flipped) by ascending edge id, an in-entry first on a tie (self loop)
        both = [(int(x) >> 1, 0, int(x)) for x in ie[ip[w]:ip[w + 1]]] + [(int(x) >> 1, 1, int(x) ^ 1) for x in oe[op[w]:op[w + 1]]]
        return np.array([x for _, _, x in sorted(both)], dtype=np.int64)
    ref_ent = np.concatenate([merged(w) for w in range(n)]) if n else np.zeros(0)
    assert np.array_equal(inc_ptr.cpu().numpy(), ref_ptr.astype(np.int32))
    assert np.array_equal(inc_ent.cpu().numpy(), ref_ent.astype(np.int32))


def test_csr_build_flags_out_of_range_endpoint(gpu):
    from dualmessagepassing_amd import _lib
    from dualmessagepassing_amd.graph import GraphIndex
    with pytest.raises(_lib.DmpError):
        GraphIndex(th.tensor([0, 7], device=gpu), th.tensor([1, 0], device=gpu), 3, None, validate=True)


@pytest.mark.parametrize("h", [1, 3, 4, 20, 64, 128, 256, 260, 512])
@pytest.mark.parametrize("name", ["er_batch", "skewed", "isolated", "hub"])
def test_seg_sum_and_gathers(name, h, gpu):
    from dualmessagepassing_amd import ops
    rng = np.random.default_rng(h)
    src, dst, rev, n = GRAPHS[name](rng)
    e = len(src)
    ix = _index(src, dst, n, rev, gpu)
    gen = th.Generator().manual_seed(h)
    m = th.randn(e, h, generator=gen)
    w = th.rand(e, generator=gen) + 0.5
    td, ts, tr = _t(dst), _t(src), _t(rev)
    mg = m.to(gpu)
    # plain: same order as index_add_ -> bit-exact
    got = ops.seg_sum_raw(mg, ix.in_ptr, ix.in_ent, n)
    assert th.equal(got.cpu(), O.seg_sum(m, td, n))
    # strided input (column slice of a wider matrix, the fused-GEMM-output case)
    wide = th.randn(e, 2 * h + 4, generator=gen).to(gpu)
    got = ops.seg_sum_raw(wide[:, 4:4 + h], ix.in_ptr, ix.in_ent, n)
    assert th.equal(got.cpu(), O.seg_sum(wide[:, 4:4 + h].cpu(), td, n))
    # weighted
    got = ops.seg_sum_raw(mg, ix.in_ptr, ix.in_ent, n, w.to(gpu))
    ref = O.seg_sum(m * w[:, None], td, n)
    assert th.allclose(got.cpu(), ref, rtol=1e-6, atol=1e-6)
    # split by flag with signs
    got = ops.seg_sum_raw(mg, ix.in_ptr, ix.in_ent, n, None, True, -1.0, 1.0)
    ref = th.cat([-O.seg_sum(m * (~tr)[:, None], td, n), O.seg_sum(m * tr[:, None], td, n)], 1)
    assert th.equal(got.cpu(), ref)
    # by source
    got = ops.seg_sum_raw(mg, ix.out_ptr, ix.out_ent, n)
    assert th.equal(got.cpu(), O.seg_sum(m, ts, n))
    # gathers
    x = th.randn(n, h, generator=gen)
    assert th.equal(ops.gather_rows_raw(x.to(gpu), ix.src32).cpu(), x[ts])
    assert th.equal(ops.gather_rows_raw(x.to(gpu), ix.dst32, w.to(gpu)).cpu(), x[td] * w[:, None])
    d2 = th.randn(n, 2 * h, generator=gen)
    got = ops.gather_select_raw(d2.to(gpu), ix.dst32, ix.rev8, h, None, -1.0, 1.0)
    ref = th.where(tr[:, None], d2[td][:, h:], -d2[td][:, :h])
    assert th.equal(got.cpu(), ref)


@pytest.mark.parametrize("h", [3, 8, 64, 128, 256])
def test_edge_combine_fwd_bwd(h, gpu):
    from dualmessagepassing_amd import ops
    rng = np.random.default_rng(h + 1)
    src, dst, rev, n = GRAPHS["er_batch"](rng)
    e = len(src)
    ix = _index(src, dst, n, rev, gpu)
    gen = th.Generator().manual_seed(h)
    G = th.randn(e, 2 * h, generator=gen, dtype=th.float64)
    P = th.randn(n, 2 * h, generator=gen, dtype=th.float64)
    b = th.randn(h, generator=gen, dtype=th.float64)
    wy = th.randn(e, h, generator=gen, dtype=th.float64)
    ts, td, tr = _t(src), _t(dst), _t(rev)
    out_deg = O.out_degrees(ts, n)
    coef64 = 2 * (1 + (1 + out_deg.double()).log2())
    Go, Po, bo = G.clone().requires_grad_(True), P.clone().requires_grad_(True), b.clone().requires_grad_(True)
    a = th.where(tr, ts, td)
    c = th.where(tr, td, ts)
    Y = Go[:, :h] + coef64[td][:, None] * Go[:, h:] + (Po[a][:, :h] - Po[c][:, h:]) + bo
    (Y * wy).sum().backward()
    Gg, Pg, bg = (t.float().to(gpu).requires_grad_(True) for t in (G, P, b))
    coef = ix.degree_coef(ix.out_deg)
    assert th.allclose(coef.cpu().double(), coef64, rtol=3e-7, atol=0)
    Yg = ops.edge_combine(Gg, Pg, bg, coef, ix)
    assert th.allclose(Yg.cpu().double(), Y.detach(), rtol=2e-6, atol=2e-6)
    (Yg * wy.float().to(gpu)).sum().backward()
    assert th.allclose(Gg.grad.cpu().double(), Go.grad, rtol=2e-6, atol=2e-6)
    assert th.allclose(Pg.grad.cpu().double(), Po.grad, rtol=2e-5, atol=2e-5)
    assert th.allclose(bg.grad.cpu().double(), bo.grad, rtol=2e-5, atol=2e-4)


# ----------------------------------------------------------------------------- full size (configs 2 and 4)
@pytest.mark.parametrize("batch,g_nodes,g_edges", [(1024, 64, 256), (1024, 512, 4096)])
def test_config2_size_properties(batch, g_nodes, g_edges, gpu):
    """BASELINE configs[1] (B=1024 x target(64, 256->512 edges)) and configs[3]'s per-GPU shard (B=1024 x target(512,
    4096->8192 edges): N = 524,288, E = 8,388,608, one [E,H] array = 4.3 GB), H=128: size-independent checks."""
    from dualmessagepassing_amd import ops
    rng = np.random.default_rng(2000)
    src, dst, rev, n, _, _ = er_batch(batch, g_nodes, g_edges, rng)
    e, h = len(src), 128
    ix = _index(src, dst, n, rev, gpu)
    # (1) sum of all-ones rows = in-degree, split by flag; exact in fp32 (small integers)
    ones = th.ones(e, h, device=gpu)
    s = ops.seg_sum_raw(ones, ix.in_ptr, ix.in_ent, n, None, True, 1.0, 1.0)
    tr, td = _t(rev).to(gpu), _t(dst).to(gpu)
    deg0 = th.bincount(td[~tr], minlength=n).float()
    deg1 = th.bincount(td[tr], minlength=n).float()
    assert th.equal(s[:, :h], deg0[:, None].expand(n, h)) and th.equal(s[:, h:], deg1[:, None].expand(n, h))
    # (2) run-to-run bitwise determinism and column-sum conservation
    gen = th.Generator(device="cpu").manual_seed(0)
    m = th.randn(e, h, generator=gen).to(gpu)
    a1 = ops.seg_sum_raw(m, ix.in_ptr, ix.in_ent, n)
    a2 = ops.seg_sum_raw(m, ix.in_ptr, ix.in_ent, n)
    assert th.equal(a1, a2)
    # fp32 per-row sums then an fp64 column sum: rounding of ~64k (config 4: ~512k) partial sums, |err| << 1e-2 (1e-1)
    assert th.allclose(a1.double().sum(0), m.double().sum(0), rtol=1e-6, atol=2e-2 * (e / 524288) ** 0.5)
    # (3) equals torch's own index_add on device within fp32 reassociation
    ref = th.zeros(n, h, device=gpu).index_add_(0, td, m)
    assert th.allclose(a1, ref, rtol=1e-5, atol=1e-5)
    # (4) linearity: seg_sum(2m + k) = 2 seg_sum(m) + k * indeg
    a3 = ops.seg_sum_raw(2 * m + 3.0, ix.in_ptr, ix.in_ent, n)
    assert th.allclose(a3, 2 * a1 + 3.0 * (deg0 + deg1)[:, None], rtol=1e-5, atol=1e-4)
    # (5) gather is the transpose of seg-sum:  <seg_sum(m), y> = <m, gather(y)>
    y = th.randn(n, h, generator=gen).to(gpu)
    lhs = (a1.double() * y.double()).sum()
    rhs = (m.double() * ops.gather_rows_raw(y, ix.dst32).double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-7 * abs(float(lhs)) + 1e-3 * (e / 524288) ** 0.5


@pytest.mark.parametrize("rows,h", [(1, 4), (37, 8), (1000, 128), (70000, 128), (5000, 256), (300, 512)])
def test_row_epilogue_kernels(rows, h, gpu):
    """gate_residual / scale_rows_colsum / relu_bwd_colsum / bwd_g_colsum / colsum / reduce_partials
    (csrc/dmp_fused.hip) against torch formulas: elementwise parts bit-exact, column sums 1e-5."""
    from dualmessagepassing_amd import fused
    gen = th.Generator().manual_seed(rows + h)
    a = th.randn(rows, h, generator=gen).to(gpu)
    b = th.randn(rows, h, generator=gen).to(gpu)
    gate = (th.rand(rows, generator=gen) < 0.6).float().to(gpu)
    assert th.equal(fused.gate_residual(a, b, gate), a + b * gate[:, None])
    assert th.equal(fused.gate_residual(None, b, gate), b * gate[:, None])
    assert th.equal(fused.gate_residual(a, b, None), a + b)
    d, cs = fused.scale_rows_colsum(a, gate)
    assert th.equal(d, a * gate[:, None])
    assert th.allclose(cs, d.double().sum(0).float(), rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    d, cs = fused.scale_rows_colsum(a, None)
    assert d.data_ptr() == a.data_ptr()
    assert th.allclose(cs, a.double().sum(0).float(), rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    dh = a.clone()
    d, cs = fused.relu_bwd_colsum_(dh, b)
    ref = th.where(b > 0, a, th.zeros_like(a))
    assert th.equal(d, ref) and d.data_ptr() == dh.data_ptr()
    assert th.allclose(cs, ref.double().sum(0).float(), rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    n = max(1, rows // 3)
    coef = th.rand(n, generator=gen).to(gpu) + 1.0
    dst = th.randint(0, n, (rows,), generator=gen).int().to(gpu)
    dg, cs = fused.bwd_g_colsum(a, coef, dst)
    assert th.equal(dg, th.cat([a, a * coef[dst.long()][:, None]], 1))
    assert th.allclose(cs, a.double().sum(0).float(), rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    assert th.allclose(fused.colsum(a), a.double().sum(0).float(), rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    # relu(a + b + bias) in place, b a column slice of a wider matrix
    wide = th.randn(rows, 3 * h, generator=gen).to(gpu)
    bias = th.randn(h, generator=gen).to(gpu)
    a2 = a.clone()
    assert fused.add_bias_relu_(a2, wide[:, h:2 * h], bias) is a2
    assert th.equal(a2, ((a + wide[:, h:2 * h]) + bias).clamp_min(0))
    # two launches give identical bits (fixed reduction tree)
    assert th.equal(fused.colsum(a), fused.colsum(a))
    part = th.randn(7, 4 * h, generator=gen).to(gpu)
    out = fused.reduce_partials(part)
    assert th.allclose(out, part.sum(0), rtol=1e-6, atol=1e-5)
    out2 = fused.reduce_partials(part, out.clone(), accumulate=True)
    assert th.allclose(out2, 2 * part.sum(0), rtol=1e-6, atol=1e-5)


def test_deferred_reductions_equal_immediate(gpu):
    """Reductions recorded inside ``deferred_reductions`` run as one multi-segment launch (two when more
    than DMP_REDUCE_MAX_SEGMENTS jobs are queued) and give the bits of the one-at-a-time kernel."""
    from dualmessagepassing_amd import fused
    gen = th.Generator().manual_seed(12)
    shapes = [(1, 4), (7, 128), (1024, 128), (134, 16384), (3, 49152)] + [(5 + i, 4 * (i + 1)) for i in range(14)]
    parts = [th.randn(s, l, generator=gen).to(gpu) for s, l in shapes]
    want = [fused.reduce_partials(p) for p in parts]
    with fused.deferred_reductions():
        got = [fused.reduce_partials(p) for p in parts]
    assert len(parts) > fused.MAX_REDUCE_SEGMENTS
    for w, g, (s, l) in zip(want, got, shapes):
        assert g.shape == (l,) and th.equal(w, g), (s, l)
    # a job list left by an exception is dropped, not launched
    try:
        with fused.deferred_reductions():
            fused.reduce_partials(parts[1])
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert not fused._deferred.stack


def test_split_k_weight_gradient_product(gpu):
    from dualmessagepassing_amd import fused
    gen = th.Generator().manual_seed(9)
    for rows in (100, 16384, 50000):
        a = th.randn(rows, 128, generator=gen).to(gpu)
        b = th.randn(rows, 256, generator=gen).to(gpu)
        ref = (a.double().t() @ b.double()).float()
        got = fused.atb(a, b)
        assert th.allclose(got, ref, rtol=1e-4, atol=2e-3)


def test_gather_select_with_base(gpu):
    from dualmessagepassing_amd import ops
    rng = np.random.default_rng(4)
    src, dst, rev, n = GRAPHS["er_batch"](rng)
    ix = _index(src, dst, n, rev, gpu)
    h = 64
    gen = th.Generator().manual_seed(2)
    d2 = th.randn(n, 2 * h, generator=gen).to(gpu)
    base = th.randn(len(src), h, generator=gen).to(gpu)
    plain = ops.gather_select_raw(d2, ix.dst32, ix.rev8, h, None, -1.0, 1.0)
    got = ops.gather_select_raw(d2, ix.dst32, ix.rev8, h, None, -1.0, 1.0, base=base)
    assert th.equal(got, base + plain)


@pytest.mark.parametrize("rows", [1, 127, 128, 129, 5000, 70001])
def test_mfma_kernels_h128(rows, gpu):
    """Fused MFMA kernels (csrc/dmp_mfma.hip) against fp64 formulas: plain GEMM (both panel counts,
    both weight layouts), edge forward (GEMM + edge_combine + relu), output (Linear + gate + residual).
    fp32 MFMA = k-ordered fma chain: errors ~1e-6 relative."""
    from dualmessagepassing_amd import _lib, fused
    from dualmessagepassing_amd._lib import ptr, stream_ptr
    lib = _lib.load()
    h = 128
    gen = th.Generator().manual_seed(rows)
    a = th.randn(rows, h, generator=gen).to(gpu)
    for ncols in (128, 256):
        w = th.randn(h, ncols, generator=gen).to(gpu)
        c = th.empty(rows, ncols, device=gpu)
        _lib.check(lib.dmp_gemm_k128(ptr(a), h, ptr(w), ncols, 0, ptr(c), ncols, rows, ncols, stream_ptr()), "gemm")
        ref = (a.double() @ w.double())
        assert th.allclose(c.double(), ref, rtol=1e-5, atol=2e-4)
        wt = w.t().contiguous()  # [ncols, 128]: transposed layout
        _lib.check(lib.dmp_gemm_k128(ptr(a), h, ptr(wt), h, 1, ptr(c), ncols, rows, ncols, stream_ptr()), "gemm_t")
        assert th.allclose(c.double(), ref, rtol=1e-5, atol=2e-4)
    # edge forward on a random graph with `rows` edges
    n = max(2, rows // 5)
    rng = np.random.default_rng(rows)
    src = rng.integers(0, n, rows).astype(np.int64)
    dst = rng.integers(0, n, rows).astype(np.int64)
    rev = rng.random(rows) < 0.5
    ix = _index(src, dst, n, rev, gpu)
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    xp = th.randn(n, 3 * h, generator=gen).to(gpu)
    bias = th.randn(h, generator=gen).to(gpu)
    coef = ix.degree_coef(ix.out_deg)
    got = fused.edge_fwd_mfma(a, wes, xp[:, h:], 3 * h, bias, coef, ix)
    g = a.double() @ wes.double()
    ts, td, tr = _t(src).to(gpu), _t(dst).to(gpu), _t(rev).to(gpu)
    ai, bi = th.where(tr, ts, td), th.where(tr, td, ts)
    ref = g[:, :h] + coef.double()[td][:, None] * g[:, h:] + xp.double()[ai][:, h:2 * h] - xp.double()[bi][:, 2 * h:] + bias.double()
    assert th.allclose(got.double(), ref.clamp_min(0), rtol=1e-5, atol=2e-4)
    # same result as the two-kernel path (GEMM + edge_combine) within fp32 re-association
    two = fused.edge_combine_raw(a @ wes, 2 * h, xp[:, h:], 3 * h, bias, coef, ix, h, relu=True)
    assert th.allclose(got, two, rtol=1e-5, atol=2e-4)
    # output kernel
    w2 = (th.randn(h, h, generator=gen) * 0.1).to(gpu)
    gate = (th.rand(rows, generator=gen) < 0.7).float().to(gpu)
    prev = th.randn(rows, h, generator=gen).to(gpu)
    for g_, p_ in ((gate, prev), (None, prev), (gate, None), (None, None)):
        got = fused.out_fwd_mfma(a, w2, bias, g_, p_)
        ref = a.double() @ w2.double().t() + bias.double()
        if g_ is not None:
            ref = ref * g_.double()[:, None]
        if p_ is not None:
            ref = ref + p_.double()
        assert th.allclose(got.double(), ref, rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("rows", [1, 129, 4097, 70001])
def test_mfma_backward_kernels_h128(rows, gpu):
    """dmp_bwd_h1_fused / dmp_bwd_z_fused against fp64 formulas and against the unfused kernels."""
    from dualmessagepassing_amd import fused, ops
    h = 128
    gen = th.Generator().manual_seed(rows + 1)
    rng = np.random.default_rng(rows + 1)
    n = max(2, rows // 5)
    src = rng.integers(0, n, rows).astype(np.int64)
    dst = rng.integers(0, n, rows).astype(np.int64)
    rev = rng.random(rows) < 0.5
    ix = _index(src, dst, n, rev, gpu)
    coef = ix.degree_coef(ix.out_deg)
    d_o = th.randn(rows, h, generator=gen).to(gpu)
    h1 = th.randn(rows, h, generator=gen).clamp_min(0).to(gpu)
    w2 = (th.randn(h, h, generator=gen) * 0.1).to(gpu)
    d_g, cs = fused.bwd_h1_mfma(d_o, w2, h1, coef, ix)
    td = _t(dst).to(gpu)
    dpre = th.where(h1 > 0, (d_o.double() @ w2.double()), th.zeros(1, dtype=th.float64, device=gpu))
    ref = th.cat([dpre, dpre * coef.double()[td][:, None]], 1)
    assert th.allclose(d_g.double(), ref, rtol=1e-5, atol=2e-4)
    assert th.allclose(cs.double(), dpre.sum(0), rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    # dPre alone with the edge gate fused in: dPre = h1 > 0 ? gate (d_o w2) : 0 (gate * d_o never materialised)
    gate = (th.rand(rows, generator=gen) * (th.rand(rows, generator=gen) > 0.3)).to(gpu)
    d_p, csg = fused.bwd_h1_mfma(d_o, w2, h1, coef, ix, both_halves=False, gate=gate)
    refg = dpre * gate.double()[:, None]
    assert d_p.shape == (rows, h) and th.allclose(d_p.double(), refg, rtol=1e-5, atol=2e-4)
    assert th.allclose(csg.double(), refg.sum(0), rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    d_p1, _ = fused.bwd_h1_mfma(d_o, w2, h1, coef, ix, both_halves=False)
    assert th.equal(d_p1, d_g[:, :h])
    # no graph at all (the node update's MLP), output into a column slice of a wider matrix
    wide = th.full((rows, 3 * h), 7.0, device=gpu)
    d_p2, cs2_ = fused.bwd_h1_mfma(d_o, w2, h1, both_halves=False, gate=gate, out=wide[:, :h])
    assert d_p2.data_ptr() == wide.data_ptr() and th.equal(wide[:, :h], d_p) and th.equal(cs2_, csg)
    assert bool((wide[:, h:] == 7.0).all())
    g2, cs2 = fused.relu_bwd_g_colsum(d_o @ w2, h1, coef, ix.dst32)      # the two-kernel path
    assert th.allclose(d_g, g2, rtol=1e-5, atol=2e-4) and th.allclose(cs, cs2, rtol=1e-4, atol=1e-3 * max(1.0, rows ** 0.5))
    # input gradient
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    d_s = th.randn(n, 2 * h, generator=gen).to(gpu)
    base = th.randn(rows, h, generator=gen).to(gpu)
    tr = _t(rev).to(gpu)
    gs = th.where(tr[:, None], d_s.double()[td][:, h:], -d_s.double()[td][:, :h])
    for b_ in (base, None):
        got = fused.bwd_z_mfma(d_g, wes, d_s, b_, coef, ix)
        ref = gs + d_g.double() @ wes.double().t()
        if b_ is not None:
            ref = ref + b_.double()
        assert th.allclose(got.double(), ref, rtol=1e-5, atol=3e-4)
        two = ops.gather_select_raw(d_s, ix.dst32, ix.rev8, h, None, -1.0, 1.0, base=b_)
        two.addmm_(d_g, wes.t())
        assert th.allclose(got, two, rtol=1e-5, atol=3e-4)


def test_mfma_kernels_are_repeatable_under_load(gpu):
    """The fused MFMA kernels at a multi-tile size, 12 back-to-back launches each: every launch must
    reproduce the first one bit for bit and match the fp64 formula (catches register / LDS hazards
    that only show when several workgroups share a SIMD)."""
    from dualmessagepassing_amd import fused
    rows, h = 150001, 128
    gen = th.Generator().manual_seed(11)
    rng = np.random.default_rng(11)
    n = rows // 7
    ix = _index(rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64), n,
                rng.random(rows) < 0.5, gpu)
    coef = ix.degree_coef(ix.out_deg)
    d_o = th.randn(rows, h, generator=gen).to(gpu)
    h1 = th.randn(rows, h, generator=gen).clamp_min(0).to(gpu)
    w2 = (th.randn(h, h, generator=gen) * 0.1).to(gpu)
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    xp = th.randn(n, 3 * h, generator=gen).to(gpu)
    d_s = th.randn(n, 2 * h, generator=gen).to(gpu)
    bias = th.randn(h, generator=gen).to(gpu)
    gate = th.rand(rows, generator=gen).to(gpu)
    calls = {
        "edge_fwd": lambda: fused.edge_fwd_mfma(d_o, wes, xp[:, h:], 3 * h, bias, coef, ix),
        "out_fwd": lambda: fused.out_fwd_mfma(h1, w2, bias, gate, d_o),
        "bwd_h1": lambda: fused.bwd_h1_mfma(d_o, w2, h1, coef, ix)[0],
        "bwd_z": lambda: fused.bwd_z_mfma(th.cat([d_o, d_o], 1), wes, d_s, h1, coef, ix),
    }
    td = th.from_numpy(ix.dst32.cpu().numpy().astype(np.int64)).to(gpu)
    dpre = th.where(h1 > 0, d_o.double() @ w2.double(), th.zeros(1, dtype=th.float64, device=gpu))
    want_h1 = th.cat([dpre, dpre * coef.double()[td][:, None]], 1)
    for name, f in calls.items():
        outs = [f() for _ in range(12)]
        th.cuda.synchronize()
        for o in outs[1:]:
            assert th.equal(o, outs[0]), name
        if name == "bwd_h1":
            assert th.allclose(outs[0].double(), want_h1, rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("rows", [31, 1000, 70001])
def test_typed_edge_kernels_equal_untyped(rows, gpu):
    """Class-typed edge_fwd / bwd_z (one W_g = A' + c_g B' panel per degree class, rows gathered and
    scattered through the class-sorted tile list) against the untyped fused kernels and fp64."""
    from dualmessagepassing_amd import fused
    h = 128
    gen = th.Generator().manual_seed(rows + 7)
    rng = np.random.default_rng(rows + 7)
    n = max(2, rows // 6)
    src, dst = rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64)
    rev = rng.random(rows) < 0.5
    ix = _index(src, dst, n, rev, gpu)
    coef = ix.degree_coef(ix.out_deg)
    ce = ix.edge_select(coef)[2].cpu().numpy()
    builds = {"device": ix.class_tiles(coef), "by_value": ix._class_tiles_by_value(coef)}
    for how, (slot_edge, tile_scale, num_tiles, bound) in builds.items():
        nt = int(num_tiles.item())
        full = slot_edge.view(-1, 32).cpu().numpy()
        se = full[:nt]
        assert sorted(se[se >= 0].tolist()) == list(range(rows)), how             # every edge in exactly one slot
        assert (full[nt:] == -1).all(), how
        ts = tile_scale[:nt].cpu().numpy()
        for t in range(nt):
            ids = se[t][se[t] >= 0]
            assert len(ids) > 0 and np.all(ce[ids] == ts[t]), how                  # one class per tile, its own scale
        assert nt <= rows // 32 + len(np.unique(ce)), how
        assert np.all(np.diff(ts) >= 0), how                                       # classes ascending
    # device builder: within a class nodes ascending, in-edges of a node in ascending edge id
    se = builds["device"][0].view(-1)[:int(builds["device"][2].item()) * 32].cpu().numpy()
    live = se[se >= 0]
    keyed = np.stack([ce[live], dst[live], live], 1)
    assert np.array_equal(keyed, keyed[np.lexsort((keyed[:, 2], keyed[:, 1], keyed[:, 0]))])
    z = th.randn(rows, h, generator=gen).to(gpu)
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    xp = th.randn(n, 3 * h, generator=gen).to(gpu)
    bias = th.randn(h, generator=gen).to(gpu)
    got = fused.edge_fwd_typed(z, wes, xp[:, h:], 3 * h, bias, coef, ix)
    ref = fused.edge_fwd_mfma(z, wes, xp[:, h:], 3 * h, bias, coef, ix)
    assert th.allclose(got, ref, rtol=1e-5, atol=2e-5)
    d_pre = th.randn(rows, 2 * h, generator=gen).to(gpu)
    d_s = th.randn(n, 2 * h, generator=gen).to(gpu)
    base = th.randn(rows, h, generator=gen).to(gpu)
    for b_ in (base, None):
        got = fused.bwd_z_typed(d_pre, 2 * h, wes, d_s, b_, coef, ix)
        ref = fused.bwd_z_mfma(d_pre, wes, d_s, b_, coef, ix)
        assert th.allclose(got, ref, rtol=1e-5, atol=2e-5)
    assert th.equal(fused.edge_fwd_typed(z, wes, xp[:, h:], 3 * h, bias, coef, ix),
                    fused.edge_fwd_typed(z, wes, xp[:, h:], 3 * h, bias, coef, ix))


@pytest.mark.parametrize("rows,keep", [(31, 0.5), (1000, 0.4), (70001, 0.46), (500, 0.0), (500, 1.0)])
def test_class_tiles_over_the_kept_edges(rows, keep, gpu):
    """``dmp_class_tiles_gated``: the tile list of the edges a 0 / 1 gate keeps -- every kept edge in exactly one slot, no
    other edge anywhere, one class per tile with its scale, classes / nodes / edge ids ascending -- and the three class-typed
    launches over it equal the launches over all edges on the kept rows (rows under a zero gate: never read, NaN there)."""
    from dualmessagepassing_amd import fused
    h = 128
    gen = th.Generator().manual_seed(rows + 11)
    rng = np.random.default_rng(rows + 11)
    n = max(2, rows // 6)
    src, dst = rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64)
    rev = rng.random(rows) < 0.5
    ix = _index(src, dst, n, rev, gpu)
    coef = ix.degree_coef(ix.out_deg)
    ce = ix.edge_select(coef)[2].cpu().numpy()
    g_np = (rng.random(rows) < keep).astype(np.float32)
    gate = th.from_numpy(g_np).to(gpu)
    gate._dmp_binary = True
    slot_edge, tile_scale, num_tiles, bound = ix.class_tiles_gated(coef, gate)
    assert bound == ix.class_tiles(coef)[3]
    nt = int(num_tiles.item())
    full = slot_edge.view(-1, 32).cpu().numpy()
    se = full[:nt]
    kept = np.nonzero(g_np)[0]
    assert sorted(se[se >= 0].tolist()) == kept.tolist()
    assert (full[nt:] == -1).all()
    ts = tile_scale[:nt].cpu().numpy()
    for t in range(nt):
        ids = se[t][se[t] >= 0]
        assert len(ids) > 0 and np.all(ce[ids] == ts[t])
    assert nt <= len(kept) // 32 + len(np.unique(ce[kept])) if len(kept) else nt == 0
    live = se.reshape(-1)[se.reshape(-1) >= 0]
    keyed = np.stack([ce[live], dst[live], live], 1)
    assert np.array_equal(keyed, keyed[np.lexsort((keyed[:, 2], keyed[:, 1], keyed[:, 0]))])
    if rows < 1000:
        return
    # the launches: kept rows as over all edges, dead rows untouched
    dead = gate == 0
    z = th.randn(rows, h, generator=gen).to(gpu)
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    xp = th.randn(n, 3 * h, generator=gen).to(gpu)
    bias = th.randn(h, generator=gen).to(gpu)
    zp = z.clone()
    zp[dead] = float("nan")
    ref = fused.edge_fwd_typed(z, wes, xp[:, h:], 3 * h, bias, coef, ix)
    got = fused.edge_fwd_typed(zp, wes, xp[:, h:], 3 * h, bias, coef, ix, dead_gate=gate)
    assert th.equal(got[~dead], ref[~dead])
    d_pre = th.randn(rows, h, generator=gen).to(gpu) * gate[:, None]
    dpp = d_pre.clone()
    dpp[dead] = float("nan")
    ref_w = fused.atb_typed(z, d_pre, coef, ix)
    got_w = fused.atb_typed(zp, dpp, coef, ix, gate=gate)
    assert bool(th.isfinite(got_w).all())
    assert float((got_w - ref_w).abs().max()) <= 2e-5 * float(ref_w.abs().max())
    d_s = th.randn(n, 2 * h, generator=gen).to(gpu)
    base = th.randn(rows, h, generator=gen).to(gpu)
    bp = base.clone()
    bp[dead] = float("nan")
    ref_z = fused.bwd_z_typed(d_pre, h, wes, d_s, base, coef, ix)
    got_z = fused.bwd_z_typed(dpp, h, wes, d_s, bp, coef, ix, gate=gate, dead_rows="zero")
    assert th.equal(got_z[~dead], ref_z[~dead]) and float(got_z[dead].abs().max()) == 0.0


@pytest.mark.parametrize("rows,h", [(1000, 128), (70001, 128), (5000, 64)])
def test_second_linear_over_the_kept_edges_tiles(rows, h, gpu):
    """``dmp_out_fwd_typed`` / ``dmp_bwd_h1_typed`` (the kept edges' tiles of a 0 / 1 gate) against the one-panel kernels over
    all rows: equal on the kept rows up to the summation order of the bf16x6 products (same products, same order inside a
    row: equal bits), the other rows untouched, the column sums to fp32 accuracy; rows under a zero gate are never read."""
    from dualmessagepassing_amd import fused
    gen = th.Generator().manual_seed(rows + h)
    rng = np.random.default_rng(rows + h)
    n = max(2, rows // 6)
    src, dst = rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64)
    ix = _index(src, dst, n, rng.random(rows) < 0.5, gpu)
    coef = ix.degree_coef(ix.out_deg)
    gate = th.from_numpy((rng.random(rows) < 0.46).astype(np.float32)).to(gpu)
    gate._dmp_binary = True
    dead = gate == 0
    tiles = fused.live_tiles(ix, coef, gate)
    assert tiles is not None
    h1 = th.randn(rows, h, generator=gen).to(gpu)
    prev = th.randn(rows, h, generator=gen).to(gpu) * gate[:, None]
    d_o = th.randn(rows, h, generator=gen).to(gpu)
    W2 = (th.randn(h, h, generator=gen) / h ** 0.5).to(gpu)
    b2 = th.randn(h, generator=gen).to(gpu)
    ref_out = fused.out_fwd_mfma(h1, W2, b2, gate, prev)
    ref_dg, ref_db, ref_rows = fused.bwd_h1_mfma(d_o, W2, h1, both_halves=False, gate=gate, slope=0.18, rows_colsum=True)
    h1p, prevp, dop = h1.clone(), prev.clone(), d_o.clone()
    for t in (h1p, prevp, dop):
        t[dead] = float("nan")
    saved = fused.dead_rows_buffer
    fused.dead_rows_buffer = lambda shape, device: th.full(shape, 7.5, dtype=th.float32, device=device)
    try:
        out = fused.out_fwd_typed(h1p, W2.t().contiguous(), b2, prevp, tiles)
        dg, db, db_rows = fused.bwd_h1_typed(dop, W2, h1p, tiles, slope=0.18)
    finally:
        fused.dead_rows_buffer = saved
    assert bool((out[dead] == 7.5).all()) and bool((dg[dead] == 7.5).all())
    s_out, s_dg = float(ref_out.abs().max()), float(ref_dg.abs().max())
    assert float((out[~dead] - ref_out[~dead]).abs().max()) <= 2e-6 * s_out
    assert float((dg[~dead] - ref_dg[~dead]).abs().max()) <= 2e-6 * s_dg
    kept = float(gate.sum())
    assert float((db - ref_db).abs().max()) <= 2e-6 * max(1.0, kept) * s_dg
    assert float((db_rows - ref_rows).abs().max()) <= 2e-6 * max(1.0, kept) * float(d_o.abs().max())


@pytest.mark.parametrize("rows,k,kpad,ascending", [(40, 3, 4, True), (1000, 7, 8, False), (70001, 9, 12, True), (70001, 16, 16, False),
                                                  (300000, 7, 8, True)])
def test_second_linear_with_the_residual_rows_from_their_codes(rows, k, kpad, ascending, gpu):
    """``dmp_out_fwd_typed_codes`` (the first layer's residual rows ``z0 = codes W_e`` as one more k-group of its second Linear, so
    that the [E, H] rows are never in memory) against ``dmp_out_fwd_typed`` reading ``z0`` made in fp64: equal on the kept rows to
    fp32 accuracy (the codes product runs as bf16x6 like the rest), the other rows untouched, rows of h1 under a zero gate and the
    padding columns of the code rows never read (NaN there); the same bits on every launch."""
    from dualmessagepassing_amd import fused
    h = 128
    gen = th.Generator().manual_seed(rows + k)
    rng = np.random.default_rng(rows + k)
    n = max(2, rows // 6)
    src, dst = rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64)
    ix = _index(src, dst, n, rng.random(rows) < 0.5, gpu)
    coef = ix.degree_coef(ix.out_deg)
    gate = th.from_numpy((rng.random(rows) < 0.46).astype(np.float32)).to(gpu)
    gate._dmp_binary = True
    dead = gate == 0
    tiles = fused.ascending_tiles(gate) if ascending else fused.live_tiles(ix, coef, gate)
    assert tiles is not None
    h1 = th.randn(rows, h, generator=gen).to(gpu)
    enc = th.full((rows, kpad), float("nan"), device=gpu)
    enc[:, :k] = th.randn(rows, k, generator=gen).to(gpu)          # (any fp32 codes, not only 0 / 1 bits)
    enc[:, :k] *= gate[:, None]
    Wc = th.randn(k, h, generator=gen).to(gpu)
    W2 = (th.randn(h, h, generator=gen) / h ** 0.5).to(gpu)
    b2 = th.randn(h, generator=gen).to(gpu)
    assert fused.out_codes_ok(h1, enc, k, Wc)
    row0 = 0 if rows == 1000 else min(rows // 3, 4097)           # the first rows: residual rows given, no codes term
    head = th.randn(row0, h, generator=gen).to(gpu) if row0 else None
    z0 = (enc[:, :k].double() @ Wc.double()).float()
    if row0:
        z0[:row0] = head
        enc[:row0] = float("nan")                                # (never read)
    h1p = h1.clone()
    h1p[dead] = float("nan")
    saved = fused.dead_rows_buffer
    fused.dead_rows_buffer = lambda shape, device: th.full(shape, 7.5, dtype=th.float32, device=device)
    try:
        ref = fused.out_fwd_typed(h1p, W2.t().contiguous(), b2, z0, tiles)
        out = fused.out_fwd_typed_codes(h1p, W2.t().contiguous(), b2, enc, k, Wc, tiles, prev=head, row0=row0)
        again = fused.out_fwd_typed_codes(h1p, W2.t().contiguous(), b2, enc, k, Wc, tiles, prev=head, row0=row0)
    finally:
        fused.dead_rows_buffer = saved
    assert bool((out[dead] == 7.5).all())
    assert th.equal(out, again)
    assert float((out[~dead] - ref[~dead]).abs().max()) <= 3e-6 * float(ref[~dead].abs().max())
    exact = z0[~dead].double() + h1[~dead].double() @ W2.double().t() + b2.double()
    assert float((out[~dead].double() - exact).abs().max()) <= 3e-6 * float(exact.abs().max())


@pytest.mark.parametrize("rows,ascending", [(40, False), (1000, True), (70001, True), (70001, False), (300000, True)])
def test_second_linear_backward_in_one_launch(rows, ascending, gpu):
    """``dmp_bwd_h1_w`` (csrc/dmp_h1w.hip: dPre AND dO^T H1 from one pass over the kept edges' tiles; four waves on rows, four on
    columns, one bf16-piece LDS image per operand read by rows and through ds_read_b64_tr_b16) against the two launches it
    replaces: dPre BIT-identical to ``dmp_bwd_h1_typed`` on the kept rows, the other rows untouched, rows under a zero gate never
    read (NaN there); the two column sums and dO^T H1 to fp32 accuracy against fp64; the same bits on every launch."""
    from dualmessagepassing_amd import fused
    h = 128
    gen = th.Generator().manual_seed(rows)
    rng = np.random.default_rng(rows)
    n = max(2, rows // 6)
    src, dst = rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64)
    ix = _index(src, dst, n, rng.random(rows) < 0.5, gpu)
    coef = ix.degree_coef(ix.out_deg)
    gate = th.from_numpy((rng.random(rows) < 0.46).astype(np.float32)).to(gpu)
    gate._dmp_binary = True
    dead = gate == 0
    tiles = fused.ascending_tiles(gate) if ascending else fused.live_tiles(ix, coef, gate)
    assert tiles is not None
    h1 = th.randn(rows, h, generator=gen).to(gpu)
    h1 = th.where(h1 > 0, h1, 0.18 * h1)                       # a saved LeakyReLU output: both signs, the sign is what counts
    h1[rng.integers(0, rows, 5)] = 0.0                         # exact zeros: not positive
    d_o = th.randn(rows, h, generator=gen).to(gpu)
    W2 = (th.randn(h, h, generator=gen) / h ** 0.5).to(gpu)
    h1p, dop = h1.clone(), d_o.clone()
    h1p[dead] = float("nan")
    dop[dead] = float("nan")
    saved = fused.dead_rows_buffer
    fused.dead_rows_buffer = lambda shape, device: th.full(shape, 7.5, dtype=th.float32, device=device)
    try:
        ref_dg, ref_db, ref_rows = fused.bwd_h1_typed(dop, W2, h1p, tiles, slope=0.18)
        ref_w = fused.atb_typed(dop, h1p, coef, ix, gate=gate, plain=True)
        assert fused.h1w_ok(dop, h1p, h)
        dg, db, db_rows, dw = fused.bwd_h1_w(dop, W2, h1p, tiles, slope=0.18)
        dg2, db2, db_rows2, dw2 = fused.bwd_h1_w(dop, W2, h1p, tiles, slope=0.18)
    finally:
        fused.dead_rows_buffer = saved
    assert bool((dg[dead] == 7.5).all())
    assert th.equal(dg[~dead], ref_dg[~dead])                  # the same MFMA sequence per element
    assert th.equal(dg, dg2) and th.equal(db, db2) and th.equal(db_rows, db_rows2) and th.equal(dw, dw2)
    keep = ~dead
    dO64, h64 = d_o[keep].double(), h1[keep].double()
    w64 = dO64.t() @ h64
    scale_w = float((dO64.abs().t() @ h64.abs()).max())        # the natural scale of an entry: sum |a||b|
    assert float((dw.double() - w64).abs().max()) <= 2e-6 * scale_w, float((dw.double() - w64).abs().max()) / scale_w
    assert float((ref_w.double() - w64).abs().max()) <= 2e-6 * scale_w
    kept = float(gate.sum())
    assert float((db_rows.double() - dO64.sum(0)).abs().max()) <= 2e-6 * max(1.0, kept) * float(d_o.abs().max())
    assert float((db - ref_db).abs().max()) <= 2e-6 * max(1.0, kept) * float(ref_dg[keep].abs().max())
    assert float((db_rows - ref_rows).abs().max()) <= 2e-6 * max(1.0, kept) * float(d_o.abs().max())


@pytest.mark.parametrize("rows", [1, 31, 4097, 70001])
@pytest.mark.parametrize("gated", [False, True])
def test_gated_weight_gradient_rows(rows, gated, gpu):
    """fused.atb_rows: (gate (.) a)^T b and the column sums of gate (.) a in one MFMA pass, against the
    fp64 product; the same bits on every launch; strided operands (a column slice of a wider matrix)."""
    from dualmessagepassing_amd import fused
    h = 128
    gen = th.Generator().manual_seed(rows + int(gated))
    wide = th.randn(rows, 2 * h, generator=gen).to(gpu)
    a, b = wide[:, h:], th.randn(rows, h, generator=gen).to(gpu)
    gate = (th.rand(rows, generator=gen) * (th.rand(rows, generator=gen) > 0.3)).to(gpu) if gated else None
    got, cs = fused.atb_rows(a, b, gate)
    ga = a.double() * (gate.double()[:, None] if gated else 1.0)
    tol = 1e-4 * max(1.0, rows ** 0.5)
    assert got.shape == (h, h) and th.allclose(got, (ga.t() @ b.double()).float(), rtol=1e-5, atol=tol)
    assert cs.shape == (h,) and th.allclose(cs, ga.sum(0).float(), rtol=1e-5, atol=tol)
    got2, cs2 = fused.atb_rows(a, b, gate)
    assert th.equal(got, got2) and th.equal(cs, cs2)
    # several 128 x 128 output blocks in one launch: [rows,256]^T [rows,384], operands that are column slices
    if rows > 1:
        a3 = th.randn(rows, 2 * h + 4, generator=gen).to(gpu)[:, :2 * h]
        b3 = th.randn(rows, 3 * h, generator=gen).to(gpu)
        got3, cs3 = fused.atb_rows(a3, b3, gate)
        g3 = a3.double() * (gate.double()[:, None] if gated else 1.0)
        assert got3.shape == (2 * h, 3 * h) and th.allclose(got3, (g3.t() @ b3.double()).float(), rtol=1e-5, atol=tol)
        assert th.allclose(cs3, g3.sum(0).float(), rtol=1e-5, atol=tol)
        assert fused.atb_rows(a3, b3, gate, colsum=False)[1] is None


@pytest.mark.parametrize("rows", [31, 1000, 70001])
def test_typed_weight_gradient(rows, gpu):
    """atb_typed: [z^T dPre | z^T (coef[dst] dPre)] over the class-sorted tiles vs fp64, and bit-stable."""
    from dualmessagepassing_amd import fused
    h = 128
    gen = th.Generator().manual_seed(rows + 9)
    rng = np.random.default_rng(rows + 9)
    n = max(2, rows // 6)
    ix = _index(rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64), n,
                rng.random(rows) < 0.5, gpu)
    coef = ix.degree_coef(ix.out_deg)
    z = th.randn(rows, h, generator=gen).to(gpu)
    d_pre = th.randn(rows, h, generator=gen).to(gpu)
    got = fused.atb_typed(z, d_pre, coef, ix)
    ce = ix.edge_select(coef)[2].double()
    want = th.cat([z.double().t() @ d_pre.double(), z.double().t() @ (d_pre.double() * ce[:, None])], 1)
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= 2e-5 * scale
    assert th.equal(got, fused.atb_typed(z, d_pre, coef, ix))
    d_g, _ = fused.bwd_h1_mfma(d_pre, (th.randn(h, h, generator=gen) * 0.1).to(gpu), z.clamp_min(0), coef, ix, both_halves=False)
    assert d_g.shape == (rows, h)


@pytest.mark.parametrize("case", ["many_classes", "long_ranges"])
def test_typed_weight_gradient_class_structure(case, gpu):
    """The weight-gradient kernel's two rare paths: (many_classes) thousands of coefficient classes of one tile
    each, so every workgroup range ends several classes (emit + accumulator / pipeline restart per tile); and
    (long_ranges) more than 64 tiles per workgroup, so the class bit mask is re-read inside a range."""
    from dualmessagepassing_amd import fused
    h = 128
    gen = th.Generator().manual_seed(21)
    rng = np.random.default_rng(21)
    if case == "many_classes":
        n, deg = 4000, 20
        dst = np.repeat(np.arange(n), deg).astype(np.int64)
        src = rng.integers(0, n, n * deg).astype(np.int64)
        perm = rng.permutation(n * deg)
        src, dst = src[perm], dst[perm]
        ix = _index(src, dst, n, rng.random(n * deg) < 0.5, gpu)
        coef = (th.rand(n, generator=gen) + 0.5).to(gpu)            # a distinct value per node: 4000 classes (value-sorted tile list)
    else:
        n, rows = 5000, 1_100_000                                   # 34 375 tiles + classes over 512 workgroups: > 64 per range
        ix = _index(rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64), n,
                    rng.random(rows) < 0.5, gpu)
        coef = ix.degree_coef(ix.out_deg)
    rows = ix.num_edges
    se, ts, nt, bound = ix.class_tiles(coef)
    tiles = int(nt)
    classes = th.unique(ts[:tiles]).numel()
    assert ((classes >= 3900 and tiles >= 3900) if case == "many_classes" else tiles > 64 * 512), (classes, tiles)
    z = th.randn(rows, h, generator=gen).to(gpu)
    d_pre = th.randn(rows, h, generator=gen).to(gpu)
    got = fused.atb_typed(z, d_pre, coef, ix)
    ce = coef[ix.dst32.long()].double()
    want = th.cat([z.double().t() @ d_pre.double(), z.double().t() @ (d_pre.double() * ce[:, None])], 1)
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= 2e-5 * scale
    assert th.equal(got, fused.atb_typed(z, d_pre, coef, ix))


def test_pingpong_driver_matches_default(gpu):
    """The experimental ping-pong driver of the fused MFMA kernels (dmp_dev_set_mfma_variant(1)) computes
    bit-identical results to the default independent-workgroup driver."""
    from dualmessagepassing_amd import _lib, fused
    lib = _lib.load()
    rows, h = 40003, 128
    gen = th.Generator().manual_seed(3)
    rng = np.random.default_rng(3)
    n = rows // 6
    ix = _index(rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64), n,
                rng.random(rows) < 0.5, gpu)
    coef = ix.degree_coef(ix.out_deg)
    z = th.randn(rows, h, generator=gen).to(gpu)
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    w2 = (th.randn(h, h, generator=gen) * 0.1).to(gpu)
    xp = th.randn(n, 3 * h, generator=gen).to(gpu)
    bias = th.randn(h, generator=gen).to(gpu)
    gate = th.rand(rows, generator=gen).to(gpu)

    def run():
        return (fused.edge_fwd_mfma(z, wes, xp[:, h:], 3 * h, bias, coef, ix), fused.out_fwd_mfma(z, w2, bias, gate, z),
                fused.bwd_h1_mfma(z, w2, z.clamp_min(0), coef, ix)[0],
                fused.bwd_z_mfma(th.cat([z, z], 1), wes, xp[:, :2 * h].contiguous(), z, coef, ix))
    # (both on the f32-input MFMA: the default driver's one-panel kernels run on the bf16 pipe otherwise, round 4)
    lib.dmp_dev_set_exact_fp32(1)
    try:
        base = run()
        try:
            lib.dmp_dev_set_mfma_variant(1)
            other = run()
        finally:
            lib.dmp_dev_set_mfma_variant(0)
    finally:
        lib.dmp_dev_set_exact_fp32(0)
    for a, b in zip(base, other):
        assert th.equal(a, b)


@pytest.mark.parametrize("h", [128, 64])
@pytest.mark.parametrize("num_layers", [1, 3, 4])
def test_fold_layers_matches_torch_algebra(num_layers, h, gpu):
    """fused.fold_layers (dmp_fold_layers / dmp_unfold_layers: one launch for all layers) against the same
    algebra in differentiable torch ops: the folded weights and, through a random cotangent, every parameter
    gradient (4 layers: more than DMP_FOLD_MAX_LAYERS, split over launches)."""
    from dualmessagepassing_amd import fused
    from dualmessagepassing_amd.dmpnn import DMPLayer
    th.manual_seed(num_layers)
    layers = [DMPLayer(h, h, num_mlp_layers=2, batch_norm=False).to(gpu) for _ in range(num_layers)]   # Linear-ReLU-Linear MLPs: the fused path
    for l in layers:
        for p in l.parameters():
            p.data.normal_(0.0, 0.3)
    got = fused.fold_layers(layers)
    assert fused._FoldLayers is not None and all(t.is_cuda for f in got for t in f)
    want = []
    for l in layers:
        nloop, in_w, out_w, nbias, eloop, src_w, dst_w, ebias, nW0, nb0, eW0, eb0 = fused._layer_params(l)
        Cn = th.cat([nloop, in_w, out_w, nbias.unsqueeze(0)], 0) @ nW0.t()
        Ce = th.cat([eloop, src_w - dst_w, dst_w, src_w, ebias.unsqueeze(0)], 0) @ eW0.t()
        wes = th.cat([Ce[:h], Ce[h:2 * h]], 1)
        want.append((Cn[h:3 * h], Cn[3 * h] + nb0, th.cat([Cn[:h], Ce[2 * h:3 * h], Ce[3 * h:4 * h]], 1),
                     wes, Ce[4 * h] + eb0))
        # the transposed copies (no gradient): [A'^T | B'^T] and the second Linears' transposes
        f = got[len(want) - 1]
        assert th.allclose(f[5], th.cat([wes[:, :h].t(), wes[:, h:].t()], 1), rtol=1e-5, atol=1e-5) and not f[5].requires_grad
        assert th.equal(f[6], l.nmlp[2].weight.t()) and th.equal(f[7], l.emlp[2].weight.t())
    gen = th.Generator().manual_seed(5)
    loss_g = loss_w = 0.0
    for fg, fw in zip(got, want):
        for a, b in zip(fg[:5], fw):
            assert a.shape == b.shape and th.allclose(a, b, rtol=1e-5, atol=1e-5), (a - b).abs().max()
            cot = th.randn(a.shape, generator=gen).to(gpu)
            loss_g = loss_g + (a * cot).sum()
            loss_w = loss_w + (b * cot).sum()
    params = [p for l in layers for p in fused._layer_params(l)]
    g_got = th.autograd.grad(loss_g, params)
    g_want = th.autograd.grad(loss_w, params)
    for i, (a, b) in enumerate(zip(g_got, g_want)):
        assert th.allclose(a, b, rtol=1e-5, atol=1e-4), (i, (a - b).abs().max().item())


@pytest.mark.parametrize("rows,k", [(1, 1), (37, 10), (70001, 10), (5000, 16)])
@pytest.mark.parametrize("gated", [False, True])
def test_smallk_gated_weight_gradient(rows, k, gated, gpu, h=128):
    """fused.smallk_atb: x^T (gate (.) d) for a narrow x (label encodings) in one pass, against fp64, bit-stable;
    d a row slice of a larger matrix (the union gradient's target rows)."""
    from dualmessagepassing_amd import fused
    gen = th.Generator().manual_seed(rows * 31 + k)
    x = (th.rand(rows, k, generator=gen) < 0.5).float().to(gpu)
    full = th.randn(rows + 5, h, generator=gen).to(gpu)
    d = full[5:]
    gate = (th.rand(rows, generator=gen) > 0.4).float().to(gpu) if gated else None
    got = fused.smallk_atb(x, d, gate)
    gd = d.double() * (gate.double()[:, None] if gated else 1.0)
    want = (x.double().t() @ gd).float()
    assert got.shape == (k, h) and th.allclose(got, want, rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    assert th.equal(got, fused.smallk_atb(x, d, gate))


@pytest.mark.parametrize("rows,k,gated", [(1, 1, True), (37, 10, False), (70001, 10, True), (5000, 16, True)])
def test_smallk_gated_weight_gradient_h64(rows, k, gated, gpu):
    """The same at the reference's shipped hidden_dim 64 (one value per lane)."""
    test_smallk_gated_weight_gradient(rows, k, gated, gpu, h=64)


@pytest.mark.parametrize("h", [128, 64])
@pytest.mark.parametrize("rows,k", [(1, 3), (4099, 10), (70001, 16)])
def test_gate_concat_from_label_encodings(rows, k, h, gpu):
    """dmpnn._gate_concat with a plain label embedding as the gated half: the union rows come from the K inputs
    per row (dmp_smallk_embed_gate) and the embedding's weight gradient from one gated pass over the upstream
    gradient (dmp_smallk_atb) -- against the unfused autograd path cat([p, gate * (enc @ W)])."""
    from dualmessagepassing_amd import dmpnn
    from dualmessagepassing_amd.embed import Embedding
    n_p = 37
    gen = th.Generator().manual_seed(rows + k)
    emb = Embedding(k, h).to(gpu)
    enc = (th.rand(rows, k, generator=gen) < 0.4).float().to(gpu)
    p = th.randn(n_p, h, generator=gen).to(gpu).requires_grad_(True)
    gate = (th.rand(rows, 1, generator=gen) > 0.3).float().to(gpu)
    cot = th.randn(n_p + rows, h, generator=gen).to(gpu)
    g = emb(enc)
    assert getattr(g, "_dmp_src", None) is not None
    out = dmpnn._gate_concat(p, g, gate)
    ref = th.cat([p, gate * (enc @ emb.weight)], 0)
    assert th.allclose(out, ref, rtol=1e-5, atol=1e-5)
    gp, gw = th.autograd.grad((out * cot).sum(), [p, emb.weight])
    rp, rw = th.autograd.grad((ref * cot).sum(), [p, emb.weight])
    assert th.equal(gp, rp) and th.allclose(gw, rw, rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    # an embedding that was modified afterwards (e.g. + id embedding) takes the generic path
    g2 = emb(enc) + 1.0
    assert getattr(g2, "_dmp_src", None) is None
    assert th.allclose(dmpnn._gate_concat(p, g2, gate), th.cat([p, gate * g2], 0), rtol=1e-6, atol=1e-6)


def test_weight_gradients_sharing_one_launch(gpu):
    """fused.atb_rows_multi: three products over the same rows (gated [R,128]^T [R,128] with column sums, [R,256]^T
    [R,128], [R,128]^T [R,384]: six output blocks) from one launch -- the bits of the one-product launches differ
    only through the tile ranges, so compare against fp64."""
    from dualmessagepassing_amd import fused
    h, rows = 128, 70001
    gen = th.Generator().manual_seed(77)
    dx = th.randn(rows, h, generator=gen).to(gpu)
    h1 = th.randn(rows, h, generator=gen).to(gpu)
    S = th.randn(rows, 2 * h, generator=gen).to(gpu)
    dXP = th.randn(rows, 3 * h, generator=gen).to(gpu)
    x = th.randn(rows, h, generator=gen).to(gpu)
    gate = (th.rand(rows, generator=gen) > 0.3).float().to(gpu)
    (a, cs), (b, n1), (c, n2) = fused.atb_rows_multi([(dx, h1, gate, True), (S, dXP[:, :h], None, False), (x, dXP, None, False)])
    assert n1 is None and n2 is None
    tol = 1e-4 * rows ** 0.5
    gd = dx.double() * gate.double()[:, None]
    assert th.allclose(a, (gd.t() @ h1.double()).float(), rtol=1e-5, atol=tol) and th.allclose(cs, gd.sum(0).float(), rtol=1e-5, atol=tol)
    assert b.shape == (2 * h, h) and th.allclose(b, (S.double().t() @ dXP[:, :h].double()).float(), rtol=1e-5, atol=tol)
    assert c.shape == (h, 3 * h) and th.allclose(c, (x.double().t() @ dXP.double()).float(), rtol=1e-5, atol=tol)
    again = fused.atb_rows_multi([(dx, h1, gate, True), (S, dXP[:, :h], None, False), (x, dXP, None, False)])
    assert th.equal(a, again[0][0]) and th.equal(b, again[1][0]) and th.equal(c, again[2][0])


@pytest.mark.parametrize("slope", [1 / 5.5, 0.01, 1.0])
@pytest.mark.parametrize("rows,h", [(70001, 128), (513, 64), (37, 20)])
def test_activation_slope_in_every_kernel(rows, h, slope, gpu):
    """LeakyReLU(slope) -- the reference's default rep / pred activation (utils/act.py:27,466) -- through every kernel
    that applies or differentiates the MLP activation: equal to torch's leaky_relu / leaky_relu_backward formulas
    (bit for bit in the streaming kernels, fp32 re-association in the MFMA ones)."""
    from dualmessagepassing_amd import fused
    F = th.nn.functional
    gen = th.Generator().manual_seed(rows + h)
    rng = np.random.default_rng(rows + h)
    n = max(2, rows // 5)
    src = rng.integers(0, n, rows).astype(np.int64)
    dst = rng.integers(0, n, rows).astype(np.int64)
    rev = rng.random(rows) < 0.5
    ix = _index(src, dst, n, rev, gpu)
    coef = ix.degree_coef(ix.out_deg)
    a = th.randn(rows, h, generator=gen).to(gpu)
    b = th.randn(rows, h, generator=gen).to(gpu)
    bias = th.randn(h, generator=gen).to(gpu)
    # forward row pass: act(a + b + bias)
    a2 = a.clone()
    fused.add_bias_relu_(a2, b, bias, slope)
    assert th.equal(a2, F.leaky_relu((a + b) + bias, slope))
    # backward row passes on the saved OUTPUT y = act(pre)
    y = F.leaky_relu(b, slope)
    want = th.ops.aten.leaky_relu_backward(a, b, slope, False)
    got, cs = fused.relu_bwd_colsum_(a.clone(), y, slope=slope)
    assert th.equal(got, want)
    assert th.allclose(cs, want.double().sum(0).float(), rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    dg, cs = fused.relu_bwd_g_colsum(a, y, coef, ix.dst32, slope)
    td = _t(dst).to(gpu)
    assert th.equal(dg, th.cat([want, want * coef[td][:, None]], 1))
    # edge_combine with the activation
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    xp = th.randn(n, 3 * h, generator=gen).to(gpu)
    g = a @ wes
    pre = fused.edge_combine_raw(g, 2 * h, xp[:, h:], 3 * h, bias, coef, ix, h, relu=False)
    out = fused.edge_combine_raw(g, 2 * h, xp[:, h:], 3 * h, bias, coef, ix, h, relu=True, slope=slope)
    assert th.equal(out, F.leaky_relu(pre, slope))
    if h != 128:
        return
    # the MFMA kernels (H = 128)
    w2 = (th.randn(h, h, generator=gen) * 0.1).to(gpu)
    ref = F.leaky_relu(pre.double(), slope)
    for fn in (fused.edge_fwd_mfma, fused.edge_fwd_typed):
        got = fn(a, wes, xp[:, h:], 3 * h, bias, coef, ix, slope)
        assert th.allclose(got.double(), ref, rtol=1e-5, atol=2e-4), fn.__name__
    d_h = a.double() @ w2.double()
    dpre = th.where(y > 0, d_h, slope * d_h)
    d_g, cs = fused.bwd_h1_mfma(a, w2, y, coef, ix, slope=slope)
    assert th.allclose(d_g.double(), th.cat([dpre, dpre * coef.double()[td][:, None]], 1), rtol=1e-5, atol=2e-4)
    assert th.allclose(cs.double(), dpre.sum(0), rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    gate = (th.rand(rows, generator=gen) * (th.rand(rows, generator=gen) > 0.3)).to(gpu)
    d_p, _ = fused.bwd_h1_mfma(a, w2, y, both_halves=False, gate=gate, slope=slope)
    assert th.allclose(d_p.double(), dpre * gate.double()[:, None], rtol=1e-5, atol=2e-4)


def test_activation_slope_outside_the_supported_range_is_refused(gpu):
    from dualmessagepassing_amd import _lib, fused
    a = th.randn(8, 16, device=gpu)
    for bad in (-0.1, 1.5):
        with pytest.raises(_lib.DmpError):
            fused.add_bias_relu_(a.clone(), a, None, bad)


@pytest.mark.parametrize("sizes_a,sizes_b", [([5, 0, 130, 64], [1, 300]), ([7] * 50, None), ([0, 0, 3], [0]), ([64] * 1024, [512] * 1024),
                                             ([3, 0, 70] * 1000, [2] * 1500)])      # (4500 graphs: three slices of the offsets scan)
def test_pool_index_device_build_equals_tensor_build(sizes_a, sizes_b, gpu):
    """ops.PoolIndex built by dmp_pool_index (two launches, sizes / flags as two pieces) == the tensor-op construction:
    every array bit for bit, and the pooled sums it produces."""
    from dualmessagepassing_amd import ops
    gen = th.Generator().manual_seed(3)
    a = th.tensor(sizes_a, dtype=th.int64, device=gpu)
    b = None if sizes_b is None else th.tensor(sizes_b, dtype=th.int64, device=gpu)
    ra, rb = int(sum(sizes_a)), int(sum(sizes_b or []))
    for with_flag in (False, True):
        fa = (th.rand(ra, generator=gen) < 0.5).to(gpu) if with_flag else None
        fb = (th.rand(rb, generator=gen) < 0.5).to(gpu) if (with_flag and b is not None) else None
        sizes = a if b is None else th.cat([a, b])
        flag = None if not with_flag else (fa if b is None else th.cat([fa, fb]))
        ref = ops.PoolIndex(sizes, flag)                                       # tensor ops (host-sized)
        got = ops.PoolIndex(a if b is None else (a, b), (fa if b is None else (fa, fb)) if with_flag else None, num_rows=ra + rb)
        assert (got.num_graphs, got.num_rows) == (ref.num_graphs, ref.num_rows) and got.num_chunks >= ref.num_chunks
        V = ref.num_chunks
        assert th.equal(got.vent, ref.vent) and th.equal(got.gptr, ref.gptr) and th.equal(got.seg32, ref.seg32)
        assert th.equal(got.vptr[:V + 1], ref.vptr) and bool((got.vptr[V:] == ra + rb).all())
        assert th.equal(got.gent[:V], ref.gent) and th.equal(got.sizes, ref.sizes)
        assert (got.flag8 is None) == (ref.flag8 is None) and (got.flag8 is None or th.equal(got.flag8, ref.flag8))
        x = th.randn(ra + rb, 16, generator=gen).to(gpu)
        assert th.equal(ops.seg_pool(x, got), ops.seg_pool(x, ref))


@pytest.mark.parametrize("h", [32, 64, 128, 256])
def test_tiled_incidence_segment_sum_equals_plain_kernel(h, gpu):
    """dmp_seg_sum2_tiled (a workgroup stages a graph tile's edge rows in LDS once and feeds both endpoint sums from
    there) gives the BITS of dmp_seg_sum2 over the incidence CSR (same CSR order of the adds): a union of small graphs
    (several per tile) and 512-edge graphs (one per tile), ragged sizes, graphs without edges, into a column slice."""
    from dualmessagepassing_amd import ops
    from dualmessagepassing_amd.collate import collate_device, union_graphs
    rng = np.random.default_rng(h)
    switch, ops.USE_TILED_SEG_SUM = ops.USE_TILED_SEG_SUM, True     # the experiment kernel (off by default: slower)
    try:
        _tiled_case(h, gpu, rng, ops, collate_device, union_graphs)
    finally:
        ops.USE_TILED_SEG_SUM = switch


def _tiled_case(h, gpu, rng, ops, collate_device, union_graphs):
    def batch(sizes):
        ls, ld, rv = [], [], []
        for n, m in sizes:
            u, v = (er_batch(1, n, m, rng)[:2] if m else (np.zeros(0, np.int64), np.zeros(0, np.int64)))
            ls.append(u); ld.append(v); rv.append(np.concatenate([np.zeros(m, bool), np.ones(m, bool)]) if m else np.zeros(0, bool))
        nn_ = np.array([s[0] for s in sizes], np.int64)
        ne_ = np.array([2 * s[1] for s in sizes], np.int64)
        g = collate_device(_t(np.concatenate(ls)).to(gpu), _t(np.concatenate(ld)).to(gpu), _t(nn_).to(gpu), _t(ne_).to(gpu),
                           int(nn_.sum()), int(ne_.sum()), edata={"is_reversed": _t(np.concatenate(rv)).to(gpu)},
                           max_nodes=int(nn_.max()), max_edges=int(ne_.max()))
        return g
    p = batch([(8, 12)] * 37 + [(3, 0), (5, 7)])
    g = batch([(64, 256)] * 9 + [(40, 100), (64, 256), (2, 1)])
    u = union_graphs(p, g)
    assert u.tiling is not None and u.tiling[4] > 1 and u.tiling[5] == 1        # 21 pattern graphs per tile, 1 target graph
    ix = u.index()
    inc_ptr, inc_ent = ix.incidence()
    N, E = u.number_of_nodes(), u.number_of_edges()
    m = th.randn(E, h, device=gpu)
    want = ops.seg_sum_raw(m, inc_ptr, inc_ent, N, None, True, 1.0, -1.0, rows_shared=True)
    got = ops.seg_sum_raw(m, inc_ptr, inc_ent, N, None, True, 1.0, -1.0, rows_shared=True, tiling=ix.tiling)
    assert th.equal(got, want)
    wide = th.full((N, 3 * h), 5.0, device=gpu)
    ops.seg_sum_raw(m, inc_ptr, inc_ent, N, None, True, 1.0, -1.0, rows_shared=True, out=wide[:, h:], tiling=ix.tiling)
    assert th.equal(wide[:, h:], want) and bool((wide[:, :h] == 5.0).all())
    # a single batch (no union) and the in-CSR work too; a graph above 512 edge rows switches the tiling off
    assert th.equal(ops.seg_sum_raw(m[:g.number_of_edges()], *g.index().incidence(), g.number_of_nodes(), None, True, 1.0, -1.0,
                                    tiling=g.index().tiling),
                    ops.seg_sum_raw(m[:g.number_of_edges()], *g.index().incidence(), g.number_of_nodes(), None, True, 1.0, -1.0))
    assert batch([(64, 300)]).tiling is None


@pytest.mark.parametrize("rows", [1, 31, 129, 5000, 70001])
@pytest.mark.parametrize("slope", [0.0, 1.0 / 5.5])
def test_mfma_kernels_h64(rows, slope, gpu):
    """The MFMA kernels at the reference's shipped hidden_dim 64 (config.py:298-301; two waves per workgroup): the plain
    product, out_fwd (Linear + gate + residual), bwd_h1 (dPre alone: gate, output slice), the class-typed edge_fwd /
    bwd_z / weight gradient and the row weight-gradient kernels (64 x 64 output blocks), each against its fp64
    formula, bit-stable from launch to launch."""
    from dualmessagepassing_amd import _lib, fused
    from dualmessagepassing_amd._lib import ptr, stream_ptr
    lib = _lib.load()
    h = 64
    gen = th.Generator().manual_seed(rows + 3)
    rng = np.random.default_rng(rows + 3)
    a = th.randn(rows, h, generator=gen).to(gpu)
    w = th.randn(h, h, generator=gen).to(gpu)
    c = th.empty(rows, h, device=gpu)
    _lib.check(lib.dmp_gemm_k64(ptr(a), h, ptr(w), h, 0, ptr(c), h, rows, stream_ptr()), "gemm64")
    assert th.allclose(c.double(), a.double() @ w.double(), rtol=1e-5, atol=2e-4)
    _lib.check(lib.dmp_gemm_k64(ptr(a), h, ptr(w.t().contiguous()), h, 1, ptr(c), h, rows, stream_ptr()), "gemm64_t")
    assert th.allclose(c.double(), a.double() @ w.double(), rtol=1e-5, atol=2e-4)
    n = max(2, rows // 5)
    src, dst = rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64)
    rev = rng.random(rows) < 0.5
    ix = _index(src, dst, n, rev, gpu)
    assert fused.typed_ok(ix, h) and fused.onepanel_ok(h)
    coef = ix.degree_coef(ix.out_deg)
    ts, td, tr = _t(src).to(gpu), _t(dst).to(gpu), _t(rev).to(gpu)
    ce = coef.double()[td][:, None]
    act = lambda x: th.where(x > 0, x, slope * x)
    # ---- out_fwd
    w2 = (th.randn(h, h, generator=gen) * 0.1).to(gpu)
    bias = th.randn(h, generator=gen).to(gpu)
    gate = (th.rand(rows, generator=gen) * (th.rand(rows, generator=gen) > 0.3)).to(gpu)
    prev = th.randn(rows, h, generator=gen).to(gpu)
    for g_, p_ in ((gate, prev), (None, prev), (gate, None), (None, None)):
        got = fused.out_fwd_mfma(a, w2, bias, g_, p_)
        ref = a.double() @ w2.double().t() + bias.double()
        if g_ is not None:
            ref = ref * g_.double()[:, None]
        if p_ is not None:
            ref = ref + p_.double()
        assert th.allclose(got.double(), ref, rtol=1e-5, atol=2e-4)
        assert th.equal(got, fused.out_fwd_mfma(a, w2, bias, g_, p_))
    # ---- bwd_h1, dPre alone
    d_o = th.randn(rows, h, generator=gen).to(gpu)
    h1 = act(th.randn(rows, h, generator=gen)).to(gpu)
    lin = d_o.double() @ w2.double()
    for g_ in (gate, None):
        d_p, cs = fused.bwd_h1_mfma(d_o, w2, h1, both_halves=False, gate=g_, slope=slope)
        x = lin * (g_.double()[:, None] if g_ is not None else 1.0)
        ref = th.where(h1 > 0, x, slope * x)
        assert d_p.shape == (rows, h) and th.allclose(d_p.double(), ref, rtol=1e-5, atol=2e-4)
        assert th.allclose(cs.double(), ref.sum(0), rtol=1e-5, atol=1e-4 * max(1.0, rows ** 0.5))
    wide = th.full((rows, 3 * h), 7.0, device=gpu)
    d_p2, cs2 = fused.bwd_h1_mfma(d_o, w2, h1, both_halves=False, gate=gate, out=wide[:, :h], slope=slope)
    d_p1, cs1 = fused.bwd_h1_mfma(d_o, w2, h1, both_halves=False, gate=gate, slope=slope)
    assert th.equal(wide[:, :h], d_p1) and th.equal(cs2, cs1) and bool((wide[:, h:] == 7.0).all())
    # ---- class-typed edge forward / input gradient / weight gradient
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    xp = th.randn(n, 3 * h, generator=gen).to(gpu)
    got = fused.edge_fwd_typed(a, wes, xp[:, h:], 3 * h, bias, coef, ix, slope)
    g = a.double() @ wes.double()
    ai, bi = th.where(tr, ts, td), th.where(tr, td, ts)
    ref = act(g[:, :h] + ce * g[:, h:] + xp.double()[ai][:, h:2 * h] - xp.double()[bi][:, 2 * h:] + bias.double())
    assert th.allclose(got.double(), ref, rtol=1e-5, atol=2e-4)
    assert th.equal(got, fused.edge_fwd_typed(a, wes, xp[:, h:], 3 * h, bias, coef, ix, slope))
    two = fused.edge_combine_raw(a @ wes, 2 * h, xp[:, h:], 3 * h, bias, coef, ix, h, relu=True, slope=slope)
    assert th.allclose(got, two, rtol=1e-5, atol=2e-4)
    d_pre = th.randn(rows, h, generator=gen).to(gpu)
    d_s = th.randn(n, 2 * h, generator=gen).to(gpu)
    gs = th.where(tr[:, None], d_s.double()[td][:, h:], -d_s.double()[td][:, :h])
    for b_ in (prev, None):
        got = fused.bwd_z_typed(d_pre, h, wes, d_s, b_, coef, ix)
        ref = gs + d_pre.double() @ wes.double()[:, :h].t() + (ce * d_pre.double()) @ wes.double()[:, h:].t()
        if b_ is not None:
            ref = ref + b_.double()
        assert th.allclose(got.double(), ref, rtol=1e-5, atol=3e-4)
        assert th.equal(got, fused.bwd_z_typed(d_pre, h, wes, d_s, b_, coef, ix))
    got = fused.atb_typed(a, d_pre, coef, ix)
    want = th.cat([a.double().t() @ d_pre.double(), a.double().t() @ (d_pre.double() * ce)], 1)
    assert got.shape == (h, 2 * h) and float((got.double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    assert th.equal(got, fused.atb_typed(a, d_pre, coef, ix))
    # ---- row weight gradients, 64 x 64 output blocks
    tol = 1e-4 * max(1.0, rows ** 0.5)
    for g_ in (gate, None):
        got, cs = fused.atb_rows(a, d_pre, g_)
        ga = a.double() * (g_.double()[:, None] if g_ is not None else 1.0)
        assert got.shape == (h, h) and th.allclose(got, (ga.t() @ d_pre.double()).float(), rtol=1e-5, atol=tol)
        assert th.allclose(cs, ga.sum(0).float(), rtol=1e-5, atol=tol)
        assert th.equal(got, fused.atb_rows(a, d_pre, g_)[0])
    a3 = th.randn(rows, 2 * h + 4, generator=gen).to(gpu)[:, :2 * h]
    b3 = th.randn(rows, 3 * h, generator=gen).to(gpu)
    got3, cs3 = fused.atb_rows(a3, b3, gate)
    g3 = a3.double() * gate.double()[:, None]
    assert got3.shape == (2 * h, 3 * h) and th.allclose(got3, (g3.t() @ b3.double()).float(), rtol=1e-5, atol=tol)
    assert th.allclose(cs3, g3.sum(0).float(), rtol=1e-5, atol=tol)
    multi = fused.atb_rows_multi([(a, d_pre, gate, True), (a3, d_pre, None, False), (a, b3, None, False)])
    one = fused.atb_rows(a, d_pre, gate)               # other workgroup ranges than the shared launch: equal up to summation order
    assert th.allclose(multi[0][0], one[0], rtol=1e-5, atol=tol) and th.allclose(multi[0][1], one[1], rtol=1e-5, atol=tol)
    assert th.allclose(multi[1][0], (a3.double().t() @ d_pre.double()).float(), rtol=1e-5, atol=tol) and multi[1][1] is None
    assert th.allclose(multi[2][0], (a.double().t() @ b3.double()).float(), rtol=1e-5, atol=tol)


def test_validate_switch_raises_on_out_of_range_entries(gpu, monkeypatch):
    """``_lib.VALIDATE``: the index builds read their status word back and raise on an edge endpoint /
    lookup index outside its range; without it the entry is flagged only (one host sync saved per build)."""
    from dualmessagepassing_amd import _lib, ops
    from dualmessagepassing_amd.graph import GraphIndex
    src = th.tensor([0, 1, 2], device=gpu)
    dst = th.tensor([1, 5, 0], device=gpu)          # 5 >= num_nodes
    x = th.randn(3, 8, device=gpu)
    monkeypatch.setattr(_lib, "VALIDATE", False)
    GraphIndex(src, dst, 3)                          # flagged, not raised
    with pytest.raises(_lib.DmpError):
        GraphIndex(src, dst, 3, validate=True)
    monkeypatch.setattr(_lib, "VALIDATE", True)
    with pytest.raises(_lib.DmpError):
        GraphIndex(src, dst, 3)
    with pytest.raises(_lib.DmpError):
        ops.take_rows(x, th.tensor([0, 3], device=gpu))
    assert th.equal(ops.take_rows(x, th.tensor([2, 0], device=gpu)), x[[2, 0]])


@pytest.mark.parametrize("h", [128, 64, 20])
@pytest.mark.parametrize("flagged,gated", [(True, True), (False, False), (True, False)])
def test_pooled_activation_backward_equals_its_two_passes(h, flagged, gated, gpu):
    """dmp_pool_relu_bwd (the last layer's activation backward from a per-graph gradient table AND the gated per-graph sums
    of the saved activation, one pass) against dmp_relu_bwd_gathered_colsum + the pooled segment sums, and fp64."""
    from dualmessagepassing_amd import fused, ops
    gen = th.Generator().manual_seed(h + 3)
    sizes = th.tensor([5, 0, 64, 65, 130, 1, 700, 33], dtype=th.int64)
    R, G = int(sizes.sum()), sizes.numel()
    flag = (th.rand(R, generator=gen) < 0.5) if flagged else None
    pool = ops.PoolIndex(sizes.to(gpu), None if flag is None else flag.to(gpu), num_rows=R)
    act = th.randn(R, h, generator=gen).to(gpu)
    table = th.randn(G, h, generator=gen).to(gpu)
    gate = (th.rand(R, generator=gen) < 0.7).float().to(gpu) if gated else None
    rowmap = fused.pool_rowmap(pool)
    slope = 1 / 5.5
    d_ref, cs_ref = fused.relu_bwd_gathered_colsum(table, rowmap, gate, act, slope)
    q_ref = fused.pool_rows(act, pool, gate)
    d, cs, q = fused.pool_relu_bwd(table, rowmap, gate, act, pool, slope)
    assert th.equal(d, d_ref)
    assert th.allclose(cs, cs_ref, rtol=1e-5, atol=1e-4)
    w = q_ref.size(1)                                                     # [G, 2H] with a flag, [G, H] without
    assert q.size(1) == 2 * h and th.allclose(q[:, :w], q_ref, rtol=1e-5, atol=1e-4)
    if not flagged:
        assert float(q[:, h:].abs().max()) == 0.0
    # fp64
    seg = th.repeat_interleave(th.arange(G), sizes).to(gpu)
    g64 = th.ones(R, device=gpu, dtype=th.float64) if gate is None else gate.double()
    live = th.ones(R, device=gpu, dtype=th.bool) if flag is None else ~flag.to(gpu)
    u = th.where(live.view(-1, 1), g64.view(-1, 1) * table.double()[seg], th.zeros(1, device=gpu, dtype=th.float64))
    want = th.where(act.double() > 0, u, slope * u)
    assert float((d.double() - want).abs().max()) <= 1e-6
    q64 = th.zeros(G, h, device=gpu, dtype=th.float64).index_add_(0, seg[live], (g64.view(-1, 1) * act.double())[live])
    assert float((q[:, :h].double() - q64).abs().max()) <= 1e-4


def _graph_batch(sizes, rng, gpu, random_flags=False):
    from dualmessagepassing_amd.collate import collate_device
    ls, ld, rv = [], [], []
    for n, m in sizes:
        u, v = (er_batch(1, n, m, rng)[:2] if m else (np.zeros(0, np.int64), np.zeros(0, np.int64)))
        ls.append(u); ld.append(v)
        rv.append(rng.random(2 * m) < 0.5 if random_flags else np.concatenate([np.zeros(m, bool), np.ones(m, bool)]))
    nn_ = np.array([s[0] for s in sizes], np.int64)
    ne_ = np.array([2 * s[1] for s in sizes], np.int64)
    return collate_device(_t(np.concatenate(ls)).to(gpu), _t(np.concatenate(ld)).to(gpu), _t(nn_).to(gpu), _t(ne_).to(gpu),
                          int(nn_.sum()), int(ne_.sum()), edata={"is_reversed": _t(np.concatenate(rv).astype(bool)).to(gpu)},
                          max_nodes=int(nn_.max()), max_edges=int(ne_.max()))


@pytest.mark.parametrize("h", [64, 128])
@pytest.mark.parametrize("random_flags", [False, True])
def test_one_pass_endpoint_sums_equal_the_incidence_segment_sum(h, random_flags, gpu):
    """dmp_seg_sum2_graphs (csrc/dmp_segacc.hip: the edge rows of a graph tile streamed ONCE, both endpoints' sums kept in
    registers, added in ascending eid) gives the BITS of dmp_seg_sum2 over the incidence CSR (rows merged by eid): a union
    of small graphs (8 per tile) and 64-node graphs (one per tile), ragged sizes, graphs without edges or nodes' worth of
    edges that are not a multiple of the 4-row blocks / super-groups, reversed flags as add_reversed_edges sets
    them and at random, into a fresh tensor and into a column slice; and agrees with an fp64 scatter."""
    from dualmessagepassing_amd import ops
    from dualmessagepassing_amd.collate import union_graphs
    rng = np.random.default_rng(1000 * h + int(random_flags))
    p = _graph_batch([(8, 12)] * 37 + [(3, 0), (5, 7), (8, 28)], rng, gpu, random_flags)
    g = _graph_batch([(64, 256)] * 9 + [(40, 100), (64, 256), (2, 1), (64, 611), (33, 33)], rng, gpu, random_flags)
    u = union_graphs(p, g)
    assert u.node_tiling is not None and u.node_tiling[4] == 8 and u.node_tiling[5] == 1
    ix = u.index()
    N, E = u.number_of_nodes(), u.number_of_edges()
    m = th.randn(E, h, device=gpu)
    inc_ptr, inc_ent = ix.incidence()
    want = ops.seg_sum_raw(m, inc_ptr, inc_ent, N, None, True, 1.0, -1.0, rows_shared=2)
    saved, ops.USE_GRAPH_SEG_SUM = ops.USE_GRAPH_SEG_SUM, True
    try:
        assert ops.graph_seg_ok(ix, m, h)
        got = ops.endpoint_sums(m, ix)
        assert th.equal(got, want)
        wide = th.full((N, 3 * h), 5.0, device=gpu)
        ops.endpoint_sums(m, ix, out=wide[:, h:])
        assert th.equal(wide[:, h:], want) and bool((wide[:, :h] == 5.0).all())
        # a strided operand (a column slice of a wider matrix)
        m2 = th.randn(E, 2 * h, device=gpu)
        assert th.equal(ops.endpoint_sums(m2[:, h:], ix), ops.seg_sum_raw(m2[:, h:], inc_ptr, inc_ent, N, None, True, 1.0, -1.0, rows_shared=2))
        # a single batch without a union
        gi = g.index()
        mg = m[:g.number_of_edges()].contiguous()
        assert th.equal(ops.endpoint_sums(mg, gi), ops.seg_sum_raw(mg, *gi.incidence(), g.number_of_nodes(), None, True, 1.0, -1.0, rows_shared=2))
    finally:
        ops.USE_GRAPH_SEG_SUM = saved
    # the definition, in fp64
    sel_a, sel_b = ix.endpoint_select()
    ref = th.zeros(N, 2 * h, dtype=th.float64, device=gpu)
    ref[:, :h].index_add_(0, sel_a.long(), m.double())
    ref[:, h:].index_add_(0, sel_b.long(), -m.double())
    assert float((got.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    # a graph with more nodes than a tile holds switches the one-pass kernel off (the caller takes the incidence CSR)
    assert _graph_batch([(65, 10)], rng, gpu).node_tiling is None


def test_one_pass_endpoint_sums_at_config_2_size(gpu):
    """bench.py's launch shape (union of 1024 pattern + 1024 target graphs, hid 128): bits of the incidence segment sum,
    twice the same bits (no arrival-order dependence), and the sum of ones = the degree counts."""
    from dualmessagepassing_amd import ops
    from dualmessagepassing_amd.collate import union_graphs
    rng = np.random.default_rng(11)
    p = _graph_batch([(8, 12)] * 1024, rng, gpu)
    g = _graph_batch([(64, 256)] * 1024, rng, gpu)
    u = union_graphs(p, g)
    ix = u.index()
    N, E, h = u.number_of_nodes(), u.number_of_edges(), 128
    m = th.randn(E, h, device=gpu)
    assert ops.graph_seg_ok(ix, m, h)
    got = ops.endpoint_sums(m, ix)
    want = ops.seg_sum_raw(m, *ix.incidence(), N, None, True, 1.0, -1.0, rows_shared=2)
    assert th.equal(got, want)
    assert th.equal(ops.endpoint_sums(m, ix), got)
    ones = ops.endpoint_sums(th.ones(E, h, device=gpu), ix)
    sel_a, sel_b = ix.endpoint_select()
    assert th.equal(ones[:, 0], th.bincount(sel_a.long(), minlength=N).float())
    assert th.equal(ones[:, h], -th.bincount(sel_b.long(), minlength=N).float())


def test_one_pass_endpoint_sums_beside_other_kernels(gpu):
    """dmp_seg_sum2_graphs keeps its sums in registers addressed through the VGPR index mode.  Inside one stream a kernel never
    shares a CU with another kernel -- two streams (and two processes on one GPU) do: a first form of the kernel (one v_add_f32
    with a relative source AND destination) gave the right sums and corrupted the waves of a GEMM running beside it in 12 % of
    the launches (scripts/stress_segacc.py).  Here: the kernel on one stream, a GEMM / a row copy / the incidence segment sum on
    another, every result against its stand-alone value."""
    from dualmessagepassing_amd import ops
    from dualmessagepassing_amd.collate import union_graphs
    rng = np.random.default_rng(12)
    u = union_graphs(_graph_batch([(8, 12)] * 512, rng, gpu), _graph_batch([(64, 256)] * 512, rng, gpu))
    ix = u.index()
    N, E, h = u.number_of_nodes(), u.number_of_edges(), 128
    m = th.randn(E, h, device=gpu)
    assert ops.graph_seg_ok(ix, m, h)
    inc = ix.incidence()
    want = ops.seg_sum_raw(m, inc[0], inc[1], N, None, True, 1.0, -1.0, rows_shared=2)
    a, b = th.randn(4096, 512, device=gpu), th.randn(512, 512, device=gpu)
    x = th.randn(E, h, device=gpu)
    want_mm, want_copy = a @ b, x.clone()
    sa, sb = th.cuda.Stream(), th.cuda.Stream()
    th.cuda.synchronize()
    bad = {"sums": 0, "gemm": 0, "copy": 0, "incidence": 0}
    for _ in range(200):
        with th.cuda.stream(sa):
            g1 = ops.endpoint_sums(m, ix)
            g2 = ops.endpoint_sums(m, ix)
        with th.cuda.stream(sb):
            mm = a @ b
            cp = x.clone()
            si = ops.seg_sum_raw(m, inc[0], inc[1], N, None, True, 1.0, -1.0, rows_shared=2)
            mm2 = a @ b
        th.cuda.synchronize()
        bad["sums"] += int(not th.equal(g1, want)) + int(not th.equal(g2, want))
        bad["gemm"] += int(not th.equal(mm, want_mm)) + int(not th.equal(mm2, want_mm))
        bad["copy"] += int(not th.equal(cp, want_copy))
        bad["incidence"] += int(not th.equal(si, want))
    assert not any(bad.values()), bad


def test_small_gemm_jobs_match_fp64(gpu):
    """dmp_small_gemm_jobs: several small products in one launch -- plain and transposed operands, up to three terms,
    an addend, operands and outputs that are column slices of wider tensors, ragged sizes (not multiples of the 16 x 64
    tile), a job with K = 0 terms' worth of nothing but the addend."""
    from dualmessagepassing_amd import fused
    gen = th.Generator().manual_seed(77)
    r = lambda *s: th.randn(*s, generator=gen).to(gpu)
    H = 128
    W0, Wes, Bn, Wx = r(20, H), r(H, 2 * H), r(2 * H, H), r(H, 3 * H)
    XX, Xn, Yn, WV0 = r(20, 3 * H), r(2, 20, H), r(4, 16, H), r(16, H)
    outs = {"M0": th.empty(20, 2 * H, device=gpu), "dWes": th.empty(H, 2 * H, device=gpu), "dBn": th.empty(2 * H, H, device=gpu),
            "dW0": th.empty(20, H, device=gpu), "dWx": th.full((H, 3 * H), 7.0, device=gpu), "dWV0": th.empty(16, H, device=gpu),
            "odd": th.empty(5, 70, device=gpu)}
    a5, b5 = r(5, 33), r(33, 70)
    jobs = [(outs["M0"], [(W0, False, Wes, False)], None),
            (outs["dWes"], [(W0, True, XX[:, :2 * H], False)], None),
            (outs["dBn"][:H], [(W0, True, Xn[0], False)], None), (outs["dBn"][H:], [(W0, True, Xn[1], False)], None),
            (outs["dW0"], [(XX[:, :2 * H], False, Wes, True), (Xn[0], False, Bn[:H], True), (Xn[1], False, Bn[H:], True)], XX[:, 2 * H:]),
            (outs["dWx"][:, H:2 * H], [(WV0, True, Yn[1], False)], None),
            (outs["dWV0"], [(Yn[b], False, Wx[:, b * H:(b + 1) * H], True) for b in range(3)], Yn[3]),
            (outs["odd"], [(a5, False, b5, False)], None)]
    fused.small_gemm_jobs(jobs)
    d = lambda t: t.double()
    want = {"M0": d(W0) @ d(Wes), "dWes": d(W0).t() @ d(XX[:, :2 * H]),
            "dBn": th.cat([d(W0).t() @ d(Xn[0]), d(W0).t() @ d(Xn[1])]),
            "dW0": d(XX[:, :2 * H]) @ d(Wes).t() + d(Xn[0]) @ d(Bn[:H]).t() + d(Xn[1]) @ d(Bn[H:]).t() + d(XX[:, 2 * H:]),
            "dWV0": sum(d(Yn[b]) @ d(Wx[:, b * H:(b + 1) * H]).t() for b in range(3)) + d(Yn[3]), "odd": d(a5) @ d(b5)}
    for k, w in want.items():
        err = float((outs[k].double() - w).abs().max())
        assert err <= 2e-5 * max(1.0, float(w.abs().max())), (k, err)
    w = d(WV0).t() @ d(Yn[1])
    assert float((outs["dWx"][:, H:2 * H].double() - w).abs().max()) <= 2e-5 * float(w.abs().max())
    assert bool((outs["dWx"][:, :H] == 7.0).all()) and bool((outs["dWx"][:, 2 * H:] == 7.0).all())   # the slice only
    with pytest.raises(Exception):
        fused.small_gemm_jobs([(outs["odd"], [(a5, False, b5, True)], None)])                       # shapes that do not chain


@pytest.mark.parametrize("rows", [40, 1000, 70001, 300000])
def test_tile_list_weight_gradients_on_the_piece_image(rows, gpu):
    """``dmp_atb2_jobs`` (csrc/dmp_h1w.hip::atb2_k: every element split into bf16 pieces once, fragments through
    ds_read_b64_tr_b16) against fp64 and against the kernels it replaces: ``atb_typed`` ([z^T d | z^T (c (.) d)] over the class
    tiles of the kept edges, and its plain form) and the node side's multi-product launch over a kept-row tile list (operands
    that are column slices); the same bits on every launch."""
    from dualmessagepassing_amd import fused
    h = 128
    gen = th.Generator().manual_seed(rows)
    rng = np.random.default_rng(rows)
    n = max(2, rows // 6)
    src, dst = rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64)
    ix = _index(src, dst, n, rng.random(rows) < 0.5, gpu)
    coef = ix.degree_coef(ix.out_deg)
    gate = th.from_numpy((rng.random(rows) < 0.46).astype(np.float32)).to(gpu)
    gate._dmp_binary = True
    keep = gate != 0
    z = th.randn(rows, h, generator=gen).to(gpu)
    d = th.randn(rows, h, generator=gen).to(gpu)
    zp, dp = z.clone(), d.clone()
    zp[~keep] = float("nan")
    dp[~keep] = float("nan")
    ce = ix.edge_select(coef)[2].double()
    z64, d64 = z[keep].double(), d[keep].double()
    want = th.cat([z64.t() @ d64, z64.t() @ (ce[keep][:, None] * d64)], dim=1)
    scale = float((z64.abs().t() @ (ce[keep][:, None].abs().clamp(min=1.0) * d64.abs())).max())
    res = {}
    for on in (True, False):
        saved = fused.USE_ATB2
        fused.USE_ATB2 = on
        try:
            res[on] = (fused.atb_typed(zp, dp, coef, ix, gate=gate), fused.atb_typed(zp, dp, coef, ix, gate=gate, plain=True))
            if on:
                again = fused.atb_typed(zp, dp, coef, ix, gate=gate)
        finally:
            fused.USE_ATB2 = saved
    assert th.equal(res[True][0], again)
    for on in (True, False):
        wide, plain = res[on]
        assert wide.shape == (h, 2 * h) and plain.shape == (h, h)
        assert float((wide.double() - want).abs().max()) <= 2e-6 * scale, (on, float((wide.double() - want).abs().max()) / scale)
        assert float((plain.double() - want[:, :h]).abs().max()) <= 2e-6 * scale
    # the node side's launch: three products, six 128 x 128 blocks, operands that are column slices, over a kept-row tile list
    tiles = fused.ascending_tiles(gate)
    S = th.randn(rows, 2 * h, generator=gen).to(gpu)
    dXP = th.randn(rows, 3 * h, generator=gen).to(gpu)
    for t in (S, dXP):
        t[~keep] = float("nan")
    got = {}
    for on in (True, False):
        saved = fused.USE_ATB2
        fused.USE_ATB2 = on
        try:
            got[on] = [r for r, _ in fused.atb_rows_multi([(dp, zp, None, False), (S, dp[:, :h], None, False), (zp, dXP, None, False)], tiles=tiles)]
        finally:
            fused.USE_ATB2 = saved
    refs = [d64.t() @ z64, S[keep].double().t() @ d64, z64.t() @ dXP[keep].double()]
    for on in (True, False):
        for r, w in zip(got[on], refs):
            assert r.shape == w.shape
            assert float((r.double() - w).abs().max()) <= 4e-6 * scale, (on, r.shape, float((r.double() - w).abs().max()) / scale)


@pytest.mark.parametrize("rows", [33, 5000, 70001])
def test_second_linear_backward_without_a_gate_over_identity_tiles(rows, gpu):
    """No gate (a model without a filter net: every row live): ``dmp_bwd_h1_w`` over the identity tile list against the two
    all-rows launches it replaces there (``bwd_h1_mfma`` + ``atb_rows``)."""
    from dualmessagepassing_amd import fused
    h = 128
    gen = th.Generator().manual_seed(rows)
    h1 = th.randn(rows, h, generator=gen).to(gpu)
    h1 = th.where(h1 > 0, h1, 0.18 * h1)
    d_o = th.randn(rows, h, generator=gen).to(gpu)
    W2 = (th.randn(h, h, generator=gen) / h ** 0.5).to(gpu)
    tiles = fused.identity_tiles(rows, gpu)
    assert tiles is not None and int(tiles[2][0]) == (rows + 31) // 32 and fused.identity_tiles(rows, gpu) is tiles
    dg, db, db_rows, dw = fused.bwd_h1_w(d_o, W2, h1, tiles, slope=0.18)
    ref_dg, ref_db = fused.bwd_h1_mfma(d_o, W2, h1, both_halves=False, gate=None, slope=0.18)
    ref_w, ref_rows = fused.atb_rows(d_o, h1, None)
    s_dg = float(ref_dg.abs().max())
    assert float((dg - ref_dg).abs().max()) <= 2e-6 * s_dg
    w64 = d_o.double().t() @ h1.double()
    scale_w = float((d_o.double().abs().t() @ h1.double().abs()).max())
    assert float((dw.double() - w64).abs().max()) <= 2e-6 * scale_w and float((ref_w.double() - w64).abs().max()) <= 2e-6 * scale_w
    assert float((db - ref_db).abs().max()) <= 2e-6 * rows * s_dg
    assert float((db_rows.double() - d_o.double().sum(0)).abs().max()) <= 2e-6 * rows * float(d_o.abs().max())


@pytest.mark.parametrize("rows,mapped,with_dst", [(1000, False, False), (70001, False, True), (70001, True, True), (300000, False, False)])
def test_input_gradient_and_class_typed_weight_gradient_in_one_launch(rows, mapped, with_dst, gpu):
    """``dmp_bwd_z_w`` (csrc/dmp_h1w.hip::dzw_k) against the two launches it replaces over the kept edges' class tiles: ``dz`` BIT-identical to
    ``dmp_bwd_z_typed_arow`` on the kept rows (zeros / untouched elsewhere), ``dWes = [z^T dPre | z^T (c dPre)]`` against fp64 and ``atb_typed``;
    NaN in every dead row of every operand; with a base TABLE + row map (the pooled last layer), with dead destinations (-1), and twice the same bits."""
    from dualmessagepassing_amd import fused
    h = 128
    gen = th.Generator().manual_seed(rows + mapped)
    rng = np.random.default_rng(rows)
    n = max(2, rows // 6)
    src, dst = rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64)
    ix = _index(src, dst, n, rng.random(rows) < 0.5, gpu)
    coef = ix.degree_coef(ix.out_deg)
    gate = th.from_numpy((rng.random(rows) < 0.46).astype(np.float32)).to(gpu)
    gate._dmp_binary = True
    gate._dmp_zero_rows = True
    dead = gate == 0
    z = th.randn(rows, h, generator=gen).to(gpu)
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    d_pre = th.randn(rows, h, generator=gen).to(gpu)
    d_s = th.randn(n, 2 * h, generator=gen).to(gpu)
    if mapped:
        G = 37
        base = th.randn(G, h, generator=gen).to(gpu)
        bmap = th.from_numpy(np.where(rng.random(rows) < 0.7, rng.integers(0, G, rows), -1).astype(np.int32)).to(gpu)
    else:
        base, bmap = th.randn(rows, h, generator=gen).to(gpu), None
        base[dead] = float("nan")
    dstm = None
    if with_dst:
        dm = ix.dst32.clone()
        dm[th.from_numpy(rng.random(rows) < 0.3).to(gpu)] = -1            # destinations whose rows of d_s were never written
        dstm = dm
    zp, dpp = z.clone(), d_pre.clone()
    zp[dead] = float("nan")
    dpp[dead] = float("nan")
    for mode in ("zero", "leave"):
        saved = fused.dead_rows_buffer
        fused.dead_rows_buffer = lambda shape, device: th.full(shape, 7.5, dtype=th.float32, device=device)
        try:
            ref_z = fused.bwd_z_typed(dpp, h, wes, d_s, base, coef, ix, base_map=bmap, gate=gate, dead_rows=mode, dst=dstm)
            both = fused.bwd_z_w(dpp, zp, wes, d_s, base, coef, ix, base_map=bmap, gate=gate, dead_rows=mode, dst=dstm)
            again = fused.bwd_z_w(dpp, zp, wes, d_s, base, coef, ix, base_map=bmap, gate=gate, dead_rows=mode, dst=dstm)
        finally:
            fused.dead_rows_buffer = saved
        assert both is not None
        dz, dw = both
        assert th.equal(dz[~dead], ref_z[~dead])
        assert bool((dz[dead] == (0.0 if mode == "zero" else 7.5)).all())
        assert th.equal(dz, again[0]) and th.equal(dw, again[1])
    saved = fused.USE_ATB2
    ref_w = fused.atb_typed(zp, dpp, coef, ix, gate=gate)
    keep = ~dead
    ce = ix.edge_select(coef)[2].double()
    z64, d64 = z[keep].double(), d_pre[keep].double()
    want = th.cat([z64.t() @ d64, z64.t() @ (ce[keep][:, None] * d64)], dim=1)
    scale = float((z64.abs().t() @ (ce[keep][:, None].abs().clamp(min=1.0) * d64.abs())).max())
    assert dw.shape == (h, 2 * h)
    assert float((dw.double() - want).abs().max()) <= 2e-6 * scale, float((dw.double() - want).abs().max()) / scale
    assert float((ref_w.double() - want).abs().max()) <= 2e-6 * scale


def test_input_gradient_and_weight_gradient_in_one_launch_without_a_gate(gpu):
    """No gate (the all-rows step): ``dmp_bwd_z_w`` over the class tiles of EVERY edge -- ``dz`` bit-identical to ``bwd_z_typed``, every row
    written; ``dWes`` against fp64."""
    from dualmessagepassing_amd import fused
    rows, h = 50000, 128
    gen = th.Generator().manual_seed(5)
    rng = np.random.default_rng(5)
    n = rows // 6
    src, dst = rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64)
    ix = _index(src, dst, n, rng.random(rows) < 0.5, gpu)
    coef = ix.degree_coef(ix.out_deg)
    z = th.randn(rows, h, generator=gen).to(gpu)
    wes = (th.randn(h, 2 * h, generator=gen) * 0.1).to(gpu)
    d_pre = th.randn(rows, h, generator=gen).to(gpu)
    d_s = th.randn(n, 2 * h, generator=gen).to(gpu)
    base = th.randn(rows, h, generator=gen).to(gpu)
    ref_z = fused.bwd_z_typed(d_pre, h, wes, d_s, base, coef, ix)
    both = fused.bwd_z_w(d_pre, z, wes, d_s, base, coef, ix)
    assert both is not None
    dz, dw = both
    assert th.equal(dz, ref_z)
    ce = ix.edge_select(coef)[2].double()
    want = th.cat([z.double().t() @ d_pre.double(), z.double().t() @ (ce[:, None] * d_pre.double())], dim=1)
    scale = float((z.double().abs().t() @ (ce[:, None].abs().clamp(min=1.0) * d_pre.double().abs())).max())
    assert float((dw.double() - want).abs().max()) <= 2e-6 * scale

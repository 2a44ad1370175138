"""Shared by tests/test_gpu_bf16x6.py and scripts/bf16x6_probe.py: every bf16x6 kernel of the library on adversarial
operands, each run twice -- as shipped (three bf16 pieces per fp32 operand, six piece products on v_mfma_f32_32x32x16_bf16)
and under ``dmp_dev_set_exact_fp32(1)`` (the same kernel on the f32-input MFMA) -- and compared with fp64."""
import numpy as np
import torch as th


def scenarios(rows, h, gen):
    """{name: [rows, h] fp32 operand}.  ``cancel``: columns k and k + h/2 hold x and -x (the weights below repeat with period
    h/2, so every exact product is 0); ``small`` / ``tiny``: values around 2^-100 / 2^-118 -- the third bf16 piece (2^-16 below the value)
    is still a normal bf16 number / lies under bf16's smallest subnormal (2^-133); ``element_binades`` / ``row_binades``: magnitudes from 2^-60 to 2^60 inside a row / row
    by row."""
    base = th.randn(rows, h, generator=gen)
    out = {"normal": base.clone()}
    out["element_binades"] = base * th.exp2(th.randint(-60, 61, (rows, h), generator=gen).float())
    out["row_binades"] = base * th.exp2(th.randint(-60, 61, (rows, 1), generator=gen).float())
    x = base[:, :h // 2]
    out["cancel"] = th.cat([x, -x], 1)
    out["small"] = base * 2.0 ** -100      # the third piece still a normal bf16 number: must be fp32-accurate
    out["tiny"] = base * 2.0 ** -118       # the third piece under bf16's range: two pieces left (2^-16)
    out["huge"] = base * 2.0 ** 100
    return out


def _rel(got, ref64, scale64):
    """largest |got - ref| / scale over the elements with a non-zero scale"""
    ok = scale64 > 0
    if not bool(ok.any()):
        return 0.0
    return float(((got.double() - ref64).abs()[ok] / scale64[ok]).max())


def _both(lib, fn):
    """fn() as shipped (bf16x6) and under the exact-fp32 switch."""
    assert not lib.dmp_dev_get_exact_fp32()
    x6 = fn()
    lib.dmp_dev_set_exact_fp32(1)
    try:
        ex = fn()
    finally:
        lib.dmp_dev_set_exact_fp32(0)
    return x6, ex


def run_all(gpu, rows=4099, h=128):
    from dualmessagepassing_amd import _lib, fused
    from dualmessagepassing_amd.graph import GraphIndex
    lib = _lib.load()
    gen = th.Generator().manual_seed(20260410)
    rng = np.random.default_rng(5)
    n = max(2, rows // 6)
    src, dst = rng.integers(0, n, rows).astype(np.int64), rng.integers(0, n, rows).astype(np.int64)
    ix = GraphIndex(th.from_numpy(src).to(gpu), th.from_numpy(dst).to(gpu), n, th.from_numpy(rng.random(rows) < 0.5).to(gpu))
    coef = ix.degree_coef(ix.out_deg)
    ce = ix.edge_select(coef)[2].double()                         # coef[dst e]
    half = (th.randn(h // 2, 2 * h, generator=gen) * 0.1)
    wes = th.cat([half, half], 0).to(gpu)                          # rows k and k + h/2 equal: "cancel" has exact result 0
    zeros_p = th.zeros(n, 3 * h, device=gpu)
    res = {}
    for name, z_cpu in scenarios(rows, h, gen).items():
        z = z_cpu.to(gpu)
        zd, wd = z.double(), wes.double()
        r = {}
        # --- edge_fwd_typed: act(z (A + c B) + P[a] - P[b] + bias) with P = 0, bias = 0, slope 1 (identity)
        ref = zd @ wd[:, :h] + ce[:, None] * (zd @ wd[:, h:])
        scale = zd.abs() @ wd[:, :h].abs() + ce[:, None] * (zd.abs() @ wd[:, h:].abs())
        x6, ex = _both(lib, lambda: fused.edge_fwd_typed(z, wes, zeros_p[:, h:], 3 * h, None, coef, ix, slope=1.0))
        r["edge_fwd_typed"] = (_rel(x6, ref, scale), _rel(ex, ref, scale), bool(th.isfinite(x6).all()))
        # --- bwd_z_typed: dPre (A + c B)^T  (no base, dS = 0)
        wT = th.cat([wes[:, :h].t(), wes[:, h:].t()], 1).contiguous()     # [A^T | B^T] as the kernel takes it
        ds0 = th.zeros(n, 2 * h, device=gpu)
        # the operand's "cancel" structure must sit on the contraction index: columns of dPre against rows of A^T -> use wes^T
        wes2 = th.cat([wes[:, :h].t().contiguous(), wes[:, h:].t().contiguous()], 1)     # [h, 2h]: (A^T | B^T) as a Wes
        w2d = wes2.double()
        ref = zd @ w2d[:, :h].t() + ce[:, None] * (zd @ w2d[:, h:].t())
        scale = zd.abs() @ w2d[:, :h].t().abs() + ce[:, None] * (zd.abs() @ w2d[:, h:].t().abs())
        x6, ex = _both(lib, lambda: fused.bwd_z_typed(z, h, wes2, ds0, None, coef, ix))
        r["bwd_z_typed"] = (_rel(x6, ref, scale), _rel(ex, ref, scale), bool(th.isfinite(x6).all()))
        # --- atb_typed: [z^T d | z^T (c d)] with d of unit scale (the contraction runs over the rows)
        d = th.randn(rows, h, generator=gen).to(gpu)
        dd = d.double()
        ref = th.cat([zd.t() @ dd, zd.t() @ (dd * ce[:, None])], 1)
        scale = th.cat([zd.abs().t() @ dd.abs(), zd.abs().t() @ (dd.abs() * ce[:, None])], 1)
        x6, ex = _both(lib, lambda: fused.atb_typed(z, d, coef, ix))
        r["atb_typed"] = (_rel(x6, ref, scale), _rel(ex, ref, scale), bool(th.isfinite(x6).all()))
        # --- out_fwd (round 4 on bf16x6): prev + gate (z W2^T + b2) with prev = 0, b2 = 0, gate = 1: z W2^T
        w2 = wes[:, :h].t().contiguous()                              # nn.Linear weight [out, in]: W2[j][k] = A[k][j]
        ones = th.ones(rows, device=gpu)
        ref = zd @ wd[:, :h]
        scale = zd.abs() @ wd[:, :h].abs()
        x6, ex = _both(lib, lambda: fused.out_fwd_mfma(z, w2, None, ones, None))
        r["out_fwd"] = (_rel(x6, ref, scale), _rel(ex, ref, scale), bool(th.isfinite(x6).all()))
        # --- bwd_h1 (round 4 on bf16x6): act'(H1) (gate z) W2 with H1 > 0 everywhere, gate = 1: z W2, W2 = A^T's storage
        w2b = wes[:, :h].contiguous()                                 # dH1 = dO @ W2 with W2 [out = k, in = j] = A[k][j]
        pos = th.ones(rows, h, device=gpu)
        x6, ex = _both(lib, lambda: fused.bwd_h1_mfma(z, w2b, pos, both_halves=False, gate=ones, slope=0.0)[0])
        r["bwd_h1"] = (_rel(x6, ref, scale), _rel(ex, ref, scale), bool(th.isfinite(x6).all()))
        res[name] = r
    return res

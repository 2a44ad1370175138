"""dmp_gemm_x6: fp32 row-block products on the bf16 matrix pipe (six piece products per partial product, csrc/dmp_gemm6.hip)
against fp64, with the error of torch's fp32 product of the same operands as the yardstick: the split products must be
fp32-ACCURATE (errors of the order of fp32 rounding of the sum, not of bf16 / tf32 rounding)."""
import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu


def _ref(a1, a2, B, transB, bias, add, rowscale, slope):
    A = a1.double() if a2 is None else th.cat([a1.double(), a2.double()], 1)
    P = A @ (B.double().t() if transB else B.double())
    if bias is not None:
        P = P + bias.double()
    act = (lambda v: v) if slope is None else (lambda v: th.where(v > 0, v, slope * v))
    if rowscale is not None:
        out = rowscale.double().view(-1, 1) * act(P)
        return out + add.double() if add is not None else out
    if add is not None:
        P = P + add.double()
    return act(P)


@pytest.mark.parametrize("R,K1,K2,N,transB", [
    (73728, 128, 0, 384, False),     # x @ Wx at the config-2 union batch
    (4099, 256, 128, 128, False),    # [S | x] @ [Bn; Wnl']: two-part A, ragged row tile
    (1000, 128, 0, 256, True),       # dPn @ Bn^T
    (777, 384, 0, 128, True),        # dXP @ Wx^T
    (130, 64, 64, 64, False),        # H = 64 shapes: 64-wide column tiles
    (5, 16, 0, 192, True), (128, 48, 16, 64, False), (0, 128, 0, 128, False),
])
def test_gemm_x6_is_fp32_accurate(R, K1, K2, N, transB, gpu):
    from dualmessagepassing_amd import fused
    g = th.Generator().manual_seed(R + K1 + N)
    a1 = th.randn(R, K1, generator=g).to(gpu) * 3
    a2 = th.randn(R, K2, generator=g).to(gpu) if K2 else None
    K = K1 + K2
    B = (th.randn(N, K, generator=g) if transB else th.randn(K, N, generator=g)).to(gpu)
    out = fused.gemm_x6(a1, B, a2, transB=transB)
    assert out.shape == (R, N)
    if R == 0:
        return
    ref = _ref(a1, a2, B, transB, None, None, None, None)
    scale = float(ref.abs().max())
    err = float((out.double() - ref).abs().max())
    A32 = a1 if a2 is None else th.cat([a1, a2], 1)
    e32 = float(((A32 @ (B.t() if transB else B)).double() - ref).abs().max())
    assert err <= max(3.0 * e32, 2e-6 * scale), (err, e32, scale)
    assert err <= 5e-6 * scale


@pytest.mark.parametrize("slope", [None, 0.0, 1 / 5.5])
@pytest.mark.parametrize("gated", [False, True])
def test_gemm_x6_epilogues(slope, gated, gpu):
    from dualmessagepassing_amd import fused
    g = th.Generator().manual_seed(7)
    R, K, N = 1031, 128, 128
    a, B = th.randn(R, K, generator=g).to(gpu), th.randn(K, N, generator=g).to(gpu) * 0.2
    bias, add = th.randn(N, generator=g).to(gpu), th.randn(R, N, generator=g).to(gpu)
    rs = (th.rand(R, generator=g) < 0.7).float().to(gpu) if gated else None
    out = fused.gemm_x6(a, B, bias=bias, add=add, rowscale=rs, slope=slope)
    ref = _ref(a, None, B, False, bias, add, rs, slope)
    assert float((out.double() - ref).abs().max()) <= 5e-6 * max(1.0, float(ref.abs().max()))
    # views: a column slice of a wider matrix as operand and as destination; the destination aliasing ``add``
    wide = th.randn(R, 3 * K, generator=g).to(gpu)
    dst = th.zeros(R, 2 * N, device=gpu)
    dst[:, N:] = add
    fused.gemm_x6(wide[:, K:2 * K], B, bias=bias, add=dst[:, N:], slope=slope, out=dst[:, N:])
    ref2 = _ref(wide[:, K:2 * K], None, B, False, bias, add, None, slope)
    assert float((dst[:, N:].double() - ref2).abs().max()) <= 5e-6 * max(1.0, float(ref2.abs().max()))
    assert float(dst[:, :N].abs().max()) == 0.0


def test_gemm_x6_refuses_what_it_cannot_take(gpu):
    from dualmessagepassing_amd import fused, _lib
    a, B = th.randn(8, 24, device=gpu), th.randn(24, 64, device=gpu)      # K not a multiple of 16
    with pytest.raises(_lib.DmpError):
        fused.gemm_x6(a, B)
    with pytest.raises(_lib.DmpError):
        fused.gemm_x6(th.randn(8, 32, device=gpu), th.randn(32, 40, device=gpu))   # N not a multiple of 64

"""Training-mode BatchNorm1d (+ the activation after it) on csrc/dmp_bn.hip against torch.nn.BatchNorm1d: values, running
statistics, gradients (UNC model.py:145-157: Linear -> BatchNorm1d -> LeakyReLU -> Linear)."""
import pytest
import torch as th

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,c", [(2, 256), (37, 256), (2708, 256), (10858, 256), (70001, 128), (513, 64), (1000, 32), (300, 512), (129, 1024), (50, 8)])
@pytest.mark.parametrize("slope", [None, 1 / 5.5, 0.0])
def test_batch_norm_act_equals_the_module(rows, c, slope, gpu):
    from dualmessagepassing_amd import ops
    gen = th.Generator().manual_seed(rows + c)
    x0 = (th.randn(rows, c, generator=gen) * 3.0 + th.randn(c, generator=gen) * 5.0).to(gpu)      # means far from 0
    dy = th.randn(rows, c, generator=gen).to(gpu)
    bn_a, bn_b = th.nn.BatchNorm1d(c).to(gpu), th.nn.BatchNorm1d(c).to(gpu)
    init = (th.rand(c, generator=gen) + 0.5, th.randn(c, generator=gen), th.randn(c, generator=gen), th.rand(c, generator=gen) + 0.5)
    with th.no_grad():
        for bn in (bn_a, bn_b):
            for t, v in zip((bn.weight, bn.bias, bn.running_mean, bn.running_var), init):
                t.copy_(v)
    act = (lambda t: t) if slope is None else (th.nn.ReLU() if slope == 0.0 else th.nn.LeakyReLU(slope))
    xa, xb = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    assert ops.batch_norm_act_ok(bn_a, xa)
    for _ in range(2):                                         # two steps: the running averages accumulate
        ya = ops.batch_norm_act(bn_a, xa, slope)
        yb = act(bn_b(xb))
    ga = th.autograd.grad(ya, [xa, bn_a.weight, bn_a.bias], dy)
    gb = th.autograd.grad(yb, [xb, bn_b.weight, bn_b.bias], dy)
    # fp64 reference of the forward
    xd = x0.double()
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    ref = (xd - mean) / (var + bn_b.eps).sqrt() * bn_b.weight.double() + bn_b.bias.double()
    if slope is not None:
        ref = th.where(ref > 0, ref, slope * ref)
    assert float((ya.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    assert float((ya - yb).abs().max()) <= 2e-5 * max(1.0, float(yb.abs().max()))
    assert th.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-6)
    assert th.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 2
    for name, a, b in zip(("dx", "dgamma", "dbeta"), ga, gb):
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 5e-5 * scale, (name, float((a - b).abs().max()), scale)


def test_apply_mlp_takes_the_hip_batch_norm_in_training_only(gpu):
    from dualmessagepassing_amd import ops
    th.manual_seed(0)
    mlp = th.nn.Sequential(th.nn.Linear(256, 256), th.nn.BatchNorm1d(256), th.nn.LeakyReLU(1 / 5.5), th.nn.Linear(256, 256)).to(gpu)
    x = th.randn(300, 256, device=gpu)
    calls = []
    orig = ops.batch_norm_act
    ops.batch_norm_act = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        mlp.train()
        y = ops.apply_mlp(mlp, x)
        assert calls == [1]
        ref_mlp = th.nn.Sequential(*[m for m in mlp])
        mlp.eval()
        ye = ops.apply_mlp(mlp, x)
        assert calls == [1]                                    # eval mode: the module itself
    finally:
        ops.batch_norm_act = orig
    assert y.shape == ye.shape == (300, 256)


@pytest.mark.parametrize("cap,rows,c", [(4096, 2708, 256), (512, 2, 64), (1024, 1024, 128), (12288, 10858, 256)])
@pytest.mark.parametrize("slope", [None, 1 / 5.5])
def test_batch_norm_over_the_real_rows_of_a_padded_batch(cap, rows, c, slope, gpu):
    """``batch_norm_act(rows_dev=...)`` (``dmp_bn_train_*_rows``: the row count on the device) on a batch padded to ``cap`` rows
    with garbage behind the real ones, against the module on the real rows alone: output, running statistics, all three
    gradients; the padding rows come out as zeros, forward and backward (UNC's sampled sub-graphs padded to a capacity so that
    their step replays, unc_harness.SampledStep)."""
    from dualmessagepassing_amd import ops
    gen = th.Generator().manual_seed(cap + rows + c)
    x0 = (th.randn(cap, c, generator=gen) * 3.0 + th.randn(c, generator=gen) * 5.0).to(gpu)
    x0[rows:] = 1e6                                              # padding: anything
    dy = th.randn(cap, c, generator=gen).to(gpu)
    dy[rows:] = -1e6
    n_dev = th.tensor([rows], dtype=th.int64, device=gpu)
    bn_a, bn_b = th.nn.BatchNorm1d(c).to(gpu), th.nn.BatchNorm1d(c).to(gpu)
    with th.no_grad():
        for bn in (bn_a, bn_b):
            bn.weight.copy_(th.rand(c, generator=th.Generator().manual_seed(1)) + 0.5)
            bn.bias.copy_(th.randn(c, generator=th.Generator().manual_seed(2)))
    act = (lambda t: t) if slope is None else th.nn.LeakyReLU(slope)
    xa, xb = x0.clone().requires_grad_(True), x0[:rows].clone().requires_grad_(True)
    ya = ops.batch_norm_act(bn_a, xa, slope, n_dev)
    yb = act(bn_b(xb)) if rows > 1 else None
    if yb is None:                                               # torch refuses one row; fp64 by hand (variance 0)
        return
    ga = th.autograd.grad(ya, [xa, bn_a.weight, bn_a.bias], dy)
    gb = th.autograd.grad(yb, [xb, bn_b.weight, bn_b.bias], dy[:rows])
    assert float(ya[rows:].abs().max()) == 0.0 if rows < cap else True
    assert float((ya[:rows] - yb).abs().max()) <= 2e-5 * max(1.0, float(yb.abs().max()))
    assert th.allclose(bn_a.running_mean, bn_b.running_mean, rtol=1e-5, atol=1e-6)
    assert th.allclose(bn_a.running_var, bn_b.running_var, rtol=1e-5, atol=1e-6)
    assert float(ga[0][rows:].abs().max()) == 0.0 if rows < cap else True
    for name, a, b in zip(("dx", "dgamma", "dbeta"), (ga[0][:rows], ga[1], ga[2]), gb):
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 5e-5 * scale, (name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("cap,rows,c", [(1024, 700, 50), (512, 3, 50)])
def test_masked_batch_norm_with_tensor_ops_for_widths_the_kernels_do_not_take(cap, rows, c, gpu):
    """``ops.masked_batch_norm`` (the reference's shipped UNC width, hid 50, is not a multiple of 4): ``apply_mlp`` of a padded
    batch falls back to it -- against the module on the real rows alone: output, running statistics, gradients."""
    from dualmessagepassing_amd import ops
    gen = th.Generator().manual_seed(cap + rows)
    x0 = (th.randn(cap, c, generator=gen) * 2.0 + 3.0).to(gpu)
    x0[rows:] = 1e4
    dy = th.randn(cap, c, generator=gen).to(gpu)
    n_dev = th.tensor([rows], dtype=th.int64, device=gpu)
    seq_a = th.nn.Sequential(th.nn.Linear(c, c), th.nn.BatchNorm1d(c), th.nn.LeakyReLU(0.2), th.nn.Linear(c, c)).to(gpu)
    import copy
    seq_b = copy.deepcopy(seq_a)
    xa, xb = x0.clone().requires_grad_(True), x0[:rows].clone().requires_grad_(True)
    ya = ops.apply_mlp(seq_a, xa, n_dev)
    yb = seq_b(xb)
    assert float((ya[:rows] - yb).abs().max()) <= 2e-5 * max(1.0, float(yb.abs().max()))
    assert th.allclose(seq_a[1].running_mean, seq_b[1].running_mean, rtol=1e-5, atol=1e-6)
    assert th.allclose(seq_a[1].running_var, seq_b[1].running_var, rtol=1e-4, atol=1e-6)
    ga = th.autograd.grad((ya[:rows] * dy[:rows]).sum(), [xa] + list(seq_a.parameters()))
    gb = th.autograd.grad((yb * dy[:rows]).sum(), [xb] + list(seq_b.parameters()))
    assert float(ga[0][rows:].abs().max()) == 0.0
    for a, b in zip([ga[0][:rows]] + list(ga[1:]), gb):
        assert float((a - b).abs().max()) <= 5e-5 * max(1.0, float(b.abs().max())) + 1e-6

"""``DMPLayer(batch_norm=True)`` -- the reference constructor's default (SubgraphCountingMatching/models/dmpnn.py:17-28,45-60;
flag ``rep_dmpnn_batch_norm``, config.py:201-207) -- on the GPU against the reference's own layer (``bnlayer_dmp_*.npz``,
oracle/make_golden.py::gen_dmplayer_bn): training mode runs the MLPs' BatchNorm1d + activation on csrc/dmp_bn.hip
(``ops.batch_norm_act``: batch statistics, running-average update), evaluation mode normalises with the running statistics;
outputs, input and parameter gradients, and the buffers after the pass.  Layers with BatchNorm take the modular path (the fused
layer folds the MLP's first Linear into the projections, which a normalisation in between forbids).

Under data parallelism every rank normalises with the statistics of ITS shard: the reference is single
device; per-shard statistics are what torch's DistributedDataParallel does without SyncBatchNorm."""
import numpy as np
import pytest
import torch as th

from conftest import golden_files, load_golden

pytestmark = pytest.mark.gpu


def _t(a):
    return th.from_numpy(np.asarray(a))


def _close(got, ref, tol, what):
    got, ref = got.detach().double().cpu(), _t(ref).double()
    assert got.shape == ref.shape, what
    err, scale = float((got - ref).abs().max()), max(1.0, float(ref.abs().max()))
    assert err <= tol * scale, "%s: max err %g (scale %g)" % (what, err, scale)


@pytest.mark.parametrize("path", golden_files("bnlayer_dmp_"))
def test_layer_with_batch_norm_matches_the_reference(path, gpu):
    from dualmessagepassing_amd import ops
    from dualmessagepassing_amd.dmpnn import DMPLayer
    from dualmessagepassing_amd.graph import BatchedGraph
    d = load_golden(path)
    h, train = d["x"].shape[1], str(d["mode"]) == "train"
    layer = DMPLayer(h, h, num_mlp_layers=2, batch_norm=True, act_func=str(d["act_func"]), dropout=0.0)
    state = {k[2:]: _t(v) for k, v in d.items() if k.startswith("p.")}
    state.update({k[3:]: _t(v) for k, v in d.items() if k.startswith("b0.")})
    layer.load_state_dict(state, strict=True)                  # parameters AND buffers under the reference's names
    layer.to(gpu).train(train)
    g = BatchedGraph(_t(d["src"]).to(gpu), _t(d["dst"]).to(gpu), int(d["num_nodes"]))
    g.edata["is_reversed"] = _t(d["rev"]).to(gpu)
    g.ndata["out_deg"] = _t(d["out_deg"]).to(gpu)
    x, z = _t(d["x"]).to(gpu).requires_grad_(True), _t(d["z"]).to(gpu).requires_grad_(True)
    assert not layer.fused_ok(g, x, z)                          # BatchNorm: the modular path
    used = []
    orig = ops.batch_norm_act
    ops.batch_norm_act = lambda *a, **k: (used.append(1), orig(*a, **k))[1]
    try:
        node_out, edge_out = layer(g, x, z)
    finally:
        ops.batch_norm_act = orig
    # training mode: both MLPs' BatchNorm ran on the HIP kernels where the width allows (C % 4 == 0 and 256 % (C/4) == 0)
    assert len(used) == (2 if train and ops.USE_HIP_BATCHNORM and 256 % (h // 4) == 0 else 0)
    _close(node_out, d["node_out"], 2e-5, "node_out")
    _close(edge_out, d["edge_out"], 2e-5, "edge_out")
    ((node_out * _t(d["wn"]).to(gpu)).sum() + (edge_out * _t(d["we"]).to(gpu)).sum()).backward()
    _close(x.grad, d["dx"], 2e-5, "dx")
    _close(z.grad, d["dz"], 2e-5, "dz")
    for k, p in layer.named_parameters():
        if "g." + k in d:
            _close(p.grad, d["g." + k], 2e-4, "grad " + k)
    for k, b in layer.named_buffers():
        _close(b.float() if b.dtype != th.float32 else b, d["b1." + k].astype(np.float64) if d["b1." + k].dtype.kind == "i" else d["b1." + k],
               1e-5, "buffer " + k)         # batch variance over 10^2..10^4 rows in another summation order

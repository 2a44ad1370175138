"""The node side of a layer over the nodes a 0 / 1 node gate keeps (round 5).

The reference multiplies the target's node rows by ``v_gate`` at the rep-net's input and after every layer
(SubgraphCountingMatching/models/dmpnn.py:245-277, basemodel.py:1515-1519): a node under a zero of the gate is a zero row in every
layer, so its aggregate, its projections, its update and every gradient row of it are dead.  The pieces that exploit that, each
against the form that computes every row: the kept-node tile list, the selectors that read a dead node's rows as zeros, the two
scatter-adds that neither sum nor store a dead node's row, the tile kernel's accumulate / activation epilogue -- and the model
with the path switched on against the path switched off, with the dead rows poisoned."""
import os
import sys

import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mask_words(keep):
    """uint32 words, bit r & 31 of word r >> 5 = keep[r] (as the int32 tensor the kernels take)."""
    k = keep.detach().cpu().numpy().astype(np.uint8)
    k = np.concatenate([k, np.zeros((-k.size) % 32, np.uint8)])
    return th.from_numpy(np.packbits(k, bitorder="little").view(np.int32).copy()).to(keep.device)


@pytest.mark.parametrize("R,p", [(1, 1.0), (31, 0.5), (32, 0.0), (4096, 0.3), (73728, 0.41), (100001, 0.9)])
def test_kept_row_tiles_are_the_kept_rows_padded_to_tiles(R, p, gpu):
    from dualmessagepassing_amd import fused
    g = th.Generator(device=gpu).manual_seed(R)
    gate = (th.rand(R, device=gpu, generator=g) < p).float()
    mask = fused.gate_row_mask(gate)
    assert th.equal(mask, _mask_words(gate != 0))
    lst, cnt = fused.kept_rows(mask, 0, R, tiles=True)
    want = th.nonzero(gate).view(-1).to(th.int32)
    n = want.numel()
    assert cnt.tolist() == [n, (n + 31) // 32]
    assert th.equal(lst[:n], want)
    assert bool((lst[n:(n + 31) // 32 * 32] == -1).all())
    lst2, cnt2 = fused.kept_rows(mask, 0, R)                      # the plain list is unchanged
    assert cnt2.tolist() == [n] and th.equal(lst2[:n], want)


def test_node_selectors_replace_dead_nodes_by_minus_one(gpu):
    from dualmessagepassing_amd.graph import GraphIndex
    g = th.Generator(device=gpu).manual_seed(5)
    N, E = 5000, 40000
    src = th.randint(0, N, (E,), device=gpu, generator=g)
    dst = th.randint(0, N, (E,), device=gpu, generator=g)
    rev = th.rand(E, device=gpu, generator=g) < 0.5
    keep = th.rand(N, device=gpu, generator=g) < 0.4
    ix = GraphIndex(src, dst, N, rev)
    sa, sb, dm = ix.edge_select_nodes(_mask_words(keep))
    u = th.where(keep[src], src, th.full_like(src, -1)).to(th.int32)
    v = th.where(keep[dst], dst, th.full_like(dst, -1)).to(th.int32)
    assert th.equal(sa, th.where(rev, u, v)) and th.equal(sb, th.where(rev, v, u)) and th.equal(dm, v)


def _batch(gpu, B=96, seed=3):
    """The index of the union of B pattern (8 nodes, 24 edges) + B target (64 nodes, 512 edges) graphs as bench.py collates them."""
    sys.path.insert(0, ROOT)
    import bench
    from dualmessagepassing_amd.collate import collate_device
    from dualmessagepassing_amd.dmpnn import prepare_joint
    cfg = dict(bench.CFG, batch=B)
    shard = bench.make_shard(cfg, seed, gpu)
    p, g = shard["p"], shard["g"]
    pattern = collate_device(p["local_src"], p["local_dst"], p["num_nodes"].clone(), p["num_edges"].clone(), p["N"], p["E"],
                             ndata=p["ndata"], edata=dict(p["edata"]), max_nodes=p["max_n"], max_edges=p["max_e"])
    graph = collate_device(g["local_src"], g["local_dst"], g["num_nodes"].clone(), g["num_edges"].clone(), g["N"], g["E"],
                           ndata=g["ndata"], edata=dict(g["edata"]), max_nodes=g["max_n"], max_edges=g["max_e"])
    return prepare_joint(pattern, graph, 128).index()


@pytest.mark.parametrize("H", [128, 64])
def test_scatter_adds_leave_out_the_dead_nodes_rows(H, gpu):
    """Forward sum over a row list / backward one-pass endpoint sums with a node mask: the kept nodes' rows equal the all-rows
    launches bit for bit, a dead node's row is not written (a sentinel survives), NaN in the skipped edge rows never shows."""
    from dualmessagepassing_amd import fused, ops
    ix = _batch(gpu)
    N, E = ix.num_nodes, ix.num_edges
    g = th.Generator(device=gpu).manual_seed(H)
    e_gate = (th.rand(E, device=gpu, generator=g) < 0.42).float()
    v_gate = (th.rand(N, device=gpu, generator=g) < 0.41).float()
    for t in (e_gate, v_gate):
        t._dmp_binary = True
        t._dmp_zero_rows = True
    M = th.randn(E, H, device=gpu, generator=g) * e_gate[:, None]
    kc = fused.keep_in_csr(ix, e_gate)
    vmask = fused.gate_row_mask(v_gate)
    rows = fused.kept_rows(vmask, 0, N)
    ref = ops.seg_sum_raw(M, kc[0], kc[1], N, None, True, -1.0, 1.0)
    Mp = M.clone()
    Mp[e_gate == 0] = float("nan")
    out = th.full((N, 2 * H), 7.5, device=gpu)
    ops.seg_sum_raw(Mp, kc[0], kc[1], N, None, True, -1.0, 1.0, out=out, rows=rows)
    keep = v_gate != 0
    assert th.equal(out[keep], ref[keep]) and bool((out[~keep] == 7.5).all())
    # ... and through the kept edges' CSR with a row per kept node (row pointers by list position: what the layer runs)
    kpi, kei = fused.NodeRows(vmask, rows, None, None).kept_incidence(ix, e_gate, in_only=True)
    out2 = th.full((N, 2 * H), 7.5, device=gpu)
    ops.seg_sum_raw(Mp, kpi, kei, N, None, True, -1.0, 1.0, out=out2, rows=rows, ptr_by_pos=True)
    assert th.equal(out2[keep], ref[keep]) and bool((out2[~keep] == 7.5).all())
    # backward: both endpoints' sums, dead nodes neither summed nor stored
    emask = fused.gate_row_mask(e_gate)
    refb = ops.endpoint_sums(M, ix, mask=emask, gate=e_gate)
    sel = ix.edge_select_nodes(vmask)
    outb = th.full((N, 2 * H), 7.5, device=gpu)
    ops.endpoint_sums(Mp, ix, out=outb, mask=emask, gate=e_gate, nodes=(vmask, sel[:2]))
    assert th.equal(outb[keep], refb[keep]) and bool((outb[~keep] == 7.5).all())
    # ... and as a segment sum over the kept edges' incidence CSR with a row per kept node (what the fused layer's backward
    # runs under both gates): the same bits, and an edge row WITHOUT a kept endpoint is never fetched (NaN there too)
    nd = fused.NodeRows(vmask, rows, None, sel)
    kp, ke = nd.kept_incidence(ix, e_gate)
    src, dst = ix.src32.long(), ix.dst32.long()
    untouched = (e_gate == 0) | ~(keep[src] | keep[dst])
    Mq = M.clone()
    Mq[untouched] = float("nan")
    outc = th.full((N, 2 * H), 7.5, device=gpu)
    ops.seg_sum_raw(Mq, kp, ke, N, None, True, 1.0, -1.0, out=outc, rows=rows, ptr_by_pos=True)
    assert th.equal(outc[keep], refb[keep]) and bool((outc[~keep] == 7.5).all())
    n = int(rows[1].item())
    full_inc = ix.incidence()
    want = sum(int(((e_gate[(full_inc[1][full_inc[0][v]:full_inc[0][v + 1]] >> 1).long()]) != 0).sum()) for v in rows[0][:200].tolist())
    assert int(kp[200].item()) == want and int(kp[n].item()) <= 2 * int((e_gate != 0).sum())


@pytest.mark.parametrize("H", [128, 64])
def test_tile_kernel_accumulates_and_activates_on_the_kept_nodes(H, gpu):
    """``out_fwd_typed`` on the kept nodes' tiles: a K = 3H product as three launches accumulating onto their own output with
    the activation on the last one, column blocks of wider operands, both weight layouts -- against fp64; dead rows untouched."""
    from dualmessagepassing_amd import fused
    g = th.Generator(device=gpu).manual_seed(H + 1)
    N = 9000
    v_gate = (th.rand(N, device=gpu, generator=g) < 0.41).float()
    vmask = fused.gate_row_mask(v_gate)
    lst, cnt = fused.kept_rows(vmask, 0, N, tiles=True)
    T = (lst, th.zeros((N + 31) // 32, device=gpu), cnt[1:2], (N + 31) // 32)
    keep = v_gate != 0
    A = th.randn(N, 3 * H, device=gpu, generator=g)
    A[~keep] = float("nan")
    W = th.randn(3 * H, H, device=gpu, generator=g) / (3 * H) ** 0.5
    bias = th.randn(H, device=gpu, generator=g)
    out = th.full((N, H), 7.5, device=gpu)
    fused.out_fwd_typed(A[:, :H], W[:H], bias, None, T, out=out)
    fused.out_fwd_typed(A[:, H:2 * H], W[H:2 * H], None, out, T, out=out)
    fused.out_fwd_typed(A[:, 2 * H:], W[2 * H:], None, out, T, out=out, slope=0.18)
    ref = th.nn.functional.leaky_relu(A[keep].double() @ W.double() + bias.double(), 0.18)
    scale = float((A[keep].double().abs() @ W.double().abs()).max())
    assert float((out[keep].double() - ref).abs().max()) <= 2e-6 * scale
    assert bool((out[~keep] == 7.5).all())
    # [out, in] layout into a column block of a wider matrix (dS = dPn Bn^T, dx = dxn + dXP Wx^T)
    Wt = th.randn(H, 3 * H, device=gpu, generator=g) / H ** 0.5            # [out = H, in = 3H]: block b is Wt[:, bH:(b+1)H]
    base = th.randn(N, H, device=gpu, generator=g)
    wide = th.full((N, 2 * H), 7.5, device=gpu)
    dst = wide[:, H:]
    for b in range(3):
        fused.out_fwd_typed(A[:, b * H:(b + 1) * H], Wt[:, b * H:(b + 1) * H], None, base if b == 0 else dst, T, out=dst, w_in_out=False)
    ref = base[keep].double() + A[keep].double() @ Wt.double().t()
    scale = float((A[keep].double().abs() @ Wt.double().abs().t()).max())
    assert float((dst[keep].double() - ref).abs().max()) <= 2e-6 * scale
    assert bool((wide[:, :H] == 7.5).all()) and bool((dst[~keep] == 7.5).all())
    # several products over the same tiles in ONE launch, then a product with two addends (the forms the fused layer uses)
    o = [th.full((N, H), 7.5, device=gpu) for _ in range(4)]
    fused.typed_jobs([dict(a=A[:, :H], W=W[:H], bias=bias, out=o[0]), dict(a=A[:, H:2 * H], W=W[H:2 * H], out=o[1]),
                      dict(a=A[:, :H], W=Wt[:, :H], w_in_out=False, prev=base, out=o[2]),
                      dict(a=A[:, 2 * H:], W=Wt[:, 2 * H:], w_in_out=False, slope=0.0, out=o[3])], T)
    fused.out_fwd_typed(A[:, 2 * H:], W[2 * H:], None, o[0], T, out=o[0], slope=0.18, prev2=o[1])
    refs = [th.nn.functional.leaky_relu(A[keep].double() @ W.double() + bias.double(), 0.18),
            A[keep, H:2 * H].double() @ W[H:2 * H].double(),
            base[keep].double() + A[keep, :H].double() @ Wt[:, :H].double().t(),
            th.relu(A[keep, 2 * H:].double() @ Wt[:, 2 * H:].double().t())]
    for got, ref in zip(o, refs):
        assert float((got[keep].double() - ref).abs().max()) <= 4e-6 * max(1.0, float(ref.abs().max())) * 3
        assert bool((got[~keep] == 7.5).all())


@pytest.mark.parametrize("lazy", [True, False])
def test_model_with_the_kept_node_path_equals_the_model_without_it(lazy, gpu):
    """bench.py's model on 96 pairs with ``DMP_NODE_ROWS`` on (dead rows poisoned with NaN) against the same model with it off:
    all 15 outputs and every parameter gradient."""
    from dualmessagepassing_amd import fused
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_compact import _model_and_batch, _outputs_and_grads
    sys.path.insert(0, ROOT)
    import bench
    cfg = dict(bench.CFG, batch=96)
    bench_, shard, step, model = _model_and_batch(cfg, gpu)
    ran = []
    real = fused.node_rows

    def spy(index, v_gate, H):
        r = real(index, v_gate, H)
        ran.append(r is not None)
        return r

    fused.USE_NODE_ROWS = False
    try:
        ref_out, ref_flat = _outputs_and_grads(bench_, cfg, shard, step, model, lazy)
    finally:
        fused.USE_NODE_ROWS = True
    fused.POISON_DEAD_ROWS = True
    fused.node_rows = spy
    try:
        out, flat = _outputs_and_grads(bench_, cfg, shard, step, model, lazy)
    finally:
        fused.POISON_DEAD_ROWS = False
        fused.node_rows = real
    assert ran and all(ran), "the kept-node path did not run"
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


@pytest.mark.parametrize("H", [128, 64])
def test_weight_gradients_over_the_kept_nodes_tiles(H, gpu):
    """``atb_rows_multi(tiles=...)``: several ``a^T b`` products over the rows of ONE tile list (rows gathered by slot) in one
    launch -- against fp64 over the kept rows; a dead row is never fetched (NaN there)."""
    from dualmessagepassing_amd import fused
    g = th.Generator(device=gpu).manual_seed(H + 7)
    N = 9000
    v_gate = (th.rand(N, device=gpu, generator=g) < 0.41).float()
    vmask = fused.gate_row_mask(v_gate)
    lst, cnt = fused.kept_rows(vmask, 0, N, tiles=True)
    T = (lst, th.zeros((N + 31) // 32, device=gpu), cnt[1:2], (N + 31) // 32)
    keep = v_gate != 0
    a1, b1 = th.randn(N, H, device=gpu, generator=g), th.randn(N, H, device=gpu, generator=g)
    a2, b2 = th.randn(N, 2 * H, device=gpu, generator=g), th.randn(N, H, device=gpu, generator=g)
    a3, b3 = th.randn(N, H, device=gpu, generator=g), th.randn(N, 3 * H, device=gpu, generator=g)
    for t in (a1, b1, a2, b2, a3, b3):
        t[~keep] = float("nan")
    (w1, _), (w2, _), (w3, _) = fused.atb_rows_multi([(a1, b1, None, False), (a2, b2, None, False), (a3, b3, None, False)], tiles=T)
    for w, a, b in ((w1, a1, b1), (w2, a2, b2), (w3, a3, b3)):
        ref = a[keep].double().t() @ b[keep].double()
        scale = float((a[keep].double().abs().t() @ b[keep].double().abs()).max())
        assert w.shape == ref.shape and bool(th.isfinite(w).all())
        assert float((w.double() - ref).abs().max()) <= 2e-6 * scale
    with pytest.raises(Exception):
        fused.atb_rows_multi([(a1, b1, v_gate, False)], tiles=T)          # the list says which rows take part: no gates here

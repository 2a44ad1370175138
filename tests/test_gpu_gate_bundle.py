"""``fused.gate_bundle``: the index arrays a step derives from its two 0 / 1 gates -- both row masks, the kept nodes' tile list, the
first layer's kept target rows, the kept edges' ascending tile list, the selectors with dead nodes as -1 -- built in three launches
(``dmp_row_mask_bits_jobs``, ``dmp_kept_rows_jobs``) against the ten single launches they replace: every array equal, bit for bit.
The arrays index the scatter-adds of SubgraphCountingMatching/models/dmpnn.py:92,163 under the ScalarFilter gates of
basemodel.py:1515-1531."""
import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu


def _t(a):
    return th.from_numpy(np.ascontiguousarray(a))


def _case(n, e, ep, seed, gpu, keep_v=0.4, keep_e=0.45):
    from dualmessagepassing_amd.graph import GraphIndex
    rng = np.random.default_rng(seed)
    src, dst = rng.integers(0, n, e).astype(np.int64), rng.integers(0, n, e).astype(np.int64)
    ix = GraphIndex(_t(src).to(gpu), _t(dst).to(gpu), n, _t(rng.random(e) < 0.5).to(gpu), validate=True)
    vg = _t((rng.random(n) < keep_v).astype(np.float32)).to(gpu)
    eg = _t((rng.random(e) < keep_e).astype(np.float32)).to(gpu)
    eg[:ep] = 1.0
    for g in (vg, eg):
        g._dmp_binary = True
        g._dmp_zero_rows = True
    return ix, vg, eg


@pytest.mark.parametrize("n,e,ep", [(5000, 70000, 32 * 40), (4096, 40001, 0), (73728, 548864, 24576), (33, 64, 32)])
def test_bundle_equals_the_single_launches(n, e, ep, gpu):
    from dualmessagepassing_amd import fused
    got = {}
    for on in (False, True):
        ix, vg, eg = _case(n, e, ep, n + e, gpu)
        saved = fused.USE_GATE_BUNDLE
        fused.USE_GATE_BUNDLE = on
        try:
            fused.gate_bundle(ix, vg, eg, want_nodes=True, l0_range=(ep, e), want_ascending=True)
            planted = on and getattr(ix, "_esel_nodes", None) is not None
            vm, em = fused.gate_row_mask(vg), fused.gate_row_mask(eg)
            nodes = fused.kept_rows(vm, 0, n, tiles=True)
            l0 = fused.kept_rows(em, ep, e)
            asc = fused.kept_rows(em, 0, e, tiles=True)
            sel = ix.edge_select_nodes(vm)
        finally:
            fused.USE_GATE_BUNDLE = saved
        assert planted == on
        th.cuda.synchronize()
        nk, ek = int(nodes[1][0]), int(asc[1][0])
        got[on] = dict(vm=vm.clone(), em=em.clone(), nodes=nodes[0][:(nk + 31) // 32 * 32].clone(), ncnt=nodes[1].clone(),
                       l0=l0[0][:int(l0[1][0])].clone(), l0cnt=l0[1].clone(), asc=asc[0][:(ek + 31) // 32 * 32].clone(), acnt=asc[1].clone(),
                       selA=sel[0].clone(), selB=sel[1].clone(), dstM=sel[2].clone())
    for k in got[False]:
        assert th.equal(got[False][k], got[True][k]), k
    # ... and they are what they say: the kept nodes ascending, padded with -1 to whole tiles
    keep = (_case(n, e, ep, n + e, gpu)[1] != 0).nonzero().view(-1).int()
    assert th.equal(got[True]["nodes"][:keep.numel()], keep) and bool((got[True]["nodes"][keep.numel():] == -1).all())
    assert got[True]["ncnt"].tolist() == [keep.numel(), (keep.numel() + 31) // 32]


def test_bundle_leaves_what_it_is_not_asked_for(gpu):
    from dualmessagepassing_amd import fused
    ix, vg, eg = _case(6000, 50000, 3200, 9, gpu)
    fused.gate_bundle(ix, vg, eg, want_nodes=False, l0_range=None, want_ascending=False)
    vm, em = fused.gate_row_mask(vg), fused.gate_row_mask(eg)          # the masks are there ...
    assert vm is not None and em is not None
    assert getattr(ix, "_esel_nodes", None) is None and not getattr(vm, "_dmp_kept_tiles", None) and not getattr(em, "_dmp_kept_rows", None)
    ref_v = ((vg.view(-1, 1) != 0) if False else None)
    bits = ((vm.view(-1, 1).to(th.int64) >> th.arange(32, device=gpu)) & 1).reshape(-1)[:6000].float()
    assert th.equal(bits, (vg != 0).float())
    # a dense gate: nothing is built (every consumer takes its unmasked form)
    ix2, vg2, eg2 = _case(6000, 50000, 3200, 10, gpu)
    eg2._dmp_dense_gate = True
    fused.gate_bundle(ix2, vg2, eg2, want_nodes=True, l0_range=(3200, 50000), want_ascending=True)
    assert getattr(vg2, "_dmp_row_mask", None) is None and getattr(ix2, "_esel_nodes", None) is None

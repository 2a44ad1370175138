"""Host-side logic of the deferred tensors and the label-code descriptor (no GPU): ``embed.DeferredRows``,
``dmpnn._LazyEdgeRows``, ``OutputDict`` resolution, ``fused.Layer0Codes`` table ranges."""
import torch as th


def test_deferred_rows_compute_once_and_report_their_shape():
    from dualmessagepassing_amd.embed import DeferredRows, materialize
    calls = []

    def fn():
        calls.append(1)
        return th.arange(12.0).view(4, 3)

    d = DeferredRows(fn, (4, 3), th.float32, th.device("cpu"))
    assert d.size() == th.Size((4, 3)) and d.size(1) == 3 and d.dim() == 2 and not d.is_cuda and not calls
    a = d.materialize()
    b = materialize(d)
    assert a is b and calls == [1]
    assert materialize(None) is None and materialize(a) is a


def test_lazy_edge_rows_split_once_with_gradients():
    from dualmessagepassing_amd.dmpnn import _LazyEdgeRows
    w = th.ones(5, 2, requires_grad=True)
    calls = []
    lazy = _LazyEdgeRows(lambda: (calls.append(1), w * 2.0)[1], 2, (5, 2), th.float32, th.device("cpu"))
    whole, p, g = lazy.whole(), lazy.part(0), lazy.part(1)
    assert p.size(0) == 2 and g.size(0) == 3 and whole.size(0) == 5 and not calls
    gp = g.materialize()
    pp = p.materialize()
    ww = whole.materialize()
    assert calls == [1] and gp.shape == (3, 2) and pp.shape == (2, 2) and ww.shape == (5, 2)
    (gp.sum() + 3.0 * pp.sum()).backward()
    assert th.equal(w.grad, th.tensor([[6.0, 6.0], [6.0, 6.0], [2.0, 2.0], [2.0, 2.0], [2.0, 2.0]]))


def test_output_dict_resolves_deferred_entries_in_place():
    from dualmessagepassing_amd.basemodel import OutputDict
    from dualmessagepassing_amd.embed import DeferredRows
    d = DeferredRows(lambda: th.zeros(2, 2), (2, 2), th.float32, th.device("cpu"))
    out = OutputDict(a=th.ones(1), g_e_rep=d)
    assert isinstance(dict.__getitem__(out, "g_e_rep"), DeferredRows)
    assert th.is_tensor(out.g_e_rep) and th.is_tensor(dict.__getitem__(out, "g_e_rep"))
    assert [k for k, _ in out.items()] == ["a", "g_e_rep"] and all(th.is_tensor(v) for v in out.values())
    assert len(out.to_tuple()) == 2 and out.get("missing") is None


def test_layer0_codes_table_ranges():
    from dualmessagepassing_amd.fused import Layer0Codes
    enc = th.zeros(10, 12)
    one = Layer0Codes(enc, 10, th.zeros(10, 128))
    assert one.tables(10, 7) == [(0, (0, 10), (0, 7))]
    two = Layer0Codes(enc, 10, th.zeros(20, 128), esplit=3, nsplit=2)
    assert two.tables(10, 7) == [(0, (0, 3), (0, 2)), (1, (3, 10), (2, 7))]
    assert Layer0Codes(enc, 10, th.zeros(20, 128), esplit=0, nsplit=0).tables(10, 7) == [(1, (0, 10), (0, 7))]   # no pattern rows
    assert Layer0Codes(enc, 10, th.zeros(20, 128), esplit=0, nsplit=2).tables(10, 7) == [(0, (0, 0), (0, 2)), (1, (0, 10), (2, 7))]   # pattern nodes without edges
    two.venc, two.VK, two.WV = th.zeros(7, 16), 16, th.zeros(16, 128)      # the node rows' two tables stacked as one
    assert two.vtables(7) == [(0, (0, 7))]
    two.VK, two.WV = 8, th.zeros(16, 128)
    assert two.vtables(7) == [(0, (0, 2)), (1, (2, 7))]

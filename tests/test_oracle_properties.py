"""Property tests (hypothesis) of the oracle's building blocks on CPU: the segment sum equals
a dense one-hot product, the layer is equivariant to node relabelling, the line graph has
sum_v indeg(v)*outdeg(v) edges and each dual edge joins head-to-tail, reversed-edge doubling
symmetrises degrees."""
import numpy as np
import torch as th
from hypothesis import given, settings, strategies as st

import dmp_oracle as O
import graph_oracle as GO


@st.composite
def graphs(draw, max_n=9, max_e=24):
    n = draw(st.integers(1, max_n))
    e = draw(st.integers(0, max_e))
    src = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
    dst = draw(st.lists(st.integers(0, n - 1), min_size=e, max_size=e))
    return n, np.array(src, np.int64), np.array(dst, np.int64)


@settings(max_examples=60, deadline=None)
@given(graphs(), st.integers(1, 5), st.integers(0, 2 ** 31 - 1))
def test_seg_sum_is_one_hot_matmul(g, h, seed):
    n, src, dst = g
    m = th.randn(len(src), h, generator=th.Generator().manual_seed(seed), dtype=th.float64)
    onehot = th.zeros(n, len(src), dtype=th.float64)
    if len(src):
        onehot[th.from_numpy(dst), th.arange(len(src))] = 1.0
    assert th.allclose(O.seg_sum(m, th.from_numpy(dst), n), onehot @ m, atol=1e-12)


@settings(max_examples=40, deadline=None)
@given(graphs(max_n=7, max_e=16), st.integers(0, 2 ** 31 - 1))
def test_dmp_layer_is_equivariant_to_node_relabelling(g, seed):
    n, src, dst = g
    gen = th.Generator().manual_seed(seed)
    h = 4
    p = {k: v.double() for k, v in O.random_dmp_params(h, h, gen).items()}
    x = th.randn(n, h, generator=gen, dtype=th.float64)
    z = th.randn(len(src), h, generator=gen, dtype=th.float64)
    rev = th.rand(len(src), generator=gen) < 0.5
    ts, td = th.from_numpy(src), th.from_numpy(dst)
    no, eo, _, _ = O.dmp_layer(p, ts, td, rev, O.out_degrees(ts, n), x, z)
    perm = th.randperm(n, generator=gen)          # new id of old node i is perm[i]
    inv = th.empty_like(perm)
    inv[perm] = th.arange(n)
    ps, pd = perm[ts], perm[td]
    no2, eo2, _, _ = O.dmp_layer(p, ps, pd, rev, O.out_degrees(ps, n), x[inv], z)
    assert th.allclose(no2[perm], no, atol=1e-9) and th.allclose(eo2, eo, atol=1e-9)


@settings(max_examples=80, deadline=None)
@given(graphs())
def test_line_graph_size_and_incidence(g):
    n, src, dst = g
    ds, dt, dn, nd, ed = GO.convert_to_dual_graph(src, dst, n, {}, {})
    indeg, outdeg = np.bincount(dst, minlength=n), np.bincount(src, minlength=n)
    assert dn == len(src) and len(ds) == int((indeg * outdeg).sum())
    # dual edge (i -> e): head of i is the tail of e, and the payload is that shared node
    assert np.array_equal(dst[ds], src[dt]) and np.array_equal(ed["id"], src[dt])
    # emission order: by e, then by ascending in-edge id
    key = dt * (len(src) + 1) + ds
    assert np.all(np.diff(key) > 0) if len(key) > 1 else True


@settings(max_examples=60, deadline=None)
@given(graphs())
def test_reversed_edges_symmetrise_degrees(g):
    n, src, dst = g
    e = len(src)
    s, t, eid, el, rev = GO.add_reversed_edges(src, dst, np.arange(e), np.zeros(e, np.int64), e + 3, 5)
    assert np.array_equal(np.bincount(s, minlength=n), np.bincount(t, minlength=n))
    assert np.array_equal(s[:e], src) and np.array_equal(t[e:], src) and rev.sum() == e
    assert np.array_equal(eid[e:], e + 3 + np.arange(e)) and np.all(el[e:] == 5)

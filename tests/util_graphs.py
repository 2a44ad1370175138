"""Synthetic graph helpers shared by tests and bench (host side, numpy)."""
import numpy as np


def er_edges(n, m, rng):
    """m distinct ordered pairs u != v drawn uniformly without replacement."""
    total = n * (n - 1)
    pick = rng.choice(total, size=m, replace=False)
    u = pick // (n - 1)
    r = pick % (n - 1)
    v = r + (r >= u)
    return u.astype(np.int64), v.astype(np.int64)


def with_rev(u, v):
    e = len(u)
    return (np.concatenate([u, v]), np.concatenate([v, u]),
            np.concatenate([np.zeros(e, bool), np.ones(e, bool)]))


def er_batch(batch, n, m, rng, add_rev=True):
    """Block-diagonal batch of ER graphs in the layer's edge order ([fwd | rev] per graph)."""
    src, dst, rev = [], [], []
    for b in range(batch):
        u, v = er_edges(n, m, rng)
        if add_rev:
            u, v, r = with_rev(u, v)
        else:
            r = np.zeros(len(u), bool)
        src.append(u + b * n)
        dst.append(v + b * n)
        rev.append(r)
    e = (2 * m if add_rev else m)
    return (np.concatenate(src), np.concatenate(dst), np.concatenate(rev), batch * n,
            np.full(batch, n, np.int64), np.full(batch, e, np.int64))


def skewed_graph(n, e, rng):
    """Heavy-tailed in-degrees (a few hubs), self loops and duplicate edges allowed."""
    w = 1.0 / (1.0 + np.arange(n)) ** 1.2
    w /= w.sum()
    dst = rng.choice(n, size=e, p=w).astype(np.int64)
    src = rng.integers(0, n, size=e).astype(np.int64)
    return src, dst

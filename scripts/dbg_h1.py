import numpy as np, torch as th, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd import fused
from dualmessagepassing_amd.graph import GraphIndex
gpu = th.device("cuda:0")
for rows in (64, 1000, 24577, 70001, 70016):
    h = 128
    gen = th.Generator().manual_seed(rows + 1)
    rng = np.random.default_rng(rows + 1)
    n = max(2, rows // 5)
    src = th.from_numpy(rng.integers(0, n, rows).astype(np.int64)).to(gpu)
    dst = th.from_numpy(rng.integers(0, n, rows).astype(np.int64)).to(gpu)
    rev = th.from_numpy(rng.random(rows) < 0.5).to(gpu)
    ix = GraphIndex(src, dst, n, rev)
    coef = ix.degree_coef(ix.out_deg)
    d_o = th.randn(rows, h, generator=gen).to(gpu)
    h1 = th.randn(rows, h, generator=gen).clamp_min(0).to(gpu)
    w2 = (th.randn(h, h, generator=gen) * 0.1).to(gpu)
    d_g, cs = fused.bwd_h1_mfma(d_o, w2, h1, coef, ix)
    dpre = th.where(h1 > 0, (d_o.double() @ w2.double()), th.zeros(1, dtype=th.float64, device=gpu))
    ref = th.cat([dpre, dpre * coef.double()[dst][:, None]], 1)
    err = (d_g.double() - ref).abs()
    bad = (err > 1e-3).nonzero()
    print(rows, "max err", float(err.max()), "bad elems", bad.shape[0], "rows", bad[:, 0].unique()[:10].tolist(), "cols", bad[:, 1].unique()[:10].tolist(),
          "colsum err", float((cs.double() - dpre.sum(0)).abs().max()))
    if bad.shape[0]:
        for (r, c) in bad[:6].tolist():
            got = float(d_g[r, c])
            t = r // 32 * 32
            blk = ref[t:t + 32]
            where = (blk - got).abs() < 1e-4
            w = where.nonzero()[:4].tolist()
            print("   bad (%d,%d) row%%32=%d got %.5f ref %.5f  unmasked %.5f  matches ref at (row-in-tile, col): %s  h1>0: %s" % (
                r, c, r % 32, got, float(ref[r, c]), float((d_o.double() @ w2.double())[r, c % 128]), w, bool(h1[r, c % 128] > 0)))

"""Dev aid: class-typed vs untyped edge kernels and the cost of building the class tile list,
on a config-2 shaped union graph (ER patterns + targets with reversed edges)."""
import os, sys, time
import numpy as np, torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd import fused
from dualmessagepassing_amd.graph import GraphIndex
gpu = th.device("cuda:0")
rng = np.random.default_rng(0)
B, h = 1024, 128
def er(b, n, m):
    pick = np.argsort(rng.random((b, n * (n - 1))), axis=1)[:, :m]
    u = pick // (n - 1); r = pick % (n - 1); v = r + (r >= u)
    off = (np.arange(b) * n)[:, None]
    return np.concatenate([u + off, v + off], 1).reshape(-1), np.concatenate([v + off, u + off], 1).reshape(-1), np.concatenate([np.zeros((b, m), bool), np.ones((b, m), bool)], 1).reshape(-1)
ps, pd, pr = er(B, 8, 12); gs, gd, gr = er(B, 64, 256)
src = th.from_numpy(np.concatenate([ps, gs + B * 8])).to(gpu); dst = th.from_numpy(np.concatenate([pd, gd + B * 8])).to(gpu)
rev = th.from_numpy(np.concatenate([pr, gr])).to(gpu)
n, e = B * 72, src.numel()
ix = GraphIndex(src, dst, n, rev)
coef = ix.degree_coef(ix.out_deg)
def timeit(f, it=20):
    for _ in range(3): f()
    th.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): f()
    th.cuda.synchronize(); return (time.perf_counter() - t) / it * 1e6
def build():
    ix._ctiles = None
    return ix.class_tiles(coef)
print("class_tiles build      %7.1f us" % timeit(build))
se, ts, nt, bound = ix.class_tiles(coef)
print("tiles in use %d of bound %d (E/32 = %d), classes %d" % (int(nt), bound, e // 32, th.unique(ts[:int(nt)]).numel()))
g = th.Generator().manual_seed(0)
z = th.randn(e, h, generator=g).to(gpu); wes = (th.randn(h, 2 * h, generator=g) * .1).to(gpu); xp = th.randn(n, 3 * h, generator=g).to(gpu)
bias = th.randn(h, generator=g).to(gpu); dpre = th.randn(e, 2 * h, generator=g).to(gpu); ds = th.randn(n, 2 * h, generator=g).to(gpu); base = th.randn(e, h, generator=g).to(gpu)
print("edge_fwd untyped       %7.1f us" % timeit(lambda: fused.edge_fwd_mfma(z, wes, xp[:, h:], 3 * h, bias, coef, ix)))
print("edge_fwd typed         %7.1f us" % timeit(lambda: fused.edge_fwd_typed(z, wes, xp[:, h:], 3 * h, bias, coef, ix)))
print("bwd_z untyped          %7.1f us" % timeit(lambda: fused.bwd_z_mfma(dpre, wes, ds, base, coef, ix)))
print("bwd_z typed            %7.1f us" % timeit(lambda: fused.bwd_z_typed(dpre, 2 * h, wes, ds, base, coef, ix)))
# ---- which torch op of the index build is slow?
E, dev, C = e, gpu, 65536
coef_e = ix.edge_select(coef)[2]
bound = E // 32 + C + 1
def T(name, f):
    print("   %-28s %8.1f us" % (name, timeit(f, 10)))
vals, order = th.sort(coef_e, stable=True)
T("sort float [E] stable", lambda: th.sort(coef_e, stable=True))
T("sort float [E]", lambda: th.sort(coef_e))
T("sort float [N]", lambda: th.sort(coef))
cidx = th.zeros(E, dtype=th.int64, device=dev)
T("cumsum neq", lambda: th.cumsum(vals[1:] != vals[:-1], 0, out=cidx[1:]))
T("bincount C=65536", lambda: th.bincount(cidx, minlength=C))
cnt = th.bincount(cidx, minlength=C); ntile = (cnt + 31) >> 5; tile_end = th.cumsum(ntile, 0); seg_start = th.cumsum(cnt, 0) - cnt
T("slot arithmetic", lambda: (tile_end - ntile)[cidx] * 32 + (th.arange(E, device=dev) - seg_start[cidx]))
slot = (tile_end - ntile)[cidx] * 32 + (th.arange(E, device=dev) - seg_start[cidx])
se2 = th.full((bound * 32,), -1, dtype=th.int32, device=dev)
T("full + index_put", lambda: th.full((bound * 32,), -1, dtype=th.int32, device=dev).index_put_((slot,), order.to(th.int32)))
T("searchsorted", lambda: th.searchsorted(tile_end, th.arange(bound, device=dev), right=True))
T("gather dst class", lambda: coef[ix.dst32.long()])
dp = dpre[:, :h].contiguous()
print("atb untyped [z^T dG]   %7.1f us" % timeit(lambda: fused.atb(z, dpre)))
print("atb typed              %7.1f us" % timeit(lambda: fused.atb_typed(z, dp, coef, ix)))
w2 = (th.randn(h, h, generator=g) * .1).to(gpu); h1 = z.clamp_min(0)
print("bwd_h1 both halves     %7.1f us" % timeit(lambda: fused.bwd_h1_mfma(dp, w2, h1, coef, ix)))
print("bwd_h1 dPre only       %7.1f us" % timeit(lambda: fused.bwd_h1_mfma(dp, w2, h1, coef, ix, both_halves=False)))
gate = th.rand(e, generator=g).to(gpu)
def old_path():
    d_o, db = fused.scale_rows_colsum(dp, gate)
    return fused.atb(d_o, h1), db
print("gate pass + bmm dW2    %7.1f us" % timeit(old_path))
print("atb_rows gated         %7.1f us" % timeit(lambda: fused.atb_rows(dp, h1, gate)))
print("bwd_h1 dPre only gated %7.1f us" % timeit(lambda: fused.bwd_h1_mfma(dp, w2, h1, coef, ix, both_halves=False, gate=gate)))

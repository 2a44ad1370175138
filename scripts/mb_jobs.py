"""Dev aid: the node-side weight-gradient launch (atb_rows_multi: 1 + 2 + 3 output blocks over N rows) and the E-row rows product."""
import os, sys, torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd import fused
gpu = th.device("cuda:0")
n, e, h = 73728, 548864, 128
g = th.Generator().manual_seed(0)
dxn, H1n, x = (th.randn(n, h, generator=g).to(gpu) for _ in range(3))
S, dXP = th.randn(n, 2 * h, generator=g).to(gpu), th.randn(n, 3 * h, generator=g).to(gpu)
gate = (th.rand(n, generator=g) < 0.7).float().to(gpu)
dzn, H1e = th.randn(e, h, device=gpu), th.randn(e, h, device=gpu)
eg = (th.rand(e, device=gpu) < 0.5).float()
def timeit(f, it=20):
    for _ in range(3): f()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): f()
    b.record(); th.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
def jobs():
    with fused.deferred_reductions():
        return fused.atb_rows_multi([(dxn, H1n, gate, True), (S, dXP[:, :h], None, False), (x, dXP, None, False)])
def rows():
    with fused.deferred_reductions():
        return fused.atb_rows(dzn, H1e, eg)
r = jobs(); ref = [t[0].clone() for t in r]
print("jobs (6 blocks, N rows): %.1f us   rows (E rows): %.1f us" % (timeit(jobs), timeit(rows)), flush=True)
a = x.double().t() @ dXP.double()
print("max rel err of x^T dXP: %.2e" % float((ref[2].double() - a).abs().max() / a.abs().max()))

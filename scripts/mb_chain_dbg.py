"""Dev aid: time dmp_edge_chain_fwd of debug builds (scripts/_dbg/libch_<knobs>.so; knobs of DMP_CH_DBG in csrc/dmp_chain.hip)."""
import ctypes, glob, os, sys
import numpy as np, torch as th
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
se, ts, nt, bound = ix.class_tiles(coef)
selA, selB, coefE = ix.edge_select(coef)
g = th.Generator().manual_seed(0)
Z = [th.randn(e, h, generator=g).to(gpu) for _ in range(3)]
wes = (th.randn(h, 2 * h, generator=g) * .1).to(gpu); xp = th.randn(n, 3 * h, generator=g).to(gpu)
bias = th.randn(h, generator=g).to(gpu); dsn = th.randn(n, 2 * h, generator=g).to(gpu)
base = [th.randn(e, h, generator=g).to(gpu) for _ in range(3)]
out = th.empty(e, h, device=gpu)
P, I64, I, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
st = th.cuda.current_stream().cuda_stream
def timeit(f, n=20):
    for i in range(3): f(i)
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): f(i)
    b.record(); th.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
w2t = (th.randn(h, h, generator=g) * .1).to(gpu); b2 = th.randn(h, generator=g).to(gpu)
gate = (th.rand(e, generator=g) < 0.7).float().to(gpu)
out2 = th.empty(e, h, device=gpu)
paths = sorted(glob.glob(os.path.join(ROOT, "scripts", "_dbg", "libch_*.so")), key=lambda s: int(s.split("_")[-1][:-3]))
for path in paths:
    lib = ctypes.CDLL(path)
    lib.dmp_edge_chain_fwd.argtypes = [P, I64, P, I64, P, I64, I64, P, P, P, P, P, P, I64, I64, I, F, P, I64, P, I64, P, P, I, P, I64, P]
    def run(i):
        rc = lib.dmp_edge_chain_fwd(Z[i % 3].data_ptr(), h, wes.data_ptr(), 2 * h, xp[:, h:].data_ptr(), 3 * h, n, bias.data_ptr(),
                                    selA.data_ptr(), selB.data_ptr(), se.data_ptr(), ts.data_ptr(), nt.data_ptr(), bound, e, h, 0.0,
                                    out.data_ptr(), h, w2t.data_ptr(), h, b2.data_ptr(), gate.data_ptr(), 1, out2.data_ptr(), h, st)
        assert rc == 0, rc
    print("%-14s edge_chain %7.1f us" % (os.path.basename(path), timeit(run)), flush=True)

"""Dev aid: time the layer-0 kernels of debug builds (scripts/_dbg/libl0_<knobs>.so; knobs of DMP_L0_VAR in csrc/dmp_layer0.hip)."""
import ctypes, glob, os, sys
import numpy as np, torch as th
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dualmessagepassing_amd.graph import GraphIndex
gpu = th.device("cuda:0")
rng = np.random.default_rng(0)
B, h, K = 1024, 128, 10
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
selA, selB, coefE = ix.edge_select(coef)
g = th.Generator().manual_seed(0)
enc = th.zeros(e, 12, device=gpu); enc[:, :K] = (th.rand(e, K, generator=g) < 0.5).float().to(gpu)
M = th.randn(K, 2 * h, generator=g).to(gpu); xp = th.randn(n, 3 * h, generator=g).to(gpu); bias = th.randn(h, generator=g).to(gpu)
D = [th.randn(e, h, generator=g).to(gpu) for _ in range(3)]; Zn = [th.randn(e, h, generator=g).to(gpu) for _ in range(3)]
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
ref = None
paths = sorted(glob.glob(os.path.join(ROOT, "scripts", "_dbg", "libl0_*.so")), key=lambda s: int(s.split("_")[-1][:-3]))
for path in paths:
    lib = ctypes.CDLL(path)
    lib.dmp_l0_edge_fwd.argtypes = [P, I64, I, P, I64, P, I64, P, P, P, P, I64, I, F, P, I64, P]
    lib.dmp_l0_bwd_w.argtypes = [P, I64, I, P, P, I64, P, I64, I64, I, P, P]
    lib.dmp_l0_bwd_w_blocks.restype = I64; lib.dmp_l0_bwd_w_blocks.argtypes = [I64]
    G = lib.dmp_l0_bwd_w_blocks(e)
    part = th.empty(G, 3 * K * h, device=gpu)
    def fwd(i):
        rc = lib.dmp_l0_edge_fwd(enc.data_ptr(), 12, K, M.data_ptr(), 2 * h, xp[:, h:].data_ptr(), 3 * h, bias.data_ptr(), coefE.data_ptr(),
                                 selA.data_ptr(), selB.data_ptr(), e, h, 0.0, out.data_ptr(), h, st)
        assert rc == 0, rc
    def bwd(i):
        rc = lib.dmp_l0_bwd_w(enc.data_ptr(), 12, K, coefE.data_ptr(), D[i % 3].data_ptr(), h, Zn[i % 3].data_ptr(), h, e, h, part.data_ptr(), st)
        assert rc == 0, rc
    tf, tb = timeit(fwd), timeit(bwd)
    bwd(0); th.cuda.synchronize()
    o, s = out.clone(), part.sum(0)
    if ref is None: ref = (o, s)
    print("%-14s fwd %7.1f us  bwd %7.1f us   (fwd equal %s, bwd maxdiff %.2e)" % (os.path.basename(path), tf, tb, th.equal(o, ref[0]),
          float((s - ref[1]).abs().max() / ref[1].abs().max())), flush=True)

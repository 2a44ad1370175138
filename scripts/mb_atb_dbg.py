"""Dev aid: time dmp_atb_typed of debug builds (scripts/_dbg/libatb_<knobs>.so; knobs: 1 no MFMAs,
2 no row loads after the prologue, 4 no partial stores, 8 contiguous rows instead of the gather)."""
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
g = th.Generator().manual_seed(0)
Z = [th.randn(e, h, generator=g).to(gpu) for _ in range(3)]
D = [th.randn(e, h, generator=g).to(gpu) for _ in range(3)]
P, I64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
st = th.cuda.current_stream().cuda_stream
def timeit(f, n=20):
    for i in range(3): f(i)
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): f(i)
    b.record(); th.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
pat = sys.argv[1] if len(sys.argv) > 1 else "libatb_*.so"
for path in sorted(glob.glob(os.path.join(ROOT, "scripts", "_dbg", pat)), key=lambda s: int(s.split("_")[-1][:-3])):
    lib = ctypes.CDLL(path)
    lib.dmp_atb_typed_blocks.restype = I64; lib.dmp_atb_typed_blocks.argtypes = [I64]
    lib.dmp_atb_typed.argtypes = [P, I64, P, I64, P, P, P, I64, I64, I, P, P, P]
    G = int(lib.dmp_atb_typed_blocks(bound))
    part = th.empty((G, 2, h * h), device=gpu)
    def run(i):
        rc = lib.dmp_atb_typed(Z[i % 3].data_ptr(), h, D[i % 3].data_ptr(), h, se.data_ptr(), ts.data_ptr(), nt.data_ptr(), bound, e, h,
                               part.data_ptr(), part[0, 1].data_ptr(), st)
        assert rc == 0, rc
    print("%-16s G=%d  %7.1f us" % (os.path.basename(path), G, timeit(run)), flush=True)

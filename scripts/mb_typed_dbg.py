"""Dev aid: time dmp_edge_fwd_typed / dmp_bwd_z_typed of debug builds (scripts/_dbg/libty_<knobs>.so; knobs:
1 no MFMAs, 2 no epilogue operand loads, 4 no stores, 8 no row loads after the prologue)."""
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
paths = sorted(glob.glob(os.path.join(ROOT, "scripts", "_dbg", "libty_*.so")), key=lambda s: int(s.split("_")[-1][:-3]))
paths = [os.path.join(ROOT, "dualmessagepassing_amd", "csrc", "libdmp_hip.so")] * 2 + paths      # the product library: bf16x6, then exact fp32
for n_path, path in enumerate(paths):
    lib = ctypes.CDLL(path)
    if hasattr(lib, "dmp_dev_set_exact_fp32"):
        lib.dmp_dev_set_exact_fp32(1 if n_path == 1 else 0)
        print("exact_fp32 =", lib.dmp_dev_get_exact_fp32())
    lib.dmp_edge_fwd_typed.argtypes = [P, I64, P, I64, P, I64, I64, P, P, P, P, P, P, I64, I64, I, F, P, I64, P]
    lib.dmp_bwd_z_typed.argtypes = [P, I64, P, I64, P, I64, I64, P, I64, P, P, F, F, P, P, P, I64, I64, I, I, P, I64, P, I64, P]
    def fwd(i):
        rc = lib.dmp_edge_fwd_typed(Z[i % 3].data_ptr(), h, wes.data_ptr(), 2 * h, xp[:, h:].data_ptr(), 3 * h, n, bias.data_ptr(),
                                    selA.data_ptr(), selB.data_ptr(), se.data_ptr(), ts.data_ptr(), nt.data_ptr(), bound, e, h,
                                    0.0, out.data_ptr(), h, st)
        assert rc == 0, rc
    def bwd(i):
        rc = lib.dmp_bwd_z_typed(Z[i % 3].data_ptr(), h, wes.data_ptr(), 2 * h, dsn.data_ptr(), 2 * h, n, base[i % 3].data_ptr(), h,
                                 ix.dst32.data_ptr(), ix.rev8.data_ptr(), -1.0, 1.0, se.data_ptr(), ts.data_ptr(), nt.data_ptr(), bound, e, h, 0, None, 0,
                                 out.data_ptr(), h, st)
        assert rc == 0, rc
    print("%-14s edge_fwd %7.1f us   bwd_z %7.1f us" % (os.path.basename(path), timeit(fwd), timeit(bwd)), flush=True)

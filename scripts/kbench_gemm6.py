"""Dev aid: dmp_gemm_x6 against torch (hipBLASLt fp32) on the node-side product shapes of bench.py's step."""
import os, sys
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd import fused
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
gpu = th.device("cuda:0")
R, H = 73728, 128
g = th.Generator().manual_seed(0)
x, S = th.randn(R, H, generator=g).to(gpu), th.randn(R, 2 * H, generator=g).to(gpu)
Wx, Bn = th.randn(H, 3 * H, generator=g).to(gpu), th.randn(2 * H, H, generator=g).to(gpu)
dPn, dXP = th.randn(R, H, generator=g).to(gpu), th.randn(R, 3 * H, generator=g).to(gpu)
def timeit(f, n=30):
    for _ in range(3): f()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); th.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
cases = [("x@Wx        K128 N384", lambda: x @ Wx, lambda: fused.gemm_x6(x, Wx), 2 * R * 128 * 384),
         ("[S|x]@[Bn;W] K384 N128", lambda: th.addmm(x @ Wx[:, :H], S, Bn), lambda: fused.gemm_x6(S, th.cat([Bn, Wx[:, :H]], 0), x), 2 * R * 384 * 128),
         ("dPn@Bn^T    K128 N256", lambda: dPn @ Bn.t(), lambda: fused.gemm_x6(dPn, Bn, transB=True), 2 * R * 128 * 256),
         ("dXP@Wx^T    K384 N128", lambda: dXP @ Wx.t(), lambda: fused.gemm_x6(dXP, Wx, transB=True), 2 * R * 384 * 128)]
for name, f_lib, f_x6, flops in cases:
    tl, tx = timeit(f_lib), timeit(f_x6)
    print("%-24s library %7.1f us (%5.1f TF/s)   x6 %7.1f us (%5.1f TF/s)" % (name, tl, flops / tl / 1e6, tx, flops / tx / 1e6), flush=True)

import ctypes, glob
P, I64, I, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
st = th.cuda.current_stream().cuda_stream
out = th.empty(R, 3 * H, device=gpu)
for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dbg", "libg6_*.so")), key=lambda s: int(s.split("_")[-1][:-3])):
    lib = ctypes.CDLL(path)
    lib.dmp_gemm_x6.argtypes = [P, I64, I, P, I64, I, P, I64, I, P, P, I64, P, I, F, P, I64, I64, I, P]
    def run():
        rc = lib.dmp_gemm_x6(x.data_ptr(), H, H, None, 0, 0, Wx.data_ptr(), 3 * H, 0, None, None, 0, None, 0, 0.0, out.data_ptr(), 3 * H, R, 3 * H, st)
        assert rc == 0, rc
    def run2():
        rc = lib.dmp_gemm_x6(dXP.data_ptr(), 3 * H, 3 * H, None, 0, 0, Wx.data_ptr(), 3 * H, 1, None, None, 0, None, 0, 0.0, out.data_ptr(), H, R, H, st)
        assert rc == 0, rc
    print("%-14s x@Wx %7.1f us   dXP@Wx^T %7.1f us" % (os.path.basename(path), timeit(run), timeit(run2)), flush=True)

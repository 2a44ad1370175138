"""Dev aid: a^T b for tall-skinny operands at the UNC shapes: bmm + sum (ops.atb_splitk) vs one library product."""
import os, sys, torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd import ops
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
gpu = th.device("cuda:0")
def timeit(f, n=50):
    for _ in range(5): f()
    th.cuda.synchronize()
    g = th.cuda.CUDAGraph()
    s = th.cuda.Stream()
    with th.cuda.stream(s):
        f(); th.cuda.synchronize()
        with th.cuda.graph(g, stream=s):
            for _ in range(10): f()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n // 10): g.replay()
    b.record(); th.cuda.synchronize()
    return a.elapsed_time(b) / (n // 10 * 10) * 1e3
for R in (2708, 10858, 21716):
    for K, N in ((256, 256),):
        a, b = th.randn(R, K, device=gpu), th.randn(R, N, device=gpu)
        t1 = timeit(lambda: ops.atb_splitk(a, b))
        t2 = timeit(lambda: a.t() @ b)
        err = float((ops.atb_splitk(a, b) - a.t() @ b).abs().max())
        print("R %6d K %d N %d: splitk (bmm + sum) %.1f us, one product %.1f us  (max diff %.2e)" % (R, K, N, t1, t2, err), flush=True)

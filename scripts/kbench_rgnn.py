"""RGCN / RGIN layer timing at the config-2 target batch (N = 65,536, E = 524,288, 32 edge types,
H = 128): typed aggregation forward and forward+backward, per layer."""
import os, sys, time
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd.graph import BatchedGraph
from dualmessagepassing_amd.rgnn import RGCNLayer, RGINLayer, typed_index, typed_linear_agg
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
gpu = th.device("cuda:0")
gen = th.Generator().manual_seed(5)
n, e, h, r = 65536, 524288, 128, 32
src = th.randint(0, n, (e,), generator=gen).to(gpu); dst = th.randint(0, n, (e,), generator=gen).to(gpu)
et = th.randint(0, r, (e,), generator=gen).to(gpu)
g = BatchedGraph(src, dst, n)


def timeit(f, iters=20):
    for _ in range(3): f()
    th.cuda.synchronize(); t = time.perf_counter()
    for _ in range(iters): f()
    th.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


x = th.randn(n, h, generator=gen).to(gpu).requires_grad_(True)
w = (th.randn(r, h, h, generator=gen) * 0.1).to(gpu).requires_grad_(True)
tix = typed_index(g, et, r)
from dualmessagepassing_amd import rgnn
for flag, label in ((True, "relation-typed MFMA kernels (dmp_rel_gemm / dmp_rel_atb)"), (False, "one library GEMM per type")):
    rgnn.USE_REL_KERNELS = flag
    print("--", label)
    print("typed_linear_agg fwd      %.3f ms" % timeit(lambda: typed_linear_agg(x.detach(), w.detach(), tix)))
    def fb():
        x.grad = w.grad = None
        typed_linear_agg(x, w, tix).square().sum().backward()
    print("typed_linear_agg fwd+bwd  %.3f ms   (algorithmic: 3 x 2EH^2 = %.1f GFLOP)" % (timeit(fb), 6 * e * h * h / 1e9))
    for name, layer in (("RGCNLayer(in)", RGCNLayer(h, h, num_rels=r).to(gpu)), ("RGINLayer", RGINLayer(h, h, num_rels=r).to(gpu))):
        def step():
            for p in layer.parameters(): p.grad = None
            x.grad = None
            layer(g, x, et)[0].square().sum().backward()
        print("%-14s fwd+bwd  %.3f ms" % (name, timeit(step)))

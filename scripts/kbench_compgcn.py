"""CompGCNLayer (SubgraphCountingMatching/models/compgcn.py:102-287) forward + backward at the BASELINE config-2 target batch
(1024 graphs of 64 nodes / 512 edges with reversed copies: N = 65,536, E = 524,288, hid 128) for the three compositions --
corr is the reference's default (config.py:170-172).  Prints the layer time and, for the fused aggregation kernel
(dmp_compgcn_agg), HIP-event time, algorithmic bytes and the fraction of the 8 TB/s HBM peak.

    python scripts/kbench_compgcn.py [--json out.json]
"""
import json, os, sys
import numpy as np, torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from util_graphs import er_batch
from dualmessagepassing_amd import _lib
from dualmessagepassing_amd.compgcn import CompGCNLayer
from dualmessagepassing_amd.graph import BatchedGraph
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
gpu = th.device("cuda:0")
B, H = 1024, 128
rng = np.random.default_rng(2)
src, dst, rev, N, bnn, bne = er_batch(B, 64, 256, rng)
E = len(src)
g = BatchedGraph(th.from_numpy(src).to(gpu), th.from_numpy(dst).to(gpu), N, th.from_numpy(bnn).to(gpu), th.from_numpy(bne).to(gpu))
g.edata["is_reversed"] = th.from_numpy(rev).to(gpu)
gen = th.Generator().manual_seed(0)
x0, z0 = th.randn(N, H, generator=gen).to(gpu), th.randn(E, H, generator=gen).to(gpu)
wn, we = th.randn(N, H, generator=gen).to(gpu), th.randn(E, H, generator=gen).to(gpu)

def timeit(f, n=20):
    for _ in range(3): f()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); th.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

out = {}
from dualmessagepassing_amd import compgcn as _cg
for comp in ("corr", "corr_fft", "mult", "sub"):
    _cg.USE_FREQ_DOMAIN_CORR = comp != "corr_fft"      # corr_fft: the reference's formulation (gathered rows + library FFTs over E rows)
    name, comp = comp, comp.split("_")[0]
    th.manual_seed(1)
    layer = CompGCNLayer(H, H, comp_opt=comp, edge_norm="both", batch_norm=False, act_func="leaky_relu").to(gpu)
    def step():
        x, z = x0.clone().requires_grad_(True), z0.clone().requires_grad_(True)
        for p in layer.parameters():
            p.grad = None
        a, b = layer(g, x, z)
        ((a * wn).sum() + (b * we).sum()).backward()
    t = timeit(step)
    # the aggregation kernel alone: width W of the rows it moves (corr: 132 floats = 66 (re, im) bins; else H)
    W = 132 if comp == "corr" else H
    ix = g.index()
    fx, fz = th.randn(N, W, generator=gen).to(gpu), th.randn(E, W, generator=gen).to(gpu)
    norm = th.rand(E, generator=gen).to(gpu)
    lib = _lib.load()
    o = th.empty(N, 2 * W, device=gpu)
    mode = {"sub": 0, "mult": 1, "corr": 2}[comp]
    def agg():
        _lib.check(lib.dmp_compgcn_agg(fx.data_ptr(), W, fz.data_ptr(), W, ix.in_ptr.data_ptr(), ix.in_ent.data_ptr(), ix.src32.data_ptr(),
                                       norm.data_ptr(), N, W, mode, o.data_ptr(), 2 * W, _lib.stream_ptr()), "agg")
    ta = timeit(agg, 50)
    nbytes = 4 * W * (E + N) + 8 * W * N + 4 * E * 3 + 4 * (N + 1)     # Fz rows, Fx rows, output, ent + src + norm, rowptr
    out[name] = {"layer_fwd_bwd_us": round(t, 1), "agg_kernel_us": round(ta, 1), "agg_bytes": nbytes,
                 "agg_gbps": round(nbytes / ta / 1e3, 1), "agg_frac_of_8TBps": round(nbytes / ta / 1e3 / 8000.0, 4)}
    print(name, out[name], flush=True)
if "--json" in sys.argv:
    json.dump({"shape": {"N": N, "E": E, "H": H}, "results": out}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)

"""Per-kernel timing of the production entry points at BASELINE config-2 target-graph size
(development aid; numbers quoted in DESIGN.md §4).  cold = inputs rotate through > 1 GiB of
distinct buffers (nothing L2 / Infinity-Cache resident), hot = same buffers every launch.
The float4 copy line is this device's streaming ceiling for comparison."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_graphs import er_batch  # noqa: E402
from dualmessagepassing_amd import _lib  # noqa: E402
from dualmessagepassing_amd.graph import GraphIndex  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(2000)
src, dst, rev, n, _, _ = er_batch(B, 64, 256, rng)
e, h = len(src), 128
ix = GraphIndex(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n, torch.from_numpy(rev).to(dev))
inc_ptr, inc_ent = ix.incidence()
coef = ix.degree_coef(ix.out_deg)
NB = 5
Z = [torch.randn(e, h, device=dev) for _ in range(NB)]
G2 = [torch.randn(e, 2 * h, device=dev) for _ in range(NB)]
P2 = [torch.randn(n, 2 * h, device=dev) for _ in range(NB)]
OUT_E = [torch.empty(e, h, device=dev) for _ in range(NB)]
OUT_E2 = [torch.empty(e, 2 * h, device=dev) for _ in range(NB)]
OUT_N2 = [torch.empty(n, 2 * h, device=dev) for _ in range(NB)]
bias = torch.randn(h, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, cold, iters=20):
    for i in range(3):
        fn(i % NB if cold else 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % NB if cold else 0)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def report(name, fn, nbytes):
    c, w = timeit(fn, True), timeit(fn, False)
    print("%-28s %6.1f MB  cold %7.1f us %6.0f GB/s | hot %7.1f us %6.0f GB/s"
          % (name, nbytes / 1e6, c, nbytes / c / 1e3, w, nbytes / w / 1e3), flush=True)


report("torch copy_ (ceiling)", lambda i: OUT_E[(i + 1) % NB].copy_(Z[i]), 8 * e * h)
report("seg_sum2 (in-CSR)", lambda i: lib.dmp_seg_sum2(Z[i].data_ptr(), h, ix.in_ptr.data_ptr(), ix.in_ent.data_ptr(), None, n, h, -1.0, 1.0, OUT_N2[i].data_ptr(), 2 * h, 0, st),
       4 * h * e + 8 * h * n + 4 * e + 4 * (n + 1))
report("seg_sum (in-CSR)", lambda i: lib.dmp_seg_sum(Z[i].data_ptr(), h, ix.in_ptr.data_ptr(), ix.in_ent.data_ptr(), None, n, h, OUT_N2[i].data_ptr(), h, 0, st),
       4 * h * e + 4 * h * n + 4 * e + 4 * (n + 1))
report("seg_sum2 (incidence)", lambda i: lib.dmp_seg_sum2(Z[i].data_ptr(), h, inc_ptr.data_ptr(), inc_ent.data_ptr(), None, n, h, 1.0, -1.0, OUT_N2[i].data_ptr(), 2 * h, 1, st),
       4 * h * e + 8 * h * n + 8 * e + 4 * (n + 1))
report("gather_select", lambda i: lib.dmp_gather_select(P2[i].data_ptr(), 2 * h, ix.dst32.data_ptr(), ix.rev8.data_ptr(), None, None, 0, e, h, -1.0, 1.0, OUT_E[i].data_ptr(), h, st),
       4 * h * (e + 2 * n) + 5 * e)
report("gather_rows", lambda i: lib.dmp_gather_rows(P2[i].data_ptr(), 2 * h, ix.dst32.data_ptr(), None, e, h, OUT_E[i].data_ptr(), h, st),
       4 * h * (e + n) + 4 * e)
report("edge_combine", lambda i: lib.dmp_edge_combine(G2[i].data_ptr(), 2 * h, P2[i].data_ptr(), 2 * h, coef.data_ptr(), bias.data_ptr(), ix.src32.data_ptr(), ix.dst32.data_ptr(), ix.rev8.data_ptr(), e, h, 0, 0.0, OUT_E[i].data_ptr(), h, st),
       4 * h * (3 * e + 2 * n) + 9 * e + 4 * n)
report("edge_combine_bwd_g", lambda i: lib.dmp_edge_combine_bwd_g(Z[i].data_ptr(), h, coef.data_ptr(), ix.dst32.data_ptr(), e, h, OUT_E2[i].data_ptr(), 2 * h, st),
       12 * h * e + 4 * e + 4 * n)

# pipeline-like: the input is produced by an elementwise kernel right before (as in the layer
# stack: residual add -> next layer's seg_sum2); A/B of the dispatch-order flag in one process
print("--- seg_sum2 right after its producer (events around the seg_sum2 only)")
for flagv in (0, 1, 0, 1):
    tot = 0.0
    evs = []
    for i in range(20):
        j = i % NB
        torch.add(Z[j], Z[(j + 1) % NB], out=OUT_E[j])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        lib.dmp_seg_sum2(OUT_E[j].data_ptr(), h, ix.in_ptr.data_ptr(), ix.in_ent.data_ptr(), None, n, h, -1.0, 1.0,
                         OUT_N2[j].data_ptr(), 2 * h, flagv, st)
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[3:])
    print("rows_shared=%d  median %.1f us  min %.1f us" % (flagv, ts[len(ts) // 2], ts[0]))

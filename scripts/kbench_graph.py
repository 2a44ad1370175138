"""Integer graph passes of the path (SURVEY §8 rows A11-A13 and the index builds behind A3) at BASELINE batch sizes:
device time per pass (HIP events, whole batch) beside the sequential C / numpy oracle on one host core.
  collate (dgl.batch semantics), add_reversed_edges, convert_to_dual_graph (directed line graph), and the per-batch
  index build of the layers (in- / out-CSR, incidence CSR, degree-class tiles)."""
import os, sys, time
import numpy as np
import torch as th
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from util_graphs import er_edges
import graph_oracle as GO
from dualmessagepassing_amd.collate import collate_device
from dualmessagepassing_amd.graph import GraphIndex
from dualmessagepassing_amd.linegraph import convert_to_dual_graph
from dualmessagepassing_amd.preprocess import add_reversed_edges
gpu = th.device("cuda:0")


def dev_ms(fn, iters=10):
    for _ in range(2): fn()
    th.cuda.synchronize()
    a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): out = fn()
    b.record(); th.cuda.synchronize()
    return a.elapsed_time(b) / iters, out


for batch, n, m, host_graphs in ((1024, 64, 256, 1024), (256, 512, 4096, 32)):
    rng = np.random.default_rng(batch)
    per = [er_edges(n, m, rng) for _ in range(batch)]
    ls, ld = np.concatenate([p[0] for p in per]), np.concatenate([p[1] for p in per])
    nn, ne = np.full(batch, n, np.int64), np.full(batch, m, np.int64)
    el = rng.integers(0, 16, batch * m)
    t = lambda a: th.from_numpy(a).to(gpu)
    tls, tld, tnn, tne, tid, tel = t(ls), t(ld), t(nn), t(ne), t(np.tile(np.arange(m), batch)), t(el)
    nl = rng.integers(0, 16, batch * n)
    nd = {"id": t(np.tile(np.arange(n), batch)), "label": t(nl)}
    mk = lambda: collate_device(tls, tld, tnn, tne, batch * n, batch * m, nd, {"id": tid, "label": tel}, max_nodes=n, max_edges=m)
    c_ms, g = dev_ms(mk)
    r_ms, r = dev_ms(lambda: add_reversed_edges(g, m, 16))
    l_ms, dg = dev_ms(lambda: convert_to_dual_graph(r))
    def index():
        ix = GraphIndex(r._src, r._dst, r.number_of_nodes(), r.edata["is_reversed"])
        ix.incidence(); ix.class_tiles(ix.degree_coef(ix.out_deg))
        return ix
    i_ms, _ = dev_ms(index)
    # host: the C oracle graph by graph on one core (a sample of the graphs, scaled)
    t0 = time.perf_counter()
    for i in range(host_graphs):
        a = GO.add_reversed_edges(per[i][0], per[i][1], np.arange(m), el[i * m:(i + 1) * m], m, 16)
    h_rev = (time.perf_counter() - t0) * batch / host_graphs * 1e3
    t0 = time.perf_counter()
    for i in range(host_graphs):
        a = GO.add_reversed_edges(per[i][0], per[i][1], np.arange(m), el[i * m:(i + 1) * m], m, 16)
        GO.convert_to_dual_graph(a[0], a[1], n, {"id": np.arange(n), "label": nl[i * n:(i + 1) * n]},
                                 {"id": a[2], "label": a[3], "is_reversed": a[4]})
    h_lg = (time.perf_counter() - t0) * batch / host_graphs * 1e3 - h_rev
    t0 = time.perf_counter(); GO.collate(ls, ld, nn, ne); h_col = (time.perf_counter() - t0) * 1e3
    E2, EL = 2 * batch * m, dg.number_of_edges()
    print("batch %d x (%d nodes, %d -> %d edges): line graph %d nodes, %d edges" % (batch, n, m, 2 * m, dg.number_of_nodes(), EL))
    print("  collate              %8.3f ms device | %9.1f ms C oracle, one core   (%.0f M edges/s on the device)" % (c_ms, h_col, batch * m / c_ms / 1e3))
    print("  add_reversed_edges   %8.3f ms device | %9.1f ms" % (r_ms, h_rev))
    print("  line-graph transform %8.3f ms device | %9.1f ms                        (%.0f M line-graph edges/s)" % (l_ms, h_lg, EL / l_ms / 1e3))
    print("  layer index build    %8.3f ms device   (in/out CSR + incidence CSR + degree-class tiles over %d edges)" % (i_ms, E2))

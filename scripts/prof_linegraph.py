"""Dev aid: the line-graph transform of the config-2 target batch (after add_reversed_edges), N times, for rocprofv3."""
import os, sys
import numpy as np, torch as th
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_graphs import er_edges
from dualmessagepassing_amd.collate import collate_device
from dualmessagepassing_amd.linegraph import convert_to_dual_graph
from dualmessagepassing_amd.preprocess import add_reversed_edges
gpu = th.device("cuda:0")
batch, n, m = 1024, 64, 256
rng = np.random.default_rng(batch)
per = [er_edges(n, m, rng) for _ in range(batch)]
t = lambda a: th.from_numpy(a).to(gpu)
g = collate_device(t(np.concatenate([p[0] for p in per])), t(np.concatenate([p[1] for p in per])), t(np.full(batch, n, np.int64)),
                   t(np.full(batch, m, np.int64)), batch * n, batch * m,
                   {"id": t(np.tile(np.arange(n), batch)), "label": t(rng.integers(0, 16, batch * n))},
                   {"id": t(np.tile(np.arange(m), batch)), "label": t(rng.integers(0, 16, batch * m))}, max_nodes=n, max_edges=m)
r = add_reversed_edges(g, m, 16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    dg = convert_to_dual_graph(r)
th.cuda.synchronize()
print(dg.number_of_nodes(), dg.number_of_edges())

"""Launch ONLY the scatter-add kernel (dmp_seg_sum2) at bench.py's launch shape (union of the config-2 pattern and
target batches) for rocprofv3 PMC / kernel-trace passes: ``python prof_seg_sum2.py [reps] [in|inc]`` -- ``in``: over
the CSR by destination (the layer's node aggregation, forward), ``inc``: over the incidence CSR (backward of the
gathered node projections: every edge row belongs to two nodes)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_graphs import er_batch  # noqa: E402
from dualmessagepassing_amd import ops  # noqa: E402
from dualmessagepassing_amd.graph import GraphIndex  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = 1024
rng = np.random.default_rng(2000)
ps, pd, pr, pn, _, _ = er_batch(B, 8, 12, rng)
gs, gd, gr, gn, _, _ = er_batch(B, 64, 256, rng)
src = np.concatenate([ps, gs + pn]); dst = np.concatenate([pd, gd + pn]); rev = np.concatenate([pr, gr])
n, e, h = pn + gn, len(src), 128
ix = GraphIndex(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n, torch.from_numpy(rev).to(dev))
zs = [torch.randn(e, h, device=dev) for _ in range(4)]
torch.cuda.synchronize()
kind = sys.argv[2] if len(sys.argv) > 2 else "in"
inc_ptr, inc_ent = ix.incidence()
torch.cuda.synchronize()
for i in range(reps):
    if kind == "in":
        s = ops.seg_sum_raw(zs[i % 4], ix.in_ptr, ix.in_ent, n, None, True, -1.0, 1.0)
    else:
        s = ops.seg_sum_raw(zs[i % 4], inc_ptr, inc_ent, n, None, True, 1.0, -1.0, rows_shared=2)
torch.cuda.synchronize()
print("rows", n, "edges", e, "H", h, "kind", kind)

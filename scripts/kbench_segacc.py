#!/usr/bin/env python3
"""The backward scatter-add at bench.py's launch shape (union of 1024 pattern + 1024 target graphs, hid 128): the
one-pass graph-tile kernel (dmp_seg_sum2_graphs) against dmp_seg_sum2 over the incidence CSR.
HIP-event time per launch over rotating inputs (4 x 281 MB: every launch reads from HBM)."""
import json
import os
import sys

import numpy as np
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bench
    from dualmessagepassing_amd import ops
    from dualmessagepassing_amd.collate import collate_device, union_graphs
    dev = th.device("cuda:0")
    H = int(os.environ.get("H", "128"))
    cfg = dict(bench.CFG, hid=H)
    shard = bench.make_shard(cfg, 0, dev)
    gs = {}
    for tag in ("p", "g"):
        s = shard[tag]
        gs[tag] = collate_device(s["local_src"], s["local_dst"], s["num_nodes"], s["num_edges"], s["N"], s["E"], ndata=s["ndata"],
                                 edata=s["edata"], max_nodes=s["max_n"], max_edges=s["max_e"])
    u = union_graphs(gs["p"], gs["g"])
    ix = u.index()
    N, E = u.number_of_nodes(), u.number_of_edges()
    ms = [th.randn(E, H, device=dev) for _ in range(4)]
    out = th.empty(N, 3 * H, device=dev)
    inc = ix.incidence()
    ix.endpoint_select()
    algo = 4 * H * E + 8 * H * N + 8 * E + 8 * (N + 1)
    res = {"H": H, "N": N, "E": E, "algorithmic_bytes": algo}

    def timeit(fn, reps=40):
        for i in range(8):
            fn(ms[i % 4])
        th.cuda.synchronize()
        ev = [(th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for i, (a, b) in enumerate(ev):
            a.record(); fn(ms[i % 4]); b.record()
        th.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        return {"min_us": round(t[0], 2), "median_us": round(t[len(t) // 2], 2), "gbps_median": round(algo / t[len(t) // 2] / 1e3, 1),
                "frac_of_8TBps": round(algo / t[len(t) // 2] / 1e3 / 8000, 4)}

    res["incidence_csr"] = timeit(lambda m: ops.seg_sum_raw(m, inc[0], inc[1], N, None, True, 1.0, -1.0, rows_shared=2, out=out[:, H:]))
    res["one_pass"] = timeit(lambda m: ops.endpoint_sums(m, ix, out=out[:, H:]))
    a = ops.endpoint_sums(ms[0], ix)
    b = ops.seg_sum_raw(ms[0], inc[0], inc[1], N, None, True, 1.0, -1.0, rows_shared=2)
    res["bit_identical"] = bool(th.equal(a, b))
    print(json.dumps(res))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The one-pass endpoint sums beside OTHER kernels on the same CUs: stream A runs dmp_seg_sum2_graphs, stream B a
stream of unrelated kernels (row copies, a GEMM, the incidence segment sum); every result of stream A and of stream B
is compared with its stand-alone value.  (Inside one stream a kernel never shares a CU with another kernel; two
processes on one GPU -- and overlapped streams -- do.)"""
import os, sys
import numpy as np
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualmessagepassing_amd import ops
from dualmessagepassing_amd.collate import collate_device, union_graphs

dev = th.device("cuda:0")
cfg = dict(bench.CFG)
shard = bench.make_shard(cfg, 0, dev)
gs = {}
for tag in ("p", "g"):
    s = shard[tag]
    gs[tag] = collate_device(s["local_src"], s["local_dst"], s["num_nodes"], s["num_edges"], s["N"], s["E"], ndata=s["ndata"],
                             edata=s["edata"], max_nodes=s["max_n"], max_edges=s["max_e"])
u = union_graphs(gs["p"], gs["g"])
ix = u.index()
N, E, H = u.number_of_nodes(), u.number_of_edges(), 128
m = th.randn(E, H, device=dev)
inc = ix.incidence()
ix.endpoint_select()
want = ops.seg_sum_raw(m, inc[0], inc[1], N, None, True, 1.0, -1.0, rows_shared=2)
a = th.randn(4096, 512, device=dev); b = th.randn(512, 512, device=dev)
want_mm = a @ b
x = th.randn(E, H, device=dev)
want_copy = x.clone()
sa, sb = th.cuda.Stream(), th.cuda.Stream()
th.cuda.synchronize()
bad = {"segacc": 0, "mm": 0, "copy": 0, "inc": 0}
iters = int(os.environ.get("ITERS", "300"))
for it in range(iters):
    with th.cuda.stream(sa):
        got = ops.endpoint_sums(m, ix)
        got2 = ops.endpoint_sums(m, ix)
    with th.cuda.stream(sb):
        mm = a @ b
        cp = x.clone()
        inc_sum = ops.seg_sum_raw(m, inc[0], inc[1], N, None, True, 1.0, -1.0, rows_shared=2)
        mm2 = a @ b
    th.cuda.synchronize()
    bad["segacc"] += int(not th.equal(got, want)) + int(not th.equal(got2, want))
    bad["mm"] += int(not th.equal(mm, want_mm)) + int(not th.equal(mm2, want_mm))
    bad["copy"] += int(not th.equal(cp, want_copy))
    bad["inc"] += int(not th.equal(inc_sum, want))
print("iterations", iters, "mismatches", bad)

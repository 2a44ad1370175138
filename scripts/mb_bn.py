"""Dev aid: BatchNorm1d(train)+LeakyReLU forward/backward, HIP op vs torch module, host and device time."""
import os, sys, time
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd import ops
gpu = th.device("cuda:0")
for rows in (2708, 10858):
    x = th.randn(rows, 256, device=gpu, requires_grad=True)
    dy = th.randn(rows, 256, device=gpu)
    bn = th.nn.BatchNorm1d(256).to(gpu)
    act = th.nn.LeakyReLU(1 / 5.5)
    def hip():
        y = ops.batch_norm_act(bn, x, 1 / 5.5)
        th.autograd.grad(y, [x, bn.weight, bn.bias], dy)
    def ref():
        y = act(bn(x))
        th.autograd.grad(y, [x, bn.weight, bn.bias], dy)
    for name, f in (("hip", hip), ("torch", ref)):
        for _ in range(5): f()
        th.cuda.synchronize()
        a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        t = time.perf_counter(); a.record()
        for _ in range(50): f()
        host = (time.perf_counter() - t) / 50 * 1e6
        b.record(); th.cuda.synchronize()
        print("rows %6d %-6s host %7.1f us/iter  device %7.1f us/iter" % (rows, name, host, a.elapsed_time(b) / 50 * 1e3), flush=True)

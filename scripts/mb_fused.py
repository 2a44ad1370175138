"""Dev aid: isolated timings of the fused MFMA entry points at the bench's union-graph size."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dualmessagepassing_amd import _lib
from dualmessagepassing_amd._lib import ptr, check
lib = _lib.load()
dev = torch.device("cuda:0")
E, N, H = 548864, 73728, 128
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cpu").manual_seed(0)
Z = [torch.randn(E, H, device=dev) for _ in range(3)]
O = [torch.empty(E, H, device=dev) for _ in range(3)]
O2 = torch.empty(E, 2 * H, device=dev)
W2 = torch.randn(H, 2 * H, device=dev) * 0.05
W1 = torch.randn(H, H, device=dev) * 0.05
P = torch.randn(N, 2 * H, device=dev)
D = torch.randn(N, 2 * H, device=dev)
coef = torch.rand(N, device=dev)
src = torch.randint(0, N, (E,), device=dev, dtype=torch.int32)
dst = torch.randint(0, N, (E,), device=dev, dtype=torch.int32)
flag = (torch.rand(E, device=dev) < 0.5).to(torch.uint8)
bias = torch.randn(H, device=dev)
gate = torch.rand(E, device=dev)
part = torch.empty(int(lib.dmp_mfma_partial_rows(E)) * 2, H, device=dev)


def timeit(f, n=20):
    for i in range(3):
        f(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        f(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def report(name, t, flops, nbytes):
    print("%-28s %7.1f us  %6.1f TF/s  %6.2f TB/s" % (name, t, flops / t / 1e6, nbytes / t / 1e6), flush=True)


import ctypes
dev_lib = ctypes.CDLL(lib._name)
keep = {}
for variant in (0, 1):
    dev_lib.dmp_dev_set_mfma_variant(variant)
    part = torch.empty(int(lib.dmp_mfma_partial_rows(E)), H, device=dev)
    print("---- variant", variant, "(0 = independent workgroups, 1 = ping-pong)")
    rows = 4 * H * E
    t = timeit(lambda i: check(lib.dmp_gemm_k128(ptr(Z[i % 3]), H, ptr(W1), H, 0, ptr(O[i % 3]), H, E, 128, st), "g"))
    report("gemm_k128 N=128", t, 2 * E * H * H, 2 * rows)
    t = timeit(lambda i: check(lib.dmp_gemm_k128(ptr(Z[i % 3]), H, ptr(W2), 2 * H, 0, ptr(O2), 2 * H, E, 256, st), "g"))
    report("gemm_k128 N=256", t, 4 * E * H * H, 3 * rows)
    t = timeit(lambda i: check(lib.dmp_edge_fwd_fused(ptr(Z[i % 3]), H, ptr(W2), 2 * H, ptr(P), 2 * H, ptr(coef), ptr(bias), ptr(src),
                                                      ptr(dst), ptr(flag), E, H, ptr(O[i % 3]), H, st), "e"))
    report("edge_fwd (NC=2, EDGE)", t, 4 * E * H * H, 2 * rows)
    t = timeit(lambda i: check(lib.dmp_out_fwd_fused(ptr(Z[i % 3]), H, ptr(W1), H, ptr(bias), ptr(gate), ptr(Z[(i + 1) % 3]), H, E, H,
                                                     ptr(O[i % 3]), H, st), "o"))
    report("out_fwd (NC=1, GATE_RES)", t, 2 * E * H * H, 3 * rows)
    t = timeit(lambda i: check(lib.dmp_bwd_h1_fused(ptr(Z[i % 3]), H, ptr(W1), H, ptr(Z[(i + 1) % 3]), H, ptr(coef), ptr(dst), E, H,
                                                    ptr(O2), 2 * H, ptr(part), st), "h"))
    report("bwd_h1 (NC=1, RELU_BWD_G)", t, 2 * E * H * H, 4 * rows)
    t = timeit(lambda i: check(lib.dmp_bwd_z_fused(ptr(Z[i % 3]), H, ptr(W2), 2 * H, ptr(D), 2 * H, ptr(Z[(i + 1) % 3]), H, ptr(coef),
                                                   ptr(dst), ptr(flag), -1.0, 1.0, E, H, ptr(O[i % 3]), H, st), "z"))
    report("bwd_z (NC=2, DZ)", t, 4 * E * H * H, 3 * rows)
    # outputs of the last call of each kernel, for cross-checking the variants
    outs = {}
    check(lib.dmp_gemm_k128(ptr(Z[0]), H, ptr(W2), 2 * H, 0, ptr(O2), 2 * H, E, 256, st), "g"); outs["gemm256"] = O2.clone()
    check(lib.dmp_edge_fwd_fused(ptr(Z[0]), H, ptr(W2), 2 * H, ptr(P), 2 * H, ptr(coef), ptr(bias), ptr(src), ptr(dst), ptr(flag), E, H, ptr(O[0]), H, st), "e"); outs["edge"] = O[0].clone()
    check(lib.dmp_out_fwd_fused(ptr(Z[0]), H, ptr(W1), H, ptr(bias), ptr(gate), ptr(Z[1]), H, E, H, ptr(O[0]), H, st), "o"); outs["out"] = O[0].clone()
    check(lib.dmp_bwd_h1_fused(ptr(Z[0]), H, ptr(W1), H, ptr(Z[1]), H, ptr(coef), ptr(dst), E, H, ptr(O2), 2 * H, ptr(part), st), "h"); outs["h1"] = O2.clone(); outs["h1cs"] = part.sum(0)
    check(lib.dmp_bwd_z_fused(ptr(Z[0]), H, ptr(W2), 2 * H, ptr(D), 2 * H, ptr(Z[1]), H, ptr(coef), ptr(dst), ptr(flag), -1.0, 1.0, E, H, ptr(O[0]), H, st), "z"); outs["dz"] = O[0].clone()
    torch.cuda.synchronize()
    if keep:
        for k in outs:
            print("   max |v1 - v0| %-8s %.3e" % (k, float((outs[k] - keep[k]).abs().max())))
    keep = outs
for n_out in (128, 256):
    B = torch.randn(H, n_out, device=dev)
    C = torch.empty(E, n_out, device=dev)
    t = timeit(lambda i: torch.mm(Z[i % 3], B, out=C))
    report("hipBLASLt default N=%d" % n_out, t, 2 * E * H * n_out, rows + 4 * E * n_out)

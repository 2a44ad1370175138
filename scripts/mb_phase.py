"""Dev aid: per-phase cycle counts of the ping-pong kernel (debug build scripts/_dbg/libmfma_80.so)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
E, N_, H = 548864, 73728, 128
st = torch.cuda.current_stream().cuda_stream
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "_dbg", "libmfma_80.so"))
P, I64, I, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
lib.dmp_gemm_k128.argtypes = [P, I64, P, I64, I, P, I64, I64, I, P]
lib.dmp_edge_fwd_fused.argtypes = [P, I64, P, I64, P, I64, I64, P, P, P, P, I64, I, P, I64, P]
lib.dmp_bwd_h1_fused.argtypes = [P, I64, P, I64, P, I64, P, P, I64, I, P, I64, P, P]
lib.dmp_out_fwd_fused.argtypes = [P, I64, P, I64, P, P, P, I64, I64, I, I, P, I64, P]
lib.dmp_bwd_z_fused.argtypes = [P, I64, P, I64, P, I64, I64, P, I64, P, P, P, F, F, I64, I, P, I64, P]
lib.dmp_dev_read_dbg.argtypes = [P]
Z = [torch.randn(E, H, device=dev) for _ in range(2)]
O = torch.empty(E, H, device=dev); O2 = torch.empty(E, 2 * H, device=dev)
W1 = torch.randn(H, H, device=dev); W2 = torch.randn(H, 2 * H, device=dev)
Pn = torch.randn(N_, 2 * H, device=dev); coef = torch.rand(N_, device=dev); bias = torch.randn(H, device=dev)
src = torch.randint(0, N_, (E,), device=dev, dtype=torch.int32); dst = torch.randint(0, N_, (E,), device=dev, dtype=torch.int32)
flag = (torch.rand(E, device=dev) < 0.5).to(torch.uint8); gate = torch.rand(E, device=dev)
coefE = torch.rand(E, device=dev); part = torch.empty(1024, H, device=dev)
d = lambda t: t.data_ptr()
calls = {
    "gemm N=128": lambda: lib.dmp_gemm_k128(d(Z[0]), H, d(W1), H, 0, d(O), H, E, 128, st),
    "gemm N=256": lambda: lib.dmp_gemm_k128(d(Z[0]), H, d(W2), 2 * H, 0, d(O2), 2 * H, E, 256, st),
    "edge_fwd": lambda: lib.dmp_edge_fwd_fused(d(Z[0]), H, d(W2), 2 * H, d(Pn), 2 * H, N_, d(bias), d(src), d(dst), d(coefE), E, H, 0.0, d(O), H, st),
    "bwd_h1": lambda: lib.dmp_bwd_h1_fused(d(Z[0]), H, d(W1), H, d(Z[1]), H, d(coefE), None, E, H, 0.0, d(O2), 2 * H, d(part), st),
    "out_fwd": lambda: lib.dmp_out_fwd_fused(d(Z[0]), H, d(W1), H, d(bias), d(gate), d(Z[1]), H, E, H, 0, d(O), H, st),
    "bwd_z": lambda: lib.dmp_bwd_z_fused(d(Z[0]), H, d(W2), 2 * H, d(Pn), 2 * H, N_, d(Z[1]), H, d(coefE), d(dst), d(flag), -1.0, 1.0, E, H, d(O), H, st),
}
for name, f in calls.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    buf = np.zeros(64, np.int64)
    lib.dmp_dev_read_dbg(buf.ctypes.data)
    buf = buf.reshape(8, 8)
    steps = buf[0, 3]
    print("%-11s %6.1f us  steps %d | per step cycles (wave0 / wave4): compute %5.0f / %5.0f  mem %5.0f / %5.0f (stage %4.0f / %4.0f)  barrier wait %5.0f / %5.0f"
          % (name, a.elapsed_time(b) * 1e3, steps, *(2 * buf[w, k] / steps for k in (0, 1) for w in (0, 4)),
             *(2 * buf[w, 4] / steps for w in (0, 4)), *(buf[w, 2] / steps for w in (0, 4))), flush=True)
    print("            epilogue: scratch writes done %5.0f / %5.0f, all stores issued %5.0f / %5.0f" % (
        *(2 * buf[w, 5] / steps for w in (0, 4)), *(2 * buf[w, 6] / steps for w in (0, 4))), flush=True)

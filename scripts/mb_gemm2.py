"""Dev aid: the E-row GEMM shapes of the fused layer under different BLAS back ends / TunableOp."""
import os, sys, time
import torch
dev = torch.device("cuda:0")
E, H = 548864, 128

def bench(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

x = torch.randn(E, H, device=dev); x2 = torch.randn(E, 2 * H, device=dev)
w11 = torch.randn(H, H, device=dev); w12 = torch.randn(H, 2 * H, device=dev); b = torch.randn(H, device=dev)
out1 = torch.empty(E, H, device=dev)
cases = [
    ("x@W   [E,128]x[128,128]", lambda: x @ w11, 2 * E * H * H),
    ("x@W^T [E,128]x[128,128]^T", lambda: x @ w11.t(), 2 * E * H * H),
    ("addmm(b, x, W^T) 128", lambda: torch.addmm(b, x, w11.t()), 2 * E * H * H),
    ("x@W   [E,128]x[128,256]", lambda: x @ w12, 2 * E * H * 2 * H),
    ("x2@W^T [E,256]x[256,128]", lambda: x2 @ w12.t(), 2 * E * H * 2 * H),
    ("addmm_ beta=1 [E,256]x[256,128]", lambda: out1.addmm_(x2, w12.t()), 2 * E * H * 2 * H),
    ("pad-N trick: x@[W|W] then slice", lambda: (x @ torch.cat([w11, w11], 1))[:, :H], 2 * E * H * H),
]
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
if tag == "rocblas":
    torch.backends.cuda.preferred_blas_library("cublas")
elif tag == "hipblaslt":
    torch.backends.cuda.preferred_blas_library("cublaslt")
print("== backend:", tag, torch.backends.cuda.preferred_blas_library(), "tunable:", os.environ.get("PYTORCH_TUNABLEOP_ENABLED"))
for name, f, fl in cases:
    t = bench(f)
    print("%-36s %8.1f us %6.1f TF" % (name, t * 1e6, fl / t / 1e12), flush=True)

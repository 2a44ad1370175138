"""Dev aid: experimental persistent fp32 MFMA GEMM (K=128) vs torch (hipBLASLt, tuned + default)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dualmessagepassing_amd import _lib
from dualmessagepassing_amd.tuning import enable_tuned_gemms
lib = _lib.load()
X = ctypes.CDLL(lib._name)
P, I64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
X.dmpx_gemm_k128.argtypes = [P, I64, P, I64, P, I64, I64, I, I, I, P]
dev = torch.device("cuda:0")
E, H = 548864, 128
st = torch.cuda.current_stream().cuda_stream
A = [torch.randn(E, H, device=dev) for _ in range(3)]

def timeit(f, n=20):
    for i in range(3): f(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): f(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for N in (128, 256):
    B = torch.randn(H, N, device=dev)
    C = torch.empty(E, N, device=dev)
    ref = A[0] @ B
    for blocks in (256, 512):
      for wide in (0, 1):
        C.zero_()
        rc = X.dmpx_gemm_k128(A[0].data_ptr(), H, B.data_ptr(), N, C.data_ptr(), N, E, N, blocks, wide, st)
        torch.cuda.synchronize()
        err = float((C - ref).abs().max())
        t = timeit(lambda i: X.dmpx_gemm_k128(A[i % 3].data_ptr(), H, B.data_ptr(), N, C.data_ptr(), N, E, N, blocks, wide, st))
        print("mfma N=%d blocks=%d wide=%d rc=%d maxerr=%.2e  %7.1f us  %6.1f TF" % (N, blocks, wide, rc, err, t, 2 * E * H * N / t / 1e6))
    t = timeit(lambda i: torch.mm(A[i % 3], B, out=C))
    print("torch default N=%d %7.1f us %6.1f TF" % (N, t, 2 * E * H * N / t / 1e6))
enable_tuned_gemms()
for N in (128, 256):
    B = torch.randn(H, N, device=dev); C = torch.empty(E, N, device=dev)
    t = timeit(lambda i: torch.mm(A[i % 3], B, out=C))
    print("torch tuned   N=%d %7.1f us %6.1f TF" % (N, t, 2 * E * H * N / t / 1e6))

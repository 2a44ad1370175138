"""Dev aid: time dmp_gemm_k128 of debug builds (scripts/_dbg/libmfma_<knobs>.so; knobs: 1 no stores,
2 no LDS transpose, 4 no row prefetch, 8 no staging writes) to see what bounds the memory phase."""
import ctypes, glob, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = torch.device("cuda:0")
E, H = 548864, 128
st = torch.cuda.current_stream().cuda_stream
Z = [torch.randn(E, H, device=dev) for _ in range(3)]
P, I64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int

def timeit(f, n=20):
    for i in range(3): f(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): f(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for path in sorted(glob.glob(os.path.join(ROOT, "scripts", "_dbg", "libmfma_*.so")), key=lambda s: int(s.split("_")[-1][:-3])):
    lib = ctypes.CDLL(path)
    lib.dmp_gemm_k128.argtypes = [P, I64, P, I64, I, P, I64, I64, I, P]
    for variant in (1,):
        out = []
        for N in (128, 256):
            W = torch.randn(H, N, device=dev); C = torch.empty(E, N, device=dev)
            t = timeit(lambda i: lib.dmp_gemm_k128(Z[i % 3].data_ptr(), H, W.data_ptr(), N, 0, C.data_ptr(), N, E, N, st))
            out.append("N=%d %6.1f us" % (N, t))
        print("%-18s variant %d  %s" % (os.path.basename(path), variant, "   ".join(out)), flush=True)

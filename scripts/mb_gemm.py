"""Micro-benchmark (development aid, not product): fp32 GEMM shapes of the DMPLayer on MI355X."""
import torch, time
dev = torch.device("cuda:0")
E, N, H = 524288, 65536, 128

def bench(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n

x = torch.randn(E, H, device=dev); dy1 = torch.randn(E, H, device=dev); dy2 = torch.randn(E, 2 * H, device=dev)
w1 = torch.randn(H, H, device=dev); w2 = torch.randn(H, 2 * H, device=dev)
for name, f, fl in [
    ("fwd  [E,H]x[H,H]", lambda: x @ w1, 2 * E * H * H),
    ("fwd  [E,H]x[H,2H]", lambda: x @ w2, 2 * E * H * 2 * H),
    ("dX   [E,2H]x[2H,H]", lambda: dy2 @ w2.t(), 2 * E * H * 2 * H),
    ("dW   x^T dy (H)", lambda: x.t() @ dy1, 2 * E * H * H),
    ("dW   x^T dy (2H)", lambda: x.t() @ dy2, 2 * E * H * 2 * H),
]:
    t = bench(f); print("%-24s %8.1f us  %6.1f TF" % (name, t * 1e6, fl / t / 1e12))
for S in (16, 32, 64, 128, 256, 512, 1024):
    for dy, tag in ((dy1, "H"), (dy2, "2H")):
        n1 = dy.size(1)
        f = lambda: torch.bmm(x.view(S, E // S, H).transpose(1, 2), dy.view(S, E // S, n1)).sum(0)
        t = bench(f); print("dW bmm split S=%-5d %-3s %8.1f us  %6.1f TF" % (S, tag, t * 1e6, 2 * E * H * n1 / t / 1e12))
# addmm activation epilogue
b = torch.randn(H, device=dev)
try:
    f = lambda: torch._addmm_activation(b, x, w1, use_gelu=False)
    t = bench(f); print("addmm+relu epilogue %8.1f us" % (t * 1e6))
    ref = torch.relu(torch.addmm(b, x, w1)); print("  max diff", float((f() - ref).abs().max()))
except Exception as e:
    print("_addmm_activation failed:", e)
t = bench(lambda: torch.relu(torch.addmm(b, x, w1))); print("addmm then relu     %8.1f us" % (t * 1e6))
t = bench(lambda: dy1.sum(0)); print("colsum dy.sum(0)    %8.1f us" % (t * 1e6))
t = bench(lambda: dy1.view(512, E // 512, H).sum(1).sum(0)); print("colsum 2-stage      %8.1f us" % (t * 1e6))
ones = torch.ones(1, E, device=dev)
t = bench(lambda: ones @ dy1); print("colsum ones@dy      %8.1f us" % (t * 1e6))

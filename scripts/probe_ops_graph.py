"""Dev probe: HIP-graph recording of single differentiable ops (each in its own process)."""
import os, subprocess, sys
VARIANTS = ["take_rows", "take_rows_small", "nn_embedding", "seg_pool_keys", "csr_build_only", "take_rows_fwd"]
if len(sys.argv) == 1:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), v], capture_output=True, text=True, timeout=300)
        tail = [l for l in (r.stdout + r.stderr).splitlines() if "amdgpu" not in l][-4:]
        print("== %-16s rc=%d  %s" % (v, r.returncode, " | ".join(t[:140] for t in tail)), flush=True)
    sys.exit(0)
variant = sys.argv[1]
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualmessagepassing_amd import ops, _lib
gpu = th.device("cuda:0")
gen = th.Generator().manual_seed(0)
W = th.randn(2708, 256, generator=gen).to(gpu).requires_grad_(True)
W2 = th.randn(2, 256, generator=gen).to(gpu).requires_grad_(True)
idx = th.randint(0, 2708, (21716,), generator=gen).to(gpu)
idx2 = th.randint(0, 2, (10858,), generator=gen).to(gpu)
emb = th.nn.Embedding(2708, 256).to(gpu)


def step():
    if variant == "take_rows":
        W.grad = None
        ops.take_rows(W, idx).square().sum().backward()
        return W.grad.sum()
    if variant == "take_rows_fwd":
        with th.no_grad():
            return ops.take_rows(W, idx).sum()
    if variant == "take_rows_small":
        W2.grad = None
        ops.take_rows_small_table(W2, idx2).square().sum().backward()
        return W2.grad.sum()
    if variant == "nn_embedding":
        emb.weight.grad = None
        emb(idx).square().sum().backward()
        return emb.weight.grad.sum()
    if variant == "seg_pool_keys":
        p = ops.PoolIndex.from_keys(idx2, 2)
        return p.vptr.sum()
    if variant == "csr_build_only":
        lib = _lib.load()
        i32 = dict(dtype=th.int32, device=gpu)
        M, N = idx.numel(), 2708
        rowptr, ent, idx32 = th.empty(N + 1, **i32), th.empty(M, **i32), th.empty(M, **i32)
        deg, status = th.empty(N, dtype=th.int64, device=gpu), th.empty(1, **i32)
        ws = th.empty(lib.dmp_csr_workspace_words(N, M), **i32)
        _lib.check(lib.dmp_csr_build(_lib.ptr(idx), None, M, N, _lib.ptr(rowptr), _lib.ptr(ent), _lib.ptr(idx32), _lib.ptr(deg),
                                     _lib.ptr(status), _lib.ptr(ws), _lib.stream_ptr()), "csr")
        return rowptr.sum()


for _ in range(3): step()
th.cuda.synchronize()
gr = th.cuda.CUDAGraph()
with th.cuda.graph(gr):
    out = step()
th.cuda.synchronize()
gr.replay(); gr.replay()
th.cuda.synchronize()
print("recorded and replayed:", float(out))

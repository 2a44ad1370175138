"""Dev aid: the incidence-CSR scatter-add at bench.py's shape with the resident workgroups per CU capped by an LDS pad
(builds csrc/dmp_agg.hip variants with -DDMP_SEG_OCC_LDS=<bytes>): does a smaller in-flight working set raise the L2 hit
rate of the rows' second read enough to pay for the lost latency hiding?"""
import ctypes, os, subprocess, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_graphs import er_batch
from dualmessagepassing_amd import _lib, ops
from dualmessagepassing_amd.graph import GraphIndex
dev = torch.device("cuda:0")
B = 1024
rng = np.random.default_rng(2000)
ps, pd, pr, pn, _, _ = er_batch(B, 8, 12, rng)
gs, gd, gr, gn, _, _ = er_batch(B, 64, 256, rng)
src = np.concatenate([ps, gs + pn]); dst = np.concatenate([pd, gd + pn]); rev = np.concatenate([pr, gr])
n, e, h = pn + gn, len(src), 128
ix = GraphIndex(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n, torch.from_numpy(rev).to(dev))
inc_ptr, inc_ent = ix.incidence()
zs = [torch.randn(e, h, device=dev) for _ in range(6)]
out = torch.empty(n, 2 * h, device=dev)
csrc = os.path.join(ROOT, "dualmessagepassing_amd", "csrc")
os.makedirs(os.path.join(ROOT, "scripts", "_dbg"), exist_ok=True)
for pad in (0, 16384, 24576, 36864, 49152, 65536):
    so = os.path.join(ROOT, "scripts", "_dbg", "libagg_%d.so" % pad)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", os.path.join(csrc, "dmp_agg.hip"), "-o", so]
    if pad:
        cmd.insert(1, "-DDMP_SEG_OCC_LDS=%d" % pad)
    subprocess.run(cmd, check=True)
    lib = ctypes.CDLL(so)
    f = lib.dmp_seg_sum2
    f.restype = ctypes.c_int
    f.argtypes = _lib.SIGNATURES["dmp_seg_sum2"][1]
    st = torch.cuda.current_stream().cuda_stream
    def run(i):
        rc = f(zs[i % 6].data_ptr(), h, inc_ptr.data_ptr(), inc_ent.data_ptr(), None, n, h, 1.0, -1.0, out.data_ptr(), 2 * h, 1, st)
        assert rc == 0, rc
    for i in range(5): run(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(30): run(i)
    b.record(); torch.cuda.synchronize()
    print("LDS pad %6d B: %.1f us per launch" % (pad, a.elapsed_time(b) / 30 * 1e3))

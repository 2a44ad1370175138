"""Dev aid: the incidence-CSR scatter-add at bench.py's shape under the DMP_SEG_VAR knobs of csrc/dmp_agg.hip (built here with
hipcc): non-temporal stores, wider workgroups (a graph's nodes side by side on one CU)."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_graphs import er_batch
from dualmessagepassing_amd import _lib
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
ref = None
for var in [int(v) for v in (sys.argv[1:] or ["0", "1", "2", "3", "4", "8", "9"])]:
    so = os.path.join(ROOT, "scripts", "_dbg", "libaggv_%d.so" % var)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-DDMP_SEG_VAR=%d" % var, "-shared",
                    os.path.join(csrc, "dmp_agg.hip"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    f = lib.dmp_seg_sum2
    f.restype = ctypes.c_int
    f.argtypes = _lib.SIGNATURES["dmp_seg_sum2"][1]
    st = torch.cuda.current_stream().cuda_stream
    def run(i):
        rc = f(zs[i % 6].data_ptr(), h, inc_ptr.data_ptr(), inc_ent.data_ptr(), None, n, h, 1.0, -1.0, out.data_ptr(), 2 * h, 2, st)
        assert rc == 0, rc
    for i in range(6): run(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(30): run(i)
    b.record(); torch.cuda.synchronize()
    run(0); torch.cuda.synchronize()
    if ref is None: ref = out.clone()
    print("DMP_SEG_VAR %2d: %.1f us per launch  (bits equal: %s)" % (var, a.elapsed_time(b) / 30 * 1e3, torch.equal(out, ref)), flush=True)

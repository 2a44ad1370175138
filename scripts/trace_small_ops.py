"""Dev aid: which Python lines issue the small aten ops of a bench step (copy_/clone/cat/fill_/zeros/add/...)."""
import os, sys, traceback, collections, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.utils._python_dispatch import TorchDispatchMode
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = dict(bench.CFG)
shard = bench.make_shard(cfg, 0, dev)
step, model = bench.build_step(cfg, shard, dev)
for _ in range(3): step()
torch.cuda.synchronize()
PAT = re.compile(os.environ.get("OPS", r"copy_|clone|cat|fill_|zero_|zeros|ones|full|add|sub|mul|div|sum|mean|index|gather|arange|_to_copy|contiguous"))
hits = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if PAT.search(name):
            shapes = [tuple(a.shape) for a in args if torch.is_tensor(a)][:2]
            st = [f for f in traceback.extract_stack() if "/root/repo" in f.filename or "dualmessagepassing" in f.filename or "bench.py" in f.filename]
            st = [f for f in st if "trace_small_ops" not in f.filename][-2:]
            where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st))
            hits[(name, str(shapes), where)] += 1
        return func(*args, **(kwargs or {}))
with Mode():
    step()
torch.cuda.synchronize()
for (name, shapes, where), c in sorted(hits.items(), key=lambda kv: kv[0][2]):
    print("%2d  %-28s %-44s %s" % (c, name[:28], shapes[:44], where))

"""Where the torch-side launches of a bench.py step come from: one eager step under a TorchDispatchMode that charges
every aten call that launches something (fills, copies, concatenations, elementwise ops, reductions, GEMMs) to the
innermost frame inside this repository."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
gpu = torch.device("cuda:0")
cfg = dict(bench.CFG, act="leaky_relu", emb="Equivariant", micro_batches=0, graph=False, gate_compact="--gate-compact" in sys.argv)
shard = bench.make_shard(cfg, 0, gpu)
step, model = bench.build_step(cfg, shard, gpu, 1)
for _ in range(4):
    step()
torch.cuda.synchronize()
NOLAUNCH = ("aten.view", "aten.empty", "aten.as_strided", "aten.slice", "aten.select", "aten.t.", "aten.transpose", "aten.detach",
            "aten.reshape", "aten._unsafe_view", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.alias", "aten.permute",
            "aten.split", "aten.unbind", "aten.narrow", "aten.lift_fresh", "aten.is_", "aten.size", "aten.stride", "aten._local_scalar",
            "aten.resize_", "aten.set_", "aten.new_empty", "aten.empty_like", "aten.empty_strided", "aten.unfold", "aten.chunk",
            "aten.record_stream", "aten._reshape_alias", "aten.view_as", "aten.contiguous")
rows = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(NOLAUNCH):
            where = "?"
            for f in reversed(traceback.extract_stack()):
                if ("/dualmessagepassing_amd/" in f.filename or f.filename.endswith("bench.py")) and "launch_sources" not in f.filename:
                    where = "%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name)
                    break
            shape = ""
            for a in args:
                if torch.is_tensor(a):
                    shape = "%s %s" % (tuple(a.shape), str(a.dtype).replace("torch.", ""))
                    break
            rows[(where, name, shape)] += 1
        return func(*args, **(kwargs or {}))


with Log():
    step()
torch.cuda.synchronize()
print("aten calls that may launch, one step:", sum(rows.values()))
for (where, name, shape), c in sorted(rows.items(), key=lambda kv: kv[0]):
    print("%2d x %-34s %-28s %s" % (c, name[:34], shape[:28], where))

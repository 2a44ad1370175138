"""Dev aid: aten / custom ops of a bench step by device time, grouped by input shapes (torch.profiler)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = dict(bench.CFG)
shard = bench.make_shard(cfg, 0, dev)
step, model = bench.build_step(cfg, shard, dev)
for _ in range(5): step()
torch.cuda.synchronize()
STEPS = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(STEPS): step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None: dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt / STEPS, e.count / STEPS, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
import re
flt = os.environ.get("FILTER")
if flt:
    rows = [r for r in rows if re.search(flt, r[2])]
for dt, c, k, s in rows[:int(os.environ.get("TOP", "90"))]:
    print("%8.1f us/step %5.1f/step  %-45s %s" % (dt, c, k[:45], s))

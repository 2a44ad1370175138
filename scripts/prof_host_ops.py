"""Dev aid: host-side cost of a bench step by op (torch.profiler CPU self time), and op / launch counts."""
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
STEPS = 10
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(STEPS): step()
    torch.cuda.synchronize()
rows = [(e.self_cpu_time_total / STEPS, e.cpu_time_total / STEPS, e.count / STEPS, e.key) for e in prof.key_averages()]
rows.sort(reverse=True)
print("ops/step %.0f   self cpu total %.0f us/step" % (sum(r[2] for r in rows), sum(r[0] for r in rows)))
for s, t, c, k in rows[:60]:
    print("%8.1f us self %8.1f us total %6.1f/step  %s" % (s, t, c, k[:70]))

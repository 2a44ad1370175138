"""Dev aid: ops of the UNC TrainModel step by device time and host self time (torch.profiler)."""
import os, sys, runpy, torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kbench_unc.py"))
from torch.profiler import profile, ProfilerActivity
step = ns["step"]
for _ in range(5): step()
th.cuda.synchronize()
STEPS = 10
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False) as prof:
    for _ in range(STEPS): step()
    th.cuda.synchronize()
rows = []
for e in prof.key_averages():
    dt = getattr(e, "self_device_time_total", None)
    if dt is None: dt = e.self_cuda_time_total
    rows.append((e.self_cpu_time_total / STEPS, dt / STEPS, e.count / STEPS, e.key))
print("ops/step %.0f  host self %.0f us/step  device %.0f us/step" % (sum(r[2] for r in rows), sum(r[0] for r in rows), sum(r[1] for r in rows)))
rows.sort(reverse=True)
for s, d, c, k in rows[:45]:
    print("%8.1f us host %8.1f us dev %6.1f/step  %s" % (s, d, c, k[:80]))

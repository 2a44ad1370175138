"""Dev aid: is the bench step host-bound?  Enqueue time of 20 steps vs their completion time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualmessagepassing_amd.tuning import enable_tuned_gemms
if not os.environ.get("NO_TUNE"): enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = dict(bench.CFG)
if os.environ.get("HID"): cfg["hid"] = int(os.environ["HID"])
shard = bench.make_shard(cfg, 0, dev)
if os.environ.get("GRAPH"): cfg["graph"] = True
step, model = bench.build_step(cfg, shard, dev)
if os.environ.get("GRAPH"):      # the step replayed from one HIP graph (dp.StepGraph), as bench.py --graph runs it
    from dualmessagepassing_amd.dp import StepGraph
    eager_step = step
    step = StepGraph(lambda: eager_step(), optimizer=eager_step.opt, max_shapes=1)
from dualmessagepassing_amd import basemodel
if os.environ.get("SIZE_CACHE_MAX"): basemodel._SIZE_CACHE_MAX = int(os.environ["SIZE_CACHE_MAX"])
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for _ in range(5): step()
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    t0 = time.perf_counter()
    per = []
    for _ in range(20):
        ts = time.perf_counter()
        step()
        per.append((time.perf_counter() - ts) * 1e3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    a1 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    print("enqueue %.2f ms/step, complete %.2f ms/step; device allocations in the timed loop: %d; reserved %.0f MB"
          % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3, a1 - a0, torch.cuda.memory_reserved() / 1e6))
    if max(per) > 2.5 * sorted(per)[10]:
        print("   per-step enqueue ms:", " ".join("%.1f" % x for x in per))

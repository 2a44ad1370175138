"""Dev aid: is the bench step host-bound?  Enqueue time of 20 steps vs their completion time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualmessagepassing_amd.tuning import enable_tuned_gemms
if not os.environ.get("NO_TUNE"): enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = dict(bench.CFG)
shard = bench.make_shard(cfg, 0, dev)
step, model = bench.build_step(cfg, shard, dev)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f ms/step, complete %.2f ms/step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))

"""Dev aid: where the HOST time of an eager bench step goes (cProfile over 30 steps, top entries by own time and by cumulative time)."""
import cProfile, io, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = dict(bench.CFG)
shard = bench.make_shard(cfg, 0, dev)
step, model = bench.build_step(cfg, shard, dev)
for _ in range(8): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(30): step()
pr.disable()
torch.cuda.synchronize()
for key, n in (("tottime", 45), ("cumulative", 60)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(n)
    print(s.getvalue()[:9000])

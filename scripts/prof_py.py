"""Dev aid: cProfile of the host side of bench steps (eager), sorted by own time."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = dict(bench.CFG, batch=int(os.environ.get('BATCH', bench.CFG['batch'])))   # BATCH=16: host-bound, no queue back-pressure in the numbers
shard = bench.make_shard(cfg, 0, dev)
step, model = bench.build_step(cfg, shard, dev)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):     # backward on this thread: the profile sees the custom functions
    for _ in range(3): step()
    pr.enable()
    for _ in range(20): step()
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(55)

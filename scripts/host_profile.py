"""Dev aid: where the host time of a bench step goes (cProfile over 30 steps, top functions by own time)."""
import cProfile, os, pstats, sys, torch, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = dict(bench.CFG)
if os.environ.get("HID"): cfg["hid"] = int(os.environ["HID"])
shard = bench.make_shard(cfg, 0, dev)
step, model = bench.build_step(cfg, shard, dev)
for _ in range(8): step()
torch.cuda.synchronize()
gc.freeze(); gc.set_threshold(200000, 20, 20)
pr = cProfile.Profile()
pr.enable()
for _ in range(30): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(40)

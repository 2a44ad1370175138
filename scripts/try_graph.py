"""Dev experiment: capture one whole bench step (collate + index builds + fwd + bwd + pack + AdamW) in a HIP graph
(torch.cuda.graph) and replay it.  Prints what fails, or eager vs replay time per step."""
import os, sys, time, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dualmessagepassing_amd.tuning import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device("cuda:0")
cfg = dict(bench.CFG)
if os.environ.get("HID"): cfg["hid"] = int(os.environ["HID"])
shard = bench.make_shard(cfg, 0, dev)
step, model = bench.build_step(cfg, shard, dev)
for _ in range(6): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("eager   %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = step()
    torch.cuda.synchronize()
    print("captured", flush=True)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print("replay  %.3f ms/step, loss %.6f" % ((time.perf_counter() - t0) / 20 * 1e3, float(loss)), flush=True)
except BaseException:
    traceback.print_exc()

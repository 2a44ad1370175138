import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
rank = int(os.environ.get("SHARD_RANK", "1"))
cfg = dict(bench.CFG, act="leaky_relu", emb="Equivariant", micro_batches=0, graph=False)
shard = bench.make_shard(cfg, rank, dev)
step, model = bench.build_step(cfg, shard, dev, 1)
for i in range(4):
    step()
    torch.cuda.synchronize()
    print("step", i, "ok", flush=True)

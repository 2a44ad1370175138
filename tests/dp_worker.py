"""Child process of tests/test_gpu_dp.py: one data-parallel rank of the PRODUCT model (full DMPNN, fused path) on
cuda:0 with the gloo backend -- what a rank of bench.py / harness.fit does for one step: its contiguous shard of the
global batch, forward, backward, gradient pack, flat all-reduce (average).  Rank 0 writes the reduced flat gradient, the
parameter vector after one AdamW step and the local loss to ``argv[1]``.  Started fresh (no GPU state is inherited)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build_batch(cfg, device, lo, hi):
    import bench
    return bench.slice_shard(cfg, bench.make_shard(cfg, 0, device), lo, hi)   # the GLOBAL batch (seeded); pairs [lo, hi)


def one_step(cfg, device, rank, world, overlap):
    import bench
    from dualmessagepassing_amd.basemodel import build_model
    from dualmessagepassing_amd.collate import collate_device
    from dualmessagepassing_amd.dmpnn import prepare_joint
    from dualmessagepassing_amd.dp import FlatAdamW, FlatGradSync, shard_range
    torch.manual_seed(0)
    model = build_model(**bench.model_config(cfg)).to(device)
    if rank != 0:                                                # de-synchronised on purpose: broadcast must repair it
        with torch.no_grad():
            for p in model.parameters():
                if p.requires_grad:                                  # the frozen encoding tables are not optimizer state
                    p.add_(0.5)
    sync = FlatGradSync(model)
    master = sync.flatten_parameters()
    sync.broadcast_parameters()
    opt = FlatAdamW([master], lr=1e-3, weight_decay=1e-5, amsgrad=True)
    lo, hi = shard_range(cfg["batch"], rank, world)
    b = build_batch(cfg, device, lo, hi)
    mk = lambda d: collate_device(d["local_src"], d["local_dst"], d["num_nodes"], d["num_edges"], d["N"], d["E"], ndata=d["ndata"],
                                  edata=d["edata"], max_nodes=d["max_n"], max_edges=d["max_e"])
    pattern, graph = mk(b["p"]), mk(b["g"])
    sync.detach_grads()
    out = model(pattern, graph)
    loss = torch.nn.functional.mse_loss(out["pred_c"].view(-1), b["counts"])
    loss.backward()
    sync.pack()
    if overlap:                                                  # the pipelined form of bench.py: structure work of the next batch in between
        work = sync.sync(async_op=True)
        prepare_joint(mk(b["p"]), mk(b["g"]), cfg["hid"])
        sync.finish(work)
    else:
        sync.sync()
    grad = sync.flat.clone()
    opt.step()
    return grad, master.data.clone(), loss.detach()


def main():
    out_path, overlap = sys.argv[1], sys.argv[2] == "1"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        cfg = dict(bench.CFG, batch=int(os.environ.get("DP_TEST_BATCH", "64")), act="leaky_relu", emb="Equivariant")
        grad, params, loss = one_step(cfg, device, rank, world, overlap)
        losses = [torch.zeros_like(loss) for _ in range(world)]
        dist.all_gather(losses, loss)
        if rank == 0:
            np.savez(out_path, grad=grad.cpu().numpy(), params=params.cpu().numpy(), losses=torch.stack(losses).cpu().numpy())
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

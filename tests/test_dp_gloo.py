"""Multi-process data-parallel path on CPU (gloo, world_size 2): the flat-buffer gradient
all-reduce reproduces the single-process global-batch gradient, parameters without a gradient
are handled, and shards partition the batch.  RCCL replaces gloo on the GPU box."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dualmessagepassing_amd.dp import FlatGradSync, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.a = torch.nn.Linear(6, 5)
        self.b = torch.nn.Linear(5, 1)
        self.unused = torch.nn.Linear(3, 3)   # never gets a gradient (like UNC nfc/efc)
        self.shared = self.a                  # aliased module, as with share_rep_net

    def forward(self, x):
        return self.b(torch.tanh(self.shared(x))).squeeze(-1)


def _data(n=24):
    g = torch.Generator().manual_seed(11)
    return torch.randn(n, 6, generator=g), torch.randn(n, generator=g)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = Tiny()
        if rank != 0:  # de-synchronise, then check broadcast_parameters restores rank 0's values
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        sync = FlatGradSync(model)
        sync.broadcast_parameters(src=0)
        x, y = _data()
        lo, hi = shard_range(x.size(0), rank, world)
        sync.zero()
        loss = torch.nn.functional.mse_loss(model(x[lo:hi]), y[lo:hi])  # mean over the local shard
        loss.backward()
        work = sync.sync(async_op=(rank % 2 == 0))
        sync.finish(work)
        out[rank] = sync.flat.clone()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_grad_allreduce_matches_global_batch():
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        flats = [out[r] for r in range(world)]
    assert torch.equal(flats[0], flats[1])
    # single-process reference on the whole batch (equal shards -> mean of means = global mean)
    model = Tiny()
    ref = FlatGradSync(model)
    x, y = _data()
    torch.nn.functional.mse_loss(model(x), y).backward()
    assert torch.allclose(flats[0], ref.flat, rtol=1e-5, atol=1e-6)
    # the never-used parameters stay exactly zero in the buffer
    n_unused = sum(p.numel() for p in model.unused.parameters())
    assert n_unused > 0 and int((ref.flat == 0).sum()) >= n_unused


def test_flat_buffer_views_and_dedup():
    model = Tiny()
    sync = FlatGradSync(model)
    uniq = list({id(p): p for p in model.parameters()}.values())
    # the aliased module is counted once; every slice starts on a 16-byte boundary (sizes rounded up to 4 elements)
    assert sync.flat.numel() == sum((p.numel() + 3) // 4 * 4 for p in uniq)
    assert all(off % 4 == 0 for off in sync.offsets) and len(sync.offsets) == len(uniq)
    model(torch.randn(4, 6)).sum().backward()
    for p in sync.params:
        assert p.grad.data_ptr() >= sync.flat.data_ptr()  # grads are views into the flat buffer
    before = sync.flat.clone()
    sync.sync()                            # world size 1: no-op
    assert torch.equal(before, sync.flat)
    sync.zero()
    assert float(sync.flat.abs().sum()) == 0.0 and all(float(p.grad.abs().sum()) == 0.0 for p in sync.params)


@pytest.mark.parametrize("n,world", [(1024, 8), (10, 3), (2, 4), (0, 2)])
def test_shard_range_partitions(n, world):
    spans = [shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1


def test_pack_equals_accumulate_into_views():
    """detach_grads()/pack() (fresh gradients, one multi-tensor copy) fills the flat buffer exactly
    like accumulating into the views, and leaves zeros for parameters without a gradient."""
    x, y = _data()
    m1, m2 = Tiny(), Tiny()
    s1, s2 = FlatGradSync(m1), FlatGradSync(m2)
    s1.zero()
    torch.nn.functional.mse_loss(m1(x), y).backward()
    s2.detach_grads()
    torch.nn.functional.mse_loss(m2(x), y).backward()
    assert all(p.grad is None for p in m2.unused.parameters())
    s2.pack()
    assert torch.equal(s1.flat, s2.flat)
    for p in s2.params:  # .grad is a view of the flat buffer again (what the optimizer / all-reduce use)
        assert p.grad is not None and p.grad.data_ptr() >= s2.flat.data_ptr()
    s2.flat.mul_(2.0)
    assert torch.equal(s2.params[0].grad, s1.params[0].grad * 2.0)


def test_flattened_parameters_step_like_per_parameter_adamw():
    """flatten_parameters(): AdamW on the one flat parameter performs exactly the per-parameter updates
    (elementwise optimizer, one parameter group), and the module keeps computing with the views."""
    x, y = _data()
    m1, m2 = Tiny(), Tiny()
    s1, s2 = FlatGradSync(m1), FlatGradSync(m2)
    master = s2.flatten_parameters()
    assert s2.flatten_parameters() is master
    o1 = torch.optim.AdamW(s1.params, lr=1e-2, weight_decay=1e-2)
    o2 = torch.optim.AdamW([master], lr=1e-2, weight_decay=1e-2)
    for _ in range(3):
        for s, m, o in ((s1, m1, o1), (s2, m2, o2)):
            s.detach_grads()
            torch.nn.functional.mse_loss(m(x), y).backward()
            s.pack()
            o.step()
    for p, q in zip(s1.params, s2.params):
        assert torch.equal(p.data, q.data)
        assert q.data.data_ptr() >= master.data.data_ptr()
    assert master.grad is s2.flat


def _one_rank_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        x, y = _data()
        res = []
        for force in (False, True):
            model = Tiny()
            sync = FlatGradSync(model, force_collective=force)
            sync.broadcast_parameters(src=0)
            sync.zero()
            torch.nn.functional.mse_loss(model(x), y).backward()
            work = sync.sync(async_op=True)
            res.append((work is not None, sync.flat.clone()))
            sync.finish(work)
            res[-1] = res[-1] + (sync.flat.clone(),)
        out[0] = res
    finally:
        dist.destroy_process_group()


def test_forced_collective_on_a_one_rank_group_is_the_identity():
    """``FlatGradSync(force_collective=True)`` on a ONE-rank group (what ``bench.py --force-collective`` does over RCCL on a
    single GPU): the all-reduce is issued (a work handle comes back), ``finish`` waits and averages over one rank, and the
    flat gradient is what the collective-free object holds."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_one_rank_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    (issued0, before0, after0), (issued1, before1, after1) = out[0]
    assert issued0 is False and issued1 is True
    assert torch.equal(before0, after0) and torch.equal(before1, after1) and torch.equal(after0, after1)


# ---- round 6: the live-parameter set is the UNION over the ranks; unequal shards at world 4; BatchNorm warning; the drain

class TwoBranch(torch.nn.Module):
    """``rev`` only takes part when the batch has a flagged row -- like ``out_weight`` / the reversed-edge parameters of a
    DMPLayer on a batch without reversed edges (dmpnn.py:111-127)."""
    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.fwd = torch.nn.Linear(6, 1)
        self.rev = torch.nn.Linear(6, 1)
        self.never = torch.nn.Linear(2, 2)

    def forward(self, x, flag):
        out = self.fwd(x[~flag]).sum()
        if bool(flag.any()):
            out = out + self.rev(x[flag]).sum()
        return out


def _live_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = TwoBranch()
        sync = FlatGradSync(model)
        master = sync.flatten_parameters()
        x, _ = _data(8)
        flag = torch.zeros(8, dtype=torch.bool)
        if rank == 1:
            flag[::2] = True                     # only rank 1's batch has "reversed" rows
        sync.detach_grads()
        model(x, flag).backward()
        local_live = [p.grad is not None for p in sync.params]
        sync.pack()
        before = sync.live.clone()
        sync.sync()
        out[rank] = dict(local=local_live, before=before, after=sync.live.clone(), flat=sync.flat.clone(),
                         dev_is_tail=master._dmp_live_dev is sync.live, host=master._dmp_live_params)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_live_parameter_set_is_the_union_over_ranks():
    """a tensor dead on this rank's batch but live on another's must be stepped HERE too (it holds a non-zero
    average after the all-reduce).  Every rank's 0 / 1 indicators ride in the tail of the all-reduce's payload; afterwards all
    ranks see the same set, and the optimizer is pointed at that device array instead of the host-side (rank-local) set."""
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_live_worker, args=(world, port, out), nprocs=world, join=True)
        r0, r1 = out[0], out[1]
    names = [n for n, _ in TwoBranch().named_parameters()]
    rev = [i for i, n in enumerate(names) if n.startswith("rev.")]
    never = [i for i, n in enumerate(names) if n.startswith("never.")]
    assert not any(r0["local"][i] for i in rev) and all(r1["local"][i] for i in rev)        # the ranks disagree locally
    assert all(float(r0["before"][i]) == 0.0 for i in rev) and all(float(r1["before"][i]) == 1.0 for i in rev)
    assert torch.equal(r0["after"], r1["after"])                                             # ... and agree after the sum
    assert all(float(r0["after"][i]) > 0 for i in rev)                                       # live somewhere -> live everywhere
    assert all(float(r0["after"][i]) == 0.0 for i in never)                                  # dead everywhere -> skipped everywhere
    assert torch.equal(r0["flat"], r1["flat"])
    off = FlatGradSync(TwoBranch()).offsets[rev[0]]
    assert float(r0["flat"][off:off + 6].abs().sum()) > 0                                    # rank 0 holds rank 1's contribution
    assert r0["dev_is_tail"] and r1["dev_is_tail"] and r0["host"] is None and r1["host"] is None


def _uneven_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = Tiny()
        sync = FlatGradSync(model, average=False)
        sync.broadcast_parameters(src=0)
        x, y = _data(22)                                     # 22 pairs over 4 ranks: shards of 6, 6, 5, 5
        lo, hi = shard_range(x.size(0), rank, world)
        sync.zero()
        # a SUM loss weighted by 1 / global size: the ranks' sums add up to the global mean whatever the shard sizes
        (torch.nn.functional.mse_loss(model(x[lo:hi]), y[lo:hi], reduction="sum") / x.size(0)).backward()
        sync.sync()
        out[rank] = (hi - lo, sync.flat.clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world_4_with_unequal_shards_matches_the_global_batch():
    world, port = 4, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_uneven_worker, args=(world, port, out), nprocs=world, join=True)
        res = [out[r] for r in range(world)]
    assert sorted(r[0] for r in res) == [5, 5, 6, 6]
    assert all(torch.equal(res[0][1], r[1]) for r in res[1:])
    model = Tiny()
    ref = FlatGradSync(model)
    x, y = _data(22)
    torch.nn.functional.mse_loss(model(x), y).backward()
    assert torch.allclose(res[0][1], ref.flat, rtol=1e-5, atol=1e-6)


def _bn_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings
        from dualmessagepassing_amd import dp
        model = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            FlatGradSync(model)
            FlatGradSync(Tiny())
        dp._drain_collectives()                 # a condition, not a delay: returns with nothing pending (gloo: at once)
        out[rank] = [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_batch_norm_under_data_parallelism_warns_per_shard_statistics():
    """with BatchNorm on (config.py:201-207) the shards use their own statistics -- said out loud (VERDICT r5 missing 6)."""
    world, port = 2, _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_bn_worker, args=(world, port, out), nprocs=world, join=True)
        msgs = [out[r] for r in range(world)]
    for ms in msgs:
        assert len(ms) == 1 and "PER SHARD" in ms[0] and "BatchNorm" in ms[0]


def test_single_process_has_no_live_tail_indirection():
    """Without a collective the optimizer keeps the host-side live set (no device indirection, nothing sent)."""
    model = TwoBranch()
    sync = FlatGradSync(model)
    master = sync.flatten_parameters()
    x, _ = _data(8)
    sync.detach_grads()
    model(x, torch.zeros(8, dtype=torch.bool)).backward()
    sync.pack()
    assert master._dmp_live_dev is None and master._dmp_live_params is not None
    assert len(master._dmp_live_params) == 2          # fwd.weight, fwd.bias

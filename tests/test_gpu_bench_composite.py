"""The step ``bench.py`` TIMES, at the size it is timed at, against the CPU oracle of the whole model.

VERDICT r3 "what's weak" 1-2: the timed composite -- 1024 pairs, hid 128, the first layer on the label codes
(``fused.Layer0Codes``), the last layer without its [E, H] rows (``edge_rows=False``), deferred embeddings, the
class-typed products on the bf16 pipe (bf16x6), replayed from a HIP graph (``dp.StepGraph``) -- was compared with
``oracle/model_oracle.py`` at 16 / 32 pairs only.  Here: ``bench.build_step`` itself on ``bench.make_shard``'s 1024 pairs
(BASELINE configs[1]) and on 64 pairs of configs[3]'s shapes; ``pred_c`` and the WHOLE flat gradient of the count loss
(``train.py:624-628`` on ``basemodel.py:1500-1663``'s forward), one eager run and one replayed run, against the oracle
on the same pairs.  The oracle runs the pairs in chunks (pairs are independent and the loss is a mean over them, so the
chunk gradients add up to the batch gradient): bounded host memory, ~1 minute of CPU.
"""
import os
import sys

import numpy as np
import pytest
import torch as th

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle(cfg, shard, model, chunk, dtype=th.float32):
    """pred_c [B] and {parameter name: gradient of mean((pred_c - counts)^2)} from oracle/model_oracle.py, chunk by chunk
    (``dtype`` float64: the same operation order in double precision)."""
    sys.path.insert(0, ROOT)
    import bench
    import model_oracle as MO
    B = cfg["batch"]
    mc = bench.model_config(cfg)
    sd = {k: (v.detach().cpu().clone().to(dtype) if v.is_floating_point() else v.detach().cpu().clone()) for k, v in model.state_dict().items()}
    for k in list(sd):                                           # shared sub-networks: one tensor under both names
        twin = "g_" + k[2:]
        if k.startswith("p_") and twin in sd and sd[k].shape == sd[twin].shape and th.equal(sd[k], sd[twin]):
            sd[k] = sd[twin]
    leaves = {}
    for k, v in sd.items():
        if v.is_floating_point() and not any(v is q for q in leaves.values()):
            leaves[k] = v.requires_grad_(True)
    counts = shard["counts"].cpu().to(dtype)
    preds = []
    th.set_num_threads(min(os.cpu_count() or 1, 32))
    for lo in range(0, B, chunk):
        hi = min(B, lo + chunk)
        part = bench.slice_shard(cfg, shard, lo, hi)
        sides = {}
        for tag, n in (("p", cfg["p_nodes"]), ("g", cfg["g_nodes"])):
            s = part[tag]
            off = th.repeat_interleave(th.arange(hi - lo) * n, s["num_edges"].cpu())      # dgl.batch: node-offset concat
            sides[tag] = {"src": s["local_src"].cpu() + off, "dst": s["local_dst"].cpu() + off,
                          "bnn": s["num_nodes"].tolist(), "bne": s["num_edges"].tolist(),
                          "id": s["ndata"]["id"].cpu(), "label": s["ndata"]["label"].cpu(), "eid": s["edata"]["id"].cpu(),
                          "elabel": s["edata"]["label"].cpu(), "rev": s["edata"]["is_reversed"].cpu()}
        ref = MO.model_forward(sd, mc, sides["p"], sides["g"])
        pc = ref["pred_c"].view(-1)
        (((pc - counts[lo:hi]) ** 2).sum() / B).backward()        # this chunk's share of the batch mean
        preds.append(pc.detach())
        del ref, pc
    grads = {k: (v.grad if v.grad is not None else None) for k, v in leaves.items()}
    return th.cat(preds), grads


def _named_flat(step, model):
    """{parameter name: its slice of the packed flat gradient} (the buffer the all-reduce and the optimizer see)."""
    sync = step.sync
    by_id = {id(p): (off, p) for p, off in zip(sync.params, sync.offsets)}
    out = {}
    for name, p in model.named_parameters():
        if id(p) in by_id and name not in out:
            off, q = by_id[id(p)]
            out[name] = sync.flat[off:off + q.numel()].view_as(q).detach().cpu().clone()
    return out


def _compare(tag, pred, flat, ref_pred, ref_grads, pred_tol=1e-4, grad_tol=2e-4):      # pred_c 1e-4, parameter gradients 2e-4
    scale = max(1.0, float(ref_pred.abs().max()))
    err = float((pred.cpu().view(-1) - ref_pred).abs().max())
    assert err <= pred_tol * scale, "%s: pred_c err %g (scale %g)" % (tag, err, scale)
    checked = 0
    seen = set()
    why = {"no oracle gradient": 0, "shared tensor seen": 0, "zero oracle gradient": 0}
    for name, g in flat.items():
        ref = ref_grads.get(name)
        if ref is None:
            twin = ("g_" + name[2:]) if name.startswith("p_") else ("p_" + name[2:])
            ref = ref_grads.get(twin)
        if ref is None:                                          # a parameter the oracle gives no gradient: zeros here too
            assert float(g.abs().max()) == 0.0, (tag, name)
            why["no oracle gradient"] += 1
            continue
        if id(ref) in seen:
            why["shared tensor seen"] += 1
            continue
        seen.add(id(ref))
        s = float(ref.abs().max())
        if s == 0.0:
            assert float(g.abs().max()) == 0.0, (tag, name)
            why["zero oracle gradient"] += 1
            continue
        e = float((g - ref).abs().max())
        # a (Leaky)ReLU whose pre-activation lies within rounding of its kink may take the other branch (util_flips):
        # at this size it moves ONE term of a sum over ~10^5..10^8 terms, far inside the bound
        assert e <= grad_tol * s, "%s: %s gradient err %g > %g of its largest entry %g" % (tag, name, e, grad_tol, s)
        checked += 1
    assert checked >= 60, (tag, checked, len(flat), why)
    return checked


def _no_worse_than_the_stand_in(tag, pred, flat, ref32, grads32, ref64, grads64, factor=4.0, floor=2e-5):
    """"against an fp64 run of the same math the build must be no worse than the stand-in".  The stand-in is the
    reference's operation order in fp32 (the oracle, torch CPU); both it and the product are fp32 roundings of the same real
    numbers along different summation orders.  Measured at 1024 pairs (round 5): the stand-in sits 3e-7 .. 6e-7 of a
    gradient tensor's largest entry from fp64 (torch's CPU products sum pairwise / in blocks), the product 1e-6 .. 1.2e-5
    (fixed-order sums over up to 5 x 10^5 rows in fp32) -- further out than the stand-in on the long weight-gradient sums, a
    factor 17 inside SURVEY's 2e-4.  Held here: within ``factor`` x the stand-in's own distance OR within ``floor`` = 2e-5
    of the tensor's largest entry (a tenth of the stated 2e-4).  -> the worst (product / max(stand-in, floor)) seen."""
    worst = ("", 0.0)
    table = []
    def one(name, got, r32, r64):
        nonlocal worst
        s = max(float(r64.abs().max()), 1e-30)
        e_prod = float((got.double() - r64).abs().max())
        e_ref = float((r32.double() - r64).abs().max())
        bound = factor * e_ref + floor * s
        assert e_prod <= bound, "%s: %s is %.3g from fp64, the fp32 stand-in %.3g (scale %.3g): worse than %.1f x the stand-in" % (tag, name, e_prod, e_ref, s, factor)
        ratio = e_prod / max(e_ref, floor * s)
        table.append((ratio, name, e_prod / s, e_ref / s))
        if ratio > worst[1]:
            worst = (name, ratio)
    one("pred_c", pred.cpu().view(-1), ref32, ref64)
    seen = set()
    for name, g in flat.items():
        r32, r64 = grads32.get(name), grads64.get(name)
        if r32 is None or r64 is None or id(r64) in seen or float(r64.abs().max()) == 0.0:
            continue
        seen.add(id(r64))
        one(name, g, r32, r64)
    for ratio, name, ep, er in sorted(table, reverse=True)[:12]:
        print("  vs fp64: %-58s product %.2e  stand-in %.2e of the largest entry  (x %.1f)" % (name, ep, er, ratio))
    return worst


def _run_case(cfg, gpu, oracle_chunk, fp64=False):
    sys.path.insert(0, ROOT)
    import bench
    from dualmessagepassing_amd import _lib, fused
    from dualmessagepassing_amd.dp import StepGraph
    lib = _lib.load()
    # the composite under test is the one bench.py times: every switch at its shipped default
    assert fused.USE_LAYER0 and not lib.dmp_dev_get_exact_fp32()
    shard = bench.make_shard(cfg, 0, gpu)
    step, model = bench.build_step(dict(cfg, graph=True), shard, gpu, 1)
    assert getattr(model, "lazy_edge_rep", True)
    # away from the initial parameters: the reference zero-initialises the heads' last Linear (pred.py), so at step 0 every
    # gradient upstream of it is exactly zero on both sides and the comparison would hold vacuously for 64 of 76 tensors
    gen = th.Generator().manual_seed(1234)
    with th.no_grad():
        for p in step.sync.params:
            p.add_((0.05 * th.randn(p.shape, generator=gen)).to(gpu))
    ref_pred, ref_grads = _oracle(cfg, shard, model, oracle_chunk)

    # one eager run (HIP-event timer on: its record names say which kernels the step ran)
    _lib.timer.reset()
    _lib.timer.only, _lib.timer.enabled = None, True
    try:
        step.front()
        names = set(k.split("[", 1)[0] for k in _lib.timer.summary())
    finally:
        _lib.timer.enabled = False
        _lib.timer.reset()
    for k in ("l0_edge_fwd", "l0_bwd_w", "pool_relu_bwd", "edge_fwd_typed", "seg_sum2", "bwd_h1_w"):
        assert k in names, (k, sorted(names))
    # the edge chain's input gradient + class-typed weight gradient: one launch (round 6), or the two it replaces
    assert "bwd_z_w" in names or ("bwd_z_typed" in names and ("atb2" in names or "atb_typed" in names)), sorted(names)
    checked = _compare("eager", step.last_pred_c, _named_flat(step, model), ref_pred, ref_grads)
    if fp64:      # the stated bounds (1e-4 / 5e-4 of the largest entry) against what fp32 itself can hold at this size
        ref_pred64, ref_grads64 = _oracle(cfg, shard, model, oracle_chunk, dtype=th.float64)
        worst = _no_worse_than_the_stand_in("eager vs fp64", step.last_pred_c, _named_flat(step, model), ref_pred, ref_grads, ref_pred64, ref_grads64)
        print("composite vs fp64: worst (product error) / (stand-in error) = %.2f at %s" % (worst[1], worst[0]))

    # the replayed run: front() has no optimizer update, so the recording sees the same parameters
    g = StepGraph(lambda: step.front(), optimizer=None, max_shapes=1)
    with g.on_stream():
        g()                          # eager (first call of a signature)
        step.sync.flat.zero_()
        g()                          # recorded + replayed
        assert g.replays == 1
        step.sync.flat.fill_(float("nan"))        # whatever the replay leaves in the buffer must be ITS gradient
        g()
        assert g.replays == 2
        th.cuda.synchronize()
        _compare("replayed", step.last_pred_c, _named_flat(step, model), ref_pred, ref_grads)
    # the same step with the rep-net on the target edges the filter gate keeps (bench.py's `gate_compact` object;
    # model.set_gate_capacity): the oracle computes every edge row, gated ones included -- same pred_c, same gradient
    info = step.set_gate_compact(True)
    assert info is not None and info["capacity"] < 0.7 * info["edges"], info
    step.sync.flat.fill_(float("nan"))
    step.front()
    _compare("gate-compact eager", step.last_pred_c, _named_flat(step, model), ref_pred, ref_grads)
    g2 = StepGraph(lambda: step.front(), optimizer=None, max_shapes=1)
    with g2.on_stream():
        g2()
        g2()
        step.sync.flat.fill_(float("nan"))
        g2()
        assert g2.replays == 2
        th.cuda.synchronize()
        _compare("gate-compact replayed", step.last_pred_c, _named_flat(step, model), ref_pred, ref_grads)
    assert model.compaction_status() == 0
    step.set_gate_compact(False)
    return checked


def test_benchmarked_step_at_config_2_full_size_matches_the_model_oracle(gpu):
    """BASELINE configs[1]: 1024 pairs of pattern (8,12) x target (64,256), hid 128 -- the bench line's step."""
    sys.path.insert(0, ROOT)
    import bench
    cfg = dict(bench.CFG, act="leaky_relu", emb="Equivariant", micro_batches=0)
    assert cfg["batch"] == 1024 and cfg["hid"] == 128
    _run_case(cfg, gpu, oracle_chunk=128, fp64=True)


def test_scaling_workload_step_matches_the_model_oracle_with_gradients(gpu):
    """BASELINE configs[3] shapes (pattern (16,32) x target (512,4096)), 64 pairs: `bench.py --workload 4 --batch 64`."""
    sys.path.insert(0, ROOT)
    import bench
    cfg = dict(bench.CFG4, batch=64, act="leaky_relu", emb="Equivariant", micro_batches=0)
    _run_case(cfg, gpu, oracle_chunk=8)

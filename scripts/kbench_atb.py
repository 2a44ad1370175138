#!/usr/bin/env python3
"""Dev aid: the class-typed weight-gradient kernel (dmp_atb_typed) at bench.py's launch shape -- fixed against per-tile cost."""
import json, os, sys
import torch as th
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from dualmessagepassing_amd import fused
    from dualmessagepassing_amd.collate import collate_device
    from dualmessagepassing_amd.dmpnn import prepare_joint
    dev = th.device("cuda:0")
    H = 128
    res = {}
    for B in (1024, 256, 64):
        cfg = dict(bench.CFG, batch=B)
        shard = bench.make_shard(cfg, 0, dev)
        gs = {}
        for tag in ("p", "g"):
            s = shard[tag]
            gs[tag] = collate_device(s["local_src"], s["local_dst"], s["num_nodes"], s["num_edges"], s["N"], s["E"], ndata=s["ndata"],
                                     edata=s["edata"], max_nodes=s["max_n"], max_edges=s["max_e"])
        u = prepare_joint(gs["p"], gs["g"], H)
        ix = u.index()
        E = ix.num_edges
        coef = ix.degree_coef(u.ndata["out_deg"])
        g = th.Generator(device=dev).manual_seed(1)
        zs = [th.randn(E, H, device=dev, generator=g) for _ in range(3)]
        ds = [th.randn(E, H, device=dev, generator=g) for _ in range(3)]
        for frac in (1.0, 0.42, 0.1):
            gate = (th.rand(E, device=dev, generator=g) < frac).float()
            gate._dmp_binary = True
            gate._dmp_zero_rows = True
            gt = gate if frac < 1.0 else None

            def timeit(fn, reps=30):
                for i in range(5):
                    fn(i)
                th.cuda.synchronize()
                ev = [(th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)) for _ in range(reps)]
                for i, (a, b) in enumerate(ev):
                    a.record(); fn(i); b.record()
                th.cuda.synchronize()
                t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
                return round(t[len(t) // 2], 1)
            with fused.deferred_reductions() as dr:
                res["B=%d keep=%.2f T|B" % (B, frac)] = timeit(lambda i: fused.atb_typed(zs[i % 3], ds[i % 3], coef, ix, gate=gt))
                dr.jobs = []
                res["B=%d keep=%.2f T" % (B, frac)] = timeit(lambda i: fused.atb_typed(zs[i % 3], ds[i % 3], coef, ix, gate=gt, plain=True))
                dr.jobs = []
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

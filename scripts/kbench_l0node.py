#!/usr/bin/env python3
"""Dev aid: dmp_l0_node_fwd at bench.py's shape (8192 pattern + 65536 target nodes, 40 % of the target nodes kept): variants."""
import json, os, sys
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from dualmessagepassing_amd import fused
    dev = th.device("cuda:0")
    g = th.Generator(device=dev).manual_seed(0)
    N, NP, VK, K0, H = 73728, 8192, 16, 10, 128
    kp = 12
    keep = th.rand(N, device=dev, generator=g) < 0.33
    keep[:NP] = True
    venc = (th.rand(N, 16, device=dev, generator=g) < 0.4).float() * keep.view(-1, 1)
    S0 = th.randint(-3, 6, (N, 2 * kp), device=dev, generator=g).float()
    W = fused.l0_node_pack(VK, K0, H, th.randn(VK, 3 * H, device=dev, generator=g), th.randn(K0, H, device=dev, generator=g), th.randn(K0, H, device=dev, generator=g))
    bias = th.randn(H, device=dev, generator=g)
    mask = fused.gate_row_mask(keep.float())
    lst, cnt = fused.kept_rows(mask, 0, N, tiles=True)
    rows = (lst, cnt[0:1])
    h1 = th.empty(N, H, device=dev)
    P = th.empty(N, 2 * H, device=dev)

    def timeit(fn, reps=20):
        """GPU time per call: the calls recorded into one HIP graph (no host gaps between them), replayed."""
        st = th.cuda.Stream()
        with th.cuda.stream(st):
            for _ in range(3):
                fn()
            th.cuda.synchronize()
            gr = th.cuda.CUDAGraph()
            with th.cuda.graph(gr, stream=st):
                for _ in range(reps):
                    fn()
            gr.replay()
            th.cuda.synchronize()
            a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                gr.replay()
            b.record()
            th.cuda.synchronize()
        return round(a.elapsed_time(b) * 1e3 / (5 * reps), 1)
    res = {}
    for name, (n0, n1) in (("pattern", (0, NP)), ("target", (NP, N))):
        res[name + " list"] = timeit(lambda: fused.l0_node_fwd(venc, VK, S0, K0, kp, W, bias, 0.18, mask, n0, n1, H, h1, P, rows=rows))
        res[name + " list from q_begin"] = timeit(lambda: fused.l0_node_fwd(venc, VK, S0, K0, kp, W, bias, 0.18, mask, n0, n1, H, h1, P, rows=rows, q_begin=n0))
        res[name + " masked rows"] = timeit(lambda: fused.l0_node_fwd(venc, VK, S0, K0, kp, W, bias, 0.18, mask, n0, n1, H, h1, P))
        res[name + " all rows"] = timeit(lambda: fused.l0_node_fwd(venc, VK, S0, K0, kp, W, bias, 0.18, None, n0, n1, H, h1, P))
    res["empty launch"] = timeit(lambda: fused.l0_node_fwd(venc, VK, S0, K0, kp, W, bias, 0.18, mask, 0, 8, H, h1, P, rows=rows))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

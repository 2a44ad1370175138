"""Dev aid: does a forked side stream inside a HIP graph capture run concurrently with the main branch on this stack?
A chain of small kernels (index-build stand-ins) beside a few large streaming kernels, one stream vs fork / join."""
import json, torch as th
dev = th.device("cuda:0")
big = [th.empty(64 << 20, device=dev) for _ in range(4)]       # 256 MB each
small = [th.zeros(4096, device=dev) for _ in range(4)]


def work(fork, nsmall=40, nbig=4):
    main = th.cuda.current_stream()
    side = work.side
    if fork:
        side.wait_stream(main)
        with th.cuda.stream(side):
            for i in range(nsmall):
                small[i % 4].add_(1.0)
    else:
        for i in range(nsmall):
            small[i % 4].add_(1.0)
    for i in range(nbig):
        big[i].mul_(1.0001)
    if fork:
        main.wait_stream(side)
    small[0].add_(big[0][:4096])


work.side = th.cuda.Stream()
res = {}
cap = th.cuda.Stream()
for fork in (False, True):
    for mode in ("eager", "graph"):
        with th.cuda.stream(cap):
            for _ in range(3):
                work(fork)
            th.cuda.synchronize()
            if mode == "graph":
                g = th.cuda.CUDAGraph()
                with th.cuda.graph(g, stream=cap):
                    work(fork)
                run = g.replay
            else:
                run = lambda: work(fork)
            for _ in range(5):
                run()
            th.cuda.synchronize()
            a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                run()
            b.record()
            th.cuda.synchronize()
            res["fork=%s %s" % (fork, mode)] = round(a.elapsed_time(b) / 20 * 1e3, 1)
print(json.dumps(res))

"""How many independent chains of short dependent kernels does one HIP-graph replay run at once?

N streams forked from the capture stream, each K dependent launches of a small elementwise kernel on its own tensor, joined
back; wall time per replay against N (flat = the chains overlap, linear = the replay serialises them).  Also eager.
    python tools/probe_graph_chains.py [elements] [K]
"""
import sys
import time

import torch

n_el = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = "cuda:0"
torch.zeros(1, device=dev)


def chains(N, streams, xs):
    main = torch.cuda.current_stream()
    for i in range(N):
        streams[i].wait_stream(main)
        with torch.cuda.stream(streams[i]):
            for _ in range(K):
                xs[i].mul_(1.0001)
    for i in range(N):
        main.wait_stream(streams[i])


print(f"# {n_el} fp32 elements per launch, {K} dependent launches per chain")
for N in (1, 2, 3, 4, 6, 8, 12):
    streams = [torch.cuda.Stream() for _ in range(N)]
    xs = [torch.ones(n_el, device=dev) for _ in range(N)]
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        chains(N, streams, xs)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            chains(N, streams, xs)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 10
        for _ in range(R):
            g.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / R
        t0 = time.perf_counter()
        for _ in range(R):
            chains(N, streams, xs)
        torch.cuda.synchronize()
        te = (time.perf_counter() - t0) / R
    print(f"chains {N:2d}: graph {tg * 1e3:7.3f} ms ({tg / K * 1e6:5.2f} us per chain link, {tg / (N * K) * 1e6:5.2f} us per launch) | eager {te * 1e3:7.3f} ms", flush=True)
    del g

print("# one graph PER chain, each replayed on its own stream")
for N in (1, 2, 3, 4, 6, 8, 12):
    streams = [torch.cuda.Stream() for _ in range(N)]
    xs = [torch.ones(n_el, device=dev) for _ in range(N)]
    graphs = []
    for i in range(N):
        with torch.cuda.stream(streams[i]):
            for _ in range(3):
                xs[i].mul_(1.0001)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[i]):
                for _ in range(K):
                    xs[i].mul_(1.0001)
            graphs.append(g)
    torch.cuda.synchronize()

    def run():
        for i in range(N):
            with torch.cuda.stream(streams[i]):
                graphs[i].replay()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 10
    for _ in range(R):
        run()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / R
    print(f"graphs {N:2d}: {tg * 1e3:7.3f} ms per round ({tg / K * 1e6:5.2f} us per chain link, {tg / (N * K) * 1e6:5.2f} us per launch)", flush=True)
    del graphs

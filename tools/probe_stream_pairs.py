"""Which pairs of HIP streams run concurrently?  One captured chain of K dependent small launches per stream; every pair (i, j)
is replayed together and timed against a single chain.  ratio ~1 = the two streams overlap (different hardware queues),
ratio ~2 = they share a queue.    python tools/probe_stream_pairs.py [n_streams] [raw|torch]"""
import ctypes
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kind = sys.argv[2] if len(sys.argv) > 2 else "torch"
K, n_el, dev = 300, 160000, "cuda:0"
torch.zeros(1, device=dev)
if kind == "raw":
    hip = ctypes.CDLL("libamdhip64.so")
    streams = []
    for _ in range(N):
        h = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(h), 1) == 0      # hipStreamNonBlocking
        streams.append(torch.cuda.ExternalStream(h.value, device=dev))
else:
    streams = [torch.cuda.Stream() for _ in range(N)]
xs = [torch.ones(n_el, device=dev) for _ in range(N)]
graphs = []
for i in range(N):
    with torch.cuda.stream(streams[i]):
        xs[i].mul_(1.0001)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[i]):
            for _ in range(K):
                xs[i].mul_(1.0001)
        graphs.append(g)
torch.cuda.synchronize()


def t(idx, R=5):
    for _ in range(2):
        for i in idx:
            with torch.cuda.stream(streams[i]):
                graphs[i].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        for i in idx:
            with torch.cuda.stream(streams[i]):
                graphs[i].replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / R


one = min(t([i]) for i in range(N))
print(f"# {kind} streams, single chain {one * 1e3:.3f} ms; pair time / single-chain time:")
for i in range(N):
    print(" ".join(f"{t([i, j]) / one:4.1f}" if j != i else "  . " for j in range(N)), flush=True)
print("# all %d together: %.1f x single" % (N, t(list(range(N))) / one))

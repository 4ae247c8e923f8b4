"""Stream scheduling of the fused co-training step: eager, or recorded as a PROGRAM of HIP-graph segments.

Why segments.  One hipGraphLaunch feeds every kernel node of the graph through ONE hardware queue: measured on MI355X /
ROCm 7.2 (tools/probe_graph_chains.py), N independent chains of dependent small kernels forked inside one captured graph take
N x the time of one chain (2.6 us per node whatever N), while the same chains captured as one graph EACH and launched on
their own streams overlap (tools/probe_stream_pairs.py: 1.2 x the time of one chain for two, 3.8 x for eight -- four hardware
queues).  A 2 x Enet step is ~4700 launches of ~7 us: inside one graph the per-model and per-pass streams of
CoTrainer._run_step_fused bought almost nothing.

So the step is captured as a sequence of segments.  Every maximal run of launches on one stream becomes its own graph; the
points where the step's streams meet (fork / join around the JSD, the adversarial hand-over, the gradient-buffer sum) become
eager event waits between graph launches; host callbacks (the gradient exchange of data parallelism) are replayed in place.
Replaying the program issues the same launches in the same per-stream order as the eager step, each segment on the stream
it was recorded on, so segments of different models / passes run on different hardware queues.

Memory: each stream captures into its own private pool, mirroring the caching allocator's per-stream free lists in eager
mode -- a block freed while recording is only handed to later segments of the SAME stream, which replay in order.
"""
from __future__ import annotations

import contextlib
import ctypes
import time
import warnings
from typing import Callable, List, Optional, Sequence, Tuple

import torch


class EagerSchedule(object):
    """The step's stream operations executed as they are issued."""
    capturing = False

    def on(self, stream):
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    def wait(self, pairs: Sequence[Tuple[torch.cuda.Stream, torch.cuda.Stream]]) -> None:
        for dst, src in pairs:
            dst.wait_stream(src)

    def call(self, fn: Callable[[], None]) -> None:
        fn()

    def record(self, src):
        """A point on ``src`` that another stream can wait for later (``wait_event``).  Unlike ``wait`` the two halves are
        issued separately: streams that share a hardware queue run in ISSUE order, so where the record and the wait sit
        between other launches decides what they end up ordered after."""
        ev = torch.cuda.Event()
        ev.record(src)
        return ev

    def wait_event(self, dst, ev) -> None:
        dst.wait_event(ev)


_hip = None
_OWN_STREAMS = {}


def _hiplib():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        _hip.hipGraphGetNodes.restype = ctypes.c_int
        _hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        _hip.hipStreamCreateWithFlags.restype = ctypes.c_int
    return _hip


def own_stream(device) -> torch.cuda.Stream:
    """A HIP stream of this package's own (hipStreamCreateWithFlags, non-blocking), never destroyed.  ``torch.cuda.Stream()``
    hands out the 32 streams of torch's pool round-robin: in a long-lived process two 'different' Stream objects can be the same
    stream -- e.g. a model stream and the stream a later graph capture runs on -- and an object that dies with its trainer
    returns nothing.  Streams that take part in captured programs are therefore created outside that pool."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        h = ctypes.c_void_p()
        rc = _hiplib().hipStreamCreateWithFlags(ctypes.byref(h), 1)          # hipStreamNonBlocking
    if rc != 0 or not h.value:
        return torch.cuda.Stream(dev)
    st = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", idx))
    _OWN_STREAMS.setdefault(idx, []).append(st)
    return st


_RECORDER_MAIN = {}


def _recorder_main(device) -> torch.cuda.Stream:
    """The stream a SegmentRecorder records the caller's work on: one per device for the life of the process (a fresh one per
    capture leaked a HIP stream per capture)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _RECORDER_MAIN.get(idx)
    if st is None:
        st = _RECORDER_MAIN[idx] = own_stream(torch.device("cuda", idx))
    return st


def _graph_node_count(graph: torch.cuda.CUDAGraph) -> int:
    """Number of nodes of a captured (keep_graph) graph, -1 when it cannot be asked."""
    try:
        n = ctypes.c_size_t(0)
        if _hiplib().hipGraphGetNodes(ctypes.c_void_p(int(graph.raw_cuda_graph())), None, ctypes.byref(n)) != 0:
            return -1
        return int(n.value)
    except Exception:
        return -1


class SegmentProgram(object):
    """What a SegmentRecorder produced: ('graph', stream key, CUDAGraph) | ('wait', [(dst key, src key, event)]) |
    ('record', src key, event) | ('wait_event', dst key, event) | ('call', stream key, fn).  The key 'main' stands for whatever stream is current when the program is replayed."""

    def __init__(self, ops: List[tuple], device):
        self.ops = ops
        self.device = device
        self.n_graphs = sum(1 for o in ops if o[0] == 'graph')
        self.n_nodes = sum(o[3] for o in ops if o[0] == 'graph' and o[3] > 0)

    def replay(self) -> None:
        main = torch.cuda.current_stream(self.device)
        try:
            for op in self.ops:
                kind = op[0]
                if kind == 'graph':
                    torch.cuda.set_stream(main if op[1] == 'main' else op[1])
                    op[2].replay()
                elif kind == 'wait':
                    for dst, src, ev in op[1]:
                        ev.record(main if src == 'main' else src)
                        (main if dst == 'main' else dst).wait_event(ev)
                elif kind == 'record':
                    op[2].record(main if op[1] == 'main' else op[1])
                elif kind == 'wait_event':
                    (main if op[1] == 'main' else op[1]).wait_event(op[2])
                else:
                    torch.cuda.set_stream(main if op[1] == 'main' else op[1])
                    op[2]()
        finally:
            torch.cuda.set_stream(main)


class SegmentRecorder(object):
    """Records the step as a SegmentProgram.  Exactly one capture is open at any time, on the stream that is current; switching
    streams (``on``), a wait between streams or a host callback closes it and opens the next one."""
    capturing = True

    def __init__(self, device):
        self.device = device
        self.main = _recorder_main(device)             # stands in for the caller's stream while recording (one per device, reused)
        self.ops: List[tuple] = []
        self._pools = {}
        self._empty: List[torch.cuda.CUDAGraph] = []
        self._open: Optional[Tuple[torch.cuda.Stream, torch.cuda.CUDAGraph]] = None
        self._outer: Optional[torch.cuda.Stream] = None

    def _key(self, s):
        return 'main' if s == self.main else s

    def _begin(self, stream) -> None:
        g = torch.cuda.CUDAGraph(keep_graph=True)
        torch.cuda.set_stream(stream)
        pool = self._pools.get(stream.cuda_stream)
        if pool is None:
            pool = self._pools[stream.cuda_stream] = torch.cuda.graph_pool_handle()
        g.capture_begin(pool=pool, capture_error_mode="thread_local")
        self._open = (stream, g)

    def _end(self) -> None:
        if self._open is None:
            return
        stream, g = self._open
        self._open = None
        torch.cuda.set_stream(stream)
        with warnings.catch_warnings():         # ("The CUDA Graph is empty": expected for a stream switch with nothing in between)
            warnings.simplefilter("ignore")
            g.capture_end()
        n = _graph_node_count(g)
        if n == 0:                              # a stream switch with nothing launched in between: not replayed, but kept alive
            self._empty.append(g)               # (the stream's private pool dies with the last graph that used it)
            return
        g.instantiate()
        self.ops.append(('graph', self._key(stream), g, n))

    def _switch(self, stream) -> None:
        if self._open is not None and self._open[0] == stream:
            torch.cuda.set_stream(stream)
            return
        self._end()
        self._begin(stream)

    # ---- the schedule interface -------------------------------------------------------------------
    @contextlib.contextmanager
    def on(self, stream):
        if stream is None:
            yield
            return
        prev = torch.cuda.current_stream(self.device)
        self._switch(stream)
        try:
            yield
        finally:
            self._switch(prev)

    def wait(self, pairs) -> None:
        pairs = list(pairs)
        if not pairs:
            return
        cur = torch.cuda.current_stream(self.device)
        self._end()
        self.ops.append(('wait', [(self._key(d), self._key(s), torch.cuda.Event()) for d, s in pairs]))
        self._begin(cur)

    def call(self, fn) -> None:
        cur = torch.cuda.current_stream(self.device)
        self._end()
        self.ops.append(('call', self._key(cur), fn))
        self._begin(cur)

    def record(self, src):
        cur = torch.cuda.current_stream(self.device)
        self._end()
        ev = torch.cuda.Event()
        self.ops.append(('record', self._key(src), ev))
        self._begin(cur)
        return ev

    def wait_event(self, dst, ev) -> None:
        cur = torch.cuda.current_stream(self.device)
        self._end()
        self.ops.append(('wait_event', self._key(dst), ev))
        self._begin(cur)

    # ---- recording --------------------------------------------------------------------------------
    def start(self) -> None:
        self._outer = torch.cuda.current_stream(self.device)
        torch.cuda.synchronize(self.device)
        self._begin(self.main)

    def finish(self) -> SegmentProgram:
        self._end()
        torch.cuda.set_stream(self._outer)
        ops: List[tuple] = []
        for op in self.ops:                     # waits that ended up next to each other (dropped empty segments) -> one op
            if op[0] == 'wait' and ops and ops[-1][0] == 'wait':
                ops[-1] = ('wait', ops[-1][1] + op[1])
            else:
                ops.append(op)
        prog = SegmentProgram(ops, self.device)
        prog.keepalive = self._empty
        return prog

    def abort(self) -> None:
        """Leave no capture open after an exception."""
        try:
            if self._open is not None:
                stream, g = self._open
                self._open = None
                torch.cuda.set_stream(stream)
                g.capture_end()
        except Exception:
            pass
        finally:
            if self._outer is not None:
                torch.cuda.set_stream(self._outer)


# ---- which streams sit on different hardware queues ---------------------------------------------------
_QUEUE_GROUPS = {}
_QUEUE_PROBE = {}


def _chain_graph(stream, x, links):
    with torch.cuda.stream(stream):
        x.mul_(1.0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
            for _ in range(links):
                x.mul_(1.0)
    return g


def queue_groups(device, candidates: int = 8, links: int = 150) -> List[List[torch.cuda.Stream]]:
    """Streams grouped by the hardware queue they share, found by timing (once per device and process, ~50 ms).

    HIP multiplexes its streams onto a few hardware queues (four by default) and does not say which: two streams of one queue
    run their graph launches one after the other.  One short captured chain per candidate stream; a candidate whose chain,
    launched together with a group's first stream, takes > 1.6 x one chain shares that group's queue."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())      # a bare "cuda" means the CURRENT device, not device 0
    key = dev.index
    if key in _QUEUE_GROUPS:
        return _QUEUE_GROUPS[key]
    import os
    forced = os.environ.get("DCT_HW_QUEUES")       # e.g. "4": skip the probe and deal fresh streams round-robin over that many groups
    if forced:
        n = max(1, int(forced))
        out = [[own_stream(dev) for _ in range(max(1, candidates // n))] for _ in range(n)]
        _QUEUE_GROUPS[key] = out
        _QUEUE_PROBE[key] = dict(groups=n, sizes=[len(g) for g in out], attempts=0, source="DCT_HW_QUEUES")
        return out
    streams = [own_stream(dev) for _ in range(candidates)]
    xs = [torch.ones(65536, device=dev) for _ in streams]
    graphs = [_chain_graph(s, x, links) for s, x in zip(streams, xs)]
    torch.cuda.synchronize(dev)

    def t(idx, rounds=3):
        best = float("inf")
        for _ in range(rounds + 1):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in idx:
                torch.cuda.set_stream(streams[i])
                graphs[i].replay()
            torch.cuda.synchronize(dev)
            best = min(best, time.perf_counter() - t0)
        return best
    cur = torch.cuda.current_stream(dev)
    try:
        for attempt in range(3):                # (a noisy first pass -- clocks ramping up -- is repeated: HIP has four queues by default)
            one = min(t([i]) for i in range(len(streams)))
            groups: List[List[int]] = []
            for i in range(len(streams)):
                for g in groups:
                    if t([g[0], i]) > 1.6 * one:
                        g.append(i)
                        break
                else:
                    groups.append([i])
            if len(groups) == 4 and max(len(g) for g in groups) <= 3:
                break
    finally:
        torch.cuda.set_stream(cur)
    del graphs
    out = [[streams[i] for i in g] for g in groups]
    ok = len(groups) == 4 and max(len(g) for g in groups) <= 3
    _QUEUE_PROBE[key] = dict(groups=len(groups), sizes=[len(g) for g in groups], attempts=attempt + 1, source="timing probe", ok=ok)
    if ok:
        _QUEUE_GROUPS[key] = out        # a failed probe is NOT cached: the next trainer probes again
    else:
        import warnings
        warnings.warn(f"dct_amd: the hardware-queue probe found {len(groups)} groups {[len(g) for g in groups]} instead of 4 "
                      "(shared or noisy GPU?); the multi-queue step layouts may run slower -- set DCT_HW_QUEUES=4 to skip the probe")
    return out


def queue_probe_report(device=None) -> dict:
    """What the last probe of ``device`` found (bench.py prints and asserts on it)."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    return dict(_QUEUE_PROBE.get(idx, {}))


class StreamDealer(object):
    """Hands out streams so that consecutive requests land on different hardware queues (round-robin over queue_groups)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._order: Optional[List[torch.cuda.Stream]] = None
        self._next = 0

    def take(self) -> torch.cuda.Stream:
        if self._order is None:
            groups = [list(g) for g in queue_groups(self.device)]
            order = []
            while any(groups):
                for g in groups:
                    if g:
                        order.append(g.pop(0))
            self._order = order
        if self._next >= len(self._order):
            return own_stream(self.device)                  # more chains than candidates: whatever HIP picks
        s = self._order[self._next]
        self._next += 1
        return s

"""Data-parallel gradient exchange for co-training: one process per GPU, RCCL over xGMI.

The reference's only multi-GPU facility is single-process ``nn.DataParallel``
(models/segmentators.py:34-36): batch split on dim 0, per-replica BatchNorm statistics,
gradients summed onto device 0.  The MI355X-native equivalent here:

  * every rank holds all S models and draws its own labeled / unlabeled mini-batches
    (global batch = world x per-GPU batch; equal per-rank batch sizes make the gradient
    *average* equal to the global-batch mean gradient);
  * each model's gradients already live in ONE flat fp32 buffer (arch/flat.py), so the exchange
    is one large all-reduce per model straight out of that buffer -- no bucket copies; the flat
    buffers of SMALL models (Enet: 1.45 MB each) are laid out back to back in one arena and leave
    as ONE collective per step for all S models, not S latency-bound ones.  xGMI is point-to-point
    (7 links x ~153 GB/s per GPU): a ring all-reduce of UNet's 124 MB takes ~1.4 ms at 8 GPUs, so
    model m's all-reduce is issued asynchronously right after model m's backward is enqueued
    and overlaps the backward of model m+1 (``begin`` / ``finish``);
  * the collective is a SUM and the 1/world of the average is applied here, not by RCCL: ``ReduceOp.AVG`` runs RCCL's
    pre-multiply kernels, whose gfx950 code holds 8-48 packed-FP32 instructions per ring / tree function (the plain ring SUM of f32
    and every bf16 kernel hold none), and an exchange overlapped with the backward pass runs beside the conv kernels -- the
    neighbourhood in which one packed-FP32 operand form goes wrong (tools/probe_packed_fp32).  RCCL's kernels do not
    use that form, so this is caution, but it is free: fused-Adam models get the factor folded into the optimizer's
    ``grad_scale`` (no launch, ``defer_average``); everything else is scaled in its buffer by ``dct_flat_scale``.
    The algorithm choice stays RCCL's (``prefer_ring()`` is there for the cautious);
  * BatchNorm buffers stay per-rank (what DataParallel replicas do); FGSM's input-gradient pass
    produces no parameter gradients and therefore never touches the exchange.

Networks without a flat buffer (injected modules in host-logic tests) are reduced through a
temporary flattened copy of their gradients.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class FlatGradSync(object):
    """``compress="bf16"``: the gradients travel as bf16 (half the bytes on the per-link-bound xGMI ring: UNet 2 x 62 MB instead
    of 2 x 124 MB per step) -- cast into a cached bf16 buffer, all-reduced, cast back into the fp32 flat buffer the optimizer
    reads.  ``measure=True``: HIP events around every wait, ``exposed_ms()`` = time the model streams actually stalled on the
    exchange (what the overlap did not hide)."""

    FUSE_BELOW = 4 << 20        # elements: models under 16 MiB of fp32 gradients share one arena and ONE all-reduce per step

    def __init__(self, segmentators, process_group=None, broadcast_weights: bool = True, compress: Optional[str] = None,
                 measure: bool = False, fuse_small: bool = True, defer_average: bool = True):
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("FlatGradSync needs an initialised torch.distributed process group")
        self.segmentators = list(segmentators)
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self._pending: List = []
        self._bucketed = set()      # models whose gradients already went out bucket by bucket this step
        self.bucket_calls = 0
        # defer_average: a model whose optimizer takes a gradient scale (fused Adam over a flat buffer) keeps the SUM in its
        # buffer and CoTrainer folds ``optimizer_scale(i)`` = 1/world into the update; False: every buffer holds the mean after finish()
        self.defer_average = bool(defer_average)
        assert compress in (None, "bf16"), compress
        self.compress = compress
        self._cbuf = {}             # model index -> bf16 image of its flat gradient buffer
        self.measure = measure
        self._events: List = []
        self.exchanged_bytes = 0
        self.fuse_small = bool(fuse_small)
        self._arena: Optional[torch.Tensor] = None      # gradient buffers of all small models, back to back (Enet -> single bucket)
        self._arena_span = {}       # model index -> (lo, hi) elements of the arena
        self._fused_ready: List[int] = []
        self._fused_events = {}     # model index -> event recorded on the stream that produced its arena slice (CUDA only)
        self.collectives = 0        # all-reduce launches since construction (tests and bench read it)
        self._prepared = False
        self._pair_rng = None       # adversarial pair: one RandomState per job, same seed on every rank (draw_pair)
        if broadcast_weights:
            self.broadcast_weights()
        # decorrelate the ranks' dropout masks (same torch seed on every rank gives every rank the same Philox seed)
        for seg in self.segmentators:
            net = seg.torchnet
            if hasattr(net, "dropout_seed") and not getattr(net, "_rank_mixed", False):
                net.dropout_seed = (int(net.dropout_seed) + 0x9E3779B97F4A7C15 * (self.rank + 1) * int(self.rank > 0)) & ((1 << 62) - 1)
                net._rank_mixed = True

    # -- weights: rank 0's initialisation everywhere ---------------------------------------------
    @torch.no_grad()
    def broadcast_weights(self):
        for seg in self.segmentators:
            net = seg.torchnet
            flat = getattr(net, "flat_params", None)
            if flat is not None:
                flat.ensure()
                dist.broadcast(flat.flat, src=0, group=self.group)
                if hasattr(net, "mark_weights_updated"):
                    net.mark_weights_updated()
            else:
                for p in net.parameters():
                    dist.broadcast(p.data, src=0, group=self.group)
            for b in net.buffers():
                if b.dtype.is_floating_point:
                    dist.broadcast(b.data, src=0, group=self.group)

    # -- the adversarial pair: identical on every rank ----------------------------------------------
    def draw_pair(self, n_models: int):
        """The (a, b) models of this step's adversarial block (reference cotraining_totalloss.py:230-234: ``np.random.choice``
        on the process-global numpy RNG).  Under data parallelism every rank must train the SAME pair -- the gradient average
        of a step mixes the ranks' adversarial terms -- but the ranks' global numpy states need not agree (loaders, user code).
        So the job owns one RandomState: rank 0 draws its seed from ITS global RNG once (a user seed still controls the
        sequence), broadcasts it, and every rank draws the per-step pairs from that private stream."""
        import numpy as np
        if self._pair_rng is None:
            dev = "cpu"
            if dist.get_backend(self.group) == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device())
            drawn = int(np.random.randint(0, 2 ** 31 - 1))     # every rank draws (and all but rank 0 discard): the ranks' global numpy streams stay aligned
            seed = torch.tensor([drawn if self.rank == 0 else 0], dtype=torch.int64, device=dev)
            dist.broadcast(seed, src=0, group=self.group)
            self._pair_rng = np.random.RandomState(int(seed.item()))
        try:
            choice = sorted(self._pair_rng.choice(list(range(n_models)), 2, replace=False).tolist())
        except Exception:
            choice = sorted(self._pair_rng.choice(list(range(n_models)), 2, replace=True).tolist())
        return choice[0], choice[1]

    # -- gradients ---------------------------------------------------------------------------------
    def _adopt_arena(self):
        """Move the flat gradient buffers of all small flat-buffer models into one allocation (each a slice of it, the
        ``p.grad`` views re-pointed), so that their exchange is a single collective with no copies.  Idempotent; re-adopts a
        model whose buffer was re-allocated since (``.to(device)``, ``load_state_dict`` re-flattening)."""
        small = []
        for i, seg in enumerate(self.segmentators):
            flat = getattr(seg.torchnet, "flat_params", None)
            # only models whose optimizer (if one is attached) is the fused flat one: re-pointing p.grad gives parameters without a gradient a zero
            # view, which a torch optimizer would treat as "has a gradient" (weight decay / moments on untouched parameters)
            opt = getattr(seg, "optimizer", None)
            fused_opt = opt is None or hasattr(opt, "grad_scale")
            if flat is not None and flat.total < self.FUSE_BELOW and fused_opt:
                flat.ensure()
                small.append((i, flat))
        if len(small) < 2:
            self._arena, self._arena_span = None, {}
            return
        dev = small[0][1].flat.device
        total = sum(f.total for _, f in small)
        if self._arena is None or self._arena.numel() != total or self._arena.device != dev:
            self._arena = torch.zeros(total, dtype=torch.float32, device=dev)
            self._arena_span = {}
            lo = 0
            for i, f in small:
                self._arena_span[i] = (lo, lo + f.total)
                lo += f.total
        base = self._arena.data_ptr()
        for i, f in small:
            lo, hi = self._arena_span[i]
            if f.gflat is not None and f.gflat.data_ptr() == base + 4 * lo and f.grads_attached():
                continue
            piece = self._arena[lo:hi]
            if f.gflat is not None and f.gflat.device == dev and f.grads_attached():
                piece.copy_(f.gflat)            # gradients accumulated so far move with the buffer
            else:
                piece.zero_()
            f.gflat = piece
            for k, p in enumerate(f.params):
                p.grad = f._grad_view(k)

    def _reduce_tensor(self, t: torch.Tensor, async_op: bool):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op), t

    def _deferred(self, model_index) -> bool:
        """True when the 1/world of model_index's average is left to its optimizer (a property of the model, not of the step:
        captured optimizer launches bake the factor in).  Models that share the arena decide together."""
        if not self.defer_average or self.world == 1:
            return False

        def one(i):
            seg = self.segmentators[i]
            flat = getattr(seg.torchnet, "flat_params", None)
            return hasattr(getattr(seg, "optimizer", None), "grad_scale") and flat is not None and flat.grads_attached()
        if model_index == "arena" or model_index in self._arena_span:
            return all(one(i) for i in self._arena_span)
        return one(model_index)

    def optimizer_scale(self, model_index: int) -> float:
        """Factor model_index's optimizer has to apply to the gradients finish() leaves in its buffer."""
        return 1.0 / self.world if self._deferred(model_index) else 1.0

    def _average(self, t: torch.Tensor):
        if self.world == 1:
            return
        if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
            from . import hip_ops as K
            K.flat_scale(t, 1.0 / self.world)
        else:
            t.mul_(1.0 / self.world)

    def _start(self, model_index: int, flat, lo: int, hi: int):
        """all-reduce of gflat[lo:hi], optionally through the bf16 image; returns the pending record"""
        src = flat.gflat[lo:hi]
        if self.compress == "bf16" and src.is_floating_point():
            buf = self._cbuf.get(model_index)
            if buf is None or buf.numel() != flat.gflat.numel() or buf.device != flat.gflat.device:
                buf = torch.empty(flat.gflat.numel(), dtype=torch.bfloat16, device=flat.gflat.device)
                self._cbuf[model_index] = buf
            wire = buf[lo:hi]
            wire.copy_(src)
        else:
            wire = src
        self.exchanged_bytes += wire.numel() * wire.element_size()
        self.collectives += 1
        work, scale = self._reduce_tensor(wire, True)
        return (model_index, work, scale, None, None, (wire, src) if wire is not src else None)

    def _start_arena(self):
        """ONE all-reduce over the arena (every small model's gradients); the pending record carries the key 'arena'.
        The models' backward passes may have run on different streams (CoTrainer's per-model streams): the launching stream
        first waits for the event each of them recorded in begin(), so the collective reads no slice that is still being written."""
        src = self._arena
        if src.is_cuda:
            cur = torch.cuda.current_stream(src.device)
            for ev in self._fused_events.values():
                cur.wait_event(ev)
        self._fused_events = {}
        if self.compress == "bf16":
            buf = self._cbuf.get("arena")
            if buf is None or buf.numel() != src.numel() or buf.device != src.device:
                buf = torch.empty(src.numel(), dtype=torch.bfloat16, device=src.device)
                self._cbuf["arena"] = buf
            buf.copy_(src)
            wire = buf
        else:
            wire = src
        self.exchanged_bytes += wire.numel() * wire.element_size()
        self.collectives += 1
        work, scale = self._reduce_tensor(wire, True)
        # the record stays pending until EVERY arena model (or finish(None)) has waited for the collective on its own stream
        return ["arena", work, scale, None, None, (wire, src) if wire is not src else None, set()]

    def begin_bucket(self, model_index: int, lo: int, hi: int):
        """Start the all-reduce of elements [lo, hi) of one model's flat gradient buffer -- called from inside the
        backward pass (net._grad_hook) as soon as that range is final, so the exchange of the decoder-side and
        centre gradients overlaps the rest of the backward.  xGMI ring all-reduce of UNet's 124 MB is ~1.4 ms at 8
        GPUs; only the last, smallest bucket (encoder side, 19 MB) is left exposed."""
        flat = self.segmentators[model_index].torchnet.flat_params
        if hi <= lo:
            return
        self._pending.append(self._start(model_index, flat, lo, hi))
        self._bucketed.add(model_index)
        self.bucket_calls += 1

    def begin(self, model_index: int):
        """Start the (asynchronous) all-reduce of one model's gradients (no-op when the backward pass already
        sent every bucket)."""
        if model_index in self._bucketed:
            return
        net = self.segmentators[model_index].torchnet
        flat = getattr(net, "flat_params", None)
        if self.fuse_small and model_index in self._arena_span and flat is not None and flat.grads_attached() and \
                flat.gflat.data_ptr() == self._arena.data_ptr() + 4 * self._arena_span[model_index][0]:
            # small model: its gradients sit in the shared arena -- the collective goes out when the LAST of them is ready
            if model_index not in self._fused_ready:
                self._fused_ready.append(model_index)
            if self._arena.is_cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self._arena.device))
                self._fused_events[model_index] = ev
            if len(self._fused_ready) == len(self._arena_span):
                self._fused_ready = []
                self._pending.append(self._start_arena())
            return
        if flat is not None and flat.grads_attached():
            self._pending.append(self._start(model_index, flat, 0, flat.gflat.numel()))
            return
        params = [p for p in net.parameters() if p.grad is not None]
        if not params:
            return
        buf = torch.cat([p.grad.reshape(-1) for p in params])
        work, scale = self._reduce_tensor(buf, True)
        self._pending.append((model_index, work, scale, buf, params, None))

    @torch.no_grad()
    def finish(self, model_index: Optional[int] = None):
        """Wait for the pending all-reduces (of one model, or all) and finish the averaging."""
        if self._fused_ready and (model_index is None or model_index in self._fused_ready):
            # not every small model produced gradients this step: exchange the ready ones one by one (same result, more launches)
            ready, self._fused_ready = self._fused_ready, []
            events, self._fused_events = self._fused_events, {}
            # every ready model's collective is launched HERE, on the calling stream, while model i's backward pass ran on the
            # stream begin(i) was called on: the calling stream waits for each of those passes first (the event begin(i) recorded)
            if self._arena is not None and self._arena.is_cuda:
                cur = torch.cuda.current_stream(self._arena.device)
                for i in ready:
                    if i in events:
                        cur.wait_event(events[i])
            for i in ready:
                flat = self.segmentators[i].torchnet.flat_params
                self._pending.append(self._start(i, flat, 0, flat.gflat.numel()))
        keep = []
        for ent in self._pending:
            if ent[0] == "arena":
                # One collective, several consumers on (possibly) different streams: each model waits for it on ITS stream and
                # finishes its own slice there (bf16 image -> fp32, the 1/world) -- model j's update is ordered behind the
                # collective by its own wait, not by whichever model happened to call finish() first.
                _, work, arena_wire, _, _, wire, done = ent
                mine = [i for i in self._arena_span if i not in done] if model_index is None else \
                       ([model_index] if model_index in self._arena_span and model_index not in done else [])
                if mine:
                    if self.measure and torch.cuda.is_available():
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        work.wait()
                        e1.record()
                        self._events.append((e0, e1))
                    else:
                        work.wait()
                    for i in mine:
                        lo, hi = self._arena_span[i]
                        if wire is not None:
                            wire[1][lo:hi].copy_(wire[0][lo:hi])
                        if not self._deferred("arena"):
                            self._average(self._arena[lo:hi])
                        done.add(i)
                if len(done) < len(self._arena_span):
                    keep.append(ent)
                continue
            if model_index is not None and ent[0] != model_index:
                keep.append(ent)
                continue
            _, work, scale, buf, params, wire = ent
            if self.measure and torch.cuda.is_available():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                work.wait()
                e1.record()
                self._events.append((e0, e1))
            else:
                work.wait()
            if wire is not None:
                wire[1].copy_(wire[0])         # bf16 image -> the fp32 gradient buffer
                scale = wire[1]
            if buf is not None or not self._deferred(ent[0]):
                self._average(scale)
            if buf is not None:
                off = 0
                for p in params:
                    n = p.grad.numel()
                    p.grad.copy_(buf[off:off + n].view_as(p.grad))
                    off += n
        self._pending = keep
        if model_index is None:
            self._bucketed.clear()
        else:
            self._bucketed.discard(model_index)

    def exposed_ms(self, reset: bool = True) -> float:
        """Milliseconds the waiting streams were blocked on gradient exchanges since the last reset (synchronises)."""
        if not self._events:
            return 0.0
        torch.cuda.synchronize()
        t = sum(a.elapsed_time(b) for a, b in self._events)
        if reset:
            self._events = []
        return float(t)

    def all_reduce(self):
        """Exchange every model's gradients and wait.  Afterwards model i's buffer holds the rank mean times
        ``1 / optimizer_scale(i)``: the SUM for fused-Adam models under ``defer_average`` (their update applies the 1/world),
        the mean itself otherwise."""
        for i in range(len(self.segmentators)):
            self.begin(i)
        self.finish()

    def prepare(self):
        """Call once the models sit on their device and before the first step (CoTrainer does): lays the small models'
        gradient buffers out in the shared arena, outside any graph capture."""
        if self.fuse_small:
            self._adopt_arena()
        self._prepared = True


def prefer_ring():
    """Optional: ask RCCL for its ring all-reduce (NCCL_ALGO=Ring unless the user chose) before the process group is created.
    Its f32 SUM kernels hold no packed-FP32 instructions at all, the tree's hold 4-16 packed adds of the default operand form --
    which the probe measured exact beside the conv kernels, so the algorithm choice is left to RCCL by default."""
    import os
    os.environ.setdefault("NCCL_ALGO", "Ring")


def init_from_env(backend: Optional[str] = None):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and initialise the
    process group: 'nccl' (= RCCL on ROCm) when a GPU is visible, else 'gloo'."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return rank, local, world
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world

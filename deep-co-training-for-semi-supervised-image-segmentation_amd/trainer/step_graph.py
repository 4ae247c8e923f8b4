"""HIP-graph replay of the fused co-training step.

The fused step (CoTrainer._run_step_fused) is a fixed launch sequence for fixed batch shapes: ~340 launches for
2 x UNet, ~5000 for 2 x Enet, each costing 15-20 us of Python + ctypes on the host -- more than the kernels take
on an MI355X, so the eager step is launch-bound.  Here the sequence is captured ONCE per step signature into a HIP
graph (torch.cuda.CUDAGraph == hipGraph on ROCm; the per-model streams fork and join inside the capture) and
replayed with one launch per step.

What makes the step replayable -- nothing in a launch may depend on a per-step host value:
  * Adam's bias-correction scalars and step count live on the device (optim.FusedAdam, dct_adam_flat_dev);
  * the dropout call counter lives on the device (dct_dropout_fwd_dev);
  * the loss weights lambda_cot / lambda_adv are read from a device tensor (the ``gscale`` argument of the loss
    backward kernels), refreshed only when a scheduler changes them;
  * the mini-batches are copied into static input buffers before each replay.
Host-side counters that the captured Python code advanced once (optimizer step count, dropout calls) are advanced
by the same amount after every replay.

A signature is captured after WARMUP eager steps (which are ordinary training steps), so lazy initialisation
(flat buffers, weight packs, kernel attributes) has happened and the capture sees the steady-state sequence.
Anything that re-allocates the weights, gradients or moments (load_state_dict, .to()) changes the signature and
leads to a new capture.

Two capture forms (CoTrainer._use_segments):
  * ONE graph (UNet): the per-model streams fork and join inside the capture; only the model streams -- pass streams are
    switched off for such a capture (their cross-stream marks have crashed hipStreamEndCapture / hipGraphLaunch on ROCm 7.2);
  * a PROGRAM of per-stream graphs (trainer/stream_sched.py; Enet, 2 x UNet + FGSM, every data-parallel step): one hipGraphLaunch
    feeds one hardware queue, so chains that should overlap are captured as separate graphs and launched on their own streams,
    with the cross-stream waits and the host callbacks (gradient exchange) replayed between them.

Data parallelism: the RCCL all-reduces are never captured; they are host callbacks of the program (UNet: one per gradient bucket,
issued from inside the backward pass; Enet: one per model).  ``segmented_graphs=False`` falls back to two graphs -- [forwards,
losses, backward passes] and [optimizer steps] -- around one eager (un-bucketed) all-reduce per model.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch



def _is_refused_capture(err: BaseException) -> bool:
    """True for the errors HIP / torch raise when a stream capture is refused or invalidated ("operation not permitted when stream is
    capturing", "capturing stream has unjoined work", "... graph ..."), False for the library's own launch / argument errors
    ("dct_amd: ... failed"), which must reach the user."""
    msg = str(err).lower()
    if "dct_amd:" in msg and "captur" not in msg:
        return False
    return "captur" in msg or "graph" in msg


class _Captured(object):
    __slots__ = ("graph", "graph_opt", "program", "lab", "unl", "out", "counters")


class StepGraphCache(object):
    WARMUP = 2
    MAX_GRAPHS = 8

    def __init__(self, trainer):
        self.tr = trainer
        self._seen: Dict[tuple, int] = {}
        self._graphs: Dict[tuple, _Captured] = {}
        self._lam_dev: Optional[torch.Tensor] = None
        self._lam_host: Tuple[float, float] = (float("nan"), float("nan"))
        self.replays = 0
        self.captures = 0

    # ------------------------------------------------------------------------------ signature
    def _signature(self, lab, unl, train_jsd, train_adv, adv_choice, lam) -> tuple:
        tr = self.tr
        sig: List = [bool(train_jsd), bool(train_adv), tuple(adv_choice) if adv_choice is not None else None,
                     lam[0] != 0.0, lam[1] != 0.0, bool(tr.model_streams), bool(tr.batch_lab_unlab), tr.grad_sync is not None,
                     bool(tr._use_segments()), bool(tr.pass_streams), bool(tr.early_backward), bool(tr.wide_forward),
                     bool(tr.adv_chain_layout)]
        for img, gt in lab:
            sig.append((tuple(img.shape), img.dtype, tuple(gt.shape), gt.dtype))
        if unl is not None:
            sig.append((tuple(unl[0].shape), unl[0].dtype))
        for seg in tr.segmentators:
            net, opt = seg.torchnet, seg.optimizer
            fp = net.flat_params
            sig.append((id(net), net.training, fp.version,
                        fp.flat.data_ptr() if fp.flat is not None else 0,
                        fp.gflat.data_ptr() if fp.gflat is not None else 0,
                        opt._m.data_ptr() if getattr(opt, "_m", None) is not None else 0,
                        getattr(net, "external_dropout_masks", None) is None,
                        bool(getattr(net, "record_dropout_masks", False))))
        return tuple(sig)

    def _lam(self) -> torch.Tensor:
        tr = self.tr
        lam = (float(tr.cot_scheduler.value), float(tr.adv_scheduler.value))
        if self._lam_dev is None or self._lam_dev.device != tr.device:
            self._lam_dev = torch.tensor(lam, dtype=torch.float32, device=tr.device)
            self._lam_host = lam
        elif lam != self._lam_host:              # once per epoch at most: two async fills
            self._lam_dev[0:1].fill_(lam[0])
            self._lam_dev[1:2].fill_(lam[1])
            self._lam_host = lam
        return self._lam_dev

    def _counters(self):
        out = []
        for seg in self.tr.segmentators:
            out.append((seg.optimizer, "_steps"))
            if hasattr(seg.torchnet, "_drop_calls"):
                out.append((seg.torchnet, "_drop_calls"))
        return out

    # ------------------------------------------------------------------------------ run
    def run(self, lab, unl, train_jsd, train_adv, adv_choice) -> dict:
        tr = self.tr
        lam_dev = self._lam()
        for seg in tr.segmentators:
            if hasattr(seg.optimizer, "refresh_lr"):
                seg.optimizer.refresh_lr()
        sig = self._signature(lab, unl, train_jsd, train_adv, adv_choice, self._lam_host)
        cap = self._graphs.get(sig)
        if cap is None:
            n = self._seen.get(sig, 0)
            self._seen[sig] = n + 1
            if n < self.WARMUP or len(self._graphs) >= self.MAX_GRAPHS:
                return tr._run_step_fused(lab, unl, train_jsd, train_adv, adv_choice, lam_dev=lam_dev)
            cap = self._capture(sig, lab, unl, train_jsd, train_adv, adv_choice, lam_dev)
            if cap is None:
                return tr._run_step_fused(lab, unl, train_jsd, train_adv, adv_choice, lam_dev=lam_dev)
            first = True
        else:
            first = False
        # The mini-batches into the static input buffers: ONE multi-tensor copy per dtype instead of 2 S + 1 launches -- the replays are
        # pipelined, and whatever is queued between two graph launches sits on the critical path (tools/phase_stamps.py: 60 us between
        # one step's last optimizer and the next step's first kernel, with five copies and three result clones in it)
        dsts = [t for pair in cap.lab for t in pair] + ([cap.unl[0]] if cap.unl is not None else [])
        srcs = [t for pair in lab for t in pair] + ([unl[0]] if cap.unl is not None else [])
        try:
            torch._foreach_copy_(dsts, srcs, non_blocking=True)
        except (RuntimeError, TypeError, AttributeError):
            for d, s_ in zip(dsts, srcs):
                d.copy_(s_, non_blocking=True)
        if cap.program is not None:            # one graph per stream segment, launched on the segments' own streams
            cap.program.replay()
        else:
            cap.graph.replay()
        if cap.graph_opt is not None:          # data parallelism: exchange the gradients between the two graphs
            for i in range(len(tr.segmentators)):
                tr.grad_sync.begin(i)
            tr.grad_sync.finish()
            cap.graph_opt.replay()
        self.replays += 1
        if not first:                     # the capture pass already advanced the host counters once
            for (obj, name), d in zip(self._counters(), cap.counters):
                setattr(obj, name, getattr(obj, name) + d)
        o = cap.out
        # the small results are copied out (callers may keep them across steps) -- as ONE stacked tensor: one launch between two replays
        # instead of one per scalar; the prediction maps are the graph's static buffers, valid until the next step
        scalars = list(o["sup"]) + [o[k] for k in ("jsd", "adv") if torch.is_tensor(o[k])]
        vals = torch.stack([v.reshape(()) for v in scalars])
        n = len(o["sup"])
        rest = iter(vals[n:])
        return dict(sup=[vals[i] for i in range(n)],
                    jsd=next(rest) if torch.is_tensor(o["jsd"]) else o["jsd"],
                    adv=next(rest) if torch.is_tensor(o["adv"]) else o["adv"],
                    preds=o["preds"], unlab_probs=o["unlab_probs"])

    def _capture(self, sig, lab, unl, train_jsd, train_adv, adv_choice, lam_dev) -> Optional[_Captured]:
        tr = self.tr
        cap = _Captured()
        cap.lab = [(torch.empty_like(img), torch.empty_like(gt)) for img, gt in lab]
        cap.unl = (torch.empty_like(unl[0]), None) if unl is not None else None
        before = [getattr(o, n) for o, n in self._counters()]
        side = [getattr(seg.torchnet, "wgrad_side_stream", None) for seg in tr.segmentators]
        for seg in tr.segmentators:          # a fork nested inside a forked stream crashes hipStreamEndCapture (ROCm 7.2)
            if (tr.model_streams or tr._use_segments()) and hasattr(seg.torchnet, "wgrad_side_stream"):   # (nor may a segment end with unjoined work)
                seg.torchnet.wgrad_side_stream = False
        torch.cuda.synchronize(tr.device)
        graph = torch.cuda.CUDAGraph()
        sync = tr.grad_sync
        cap.graph_opt = None
        cap.program = None
        # no garbage collection while a capture is open: a collected object of an earlier trainer (streams, events, graphs)
        # would be destroyed with HIP calls that are not permitted during capture
        import gc
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            if tr._use_segments():
                from .stream_sched import EagerSchedule, SegmentRecorder
                rec = SegmentRecorder(tr.device)
                tr._sched = rec
                try:
                    rec.start()
                    cap.out = tr._run_step_fused(cap.lab, cap.unl, train_jsd, train_adv, adv_choice, lam_dev=lam_dev)
                    cap.program = rec.finish()
                except RuntimeError as e:
                    # a capture the runtime refuses must not take the training down: nothing was launched while recording, the
                    # host counters the recorded pass advanced are put back, and this signature stays on eager launches
                    rec.abort()
                    tr._sched = EagerSchedule()
                    for (o, n), v in zip(self._counters(), before):
                        setattr(o, n, v)
                    for seg in tr.segmentators:      # weight packs "built" by recorded-only launches do not exist: rebuild next step
                        if hasattr(seg.torchnet, "mark_weights_updated"):
                            seg.torchnet.mark_weights_updated()
                    if not _is_refused_capture(e):
                        raise                         # an argument / kernel error is not a refused capture: do not hide it
                    self._seen[sig] = -(1 << 30)
                    import warnings
                    warnings.warn(f"dct_amd: capturing the step as a program of HIP graphs failed ({e}); this step shape runs eagerly")
                    return None
                except BaseException:
                    rec.abort()
                    raise
                finally:
                    tr._sched = EagerSchedule()
                graph = None
            elif sync is None:
                # ONE graph: only the model streams fork inside it (the layout captured since round 1).  Pass streams add forks whose
                # marks are waited for by other forked streams; hipStreamEndCapture / hipGraphLaunch of such captures have crashed
                # on ROCm 7.2, and the sequential accumulation they replace is bit-identical (tests/test_stream_sched_gpu.py)
                keep_pass, tr.pass_streams = tr.pass_streams, False
                try:
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        cap.out = tr._run_step_fused(cap.lab, cap.unl, train_jsd, train_adv, adv_choice, lam_dev=lam_dev)
                finally:
                    tr.pass_streams = keep_pass
            else:
                # no collective inside a capture: graph 1 ends after the backward passes, graph 2 holds the optimizer steps
                tr.grad_sync, tr._defer_optimizer = None, True
                keep_pass, tr.pass_streams = tr.pass_streams, False
                try:
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        cap.out = tr._run_step_fused(cap.lab, cap.unl, train_jsd, train_adv, adv_choice, lam_dev=lam_dev)
                    tr._defer_optimizer = False
                    cap.graph_opt = torch.cuda.CUDAGraph()
                    tr._opt_phase_sync = sync          # (the exchange leaves a SUM: the update folds its 1/world in)
                    with torch.cuda.graph(cap.graph_opt, capture_error_mode="thread_local"):
                        tr._optimizer_phase(None)      # S small launches on the capture stream
                finally:
                    tr._opt_phase_sync = None
                    tr.grad_sync, tr._defer_optimizer = sync, False
                    tr.pass_streams = keep_pass
        finally:
            if gc_was_on:
                gc.enable()
            for seg, s in zip(tr.segmentators, side):
                if s is not None:
                    seg.torchnet.wgrad_side_stream = s
        cap.graph = graph
        cap.counters = [getattr(o, n) - b for (o, n), b in zip(self._counters(), before)]
        for (o, n), d in zip(self._counters(), cap.counters):
            if n == "_drop_calls" and d % 2:       # the device counter's two words are used in turn (dct_dropout_fwd_dev): replays must stay in turn
                raise RuntimeError("dct_amd: a captured step must hold an even number of dropout launches per network")
        # a re-allocation during the capture (first gradient buffer, moments) would have changed the signature
        if self._signature(lab, unl, train_jsd, train_adv, adv_choice, self._lam_host) != sig:
            raise RuntimeError("dct_amd: weights / gradients / Adam moments were (re)allocated while the step was being "
                               "captured; they would live in the graph's private pool")
        self._graphs[sig] = cap
        self.captures += 1
        return cap

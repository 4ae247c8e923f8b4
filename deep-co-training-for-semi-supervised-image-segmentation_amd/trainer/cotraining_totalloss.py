"""``CoTrainer``: N-model deep co-training with the reference's interface
(/root/reference/generalframework/trainer/cotraining_totalloss.py:28-482).

The per-step body the reference inlines in ``_train_loop`` (:203-248) is factored out as
``_run_step``.  It has two implementations with identical observable behaviour:

* the *fused* path (dct_amd HIP networks + the stock loss modules on a HIP device): network
  forwards are single autograd nodes over hand-written kernels; every loss value and its
  logit-gradient come from one fused kernel each (CE, JSD-from-logits, KL-from-logits), the loss
  weights are folded into those kernels, and ``total.backward()`` becomes ONE
  ``torch.autograd.backward(heads, dlogits)`` call -- same single backward through every graph;
* the *generic* path (any nn.Module / any criterion callables): the reference's op sequence
  through the public module APIs.  Host-logic tests drive it with injected CPU modules.

Same op order as the reference: S supervised forwards, then S unlabeled forwards, then the
FGSM block on the cached batches, then ``zero_grad`` (after the forwards), one backward,
all optimizers step.  The numpy RNG is consumed once per step when ``train_adv``.
"""
from __future__ import annotations

import dataclasses

import contextlib
import os
from operator import itemgetter
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import yaml
from torch import Tensor, nn

from .. import ModelMode
from .. import scheduler
from ..metrics import AverageValueMeter, DiceMeter
from ..models import Segmentator
from ..utils import iterator_, map_, dict_merge, tqdm_
from ..utils.AEGenerator import FSGMGenerator
from .stream_sched import EagerSchedule, StreamDealer, own_stream
from .trainer import Trainer


# HIP streams are shared by every trainer of the process (model stream i, pass stream (i, k)): a test suite or a sweep builds
# dozens of trainers, and stream objects that die with an old trainer would be destroyed whenever the cyclic GC runs --
# including in the middle of a later trainer's graph capture, where hipStreamDestroy is not permitted.
_STREAM_POOL = {}
POOL_STREAMS = os.environ.get("DCT_POOL_STREAMS", "1") != "0"


def _pooled_stream(device, *key, dealer=None):
    """The stream of role ``key`` (("model", i), ("pass", i, k), ...) on ``device``.  With a dealer it comes from the cached,
    bounded set of probe streams (one per hardware queue in turn); otherwise ONE own stream per (device, role) for the life of
    the process -- a long-lived process that builds many trainers (sweeps, the test suite) must not create streams without
    bound: they are never destroyed and all map onto the same four hardware queues.  Two live trainers then share
    their role streams, which only serialises them against each other.  DCT_POOL_STREAMS=0 restores a fresh stream per request."""
    if dealer is not None:
        return dealer.take()
    if not POOL_STREAMS:
        return own_stream(device)
    k = (str(device),) + key
    st = _STREAM_POOL.get(k)
    if st is None:
        st = _STREAM_POOL[k] = own_stream(device)
    return st


def fix_seed(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class _NullWriter(object):
    def add_scalars(self, *a, **k):
        pass


def _make_writer(save_dir):
    try:
        from tensorboardX import SummaryWriter  # optional, as in the reference (:6)
        return SummaryWriter(str(save_dir))
    except Exception:
        return _NullWriter()


@dataclasses.dataclass
class ExecutionPlan:
    """Layout switches of the fused step (none changes a result bit; each is an A/B handle whose default is the measured optimum --
    DESIGN.md 4.3).  ``CoTrainer.<name>`` reads and writes the field of the same name (``tr.model_streams = False`` still works)."""
    batch_lab_unlab: bool = True        # one B_l+B_u pass per batch-independent net (see _run_step_fused)
    model_streams: bool = True          # one HIP stream per model in the fused step (see _streams)
    spread_streams: bool = True         # deal the model / pass streams over different hardware queues (stream_sched.queue_groups)
    pass_streams: bool = True           # nets that support it (Enet): the backward passes of one model run on separate streams
    group_one: bool = True              # labeled and unlabeled passes in ONE group where the batch shapes agree (False: two groups, two queues)
    leaf_offload: bool = True           # ... with the adversarial backward pass's weight gradients on the queue that leaves idle
    group_passes: bool = True           # ... and issue the 2S co-training passes as grouped launches where the networks can (Enet)
    wide_forward: bool = True           # networks with deferred running statistics: lay the step out on four hardware queues (_run_step_wide)
    adv_chain_layout: bool = True       # two batch-independent networks (UNet) with FGSM: see _run_step_adv_chain
    early_backward: bool = True         # start the labeled / unlabeled backward passes right after the JSD, beside the adversarial block
    grad_overwrite: bool = True         # nets that support it: first backward pass of a step writes the gradients (no zero fill)
    segmented_graphs: Optional[bool] = None   # capture the step as one graph per stream segment (different hardware queues) instead of one
                                        # graph with forked streams inside (one queue).  None = by network: those made of many short launches
                                        # ask for it (Enet: 39 -> 22 ms per cfg4 step), those whose kernels fill the chip do not (UNet)
    use_hip_graph: bool = True          # replay the fused step from a captured HIP graph (trainer/step_graph.py)
    ddp_segmented_graph: bool = True    # data parallelism: replay the step as graph segments around the eager all-reduces
    force_loss_scale: Optional[float] = None  # tests: a power of two applied to every loss gradient and divided out by the optimizers


def _plan_property(name):
    return property(lambda self: getattr(self.plan, name), lambda self, value: setattr(self.plan, name, value))


class CoTrainer(Trainer):

    def __init__(self, segmentators: List[Segmentator],
                 labeled_dataloaders: List,
                 unlabeled_dataloader,
                 val_dataloader,
                 criterions: Dict[str, nn.Module],
                 max_epoch: int = 100,
                 save_dir: str = 'tmp',
                 device: str = 'cpu',
                 axises: List[int] = [1, 2, 3],
                 checkpoint: Union[List[str], None] = None,
                 metricname: str = 'metrics.csv',
                 adv_scheduler_dict: dict = None,
                 cot_scheduler_dict: dict = None,
                 adv_training_dict: dict = {},
                 use_tqdm: bool = True,
                 whole_config=None,
                 steps_per_epoch: int = 300,
                 grad_sync=None) -> None:
        self.max_epoch = max_epoch
        self.segmentators = segmentators
        self.labeled_dataloaders = labeled_dataloaders
        self.unlabeled_dataloader = unlabeled_dataloader
        self.val_dataloader = val_dataloader

        # same contract checks as the reference (:54-64)
        assert self.segmentators.__len__() == self.labeled_dataloaders.__len__()
        assert self.segmentators.__len__() >= 1
        assert set(map_(id, self.segmentators)).__len__() == self.segmentators.__len__()
        assert set(map_(id, self.labeled_dataloaders)).__len__() == self.segmentators.__len__()
        assert set(map_(lambda x: x.batch_size, self.labeled_dataloaders)).__len__() == 1
        self.criterions = criterions
        assert set(self.criterions.keys()) == {'jsd', 'sup', 'adv'}

        self.save_dir = Path(save_dir)
        self.save_dir.mkdir(parents=True, exist_ok=True)
        self.writer = _make_writer(self.save_dir)
        if whole_config:
            with open(Path(self.save_dir, 'config.yml'), 'w') as outfile:
                yaml.dump(whole_config, outfile, default_flow_style=False)

        self.device = torch.device(device)
        self.C = self.segmentators[0].arch_params['num_classes']
        self.axises = axises
        self.best_scores = np.zeros(self.segmentators.__len__())
        self.start_epoch = 0
        self.metricname = metricname
        self.n_batch = steps_per_epoch      # the reference hard-codes 300 (:191)

        self.cot_scheduler = getattr(scheduler, cot_scheduler_dict['name'])(
            **{k: v for k, v in cot_scheduler_dict.items() if k != 'name'})
        self.adv_scheduler = getattr(scheduler, adv_scheduler_dict['name'])(
            **{k: v for k, v in adv_scheduler_dict.items() if k != 'name'})
        self.adv_training_dict = adv_training_dict

        if checkpoint is not None:
            self._load_checkpoint(checkpoint)

        self.to(self.device)
        self.use_tqdm = use_tqdm and tqdm_ is not None
        self.grad_sync = grad_sync          # dct_amd.ddp.FlatGradSync or None (single process)
        if grad_sync is not None and hasattr(grad_sync, "prepare"):
            grad_sync.prepare()             # small models' gradient buffers -> one arena, one collective per step
        self.plan = ExecutionPlan()         # how the fused step is laid out on streams / queues / graphs (every switch in one place)
        self._stream_pool = None
        self._sched = EagerSchedule()       # stream operations of the fused step: eager, or recorded (trainer/stream_sched.py)
        self._dealer = None
        self._pass_pool = None
        self._pass_bufs = {}
        self._qstreams = None
        self._step_hint_adv_chain = False
        self._overwrite_models = set()
        self._pass_early = {}
        self._step_graphs = None
        self.last_step = None
        self._defer_optimizer = False       # segmented capture: _finish_step stops after the backward passes (see _optimizer_phase)

    def to(self, device: torch.device):
        [segmentator.to(device) for segmentator in self.segmentators]
        [criterion.to(device) for _, criterion in self.criterions.items() if hasattr(criterion, "to")]

    # ------------------------------------------------------------------------------ epochs
    def start_training(self, train_jsd=False, train_adv=False, save_train=False, save_val=False,
                       augment_labeled_data=False, augment_unlabeled_data=False):
        S = len(self.segmentators)
        metrics = {k: torch.zeros(self.max_epoch, S, self.C, 2, dtype=torch.float)
                   for k in ('train_dice', 'train_unlab_dice', 'val_dice', 'val_batch_dice')}
        for epoch in range(self.start_epoch, self.max_epoch):
            train_dice, train_unlab_dice = self._train_loop(
                labeled_dataloaders=self.labeled_dataloaders, unlabeled_dataloader=self.unlabeled_dataloader,
                epoch=epoch, mode=ModelMode.TRAIN, save=save_train, train_jsd=train_jsd, train_adv=train_adv,
                augment_labeled_data=augment_labeled_data, augment_unlabeled_data=augment_unlabeled_data)
            with torch.no_grad():
                val_dice, val_batch_dice = self._eval_loop(val_dataloader=self.val_dataloader, epoch=epoch,
                                                           mode=ModelMode.EVAL, save=save_val)
            self.schedulerStep()
            for k, v in (('train_dice', train_dice), ('train_unlab_dice', train_unlab_dice),
                         ('val_dice', val_dice), ('val_batch_dice', val_batch_dice)):
                assert metrics[k][epoch].shape == v.shape
                metrics[k][epoch] = v
            if self._is_main():
                self._write_metrics(metrics)
            current_metric = val_batch_dice[:, self.axises, 0].mean(1)
            if self._is_main():
                self.checkpoint(current_metric, epoch)

    def _is_main(self) -> bool:
        return self.grad_sync is None or self.grad_sync.rank == 0

    def _write_metrics(self, metrics):
        for k, v in metrics.items():
            np.save(self.save_dir / f'{k}.npy', v.data.numpy())
        try:
            import pandas as pd
        except Exception:
            return
        for s in range(self.segmentators.__len__()):
            df = pd.DataFrame({
                **{f"train_dice_{i}": metrics["train_dice"][:, s, i, 0] for i in self.axises},
                **{f"train_unlab_dice_{i}": metrics["train_unlab_dice"][:, s, i, 0] for i in self.axises},
                **{f"val_dice_{i}": metrics["val_dice"][:, s, i, 0] for i in self.axises},
                **{f"val_batch_dice_{i}": metrics["val_batch_dice"][:, s, i, 0] for i in self.axises}})
            df.to_csv(Path(self.save_dir, self.metricname.replace('.csv', f'_{s}.csv')), float_format="%.4f",
                      index_label="epoch")

    # ------------------------------------------------------------------------------ the step
    def _fused_ok(self) -> bool:
        from ..loss.loss import CrossEntropyLoss2d, JSD_2D
        if self.device.type != 'cuda':
            return False
        if type(self.criterions['sup']) is not CrossEntropyLoss2d or type(self.criterions['jsd']) is not JSD_2D:
            return False
        return all(hasattr(s.torchnet, "flat_params") and hasattr(s.torchnet, "plan_forward")
                   for s in self.segmentators) and len(self.segmentators) <= 8

    def _draw_adv_choice(self) -> Tuple[int, int]:
        S = len(self.segmentators)
        if self.grad_sync is not None and hasattr(self.grad_sync, "draw_pair"):
            return self.grad_sync.draw_pair(S)       # data parallelism: the same pair on every rank
        try:
            choice = sorted(np.random.choice(list(range(S)), 2, replace=False).tolist())
        except Exception:
            choice = sorted(np.random.choice(list(range(S)), 2, replace=True).tolist())
        return choice[0], choice[1]

    def _run_step(self, lab_batches: Sequence[Tuple[Tensor, Tensor]], unlab_batch: Optional[Tuple[Tensor, Tensor]],
                  train_jsd: bool, train_adv: bool, adv_choice: Optional[Tuple[int, int]] = None) -> dict:
        """One co-training step == reference lines :207-248.  Returns
        dict(sup=[Tensor], jsd=Tensor|0, adv=Tensor|0, preds=[Tensor], unlab_probs=[Tensor])."""
        if self.grad_sync is not None and not getattr(self.grad_sync, "_prepared", True):
            self.grad_sync.prepare()        # (a grad_sync attached after construction) before anything is captured
        if train_adv and adv_choice is None:
            adv_choice = self._draw_adv_choice()
        # the kernels take raw pointers: images fp32, labels int64, both dense (a loader may hand over uint8 labels or
        # half images; the reference's modules would cast or raise, the C ABI would reinterpret the bytes)
        def _img(t):
            return t.to(device=self.device, dtype=torch.float32).contiguous()

        def _gt(t):
            return t.to(device=self.device, dtype=torch.int64).contiguous()
        lab = [(_img(img), _gt(gt)) for img, gt in lab_batches]
        unl = None
        if unlab_batch is not None and (train_jsd or train_adv):
            unl = (_img(unlab_batch[0]), _gt(unlab_batch[1]) if unlab_batch[1] is not None else None)
        self._step_hint_adv_chain = bool(
            self.adv_chain_layout and train_jsd and train_adv and unl is not None and adv_choice is not None and
            len(self.segmentators) == 2 and adv_choice[0] != adv_choice[1] and self.batch_lab_unlab and self.model_streams and
            self.grad_sync is None and
            all(getattr(s.torchnet, "batch_independent", False) and getattr(s.torchnet, "supports_grad_overwrite", False) and
                getattr(s.torchnet, "external_dropout_masks", None) is None for s in self.segmentators))
        if self._fused_ok():
            # replay needs every per-step scalar on the device: only the fused Adam keeps its step count / lr there
            graphable = all(hasattr(s.optimizer, "refresh_lr") and hasattr(s.optimizer, "_steps") for s in self.segmentators)
            # data parallelism: the RCCL all-reduces are not captured.  The step is recorded as a program of graph segments
            # (trainer/stream_sched.py) in which every gradient exchange -- UNet's buckets handed out from inside the backward
            # pass, Enet's one buffer per model -- is a host callback between two segments
            segmented = self.grad_sync is not None and self.ddp_segmented_graph
            if self.use_hip_graph and graphable and (self.grad_sync is None or segmented) and \
                    all(s.torchnet.training for s in self.segmentators):
                if self._step_graphs is None:
                    from .step_graph import StepGraphCache
                    self._step_graphs = StepGraphCache(self)
                out = self._step_graphs.run(lab, unl, train_jsd, train_adv, adv_choice)
            else:
                out = self._run_step_fused(lab, unl, train_jsd, train_adv, adv_choice)
        else:
            out = self._run_step_generic(lab, unl, train_jsd, train_adv, adv_choice)
        self.last_step = out
        return out

    def _streams(self):
        """One HIP stream per model for the fused step, or None.  The S networks are independent between the
        points where the losses couple them (JSD, the FGSM hand-over), and many of their launches cannot fill
        256 CUs alone (deep UNet levels: a few hundred tiles; every kernel's last round of blocks): queued on
        separate streams, the tail of one model's kernel is filled with blocks of the other's.  Ordering is by
        stream events only (wait_stream); no host synchronisation is added."""
        if not self.model_streams or self.device.type != 'cuda' or len(self.segmentators) < 2:
            return None
        if self._stream_pool is None or len(self._stream_pool) != len(self.segmentators):
            self._stream_pool = [_pooled_stream(self.device, "model", i, dealer=self._stream_dealer()) for i in range(len(self.segmentators))]
        return self._stream_pool

    def _use_segments(self) -> bool:
        if self.segmented_graphs is not None:
            return bool(self.segmented_graphs)
        if self._step_hint_adv_chain:
            return True         # two batch-independent nets + FGSM: the adversarial chain gets a hardware queue (_run_step_adv_chain)
        if self.grad_sync is not None and self.ddp_segmented_graph:
            return True         # data parallelism: the gradient exchanges are host callbacks BETWEEN graph segments
        return any(getattr(seg.torchnet, "prefers_segmented_graphs", False) for seg in self.segmentators)

    def _stream_dealer(self):
        if not self.spread_streams or self.device.type != 'cuda' or not self._use_segments():
            return None         # (one captured graph runs on one hardware queue whatever streams were forked inside it)
        if self._dealer is None:
            self._dealer = StreamDealer(self.device)
        return self._dealer

    def _pass_parallel_ok(self, net, model_passes, streams) -> bool:
        return bool(self.pass_streams and streams is not None and 1 < len(model_passes) <= 3 and
                    getattr(net, "supports_pass_streams", False))

    def _pass_streams_for(self, i, n):
        if self._pass_pool is None:
            self._pass_pool = {}
        pool = self._pass_pool.setdefault(i, [])
        while len(pool) < n:
            pool.append(_pooled_stream(self.device, "pass", i, len(pool), dealer=self._stream_dealer()))
        return pool[:n]

    def _pass_buffer(self, i, k, fp):
        """Flat gradient buffer of backward pass k of model i (eager steps reuse it; a capture takes it from the graph's pool)."""
        capturing = torch.cuda.is_current_stream_capturing()
        key = (i, k, fp.total, str(self.device))
        buf = None if capturing else self._pass_bufs.get(key)
        if buf is None:
            buf = torch.empty(fp.total, dtype=torch.float32, device=self.device)
            if not capturing:
                self._pass_bufs[key] = buf
        return buf

    def _start_passes(self, i, net, model_passes, side_streams, bufs):
        """Queue backward passes of model i on ``side_streams`` (which already wait for the passes' inputs), each into its own
        gradient buffer appended to ``bufs``.  -> (flat parameters, bufs)."""
        fp = net.flat_params
        for (tape, dl), st in zip(model_passes, side_streams):
            buf = self._pass_buffer(i, len(bufs), fp)
            bufs.append(buf)
            with self._sched.on(st):
                buf.zero_()
                net.plan_backward(tape, dl, need_dx=False, need_dw=True, grad_buffer=buf)
        return fp, bufs

    # Diagnostic (tools/phase_stamps.py): with ``self.phase_stamps`` = a zeroed int64 device tensor [S, 4, 1 + PHASE_RING], one-thread stamp
    # launches date the phases of the step on each model's stream (site k = 0 forward starts, 1 forward + loss done, 2 backward done,
    # 3 optimizer done) into a ring per site.  They are kernel nodes: a captured step replays them, so pipelined replays leave a timeline.
    phase_stamps = None
    fgsm_shares_encoder = True         # three-queue adversarial step: the FGSM generator's forward pass takes the encoder of b's joint pass from its tape
    adv_chain_late_b = 2            # three-queue adversarial step: model b's backward pass behind the JSD (0), the adversarial batch (1), model a's adversarial forward pass (2)
    PHASE_RING = 64

    def _stamp(self, model: int, k: int):
        buf = self.phase_stamps
        if buf is None:
            return
        from .. import _lib
        _lib.check(_lib.load().dct_stamp(buf[model, k].data_ptr(), self.PHASE_RING, torch.cuda.current_stream(self.device).cuda_stream), "dct_stamp")

    def _finish_step(self, backward_calls, streams=None):
        """zero_grad (after the forwards, :245) -> backward (:246-247) -> [gradient all-reduce] -> step (:248).
        ``backward_calls``: list of (model index or None, callable).  With data parallelism each
        model's all-reduce starts as soon as its backward is enqueued and overlaps the next one.
        With ``streams`` model i's zero_grad / backward / all-reduce / Adam are all queued on streams[i]."""
        def on(i):
            return self._sched.on(streams[i]) if streams is not None and i is not None else contextlib.nullcontext()
        for i, seg in enumerate(self.segmentators):
            if i in self._overwrite_models:      # the first backward pass of this model writes every gradient element
                continue
            with on(i):
                seg.optimizer.zero_grad()
        self._pass_pending = dict(getattr(self, "_pass_early", None) or {})      # passes already queued beside the adversarial block
        for idx, call in backward_calls:
            with on(idx):
                call()
                if idx is not None:
                    self._stamp(idx, 2)
                if self.grad_sync is not None and idx is not None and idx not in self._pass_pending:
                    self._sched.call(lambda idx=idx: self.grad_sync.begin(idx))
        if self._pass_pending:
            # some backward passes ran on their own streams into their own gradient buffers: every stream joins the origin
            # stream (never a forked one: see _run_step_fused), the buffers are added there in pass order, and what follows
            # (gradient exchange, optimizers) is queued on the origin stream
            self._pass_join()
            for idx, (flat, bufs) in sorted(self._pass_pending.items()):
                from .. import hip_ops as K
                if len(bufs) == 1:
                    flat.gflat.copy_(bufs[0])
                elif len(bufs) <= 3 and flat.gflat.is_cuda and flat.gflat.numel() % 4 == 0:
                    K.flat_sum(flat.gflat, bufs[0], bufs[1], bufs[2] if len(bufs) == 3 else None)      # ((lab + unl) + adv), one launch
                else:
                    torch.add(bufs[0], bufs[1], out=flat.gflat)
                    for buf in bufs[2:]:
                        flat.gflat.add_(buf)
                if self.grad_sync is not None:
                    self._sched.call(lambda idx=idx: self.grad_sync.begin(idx))
            streams = None
            self._pass_pending = {}
        if self._defer_optimizer:
            return
        if self.grad_sync is not None and any(idx is None for idx, _ in backward_calls):
            self._sched.call(lambda: self.grad_sync.all_reduce())
        self._optimizer_phase(streams)

    def _optimizer_phase(self, streams=None):
        """[wait for model i's gradient exchange] -> optimizer step (:248), per model on its stream."""
        def on(i):
            return self._sched.on(streams[i]) if streams is not None else contextlib.nullcontext()
        unscale = getattr(self, "_grad_unscale", 1.0)
        sync = self.grad_sync if self.grad_sync is not None else getattr(self, "_opt_phase_sync", None)
        for i, seg in enumerate(self.segmentators):
            with on(i):
                if self.grad_sync is not None:
                    self._sched.call(lambda i=i: self.grad_sync.finish(i))     # model i's all-reduce only: later ones overlap this Adam launch
                if hasattr(seg.optimizer, "grad_scale"):
                    # fused Adam: the inverse loss scale and, under data parallelism, the 1/world of the gradient average
                    # (the exchange is a SUM: ddp.py) are folded into the update
                    seg.optimizer.grad_scale = unscale * (sync.optimizer_scale(i) if hasattr(sync, "optimizer_scale") else 1.0)
                elif unscale != 1.0:
                    flat = getattr(seg.torchnet, "flat_params", None)
                    if flat is not None and flat.grads_attached():
                        flat.gflat.mul_(unscale)
                seg.optimizer.step()
                self._stamp(i, 3)

    def _run_step_generic(self, lab, unl, train_jsd, train_adv, adv_choice) -> dict:
        S = len(self.segmentators)
        supervisedLoss, jsdLoss, advLoss = 0, 0, 0
        sup, preds, unlab_preds = [], [], []
        for i in range(S):
            img, gt = lab[i]
            pred = self.segmentators[i].predict(img, logit=True)
            sup_loss = self.criterions.get('sup')(pred, gt.squeeze(1))
            sup.append(sup_loss.detach())
            preds.append(pred.detach())
            supervisedLoss = supervisedLoss + sup_loss
        if train_jsd:
            unlab_preds = map_(lambda x: x.predict(unl[0], logit=False), self.segmentators)
            jsdLoss = self.criterions.get('jsd')(unlab_preds).mean()
        if train_adv:
            a, b = adv_choice
            advLoss = self._adv_from_batches((self.segmentators[a], self.segmentators[b]), lab[b], unl[0],
                                             **self.adv_training_dict)
        totalLoss = supervisedLoss + self.cot_scheduler.value * jsdLoss + self.adv_scheduler.value * advLoss
        self._grad_unscale = 1.0
        self._finish_step([(None, totalLoss.backward)])
        return dict(sup=sup, jsd=jsdLoss.detach() if train_jsd else 0, adv=advLoss.detach() if train_adv else 0,
                    preds=preds, unlab_probs=[p.detach() for p in unlab_preds])

    def _run_step_fused(self, lab, unl, train_jsd, train_adv, adv_choice, lam_dev=None) -> dict:
        """The step as one launch sequence over the networks' execution plans (plan_forward / plan_backward: no
        autograd graph), the fused loss kernels and the flat Adam.  ``lam_dev`` (float32[2] device tensor holding
        lambda_cot, lambda_adv): the loss weights are then read on the device, as a captured graph needs."""
        from .. import hip_ops as K
        from ..loss.loss import _nchw
        S, C = len(self.segmentators), self.C
        nets = [s.torchnet for s in self.segmentators]
        ignore = self.criterions['sup'].ignore_index
        lam_cot, lam_adv = float(self.cot_scheduler.value), float(self.adv_scheduler.value)
        # fp16 networks: per-pixel gradients of a mean over ~1e6 pixels sit in half's subnormal range, so every loss
        # gradient is scaled by a power of two (>= the pixel count of a labeled batch) and the optimizers divide it out again
        # (include/dct.h, DCT_F16).  1.0 -- and bit-identical arithmetic -- for bf16 / fp32 networks.
        gs = 1.0
        if self.force_loss_scale is not None:
            gs = float(self.force_loss_scale)
        elif any(getattr(n, "compute_dtype", None) == torch.float16 for n in nets):
            gs = float(2 ** min(24, max(10, (lab[0][0].shape[0] * lab[0][0].shape[2] * lab[0][0].shape[3] - 1).bit_length())))
        g_cot = dict(gscale=lam_dev[0:1], gmul=gs) if lam_dev is not None else dict(gmul=lam_cot * gs)
        g_adv = dict(gscale=lam_dev[1:2], gmul=gs) if lam_dev is not None else dict(gmul=lam_adv * gs)
        self._grad_unscale, self._loss_scale = 1.0 / gs, gs
        passes: List[List[Tuple[object, Tensor]]] = [[] for _ in range(S)]     # per model: (tape, dlogits) to back-propagate
        sup, preds = [], []
        # Networks whose samples do not interact (UNet: no BatchNorm) run the labeled and the
        # unlabeled batch as ONE pass of B_l + B_u images: same per-pixel results, twice the GEMM
        # rows per launch and half the launches.  Nets with batch statistics keep separate passes
        # (three separate BN-statistics batches per model per step).
        fuse = bool(train_jsd and unl is not None and self.batch_lab_unlab and
                    all(getattr(n, "batch_independent", False) and
                        getattr(n, "external_dropout_masks", None) is None for n in nets))
        streams = self._streams()
        main = torch.cuda.current_stream(self.device)
        if self._wide_ok(nets, streams, train_jsd, unl, fuse):
            return self._run_step_wide(lab, unl, train_adv, adv_choice, nets, gs, g_cot, g_adv, lam_cot, lam_adv, ignore)
        if fuse and self._adv_chain_ok(nets, streams, train_jsd, train_adv, adv_choice, unl):
            return self._run_step_adv_chain(lab, unl, adv_choice, nets, gs, g_cot, g_adv, lam_cot, lam_adv, ignore)

        def on(i):
            return self._sched.on(streams[i]) if streams is not None else contextlib.nullcontext()

        # backward-pass streams (two per model that supports them): like the model streams they enter a capture only through a
        # wait on the origin stream -- a fork nested inside a forked stream crashes hipStreamEndCapture on ROCm 7.2 -- so they
        # are forked and joined together with the model streams and idle until the backward passes are queued
        side = []
        if streams is not None and self.pass_streams:
            for i in range(S):
                if getattr(nets[i], "supports_pass_streams", False):
                    side += self._pass_streams_for(i, 2)

        def fork():
            if streams is not None:
                self._sched.wait([(st, main) for st in list(streams) + side])

        def join():
            if streams is not None:
                self._sched.wait([(main, st) for st in list(streams) + side])
        fork()
        joined_after_forwards = False       # becomes True at the JSD join: labeled / unlabeled pass inputs are then final on main
        self._pass_join = join
        full = []                                                              # fuse: (tape, logits, dlogits) of the joint pass
        for i in range(S):                                                     # :208-218
            with on(i):
                self._stamp(i, 0)
                img, gt = lab[i]
                B_l = img.shape[0]
                if fuse:
                    lp_all, tape = nets[i].plan_forward(torch.cat((img, unl[0]), dim=0), True)
                    dl_all = torch.empty_like(lp_all)
                    full.append((tape, lp_all, dl_all))
                    lp, dl_out = lp_all[:B_l], dl_all[:B_l]
                else:
                    lp, tape = nets[i].plan_forward(img, True)
                    dl_out = torch.empty_like(lp)
                    passes[i].append((tape, dl_out))
                t = gt.reshape(-1)
                out = K.ce_step(lp, t, C, dl_out, gmul=gs, ignore_index=ignore)       # loss value + count and the logit gradient: two launches
                sup.append(out[0])
                preds.append(_nchw(lp))
                self._stamp(i, 1)
        jsd, unlab_probs = 0, []
        if train_jsd:                                                          # :219-227
            if fuse:
                B_l = lab[0][0].shape[0]
                lps = [f[1][B_l:] for f in full]
                dl_outs = [f[2][B_l:] for f in full]
            else:
                lps, dl_outs, utapes = [], [], []
                for i in range(S):
                    with on(i):
                        lp_u, tape = nets[i].plan_forward(unl[0], True)
                        lps.append(lp_u)
                        utapes.append(tape)
                        dl_outs.append(torch.empty_like(lp_u))
            joined_after_forwards = True
            join()                                  # the JSD couples all S models: main stream, then fork again
            # value, the S softmax maps and the S logit gradients in ONE pass over the logits (dct_jsd_logits_step: bit for bit the five separate launches)
            jsd1, probs = K.jsd_logits_step(lps, C, dl_outs if lam_cot != 0.0 else None, True, **g_cot)
            jsd = jsd1[0]
            unlab_probs = [_nchw(p_) for p_ in probs]
            if lam_cot != 0.0:
                if not fuse:
                    for i in range(S):
                        passes[i].append((utapes[i], dl_outs[i]))
            elif fuse:
                for d in dl_outs:
                    d.zero_()
            fork()
        for i, f in enumerate(full):
            passes[i].append((f[0], f[2]))
        # Early backward.  After the JSD the labeled and unlabeled logit gradients are final, and the adversarial block that
        # follows is one dependent chain (FGSM forward + input gradient on model b, then model a's forward on the perturbed
        # batch) that leaves every other hardware queue idle.  Nets whose passes write their own gradient buffers (pass
        # streams) start those two backward passes now, on their pass streams; the adversarial pass joins them later on the
        # model's stream.  Nothing the adversarial block writes is read by them (weights are constant until the optimizers,
        # saved tensors and BatchNorm batch statistics are per pass, running statistics are only touched by forwards).  The
        # sum order of the pass buffers, ((lab + unl) + adv), is unchanged.  The tapes stay referenced until the step ends:
        # they were allocated on the model's stream, which keeps allocating while the pass streams still read them.
        self._pass_early = {}
        keep_alive = []
        adv = 0
        if train_adv:                                                          # :233-244 -> :371-392
            a, b = adv_choice
            eps = float(self.adv_training_dict.get('eplision', 0.05))
            img_b, gt_b = lab[b]
            with on(b):
                x = torch.cat((img_b, unl[0]), dim=0)
                x_adv, noise, lp_real, _ = self._fgsm_fused(nets[b], x, gt_b, eps, ignore)
            # (queued after the FGSM chain and its completion mark, and before model a's wait for that mark: streams that share
            # a hardware queue run in issue order -- the critical chain goes first, its mark must not land behind another
            # stream's segment, and nothing may be parked behind model a's blocked wait)
            fgsm_done = self._sched.record(streams[b]) if streams is not None and a != b else None
            if train_adv and self.early_backward and joined_after_forwards and streams is not None and self.pass_streams:
                for i in range(S):
                    if (1 <= len(passes[i]) <= 2 and getattr(nets[i], "supports_pass_streams", False) and
                            nets[i].flat_params.grads_attached() and (self.grad_sync is None or not hasattr(nets[i], "grad_bucket_ranges"))):
                        self._pass_early[i] = self._start_passes(i, nets[i], passes[i], self._pass_streams_for(i, 2), [])
                        keep_alive.append(list(passes[i]))
                        passes[i].clear()
            if fgsm_done is not None:
                self._sched.wait_event(streams[a], fgsm_done)  # the adversarial images and the detached target come from model b
            with on(a):
                lp_adv, tape = nets[a].plan_forward(x_adv, True)
                adv = K.kl_logits_fwd(lp_adv, lp_real, C)[0]
                if lam_adv != 0.0:
                    da = K.kl_logits_bwd(lp_adv, lp_real, C, torch.empty_like(lp_adv), **g_adv)
                    passes[a].append((tape, da))

        def pass_parallel(i):
            """The backward passes of model i on separate streams, each into its own flat gradient buffer (Enet: 1.45 MB): all
            but the last on side streams, the last (the adversarial pass when there is one -- the only pass whose inputs are
            queued on the model streams after the last fork) on the model's stream.  After the join the buffers are summed into
            the gradient buffer on the origin stream in pass order (_finish_step): ((lab + unl) + adv), bit for bit what the
            in-place accumulation of sequential passes produces."""
            net, fp = nets[i], nets[i].flat_params
            cur = torch.cuda.current_stream(self.device)
            started = self._pass_early.get(i)
            bufs = list(started[1]) if started is not None else []
            extra = self._pass_streams_for(i, 2)[len(bufs):]
            side_passes, last = (passes[i][:-1], passes[i][-1]) if passes[i] else ([], None)
            if side_passes:
                if not joined_after_forwards:      # the passes' inputs were produced on the model stream after the last fork
                    self._sched.wait([(extra[k], cur) for k in range(len(side_passes))])
                _, bufs = self._start_passes(i, net, side_passes, extra, bufs)
            if last is not None:
                buf = self._pass_buffer(i, len(bufs), fp)
                bufs.append(buf)
                buf.zero_()
                net.plan_backward(last[0], last[1], need_dx=False, need_dw=True, grad_buffer=buf)
            self._pass_pending[i] = (fp, bufs)
            passes[i].clear()

        def backward_of(i):
            def run():
                if i in self._pass_early or (self._pass_parallel_ok(nets[i], passes[i], streams) and
                                             nets[i].flat_params.grads_attached()):
                    return pass_parallel(i)
                # data parallelism: during the LAST backward pass of a model its gradient buckets go out as they
                # complete (earlier passes only accumulate)
                ranges = nets[i].grad_bucket_ranges() if (self.grad_sync is not None and hasattr(nets[i], "grad_bucket_ranges")) else None
                for k, (tape, dl) in enumerate(passes[i]):
                    if ranges is not None and k == len(passes[i]) - 1:
                        nets[i]._grad_hook = lambda b, i=i, r=ranges: self._sched.call(
                            lambda: self.grad_sync.begin_bucket(i, r[b][0], r[b][1]))
                    try:
                        if i in self._overwrite_models:
                            nets[i].plan_backward(tape, dl, need_dx=False, need_dw=True, overwrite=(k == 0))
                        else:
                            nets[i].plan_backward(tape, dl, need_dx=False, need_dw=True)
                    finally:
                        if ranges is not None:
                            nets[i]._grad_hook = None
                passes[i].clear()
            return run
        # nets whose every parameter gets a gradient in every pass, with the flat gradient buffer already attached:
        # the first pass overwrites instead of zero_grad + accumulate (the reference's zero_grad at :245 has the same effect)
        self._overwrite_models = {i for i in range(S) if self.grad_overwrite and passes[i] and
                                  getattr(nets[i], "supports_grad_overwrite", False) and nets[i].flat_params.grads_attached()}
        # (pass-parallel models write the whole gradient buffer as the sum of their pass buffers: no zero fill either)
        self._overwrite_models |= {i for i in range(S) if self._pass_parallel_ok(nets[i], passes[i], streams) and
                                   nets[i].flat_params.grads_attached()}
        self._overwrite_models |= set(self._pass_early)
        try:
            self._finish_step([(i, backward_of(i)) for i in range(S)], streams)
        finally:
            self._overwrite_models = set()
            self._pass_early = {}
            self._pass_join = None          # (a closure over this step's tapes and streams: not kept past the step)
        join()
        del keep_alive
        return dict(sup=sup, jsd=jsd, adv=adv, preds=preds, unlab_probs=unlab_probs)

    # ------------------------------------------------------------------------------ the step on four hardware queues
    def _wide_ok(self, nets, streams, train_jsd, unl, fuse) -> bool:
        """Every network defers its running statistics and writes per-pass gradient buffers, the gradient buffers exist, and the
        device has four distinguishable hardware queues."""
        if not (self.wide_forward and streams is not None and self.pass_streams and train_jsd and unl is not None and not fuse):
            return False
        if torch.cuda.is_current_stream_capturing() and not self._sched.capturing:
            return False        # ONE graph being captured: the JSD's join into a forked stream crashes hipStreamEndCapture (ROCm 7.2)
        if not all(getattr(n, "supports_deferred_running_stats", False) and getattr(n, "supports_pass_streams", False) and
                   n.training and n.flat_params.grads_attached() for n in nets):
            return False
        return self._queue_streams() is not None

    def _queue_streams(self):
        """One stream per hardware queue (stream_sched.queue_groups), or None when the probe found fewer than four."""
        if self._qstreams is None:
            from .stream_sched import queue_groups
            groups = queue_groups(self.device)
            self._qstreams = [g[0] for g in groups][:4] if len(groups) >= 4 else False
        return self._qstreams or None

    def _run_step_wide(self, lab, unl, train_adv, adv_choice, nets, gs, g_cot, g_adv, lam_cot, lam_adv, ignore) -> dict:
        """The step laid out on the device's four hardware queues (JSD on, networks with deferred running statistics).

        With the running-statistics updates taken out of the forward passes (arch/enet.py::plan_forward(defer_running=True); they
        are applied at the end in the reference's order: labeled, unlabeled, FGSM, adversarial) the 2S + 2 forward passes of a step
        have no order among themselves, and with per-pass gradient buffers neither have the backward passes.  What remains are
        the data dependencies, i.e. two chains:
          adversarial: FGSM forward + input gradient on model b -> forward of model a on the perturbed batch -> its backward;
          co-training: 2S labeled / unlabeled forwards -> JSD -> their 2S backward passes.
        The adversarial chain gets a queue of its own and is issued first (it is the longer one: 13 ms of launches against
        2 + 4 per co-training pass for 2 x Enet, tools/probe_step_program.py); the 2S passes share the other three, each pass
        staying on one queue from its forward to its backward.  Streams that share a hardware queue run in issue order, so a
        'stream' here IS a queue: one per group of stream_sched.queue_groups.  Results are bit for bit those of the sequential
        step: same kernels on the same operands, gradient buffers summed in pass order ((lab + unl) + adv)."""
        from .. import hip_ops as K
        from ..loss.loss import _nchw
        S, C = len(nets), self.C
        sched = self._sched
        main = torch.cuda.current_stream(self.device)
        Q = self._queue_streams()
        adv_q = Q[3] if train_adv else None
        free = Q[:3] if train_adv else Q
        lab_q = [free[i % len(free)] for i in range(S)]
        unl_q = [free[(S + i) % len(free)] for i in range(S)]
        sched.wait([(st, main) for st in Q])
        self._pass_join = lambda: sched.wait([(main, st) for st in Q])
        bufs = [[None, None, None] for _ in range(S)]          # per model: gradient buffers of the (lab, unl, adv) passes
        tapes = [[] for _ in range(S)]                         # per model: tapes in the reference's forward order
        fp = [n.flat_params for n in nets]

        def backward(i, k, tape, dl):
            buf = bufs[i][k] = self._pass_buffer(i, k, fp[i])
            buf.zero_()
            nets[i].plan_backward(tape, dl, need_dx=False, need_dw=True, grad_buffer=buf)

        adv, adv_tapes = 0, {}
        grouped = self._group_passes_ok(nets, lab, unl)
        leaf_keep = None
        if train_adv:                                                          # :233-244 -> :371-392
            a, b = adv_choice
            eps = float(self.adv_training_dict.get('eplision', 0.05))
            with sched.on(adv_q):
                x = torch.cat((lab[b][0], unl[0]), dim=0)
                x_adv, noise, lp_real, ftape = self._fgsm_fused(nets[b], x, lab[b][1], eps, ignore, defer_running=True)
                lp_adv, atape = nets[a].plan_forward(x_adv, True, defer_running=True)
                adv = K.kl_logits_fwd(lp_adv, lp_real, C)[0]
                if lam_adv != 0.0:
                    da = K.kl_logits_bwd(lp_adv, lp_real, C, torch.empty_like(lp_adv), **g_adv)
                    if grouped and self.leaf_offload:
                        # the chain ends with model a's backward pass: its weight gradients (a third of its launches, leaves of
                        # the data-gradient chain) go to the queue the grouped co-training passes leave idle, a few blocks at a time
                        side_q = free[2]
                        buf = bufs[a][2] = self._pass_buffer(a, 2, fp[a])
                        buf.zero_()
                        with K.LeafSide() as side:
                            def leaves_out():
                                ev = sched.record(adv_q)
                                sched.wait_event(side_q, ev)
                                with sched.on(side_q):
                                    side.flush()
                            nets[a].plan_backward(atape, da, need_dx=False, need_dw=True, grad_buffer=buf, leaf_hook=leaves_out)
                            leaves_out()
                        leaf_keep = side.kept          # referenced until the queues are joined (end of this function)
                    else:
                        backward(a, 2, atape, da)
            adv_tapes = {"fgsm": (b, ftape), "adv": (a, atape)}
        if grouped:
            out = self._wide_grouped_tail(lab, unl, train_adv, nets, gs, g_cot, lam_cot, ignore, free, bufs, tapes, fp, adv, adv_tapes)
            del leaf_keep
            return out
        sup, preds, lab_pass = [], [], []
        for i in range(S):                                                     # :208-218
            with sched.on(lab_q[i]):
                img, gt = lab[i]
                lp, tape = nets[i].plan_forward(img, True, defer_running=True)
                dl = torch.empty_like(lp)
                t = gt.reshape(-1)
                out = K.ce_step(lp, t, C, dl, gmul=gs, ignore_index=ignore)       # loss value + count and the logit gradient: two launches
                sup.append(out[0])
                preds.append(_nchw(lp))
                lab_pass.append((tape, dl))
                tapes[i].append(tape)
        lps, dl_outs, utapes = [], [], []
        for i in range(S):                                                     # :219-227
            with sched.on(unl_q[i]):
                lp_u, tape = nets[i].plan_forward(unl[0], True, defer_running=True)
                lps.append(lp_u)
                utapes.append(tape)
                dl_outs.append(torch.empty_like(lp_u))
                tapes[i].append(tape)
        # the JSD couples all S models.  It runs on the first co-training queue, NOT on the origin stream: the origin stream
        # sits on one of the four hardware queues too -- here the adversarial chain's, behind which the JSD would wait 10 ms
        jq = free[0]
        sched.wait([(jq, st) for st in free[1:]])
        with sched.on(jq):
            # value, the S softmax maps and the S logit gradients in ONE pass over the logits (dct_jsd_logits_step: bit for bit the five separate launches)
            jsd1, probs = K.jsd_logits_step(lps, C, dl_outs if lam_cot != 0.0 else None, True, **g_cot)
            jsd = jsd1[0]
            unlab_probs = [_nchw(p_) for p_ in probs]
        sched.wait([(st, jq) for st in free[1:]])
        for i in range(S):
            with sched.on(lab_q[i]):
                backward(i, 0, *lab_pass[i])
        if lam_cot != 0.0:
            for i in range(S):
                with sched.on(unl_q[i]):
                    backward(i, 1, utapes[i], dl_outs[i])
        if train_adv:
            tapes[adv_tapes["fgsm"][0]].append(adv_tapes["fgsm"][1])
            tapes[adv_tapes["adv"][0]].append(adv_tapes["adv"][1])
        self._pass_early = {i: (fp[i], [bf for bf in bufs[i] if bf is not None]) for i in range(S)}
        self._overwrite_models = set(range(S))
        try:
            self._finish_step([], None)             # join, sum the pass buffers in order, [gradient exchange], optimizers
        finally:
            self._overwrite_models = set()
            self._pass_early = {}
            self._pass_join = None
        for i in range(S):                          # running statistics: labeled, unlabeled, FGSM, adversarial (reference order)
            nets[i].apply_running_updates(tapes[i])
        return dict(sup=sup, jsd=jsd, adv=adv, preds=preds, unlab_probs=unlab_probs)

    def _group_passes_ok(self, nets, lab, unl) -> bool:
        """The 2S co-training passes as grouped launches (include/dct.h "grouped passes"): networks whose plan records into a
        K.PassGroup, equal labeled batch shapes, and no more members per group than the library packs into one launch."""
        from .. import hip_ops as K
        if not (self.group_passes and all(getattr(n, "supports_pass_groups", False) for n in nets)):
            return False
        if len({tuple(b[0].shape) for b in lab}) != 1:
            return False
        return len(nets) <= K.group_max()

    def _wide_grouped_tail(self, lab, unl, train_adv, nets, gs, g_cot, lam_cot, ignore, free, bufs, tapes, fp, adv, adv_tapes) -> dict:
        """Co-training half of `_run_step_wide` with the 2S passes grouped: the S labeled and the S unlabeled forward passes (one
        group of 2S when the two batch shapes agree, else two groups of S on two queues) are ONE chain of ~210 launches instead
        of 2S chains sharing three queues, and so are the 2S backward passes (~530 launches).  Same kernel bodies on the same
        operands, same per-pass gradient buffers summed in the same order: bit-identical to the ungrouped layout."""
        from .. import hip_ops as K
        from ..loss.loss import _nchw
        S, C = len(nets), self.C
        sched = self._sched
        one = self.group_one and tuple(lab[0][0].shape) == tuple(unl[0].shape) and 2 * S <= K.group_max()
        q_lab, q_unl = free[0], (free[0] if one else free[1])
        for n in nets:
            n.flat_params.ensure()
        lab_out, unl_out = [None] * S, [None] * S

        def forward_group(members):
            with K.PassGroup(len(members)) as grp:
                for m, (i, kind) in enumerate(members):
                    grp.member(m)
                    x = lab[i][0] if kind == 0 else unl[0]
                    lp, tape = nets[i].plan_forward(x, True, defer_running=True)
                    (lab_out if kind == 0 else unl_out)[i] = (lp, tape)

        lab_members = [(i, 0) for i in range(S)]
        unl_members = [(i, 1) for i in range(S)]
        if one:
            with sched.on(q_lab):
                forward_group(lab_members + unl_members)
        else:
            with sched.on(q_lab):
                forward_group(lab_members)
            with sched.on(q_unl):
                forward_group(unl_members)
        for i in range(S):                      # tapes per model in the reference's forward order: labeled, unlabeled
            tapes[i].append(lab_out[i][1])
            tapes[i].append(unl_out[i][1])
        sup, preds, dls = [], [], []
        with sched.on(q_lab):                                                  # :208-218
            for i in range(S):
                lp, gt = lab_out[i][0], lab[i][1]
                dl = torch.empty_like(lp)
                t = gt.reshape(-1)
                out = K.ce_step(lp, t, C, dl, gmul=gs, ignore_index=ignore)       # loss value + count and the logit gradient: two launches
                sup.append(out[0])
                preds.append(_nchw(lp))
                dls.append(dl)
        lps = [unl_out[i][0] for i in range(S)]
        dl_outs = [torch.empty_like(lp) for lp in lps]
        with sched.on(q_unl):                                                  # :219-227
            # value, the S softmax maps and the S logit gradients in ONE pass over the logits (dct_jsd_logits_step: bit for bit the five separate launches)
            jsd1, probs = K.jsd_logits_step(lps, C, dl_outs if lam_cot != 0.0 else None, True, **g_cot)
            jsd = jsd1[0]
            unlab_probs = [_nchw(p_) for p_ in probs]

        def backward_group(members):
            # what does not record (zero fill of the pass buffers, the cast of the loss gradients) goes first
            prepared = []
            for i, kind in members:
                buf = bufs[i][kind] = self._pass_buffer(i, kind, fp[i])
                buf.zero_()
                dl = dls[i] if kind == 0 else dl_outs[i]
                cd = nets[i].compute_dtype
                if dl.dtype != cd:
                    dl = K.cast(dl, torch.empty(dl.shape, dtype=cd, device=dl.device))
                prepared.append((i, kind, buf, dl))
            with K.PassGroup(len(members)) as grp:
                for m, (i, kind, buf, dl) in enumerate(prepared):
                    grp.member(m)
                    tape = (lab_out if kind == 0 else unl_out)[i][1]
                    nets[i].plan_backward(tape, dl, need_dx=False, need_dw=True, grad_buffer=buf)

        unl_bwd = unl_members if lam_cot != 0.0 else []
        if one:
            with sched.on(q_lab):
                backward_group(lab_members + unl_bwd)
        else:
            with sched.on(q_lab):
                backward_group(lab_members)
            if unl_bwd:
                with sched.on(q_unl):
                    backward_group(unl_bwd)
        if train_adv:
            tapes[adv_tapes["fgsm"][0]].append(adv_tapes["fgsm"][1])
            tapes[adv_tapes["adv"][0]].append(adv_tapes["adv"][1])
        self._pass_early = {i: (fp[i], [bf for bf in bufs[i] if bf is not None]) for i in range(S)}
        self._overwrite_models = set(range(S))
        try:
            self._finish_step([], None)             # join, sum the pass buffers in order, [gradient exchange], optimizers
        finally:
            self._overwrite_models = set()
            self._pass_early = {}
            self._pass_join = None
        for i in range(S):                          # running statistics: labeled, unlabeled, FGSM, adversarial (reference order)
            nets[i].apply_running_updates(tapes[i])
        return dict(sup=sup, jsd=jsd, adv=adv, preds=preds, unlab_probs=unlab_probs)

    def _adv_chain_ok(self, nets, streams, train_jsd, train_adv, adv_choice, unl) -> bool:
        if not (self._step_hint_adv_chain and streams is not None and self.grad_overwrite):
            return False
        if torch.cuda.is_current_stream_capturing() and not self._sched.capturing:
            return False        # ONE graph being captured: joins into forked streams crash hipStreamEndCapture (ROCm 7.2)
        if not all(n.training and n.flat_params.grads_attached() for n in nets):
            return False
        return self._queue_streams() is not None

    def _run_step_adv_chain(self, lab, unl, adv_choice, nets, gs, g_cot, g_adv, lam_cot, lam_adv, ignore) -> dict:
        """Two batch-independent networks (UNet), CE + JSD + FGSM, on three hardware queues.

        In the sequential layout the adversarial block is a tail of its own: both joint forwards -> JSD -> FGSM forward + input
        gradient on model b ALONE (2.6 of a 12.6 ms cfg3 step) -> model a's forward on the perturbed batch ALONE (1.2 ms) -> the
        backward passes, model a's two one after the other while model b's queue idles (tools/probe_step_program.py cfg3).  But the
        FGSM chain needs nothing from the JSD: it is queued on model b's queue right behind b's joint forward, followed there by
        model a's adversarial forward; the JSD and model b's backward pass take a third queue; model a's queue runs its forward,
        its backward, then -- once the adversarial forward has arrived -- the adversarial backward, accumulating as before.
        Per network the order of forward passes (dropout counter) and of gradient accumulation is the sequential one, so the
        results are bit for bit the same.  Optimizers: model b's waits for the adversarial chain (the last reader of b's weights)."""
        from .. import hip_ops as K
        from ..loss.loss import _nchw
        C = self.C
        a, b = adv_choice
        sched = self._sched
        main = torch.cuda.current_stream(self.device)
        Q = self._queue_streams()
        qa, qb, qj = Q[0], Q[1], Q[2]
        used = (qa, qb, qj)
        q_of = {a: qa, b: qb}
        sched.wait([(st, main) for st in used])
        full, sup, preds, fwd_done = {}, [None, None], [None, None], {}
        B_l = lab[0][0].shape[0]
        # The FGSM generator's clean forward pass of model b runs over the batch of b's joint pass with the same weights: everything in front of the
        # first dropout -- stem + eight encoder convolutions, half of a forward pass, on the chain that gates the rest of the step -- is taken from
        # the joint pass's tape instead of computed again (UNet.plan_forward(reuse=...); same kernels' outputs, so the same bits).
        share_encoder = bool(self.fgsm_shares_encoder and getattr(nets[b], "supports_forward_reuse", False) and nets[b].training)
        joint_x_b = None
        # (tools/phase_stamps.py --config cfg3: row a = forward starts / forward + loss done / backward done / optimizer done (behind the adversarial
        #  backward pass), row b = forward starts / adversarial batch ready / adversarial forward done / optimizer done (third queue, behind b's backward))
        for i in (b, a):                                                       # :208-227 (joint labeled + unlabeled pass)
            with sched.on(q_of[i]):
                self._stamp(0 if i == a else 1, 0)
                img, gt = lab[i]
                xj = torch.cat((img, unl[0]), dim=0)
                if i == b and share_encoder:
                    joint_x_b = xj
                    lp_all, tape = nets[i].plan_forward(xj, True, keep_predrop=True)
                else:
                    lp_all, tape = nets[i].plan_forward(xj, True)
                dl_all = torch.empty_like(lp_all)
                full[i] = (tape, lp_all, dl_all)
                lp, t = lp_all[:B_l], gt.reshape(-1)
                out = K.ce_step(lp, t, C, dl_all[:B_l], gmul=gs, ignore_index=ignore)       # loss value + count and the logit gradient: two launches
                sup[i] = out[0]
                preds[i] = _nchw(lp)
                if i == a:
                    self._stamp(0, 1)
            fwd_done[i] = sched.record(q_of[i])
        eps = float(self.adv_training_dict.get('eplision', 0.05))              # :233-244 -> :371-392
        with sched.on(qb):
            x = joint_x_b if joint_x_b is not None else torch.cat((lab[b][0], unl[0]), dim=0)
            x_adv, noise, lp_real, _ = self._fgsm_fused(nets[b], x, lab[b][1], eps, ignore, reuse=full[b][0] if joint_x_b is not None else None)
            self._stamp(1, 1)
        fgsm_done = sched.record(qb)
        sched.wait_event(qb, fwd_done[a])           # model a's passes keep their order (dropout counter, weight packs)
        da = None
        with sched.on(qb):
            lp_adv, atape = nets[a].plan_forward(x_adv, True)
            adv = K.kl_logits_fwd(lp_adv, lp_real, C)[0]
            if lam_adv != 0.0:
                da = K.kl_logits_bwd(lp_adv, lp_real, C, torch.empty_like(lp_adv), **g_adv)
            self._stamp(1, 2)
        adv_done = sched.record(qb)
        sched.wait_event(qj, fwd_done[a])
        sched.wait_event(qj, fwd_done[b])
        with sched.on(qj):
            lps = [full[i][1][B_l:] for i in range(2)]
            dl_outs = [full[i][2][B_l:] for i in range(2)]
            # value, the S softmax maps and the S logit gradients in ONE pass over the logits (dct_jsd_logits_step: bit for bit the five separate launches)
            jsd1, probs = K.jsd_logits_step(lps, C, dl_outs if lam_cot != 0.0 else None, True, **g_cot)
            jsd = jsd1[0]
            unlab_probs = [_nchw(p_) for p_ in probs]
            if lam_cot == 0.0:
                for d in dl_outs:
                    d.zero_()
        jsd_done = sched.record(qj)
        if int(self.adv_chain_late_b) == 2:
            # (behind model a's adversarial FORWARD pass: with the FGSM chain shortened by the shared encoder this is another 0.5 % ahead -- 8.92 against
            #  8.97 ms, three rounds; behind the JSD: 9.19)
            sched.wait_event(qj, adv_done)
        elif self.adv_chain_late_b:
            # Model b's backward pass waits for the adversarial BATCH: until then the FGSM chain shares the device with model a's backward pass only and
            # delivers sooner; behind it, model a's adversarial forward + backward -- 3.1 ms that used to run alone (tools/phase_stamps.py --config
            # cfg3: 37 % of the step at depth 1) -- have model b's backward pass beside them.  Same launches, same order per network: same bits.
            # cfg3 10.05 -> 9.74 ms (-3.1 %, three rounds on one box).  Gated on the FGSM generator's FORWARD pass instead: 9.93; on model a's
            # adversarial forward pass: 9.72 (level with this).
            sched.wait_event(qj, fgsm_done)
        with sched.on(qj):                          # model b's backward: its own queue is busy with the adversarial chain
            nets[b].plan_backward(full[b][0], full[b][2], need_dx=False, need_dw=True, overwrite=True)
        sched.wait_event(qa, jsd_done)
        with sched.on(qa):
            nets[a].plan_backward(full[a][0], full[a][2], need_dx=False, need_dw=True, overwrite=True)
            self._stamp(0, 2)
        sched.wait_event(qa, adv_done)
        if da is not None:
            with sched.on(qa):
                nets[a].plan_backward(atape, da, need_dx=False, need_dw=True)
        sched.wait_event(qj, adv_done)              # the adversarial chain read model b's weights
        opt_streams = [None, None]
        opt_streams[a], opt_streams[b] = qa, qj
        self._optimizer_phase(opt_streams)
        sched.wait([(main, st) for st in used])
        return dict(sup=sup, jsd=jsd, adv=adv, preds=preds, unlab_probs=unlab_probs)

    def _fgsm_fused(self, net, x, gt, eps, ignore, defer_running=False, reuse=None):
        """FSGMGenerator (AEGenerator.py:16-51) on the fused kernels: forward, pseudo-label the
        unlabeled tail, CE, backward to the input only, x + eps*sign(g).  Returns the physical
        NHWC logits of the clean pass (their softmax is the detached KL target) and the tape."""
        from .. import hip_ops as K
        C = self.C
        if reuse is not None:
            lp, tape = net.plan_forward(x, True, reuse=reuse)
        else:
            lp, tape = net.plan_forward(x, True, defer_running=True) if defer_running else net.plan_forward(x, True)
        t = gt.reshape(-1)
        if x.shape[0] > gt.shape[0]:
            pseudo = K.argmax(lp, C)
            t = torch.cat((t, pseudo[t.numel():]))
        # only the sign of the input gradient is used: the (power-of-two) scale keeps it out of half's subnormals
        dl = torch.empty_like(lp)
        K.ce_step(lp, t, C, dl, gmul=getattr(self, "_loss_scale", 1.0), ignore_index=ignore)
        gx = net.plan_backward(tape, dl, need_dx=True, need_dw=False)
        x_adv, noise = K.fgsm_step(x.detach().contiguous(), gx.contiguous(), eps)
        return x_adv, noise, lp, tape

    # ------------------------------------------------------------------------------ loops
    def _train_loop(self, labeled_dataloaders: List, unlabeled_dataloader, epoch: int, mode: ModelMode, save: bool,
                    augment_labeled_data=False, augment_unlabeled_data=False, train_jsd=False, train_adv=False):
        fix_seed(epoch)
        S = len(self.segmentators)
        diceMeters = [DiceMeter(report_axises=self.axises, method='2d', C=self.C) for _ in range(S)]
        unlabdiceMeters = [DiceMeter(report_axises=self.axises, method='2d', C=self.C) for _ in range(S)]
        suplossMeters = [AverageValueMeter() for _ in range(S)]
        jsdlossMeter = AverageValueMeter()
        advlossMeter = AverageValueMeter()

        [segmentator.set_mode(mode) for segmentator in self.segmentators]
        for l_dataloader in labeled_dataloaders:
            l_dataloader.dataset.set_mode(ModelMode.TRAIN if augment_labeled_data else ModelMode.EVAL)
        unlabeled_dataloader.dataset.training = ModelMode.TRAIN if augment_unlabeled_data else ModelMode.EVAL
        assert self.segmentators[0].training

        desc = f">>   Training   ({epoch})" if mode == ModelMode.TRAIN else f">> Validating   ({epoch})"
        n_batch = self.n_batch
        fake_labeled_iterators = [iterator_(x) for x in labeled_dataloaders]
        fake_unlabeled_iterator = iterator_(unlabeled_dataloader)
        self._iters = (fake_labeled_iterators, fake_unlabeled_iterator)
        report_iterator = iterator_(['label', 'unlab'])
        report_status = 'label'
        n_batch_iter = tqdm_(range(n_batch)) if self.use_tqdm else range(n_batch)

        for batch_num in n_batch_iter:
            if batch_num % 30 == 0 and train_jsd and self.cot_scheduler.value > 0:
                report_status = report_iterator.__next__()
            lab_batches, paths = [], []
            for it in fake_labeled_iterators:
                [[img, gt], _, path] = it.__next__()
                lab_batches.append((img, gt))
                paths.append(path)
            unlab_batch, unl_path = None, None
            if train_jsd:
                [[unlab_img, unlab_gt], _, unl_path] = fake_unlabeled_iterator.__next__()
                unlab_batch = (unlab_img, unlab_gt)
            elif train_adv:
                [[unlab_img, unlab_gt], _, unl_path] = fake_unlabeled_iterator.__cache__()   # as :377 does
                unlab_batch = (unlab_img, unlab_gt)
            out = self._run_step(lab_batches, unlab_batch, train_jsd, train_adv)
            for i in range(S):
                diceMeters[i].add(out['preds'][i], lab_batches[i][1].to(self.device))
                suplossMeters[i].add(out['sup'][i])
                if save:
                    self._save_images(out['preds'][i].max(1)[1], paths[i], 'train', epoch, str(i))
            if train_jsd:
                for i in range(S):
                    unlabdiceMeters[i].add(out['unlab_probs'][i], unlab_batch[1].to(self.device))
                    if save:
                        self._save_images(out['unlab_probs'][i].max(1)[1], unl_path, 'unlab', epoch, str(i))
                jsdlossMeter.add(out['jsd'])
            if train_adv:
                advlossMeter.add(out['adv'])
            if self.use_tqdm and (batch_num % 10 == 0 or batch_num == n_batch - 1):
                nice_dict = self._report_dict(diceMeters if report_status == 'label' else unlabdiceMeters)
                n_batch_iter.set_postfix({f'{k}_{k_}': f'{v[k_]:.2f}' for k, v in nice_dict.items() for k_ in v.keys()})
                n_batch_iter.set_description(report_status + ': ' + ','.join(
                    [f'L{i}:{suplossMeters[i].value()[0]:.3f}' for i in range(S)]))

        lab_dsc_dict = self._dsc_dict(diceMeters)
        unlab_dsc_dict = self._dsc_dict(unlabdiceMeters)
        self.upload_dicts('labeled dataset', lab_dsc_dict, epoch)
        self.upload_dicts('unlabeled dataset', unlab_dsc_dict, epoch)
        nice_dict = self._report_dict(diceMeters)
        if self._is_main():
            print(f"{desc} " + ', '.join([f'{k}_{k_}:{v[k_]:.2f}' for k, v in nice_dict.items() for k_ in v.keys()]))
        self.train_loss_summary = dict(sup=[m.value()[0] for m in suplossMeters], jsd=jsdlossMeter.value()[0],
                                       adv=advlossMeter.value()[0])
        return torch.stack([torch.stack(diceMeters[i].value()[1], dim=1) for i in range(S)]).cpu(), torch.stack(
            [torch.stack(unlabdiceMeters[i].value()[1], dim=1) for i in range(S)]).cpu()

    def _dsc_dict(self, meters):
        vals = [m.value() for m in meters]
        return {f"S{i}": {f"DSC{n}": float(vals[i][1][0][n]) for n in self.axises} for i in range(len(meters))}

    def _report_dict(self, meters):
        d = self._dsc_dict(meters)
        mean = {f"S{i}": {"DSC": float(meters[i].value()[0][0])} for i in range(len(meters))}
        return dict_merge(d, mean, re=True)

    def _eval_loop(self, val_dataloader, epoch: int, mode: ModelMode = ModelMode.EVAL, save: bool = False):
        [segmentator.set_mode(mode) for segmentator in self.segmentators]
        val_dataloader.dataset.set_mode(ModelMode.EVAL)
        assert not self.segmentators[0].training
        desc = f">> Validating   ({epoch})"
        S = self.segmentators.__len__()
        coefdiceMeters = [DiceMeter(report_axises=self.axises, method='2d', C=self.C) for _ in range(S)]
        batchdiceMeters = [DiceMeter(report_axises=self.axises, method='3d', C=self.C) for _ in range(S)]
        vallossMeters = [AverageValueMeter() for _ in range(S)]
        val_iter = tqdm_(val_dataloader) if self.use_tqdm else val_dataloader
        for batch_num, [(img, gt), _, path] in enumerate(val_iter):
            img, gt = img.to(self.device), gt.to(self.device)
            preds = map_(lambda x: x.predict(img, logit=True), self.segmentators)
            loss = map_(lambda pred: self.criterions.get('sup')(pred, gt.squeeze(1)), preds)
            for i in range(S):
                coefdiceMeters[i].add(preds[i], gt)
                batchdiceMeters[i].add(preds[i], gt)
                vallossMeters[i].add(loss[i].detach())
                if save:
                    self._save_images(preds[i].max(1)[1], path, 'eval', epoch, str(i))
        dsc_dict = self._dsc_dict(batchdiceMeters)
        self.upload_dicts('val_data', dsc_dict, epoch)
        nice_dict = self._report_dict(batchdiceMeters)
        if self._is_main():
            print(f"{desc} " + ', '.join([f'{k}_{k_}: {v[k_]:.2f}' for k, v in nice_dict.items() for k_ in v.keys()]))
        return torch.stack([torch.stack(coefdiceMeters[i].value()[1], dim=1) for i in range(S)]).cpu(), torch.stack(
            [torch.stack(batchdiceMeters[i].value()[1], dim=1) for i in range(S)]).cpu()

    # ------------------------------------------------------------------------------ adversarial term
    def _adv_from_batches(self, segmentators: Sequence[Segmentator], lab_b: Tuple[Tensor, Tensor], unl_img: Tensor,
                          eplision: float = 0.05) -> Tensor:
        """KL( a(x_adv) || b(x).detach() ) with x = cat(labeled batch of b, unlabeled batch), x_adv from
        FGSM on b (:371-392,440-442)."""
        assert segmentators.__len__() == 2, 'only implemented for 2 segmentators'
        img_2, gt_2 = lab_b
        fsgm = self._fsgm_cls(segmentators[1].torchnet, eplision=eplision)
        img_adv, noise, real_preds = fsgm(torch.cat((img_2, unl_img), dim=0), gt=gt_2, criterion=self.criterions['sup'])
        adv_preds = segmentators[0].predict(img_adv, logit=False)
        kl = self._kl_cls(reduce=True)
        adv_losses = [kl(adv_preds, real_preds.detach())]
        return sum(adv_losses) / adv_losses.__len__()

    _fsgm_cls = FSGMGenerator

    @property
    def _kl_cls(self):
        from ..loss.loss import KL_Divergence_2D
        return getattr(self, "_kl_override", None) or KL_Divergence_2D

    def _FSGM_adv_training(self, segmentators: Sequence[Segmentator], lab_data_iterators: Sequence[iterator_],
                           unlab_data_iterator: iterator_, eplision: float = 0.05):
        """Reference signature (:371-374): re-uses THIS step's batches through ``__cache__``."""
        assert segmentators.__len__() == 2, 'only implemented for 2 segmentators'
        [[unl_img, _], _, _] = unlab_data_iterator.__cache__()
        [[img_2, gt_2], _, _] = lab_data_iterators[1].__cache__()
        return self._adv_from_batches(segmentators, (img_2.to(self.device), gt_2.to(self.device)),
                                      unl_img.to(self.device), eplision)

    # ------------------------------------------------------------------------------ misc
    def _save_images(self, segs: Tensor, names, mode: str, it: int, seg_num=None):
        from PIL import Image
        for seg, name in zip(segs, names):
            p = Path(self.save_dir, f"iter{it:03d}", mode, *( [seg_num] if seg_num is not None else []), name).with_suffix(".png")
            p.parent.mkdir(parents=True, exist_ok=True)
            Image.fromarray(seg.cpu().numpy().astype(np.uint8)).save(str(p))

    def upload_dicts(self, name, dicts, epoch):
        for k, v in dicts.items():
            self.writer.add_scalars(name + '/' + k, v, epoch)

    def schedulerStep(self):
        for segmentator in self.segmentators:
            segmentator.schedulerStep()
        self.cot_scheduler.step()
        self.adv_scheduler.step()

    def _load_checkpoint(self, checkpoint):
        paths = sorted(Path(checkpoint).glob('best_*.pth')) or sorted(Path(checkpoint).glob('last*.pth'))
        for i, cp in enumerate(paths[:len(self.segmentators)]):
            state_dict = torch.load(cp, map_location=torch.device('cpu'), weights_only=False)
            self.segmentators[i].load_state_dict(state_dict['segmentator'])
            self.best_scores[i] = state_dict['best_score']
            print(f'>>>  {cp} has been loaded successfully. Best score {self.best_scores[i]:.3f} @ {state_dict["best_epoch"]}.')
            self.segmentators[i].train()

    def checkpoint(self, metric, epoch, filename='best.pth'):
        assert isinstance(metric, Tensor)
        assert metric.__len__() == self.segmentators.__len__()
        for i, score in enumerate(metric):
            self.best_score = self.best_scores[i]
            self.segmentator = self.segmentators[i]
            super().checkpoint(score, epoch, filename=f'best_{i}.pth')
            self.best_scores[i] = self.best_score


for _f in dataclasses.fields(ExecutionPlan):
    setattr(CoTrainer, _f.name, _plan_property(_f.name))
del _f

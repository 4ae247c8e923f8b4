"""Two data-parallel ranks on ONE GPU (both processes on cuda:0, torch.distributed over gloo, which moves CUDA tensors
through the host): the multi-rank control flow of the fused step on the real kernels -- weight broadcast, per-model
streams, gradient buckets handed to FlatGradSync.begin_bucket from inside the backward pass, per-model wait, fused Adam.
(The RCCL transport itself needs one GPU per rank; tests/test_step_gpu.py covers it single-rank.)"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _trainer(tmp, rank, with_sync, defer=True):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import FakeLoader, batches
    from dct_amd.loss import get_loss_fn
    from dct_amd.models import Segmentator
    from dct_amd.trainer import CoTrainer
    C, H, B, n = 3, 176, 1, 3
    segs = []
    for seed in (5 + 10 * rank, 6 + 10 * rank):      # ranks start from DIFFERENT weights: the broadcast must equalise them
        torch.manual_seed(seed)
        segs.append(Segmentator({"name": "unet", "num_classes": C, "compute_dtype": torch.bfloat16, "dropout_p": 0.0},
                                {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4},
                                {"name": "StepLR", "step_size": 90, "gamma": 0.1}))
    lab = [FakeLoader(batches(100 * rank + 31 + i, n, B, H, C), B) for i in range(2)]
    unl = FakeLoader(batches(100 * rank + 41, n, B, H, C), B)
    crit = {"sup": get_loss_fn("cross_entropy"), "jsd": get_loss_fn("jsd"), "adv": get_loss_fn("jsd")}
    tr = CoTrainer(segs, lab, unl, unl, crit, max_epoch=1, save_dir=tmp, device="cuda:0", axises=[1, 2],
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.5},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.05},
                   adv_training_dict={"eplision": 0.03}, use_tqdm=False, steps_per_epoch=n)
    for s in segs:
        s.train()
    if with_sync:
        from dct_amd.ddp import FlatGradSync
        tr.grad_sync = FlatGradSync(segs, defer_average=defer)
    return tr, lab, unl, n


def _worker(rank, world, port, out, defer=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    tr, lab, unl, n = _trainer(os.path.join(out, f"r{rank}"), rank, True, defer=defer)
    for k in range(n):
        lb = [(lab[i][k][0][0], lab[i][k][0][1]) for i in range(2)]
        o = tr._run_step(lb, (unl[k][0][0], unl[k][0][1]), True, True, (0, 1))
        assert all(torch.isfinite(v) for v in o["sup"])
    assert [tr.grad_sync.optimizer_scale(i) for i in range(2)] == ([0.5, 0.5] if defer else [1.0, 1.0])
    torch.cuda.synchronize()
    w = [torch.cat([p.detach().flatten() for p in s.torchnet.parameters()]).cpu() for s in tr.segmentators]
    torch.save(dict(w=w, buckets=tr.grad_sync.bucket_calls), os.path.join(out, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_one_gpu_fused_step_stays_in_sync(tmp_path):
    """... and the two places the 1/world of the average can be applied -- folded into the fused Adam's gradient scale (default:
    the buffers keep the SUM) or multiplied into the buffers by dct_flat_scale -- end in the same weights, bit for bit (world 2)."""
    world, res = 2, {}
    for defer in (True, False):
        port, out = _free_port(), os.path.join(str(tmp_path), f"d{int(defer)}")
        os.makedirs(out)
        mp.spawn(_worker, args=(world, port, out, defer), nprocs=world, join=True)
        r0 = torch.load(os.path.join(out, "w0.pt"), weights_only=False)
        r1 = torch.load(os.path.join(out, "w1.pt"), weights_only=False)
        assert r0["buckets"] == r1["buckets"] == 3 * 2 * 3          # 3 buckets x 2 models x 3 steps, from inside the backward
        for a, b in zip(r0["w"], r1["w"]):
            assert torch.isfinite(a).all()
            assert torch.equal(a, b)                                  # same start (broadcast) + same averaged gradients
        res[defer] = r0["w"]
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


def _unet_prog_worker(rank, world, port, out, graph, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dev = rank if backend == "nccl" else 0          # RCCL: one GPU per rank; gloo: both ranks on cuda:0
    torch.cuda.set_device(dev)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import FakeLoader, batches
    from dct_amd.ddp import FlatGradSync
    from dct_amd.loss import get_loss_fn
    from dct_amd.models import Segmentator
    from dct_amd.trainer import CoTrainer
    C, H, B, n = 3, 176, 1, 8
    segs = []
    for seed in (5 + 10 * rank, 6 + 10 * rank):
        torch.manual_seed(seed)
        segs.append(Segmentator({"name": "unet", "num_classes": C, "compute_dtype": torch.bfloat16, "dropout_p": 0.0},
                                {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4}, {"name": "StepLR", "step_size": 90, "gamma": 0.1}))
    lab = [FakeLoader(batches(100 * rank + 31 + i, n, B, H, C), B) for i in range(2)]
    unl = FakeLoader(batches(100 * rank + 41, n, B, H, C), B)
    crit = {"sup": get_loss_fn("cross_entropy"), "jsd": get_loss_fn("jsd"), "adv": get_loss_fn("jsd")}
    tr = CoTrainer(segs, lab, unl, unl, crit, max_epoch=1, save_dir=os.path.join(out, f"r{rank}"), device=f"cuda:{dev}", axises=[1, 2],
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.5},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.05},
                   adv_training_dict={"eplision": 0.03}, use_tqdm=False, steps_per_epoch=n)
    for s_ in segs:
        s_.train()
    tr.grad_sync = FlatGradSync(segs)
    tr.use_hip_graph = graph
    for k in range(n):
        lb = [(lab[i][k][0][0], lab[i][k][0][1]) for i in range(2)]
        tr._run_step(lb, (unl[k][0][0], unl[k][0][1]), True, k % 2 == 0, (0, 1) if k % 2 == 0 else None)   # FGSM every other step
    torch.cuda.synchronize()
    g = tr._step_graphs
    w = [torch.cat([p.detach().flatten() for p in s_.torchnet.parameters()]).cpu() for s_ in tr.segmentators]
    torch.save(dict(w=w, buckets=tr.grad_sync.bucket_calls, replays=0 if g is None else g.replays,
                    programs=0 if g is None else sum(1 for c in g._graphs.values() if c.program is not None and
                                                     sum(1 for o in c.program.ops if o[0] == 'call') >= 6)),
               os.path.join(out, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _unet_prog_case(tmp_path, backend):
    res = {}
    for graph in (False, True):
        out = os.path.join(str(tmp_path), f"g{int(graph)}")
        os.makedirs(out)
        mp.spawn(_unet_prog_worker, args=(2, _free_port(), out, graph, backend), nprocs=2, join=True)
        r0 = torch.load(os.path.join(out, "w0.pt"), weights_only=False)
        r1 = torch.load(os.path.join(out, "w1.pt"), weights_only=False)
        for a, b in zip(r0["w"], r1["w"]):
            assert torch.isfinite(a).all() and torch.equal(a, b)
        assert r0["buckets"] == 3 * 2 * 8
        if graph:     # both step signatures (with / without FGSM) were captured as programs with the bucket hand-overs between graphs
            assert r0["replays"] >= 2 and r0["programs"] >= 1
        res[graph] = r0
    for a, b in zip(res[False]["w"], res[True]["w"]):
        assert torch.equal(a, b)                    # replayed segments around the exchanges == eager launches around them


@pytest.mark.timeout(900)
def test_unet_two_ranks_program_around_bucketed_exchange_equals_eager(tmp_path):
    """2 x UNet under data parallelism: the step replayed as graph segments with the three gradient buckets per model handed to
    the exchange BETWEEN segments (host callbacks of the program) must leave exactly the weights of the eager step."""
    _unet_prog_case(tmp_path, "gloo")


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank")
def test_unet_two_ranks_rccl_program_equals_eager(tmp_path):
    """The same over RCCL, one GPU per rank (runs only where two devices are visible)."""
    _unet_prog_case(tmp_path, "nccl")


# ---------------------------------------------------------------------------------------------------------------------
# Enet under data parallelism: the step replays two captured graphs around one eager all-reduce per model
# (trainer/step_graph.py); optional bf16 gradient exchange (ddp.FlatGradSync(compress="bf16")).
def _enet_trainer(tmp, rank, compress, graph):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import FakeLoader, blob_batches
    from dct_amd.ddp import FlatGradSync
    from dct_amd.loss import get_loss_fn
    from dct_amd.models import Segmentator
    from dct_amd.trainer import CoTrainer
    C, H, B, n = 3, 64, 2, 6
    segs = []
    for seed in (5 + 10 * rank, 6 + 10 * rank):
        torch.manual_seed(seed)
        segs.append(Segmentator({"name": "enet", "num_classes": C, "compute_dtype": torch.bfloat16},
                                {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4}, {"name": "StepLR", "step_size": 90, "gamma": 0.1}))
    lab = [FakeLoader(blob_batches(100 * rank + 31 + i, n, B, H, C), B) for i in range(2)]
    unl = FakeLoader(blob_batches(100 * rank + 41, n, B, H, C), B)
    crit = {"sup": get_loss_fn("cross_entropy"), "jsd": get_loss_fn("jsd"), "adv": get_loss_fn("jsd")}
    tr = CoTrainer(segs, lab, unl, unl, crit, max_epoch=1, save_dir=tmp, device="cuda:0", axises=[1, 2],
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.5},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.05},
                   adv_training_dict={"eplision": 0.03}, use_tqdm=False, steps_per_epoch=n)
    for s in segs:
        s.train()
    tr.grad_sync = FlatGradSync(segs, compress=compress, measure=True)
    tr.ddp_segmented_graph = graph
    return tr, lab, unl, n


def _enet_worker(rank, world, port, out, compress, graph, sup_only=False, model_streams=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    tr, lab, unl, n = _enet_trainer(os.path.join(out, f"r{rank}"), rank, compress, graph)
    tr.model_streams = model_streams
    sups = []
    for k in range(n):
        lb = [(lab[i][k][0][0], lab[i][k][0][1]) for i in range(2)]
        if sup_only:
            o = tr._run_step(lb, (unl[k][0][0], unl[k][0][1]), False, False, None)
        else:
            o = tr._run_step(lb, (unl[k][0][0], unl[k][0][1]), True, True, (0, 1))
        sups.append([float(v) for v in o["sup"]])
    torch.cuda.synchronize()
    g = tr._step_graphs
    w = [torch.cat([p.detach().flatten() for p in s.torchnet.parameters()]).cpu() for s in tr.segmentators]
    torch.save(dict(w=w, sups=sups, captures=0 if g is None else g.captures, replays=0 if g is None else g.replays,
                    two_graphs=g is not None and all((c.graph_opt is not None) or (c.program is not None and any(o[0] == 'call' for o in c.program.ops))
                                                      for c in g._graphs.values()),   # the exchange sits BETWEEN captured graphs
                    exposed=tr.grad_sync.exposed_ms(), bytes=tr.grad_sync.exchanged_bytes,
                    steps=[s.optimizer._steps for s in tr.segmentators]), os.path.join(out, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_enet_two_ranks_segmented_graph_equals_eager_exchange(tmp_path):
    """Six data-parallel Enet steps (JSD + FGSM) on two ranks: (a) eager launches with the per-model exchange, (b) two eager
    steps, then the [forward + backward] / [optimizers] graphs replayed around the exchange.  Both ranks must end identical,
    and (b) must equal (a) bit for bit."""
    res = {}
    for graph in (False, True):
        out = str(tmp_path / f"g{int(graph)}")
        os.makedirs(out)
        mp.spawn(_enet_worker, args=(2, _free_port(), out, None, graph), nprocs=2, join=True)
        r0 = torch.load(os.path.join(out, "w0.pt"), weights_only=False)
        r1 = torch.load(os.path.join(out, "w1.pt"), weights_only=False)
        for a, b in zip(r0["w"], r1["w"]):
            assert torch.isfinite(a).all() and torch.equal(a, b)
        assert r0["steps"] == [6, 6]
        if graph:
            assert r0["captures"] == 1 and r0["replays"] == 3 and r0["two_graphs"]     # step 0 allocates (own signature), 1-2 eager, capture at 3
        else:
            assert r0["captures"] == 0
        res[graph] = r0
    assert res[False]["sups"] == res[True]["sups"]
    for a, b in zip(res[False]["w"], res[True]["w"]):
        assert torch.equal(a, b)


@pytest.mark.timeout(900)
def test_enet_two_ranks_bf16_gradient_exchange(tmp_path):
    """compress="bf16": half the bytes on the wire, ranks still bit-identical to each other, weights within bf16 rounding of the
    fp32 exchange after six steps."""
    res = {}
    for compress in (None, "bf16"):
        out = str(tmp_path / f"c{compress}")
        os.makedirs(out)
        mp.spawn(_enet_worker, args=(2, _free_port(), out, compress, True), nprocs=2, join=True)
        r0 = torch.load(os.path.join(out, "w0.pt"), weights_only=False)
        r1 = torch.load(os.path.join(out, "w1.pt"), weights_only=False)
        for a, b in zip(r0["w"], r1["w"]):
            assert torch.equal(a, b)
        res[compress] = r0
    assert res["bf16"]["bytes"] * 2 == res[None]["bytes"]
    for a, b in zip(res[None]["w"], res["bf16"]["w"]):
        assert ((a - b).norm() / a.norm()).item() < 3e-2          # six Adam steps of lr 1e-3 on sign-like updates (measured 1.2e-2)
    assert res[None]["exposed"] >= 0.0


@pytest.mark.timeout(900)
def test_enet_two_ranks_supervised_only_arena_exchange_is_ordered_against_the_model_streams(tmp_path):
    """Supervised-only steps (start_training's default: no JSD, no adversarial term) of 2 x Enet on per-model streams: each model's
    ONE backward pass runs on its own stream and the fused arena collective is launched from the last model's -- it must wait for
    the other model's stream (event per model in FlatGradSync.begin), and every model's Adam must wait for the collective on ITS
    stream.  The weights must equal, bit for bit, the run with all models queued on one stream, and the ranks each other."""
    res = {}
    for streams in (False, True):
        out = str(tmp_path / f"s{int(streams)}")
        os.makedirs(out)
        mp.spawn(_enet_worker, args=(2, _free_port(), out, None, False, True, streams), nprocs=2, join=True)
        r0 = torch.load(os.path.join(out, "w0.pt"), weights_only=False)
        r1 = torch.load(os.path.join(out, "w1.pt"), weights_only=False)
        for a, b in zip(r0["w"], r1["w"]):
            assert torch.isfinite(a).all() and torch.equal(a, b)
        assert r0["steps"] == [6, 6]
        res[streams] = r0
    assert res[False]["sups"] == res[True]["sups"]
    for a, b in zip(res[False]["w"], res[True]["w"]):
        assert torch.equal(a, b)

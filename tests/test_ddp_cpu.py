"""Data-parallel gradient exchange on CPU: two processes over gloo (world_size 2) run CoTrainer._run_step
on different per-rank batches through FlatGradSync and must end with (a) identical weights on both
ranks and (b) the weights a single process gets from the averaged gradient -- the N>1 path of bench.py.
Arithmetic comes from injected oracle modules (the product kernels are HIP-only)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(tmp, rank_seed_offset):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from helpers import FakeLoader, batches
    from test_host_logic_cpu import OracleCE, OracleJSD, OracleKL, OracleFGSM, OracleDice
    from dct_amd.models import Segmentator
    from dct_amd.trainer import CoTrainer
    from dct_amd.trainer import cotraining_totalloss as mod
    mod.DiceMeter = OracleDice
    C, H, B = 2, 176, 1
    segs = []
    for s in (21, 22):
        torch.manual_seed(s)                        # identical initial weights on every rank
        net = oracle.build_net("unet", C, dropout_p=0.0)
        segs.append(Segmentator({"name": "unet", "num_classes": C}, {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4},
                                {"name": "StepLR", "step_size": 90, "gamma": 0.1}, torchnet=net,
                                softmax_fn=oracle.softmax_channels))
    lab = [FakeLoader(batches(31 + i + rank_seed_offset, 1, B, H, C), B) for i in range(2)]
    unl = FakeLoader(batches(41 + rank_seed_offset, 1, B, H, C), B)
    crit = {"sup": OracleCE(), "jsd": OracleJSD(), "adv": OracleJSD()}
    tr = CoTrainer(segs, lab, unl, unl, crit, max_epoch=1, save_dir=tmp, device="cpu", axises=[1],
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.5},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.05},
                   adv_training_dict={"eplision": 0.03}, use_tqdm=False, steps_per_epoch=1)
    tr._fsgm_cls = OracleFGSM
    tr._kl_override = OracleKL
    for s in segs:
        s.train()
    return tr, lab, unl


def _step(tr, lab, unl):
    lb = [(lab[i][0][0][0], lab[i][0][0][1]) for i in range(2)]
    ub = (unl[0][0][0], unl[0][0][1])
    return tr._run_step(lb, ub, True, True, (0, 1))


def _weights(tr):
    return [torch.cat([p.detach().flatten() for p in s.torchnet.parameters()]).clone() for s in tr.segmentators]


def _worker(rank, world, port, tmp, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from dct_amd import ddp
    r, _, w = ddp.init_from_env("gloo")
    assert (r, w) == (rank, world)
    tr, lab, unl = _build(os.path.join(tmp, f"r{rank}"), 100 * rank)
    # perturb rank 1's initial weights: broadcast_weights must restore rank 0's
    if rank == 1:
        with torch.no_grad():
            for p in tr.segmentators[0].torchnet.parameters():
                p.add_(0.5)
    tr.grad_sync = ddp.FlatGradSync(tr.segmentators)
    _step(tr, lab, unl)
    torch.save([w_.numpy() for w_ in _weights(tr)], os.path.join(out, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_gloo_step_equals_averaged_gradient_step(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path)
    mp.spawn(_worker, args=(world, port, out, out), nprocs=world, join=True)
    w0 = torch.load(os.path.join(out, "w0.pt"), weights_only=False)
    w1 = torch.load(os.path.join(out, "w1.pt"), weights_only=False)
    for a, b in zip(w0, w1):
        np.testing.assert_array_equal(a, b)          # ranks stay bit-identical
    # single-process reference: average the two ranks' gradients by hand, then one Adam step
    # (the parent keeps its ATen thread count: other tests in this process pin last-bit numerics to it)
    trs = [_build(os.path.join(out, f"s{r}"), 100 * r) for r in range(world)]
    grads = []
    for tr, lab, unl in trs:
        steps = [s.optimizer.step for s in tr.segmentators]
        for s in tr.segmentators:
            s.optimizer.step = lambda: None          # gradients only
        _step(tr, lab, unl)
        grads.append([[p.grad.clone() for p in s.torchnet.parameters()] for s in tr.segmentators])
        for s, st in zip(tr.segmentators, steps):
            s.optimizer.step = st
    tr0 = trs[0][0]
    for m, seg in enumerate(tr0.segmentators):
        for k, p in enumerate(seg.torchnet.parameters()):
            p.grad = (grads[0][m][k] + grads[1][m][k]) / world
        seg.optimizer.step()
    for a, b in zip(w0, _weights(tr0)):
        # gloo sums then scales, the hand average divides once: 1 ulp apart in g, which Adam's first step
        # (lr * g / (|g| + eps)) turns into <= 1e-5 on elements whose |g| is comparable to eps = 1e-8
        np.testing.assert_allclose(a, b.numpy(), rtol=0, atol=2e-5)


def _bucket_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import types
    import torch.distributed as dist
    from dct_amd import ddp
    from dct_amd.arch.flat import FlatParams
    ddp.init_from_env("gloo")
    segs = []
    for m in range(3):
        net = torch.nn.Sequential(torch.nn.Linear(40, 30), torch.nn.Linear(30, 7))
        net.flat_params = FlatParams(list(net.parameters()))
        net.flat_params.ensure()
        net.flat_params.ensure_grads()
        net.flat_params.gflat.copy_(torch.arange(net.flat_params.total, dtype=torch.float32) * (rank + 1) + 10 * m)
        segs.append(types.SimpleNamespace(torchnet=net))
    # model 2's optimizer takes a gradient scale (the fused Adam does): its buffer keeps the SUM, the 1/world goes to the update
    segs[2].optimizer = types.SimpleNamespace(grad_scale=1.0)
    sync = ddp.FlatGradSync(segs, broadcast_weights=False)
    assert [sync.optimizer_scale(m) for m in range(3)] == [1.0, 1.0, 1.0 / world]
    total = segs[0].torchnet.flat_params.total
    cuts = [(total // 2, total), (total // 5, total // 2), (0, total // 5)]     # completion order of a backward pass
    for lo, hi in cuts:
        sync.begin_bucket(0, lo, hi)
    sync.begin(0)                   # every bucket already went out: must not reduce a second time
    sync.begin(1)                   # model 1: one whole-buffer exchange
    sync.begin(2)
    for m in range(3):
        sync.finish(m)
    torch.save([s.torchnet.flat_params.gflat.clone() for s in segs], os.path.join(out, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_bucketed_exchange(tmp_path):
    """FlatGradSync.begin_bucket (the in-backward exchange of UNet's gradient buckets): three slices of model 0 and the
    whole buffer of model 1 end as the rank mean on both ranks, each element reduced exactly once; model 2, whose optimizer
    folds the 1/world into its update (FlatGradSync.optimizer_scale), keeps the rank SUM."""
    world, port = 2, _free_port()
    out = str(tmp_path)
    mp.spawn(_bucket_worker, args=(world, port, out), nprocs=world, join=True)
    g0 = torch.load(os.path.join(out, "g0.pt"), weights_only=False)
    g1 = torch.load(os.path.join(out, "g1.pt"), weights_only=False)
    for m in range(3):
        n = g0[m].numel()
        want = (torch.arange(n, dtype=torch.float32) * 1 + 10 * m + torch.arange(n, dtype=torch.float32) * 2 + 10 * m) / (2 if m < 2 else 1)
        assert torch.equal(g0[m], g1[m])
        np.testing.assert_allclose(g0[m].numpy(), want.numpy(), rtol=1e-6)


def _arena_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import types
    import torch.distributed as dist
    from dct_amd import ddp
    from dct_amd.arch.flat import FlatParams
    ddp.init_from_env("gloo")
    np.random.seed(1000 + 17 * rank)            # the ranks' global numpy states DIFFER (loaders, user code)
    segs = []
    for m in range(3):
        net = torch.nn.Sequential(torch.nn.Linear(40, 30), torch.nn.Linear(30, 7 + m))
        net.flat_params = FlatParams(list(net.parameters()))
        segs.append(types.SimpleNamespace(torchnet=net))
    sync = ddp.FlatGradSync(segs, broadcast_weights=False)
    sync.prepare()
    flats = [s.torchnet.flat_params for s in segs]
    assert sync._arena is not None and sync._arena.numel() == sum(f.total for f in flats)
    for m, f in enumerate(flats):
        assert f.grads_attached() and f.gflat.data_ptr() == sync._arena.data_ptr() + 4 * sync._arena_span[m][0]
    log = {"pairs": [], "collectives": []}
    for step in range(3):
        for m, f in enumerate(flats):
            f.gflat.copy_(torch.arange(f.total, dtype=torch.float32) * (rank + 1) + 10 * m + step)
        c0 = sync.collectives
        for m in range(3):
            sync.begin(m)               # the collective leaves with the LAST model's begin
        for m in range(3):
            sync.finish(m)
        log["collectives"].append(sync.collectives - c0)
        log["pairs"].append(sync.draw_pair(3))
    # a model whose buffer was re-allocated (e.g. .to(device)) is re-adopted with its gradients
    flats[1].gflat = flats[1].gflat.clone()
    for k, p in enumerate(flats[1].params):
        p.grad = flats[1]._grad_view(k)
    keep = flats[1].gflat.clone()
    sync.prepare()
    assert flats[1].gflat.data_ptr() == sync._arena.data_ptr() + 4 * sync._arena_span[1][0] and torch.equal(flats[1].gflat, keep)
    # only two of the three models have gradients this step: falls back to one exchange each, still the rank mean
    for m in (0, 2):
        flats[m].gflat.fill_(float(rank + 1))
        sync.begin(m)
    c0 = sync.collectives
    sync.finish()
    log["partial"] = (sync.collectives - c0, float(flats[0].gflat[0]), float(flats[2].gflat[-1]))
    log["g"] = [f.gflat.clone() for f in flats]
    torch.save(log, os.path.join(out, f"a{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_small_models_share_one_collective_and_one_adversarial_pair(tmp_path):
    """the S Enet-sized models' gradients leave as ONE flat collective per step (FlatGradSync arena), and the
    adversarial pair is identical on every rank although the ranks' global numpy RNG states differ (draw_pair)."""
    world, port = 2, _free_port()
    out = str(tmp_path)
    mp.spawn(_arena_worker, args=(world, port, out), nprocs=world, join=True)
    a0 = torch.load(os.path.join(out, "a0.pt"), weights_only=False)
    a1 = torch.load(os.path.join(out, "a1.pt"), weights_only=False)
    assert a0["collectives"] == [1, 1, 1] and a1["collectives"] == [1, 1, 1]
    assert a0["pairs"] == a1["pairs"] and all(0 <= a < b <= 2 for a, b in a0["pairs"])
    assert a0["partial"] == a1["partial"] == (2, 1.5, 1.5)
    for m in range(3):
        assert torch.equal(a0["g"][m], a1["g"][m])
    n = a0["g"][1].numel()                      # model 1 kept the mean of step 2
    want = (torch.arange(n, dtype=torch.float32) * 1 + torch.arange(n, dtype=torch.float32) * 2) / 2 + 10 + 2
    np.testing.assert_allclose(a0["g"][1].numpy(), want.numpy(), rtol=1e-6)


def _dry_bench(extra, timeout=850):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "1", "--warmup", "0"] + extra,
                          env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) must start its two ranks itself and leave ONE JSON line with
    n_gpus = 2 as the last line of its output: --dry-launch runs that path on CPU (gloo, oracle-injected networks) through the
    SAME `bench.measure()` the MI355X run uses -- warm-up, timed regions and every leg behind them (the per-launch event leg
    included, with a CPU stand-in for the profiler) -- and counts the collectives each rank issues: the sequences must agree."""
    import json
    r = _dry_bench([])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "dp2" and line["value"] > 0
    assert all(v == v for v in line["losses_last_step"]["sup"])
    col = line["collectives"]
    assert col["same_sequence_on_every_rank"] and col["per_rank"][0] == col["per_rank"][1] > 0 and col["event_leg_ran"]


@pytest.mark.timeout(600)
def test_bench_rank_conditional_step_leg_is_caught():
    """The round-4 defect -- a leg behind the timed region that runs steps (hence gradient all-reduces) on rank 0 only while the
    other ranks go on to the barrier -- re-introduced through a self-test switch: the launch must fail (gloo's collective
    timeout), not print a line.  This is what makes the test above a test of the post-timed control flow."""
    r = _dry_bench(["--dry-timeout", "20", "--dry-break-rank0-leg"], timeout=550)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]

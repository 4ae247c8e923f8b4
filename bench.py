#!/usr/bin/env python3
"""Headline benchmark: co-train imgs/sec (lab+unlab), 2xUNet ACDC-shaped 256x256 slices
(BASELINE.json configs[1]: CE + JSD consistency, bs 8+8 per GPU, bf16), synthetic data.

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N --config cfg4 --global-batch 64+64   (fixed global batch split over the ranks: "scaling": "strong")

A "step" is one pass of the hot path (CoTrainer._run_step) over one batch: 2 supervised
forwards + CE, 2 unlabeled forwards + JSD, one backward through all four graphs, gradient
all-reduce (N>1), fused Adam on both models.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line (contract in the task statement), including

  roofline     : dominant conv kernel class -- algorithmic FLOPs / time in
                 that class measured with HIP events around every launch on its stream
  cpu_baseline : the CPU oracle's co-training step timed on this host's cores on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_F32_TFLOPS = 157.3

CONFIGS = {
    # name: arch, H, C, B_l, B_u, train_adv, S
    "cfg2": dict(arch="unet", H=256, C=4, B_l=8, B_u=8, train_adv=False, S=2,
                 desc="2xUNet co-training (CE + JSD), ACDC-shaped 256x256 C=4, bs 8+8 per GPU"),
    # diagnostic (not a BASELINE config): ONE of cfg2's two networks, same batches -- what the second model's chain costs beside the first
    "cfg2s1": dict(arch="unet", H=256, C=4, B_l=8, B_u=8, train_adv=False, S=1,
                   desc="diagnostic: ONE UNet of cfg2 (CE + the degenerate one-model JSD), 256x256 C=4, bs 8+8"),
    "cfg3": dict(arch="unet", H=256, C=4, B_l=8, B_u=8, train_adv=True, S=2,
                 desc="2xUNet co-training (CE + JSD + FGSM eps .03), 256x256 C=4, bs 8+8 per GPU"),
    # BASELINE configs[3] / [4]: the Enet configurations (HBM/launch bound; roofline leg reports HBM GB/s)
    "cfg4": dict(arch="enet", H=200, C=2, B_l=8, B_u=8, train_adv=True, S=2,
                 desc="2xEnet co-training (CE + JSD + FGSM), spinal-cord-GM-shaped 200x200 C=2, bs 8+8 per GPU"),
    # "cfg4 ... arch enet as in script/GM/check.sh:12 (also report unet, valid at 200)"
    "cfg4u": dict(arch="unet", H=200, C=2, B_l=8, B_u=8, train_adv=True, S=2,
                  desc="2xUNet co-training (CE + JSD + FGSM), spinal-cord-GM-shaped 200x200 C=2, bs 8+8 per GPU"),
    "cfg5": dict(arch="enet", H=320, C=2, B_l=4, B_u=16, train_adv=True, S=3,
                 desc="3xEnet co-training (CE + JSD + FGSM), prostate-shaped 320x320 C=2, lab:unlab 1:4 (4+16 per GPU)"),
}

# conv in+out activation elements per image, forward (measured on the reference modules)
ENET_ACT_ELEMS = {200: 9.48e6, 256: 15.66e6, 320: 24.27e6, 64: 0.97e6}


def unet_encoder_gemm_flops(H: int, W: int):
    """The part of unet_conv_flops' MFMA share that lies in front of the first dropout (the eight encoder convolutions without the stem)."""
    g, h, w, cin = 0.0, H, W, 1
    for lvl, width in enumerate((64, 128, 256, 512)):
        if lvl:
            g += 2.0 * (h - 2) * (w - 2) * 9 * cin * width
        g += 2.0 * (h - 4) * (w - 4) * 9 * width * width
        h, w, cin = (h - 4 + 1) // 2, (w - 4 + 1) // 2, width
    return g


def unet_conv_flops(H: int, W: int, C: int):
    """Forward conv+convT FLOPs (2*MAC) of one image through the reference UNet, split into the part
    the MFMA implicit-GEMM kernels execute and the small stem/head layers.  256x256/C=4 -> 34.51 GF."""
    gemm = 0.0
    small = 0.0
    h, w, cin = H, W, 1
    skips = []
    for lvl, width in enumerate((64, 128, 256, 512)):
        f = 2.0 * (h - 2) * (w - 2) * 9 * cin * width
        if lvl == 0:
            small += f
        else:
            gemm += f
        gemm += 2.0 * (h - 4) * (w - 4) * 9 * width * width
        h, w, cin = (h - 4 + 1) // 2, (w - 4 + 1) // 2, width
    gemm += 2.0 * (h - 2) * (w - 2) * 9 * 512 * 1024 + 2.0 * (h - 4) * (w - 4) * 9 * 1024 * 1024
    gemm += 2.0 * (h - 4) * (w - 4) * 4 * 1024 * 512
    h, w = 2 * (h - 4), 2 * (w - 4)
    for ci, feat, co in ((1024, 512, 256), (512, 256, 128), (256, 128, 64)):
        gemm += 2.0 * (h - 2) * (w - 2) * 9 * ci * feat + 2.0 * (h - 4) * (w - 4) * 9 * feat * feat
        gemm += 2.0 * (h - 4) * (w - 4) * 4 * feat * co
        h, w = 2 * (h - 4), 2 * (w - 4)
    gemm += 2.0 * (h - 2) * (w - 2) * 9 * 128 * 64 + 2.0 * (h - 4) * (w - 4) * 9 * 64 * 64
    small += 2.0 * (h - 4) * (w - 4) * 64 * C
    return gemm, small


def pmc_traffic(config):
    """HBM bytes per launch of the dominant conv kernel from the committed rocprofv3 --pmc passes
    (profiles/*_pmc_traffic.json, produced by tools/pmc_traffic.py on this same command); None when absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{config}_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            d["source"] = os.path.relpath(path, ROOT)
            d["archived"] = "separate rocprofv3 --pmc passes of this command (committed file); not measured by this run"
            return d
        except Exception:
            continue
    return None


def pmc_mfma_busy(config):
    """MFMA-pipe busy share per kernel class from the newest committed profiles/r*_<config>_pmc_mfma_busy.txt: counter evidence of an
    EARLIER rocprofv3 --pmc pass of this command (tools/pmc_mfma.py); None when absent.  (The shader clock is measured by the run
    itself: `shader_clock_ghz_during_the_step`.)"""
    import glob
    out = {}
    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{config}_pmc_mfma_busy.txt")), reverse=True)[0]
        with open(path) as f:
            for line in f:
                t = line.split()
                if len(t) >= 3 and t[2].endswith("%"):
                    out[t[0]] = float(t[2].rstrip("%")) / 100.0
        return {"mfma_pipe_busy_share_of_simd_cycles": out,
                "source": f"{os.path.relpath(path, ROOT)} (committed rocprofv3 --pmc pass of this command)"}
    except Exception:
        return None


def make_trainer(cfg, dtype, device, rank, world, grad_sync_factory, n_batches=4, data="blob"):
    from dct_amd.loss import get_loss_fn
    from dct_amd.models import Segmentator
    from dct_amd.trainer import CoTrainer
    from helpers import FakeLoader, batches, blob_batches
    S, C, H = cfg["S"], cfg["C"], cfg["H"]
    torch.manual_seed(1234)        # identical initial weights on every rank
    segs = [Segmentator({"name": cfg["arch"], "num_classes": C, "compute_dtype": dtype},
                        {"name": "Adam", "lr": 1e-3, "weight_decay": 1e-4},
                        {"name": "StepLR", "step_size": 90, "gamma": 0.1}) for _ in range(S)]

    def dev_batches(seed, B):
        # blob-structured slices (tests/helpers.py): the nets learn them, so the timed steps run on the operand statistics
        # of a network that is training (i.i.d. random labels drive it to the trivial uniform predictor)
        # --data iid: the inputs SURVEY.md 8d prescribes (torch.rand images in [0, 1), torch.randint labels, seeded)
        make = batches if data == "iid" else blob_batches
        return [[[b[0][0].to(device), b[0][1].to(device)], None, b[2]] for b in make(seed, n_batches, B, H, C)]

    base = 1234 + 1000 * rank      # reference default seed (config/ACDC_config_cotraing.yaml:79) + rank
    lab = [FakeLoader(dev_batches(base + 1 + i, cfg["B_l"]), cfg["B_l"]) for i in range(S)]
    unl = FakeLoader(dev_batches(base + 99, cfg["B_u"]), cfg["B_u"])
    crit = {"sup": get_loss_fn("cross_entropy"), "jsd": get_loss_fn("jsd"), "adv": get_loss_fn("jsd")}
    tr = CoTrainer(segs, lab, unl, unl, crit, max_epoch=1, save_dir=tempfile.mkdtemp(prefix="dct_bench_"),
                   device=str(device), axises=list(range(1, C)),
                   cot_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.5},
                   adv_scheduler_dict={"name": "ConstantScheduler", "begin_epoch": 0, "max_value": 0.05},
                   adv_training_dict={"eplision": 0.03}, use_tqdm=False)
    for s in segs:
        s.train()
    if world > 1:
        tr.grad_sync = grad_sync_factory(segs)
    return tr, lab, unl


SETUP_STEPS = 6


def host_cores() -> int:
    """Cores this process may actually use: scheduler affinity capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host on the GPU boxes and oversubscribes ATen's pool)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(cfg, seconds_budget=25.0):
    """The oracle's step (fp32, ATen CPU kernels = what the reference executes) on a bounded sample:
    the configuration's own batch (cfg2: 8+8 = 24 images per step), one warm-up step, then as many timed steps as fit the
    budget (at least one)."""
    import oracle
    from helpers import blob_batches
    S, C, H = cfg["S"], cfg["C"], cfg["H"]
    B_l, B_u = cfg["B_l"], cfg["B_u"]
    threads = host_cores()
    torch.set_num_threads(threads)
    models = []
    for s in range(S):
        torch.manual_seed(100 + s)
        models.append(oracle.OracleModel.make(oracle.build_net(cfg["arch"], C).train()))
    lab = [tuple(blob_batches(5 + s, 1, B_l, H, C)[0][0]) for s in range(S)]
    unl = blob_batches(99, 1, B_u, H, C)[0][0][0]

    def step():
        oracle.cotrain_step(models, lab, unl, True, cfg["train_adv"], lam_cot=0.5, lam_adv=0.05, eps=0.03)

    t0 = time.perf_counter()
    step()  # warm-up
    warm = time.perf_counter() - t0
    n, t0 = 0, time.perf_counter()
    while n < 1 or (n < 8 and (time.perf_counter() - t0) + warm * 1.2 < seconds_budget - warm):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    imgs = S * B_l + B_u
    return {"value": imgs / dt, "unit": "imgs/sec", "cores": threads, "kind": "port",
            "sample": f"oracle (PyTorch-CPU fp32 restatement of the reference step) {S}x{cfg['arch']} {H}x{H} C={C}, "
                      f"bs {B_l}+{B_u} ({imgs} imgs/step, the benchmarked batch), {n} timed steps after 1 warm-up, {dt:.2f} s/step, "
                      f"{threads} threads"}


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) through torch.distributed.run and
    pass their output through -- rank 0's JSON line stays the last line.  Runs BEFORE this process touches the GPU (a
    process that has initialised HIP must not be replaced or forked into ranks); the reference's analogue is one command
    driving N GPUs through nn.DataParallel (generalframework/models/segmentators.py:34-36)."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


class HipLegs:
    """Device side of the measurement legs on the MI355X: synchronisation and the library's per-launch HIP-event profiler."""
    kind = "hip"

    def __init__(self, device):
        from dct_amd import _lib
        self.device, self._lib = device, _lib

    def sync(self):
        torch.cuda.synchronize()

    def scalar(self, v):
        return torch.tensor([v], dtype=torch.float64, device=self.device)

    def clock_begin(self, seconds: float):
        """One sleeping wave on a stream of its own for `seconds`: shader-clock cycles over reference ticks of its life
        (include/dct.h dct_clock_probe) = the clock the chip holds while whatever is launched next runs."""
        self._clk_out = torch.zeros(2, dtype=torch.int64, device=self.device)
        self._clk_stream = torch.cuda.Stream(device=self.device)
        ticks = max(1, min(int(seconds * 1e8), 100_000_000))
        self._lib.check(self._lib.load().dct_clock_probe(self._clk_out.data_ptr(), ticks, self._clk_stream.cuda_stream), "dct_clock_probe")

    def clock_end(self):
        self._clk_stream.synchronize()
        cyc, ref = (int(v) for v in self._clk_out.tolist())
        return {"ghz": 0.1 * cyc / ref if ref else None, "window_ms": ref / 1e5,
                "how": "one sleeping wave beside the replayed step: d(s_memtime) / d(s_memrealtime) x 100 MHz over its life"}

    def prof_begin(self, record: bool):
        self._lib.prof_read(reset=True)
        self._lib.prof_enable(bool(record))

    def prof_end(self, record: bool):
        self._lib.prof_enable(False)
        return self._lib.prof_read(reset=True) if record else None


class DryLegs:
    """CPU stand-in for `--dry-launch`: the same control flow without a device (nothing to synchronise, no event profiler)."""
    kind = "dry"
    device = torch.device("cpu")

    def sync(self):
        pass

    def scalar(self, v):
        return torch.tensor([v], dtype=torch.float64)

    def clock_begin(self, seconds: float):
        pass

    def clock_end(self):
        return None

    def prof_begin(self, record: bool):
        self._on = bool(record)

    def prof_end(self, record: bool):
        return {"igemm": {"ms": 0.0, "launches": 0}} if record else None


def measure(args, tr, one_step, legs, rank, world, ddp_on, meters_leg=None, operand_leg=None, enter_event_leg=None):
    """W warm-up steps, the timed regions, and every leg behind them.  ONE rule keeps `--gpus N` from hanging: a leg that runs
    `one_step` (which issues the gradient all-reduces when N > 1) is executed by EVERY rank; what is rank-conditional is only
    who records or prints.  (Round 4 ran the HIP-event leg on rank 0 alone while the other ranks went on to the barrier:
    mismatched collectives on one communicator.)  `--dry-launch` drives this same function on CPU with a collective counter per
    rank (tests/test_ddp_cpu.py::test_bench_launches_its_own_ranks)."""
    import torch.distributed as dist
    for i in range(args.warmup):
        one_step(i)
    legs.sync()
    if ddp_on:
        tr.grad_sync.exposed_ms(reset=True)
        tr.grad_sync.exchanged_bytes = 0
        dist.barrier()
    legs.sync()
    # The timed region: EXACTLY K steps between barrier + synchronize on both sides -- taken `--repeats` times back to back (default 3)
    # and the MEDIAN region reported: a region is 0.1-0.3 s, and run-to-run noise on one box is of the size of a round's gains.
    regions, out = [], None
    reps = max(1, args.repeats)
    for rep in range(reps):
        if ddp_on:
            dist.barrier()
        legs.sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = one_step(args.warmup + rep * args.steps + i)
        legs.sync()
        if ddp_on:
            dist.barrier()
        legs.sync()
        t_rep = time.perf_counter() - t0
        if ddp_on:                      # MAX over ranks, per region
            tt = legs.scalar(t_rep)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_rep = float(tt.item())
        regions.append(t_rep)
    res = {"regions": regions, "elapsed": sorted(regions)[len(regions) // 2], "out": out, "exchange": None, "meters_ms": None,
           "operand_stats": None, "prof": None, "clock": None,
           # (the timed steps' layout: the legs below switch to one stream, where the adversarial step recomputes the FGSM generator's encoder)
           "shared_fgsm_encoder": bool(getattr(tr, "fgsm_shares_encoder", False) and getattr(tr, "_step_hint_adv_chain", False))}
    if ddp_on:
        ex = legs.scalar(tr.grad_sync.exposed_ms(reset=True) / (args.steps * reps))
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        graphs = getattr(tr, "_step_graphs", None)
        res["exchange"] = {"exposed_allreduce_ms_per_step_max_over_ranks": float(ex.item()),
                           "bytes_per_step_per_rank": tr.grad_sync.exchanged_bytes / (args.steps * reps),
                           "wire_dtype": "bf16" if args.grad_compress == "bf16" else "f32",
                           "mode": "captured graph segments around the eager all-reduces" if graphs is not None and graphs.captures
                           else "eager launches, bucketed all-reduce from inside the backward pass"}
    base = args.warmup + args.steps * reps
    # The shader clock UNDER this load: the K steps once more (every rank: their all-reduces must meet) with one sleeping probe wave
    # beside them on rank 0, alive for ~80 % of the time the K steps take (the back-to-back single-layer loops of round 3's clock
    # study are power-capped to 1.77 GHz; the step is not).
    if not args.no_clock_probe:
        if rank == 0:
            legs.clock_begin(0.8 * res["elapsed"])
        for i in range(args.steps):
            one_step(base + i)
        legs.sync()
        if rank == 0:
            res["clock"] = legs.clock_end()
        base += args.steps
    # The step with the in-step meters on (SURVEY.md 8d asks for it separately).  `world == 1` is the same on every rank.
    if meters_leg is not None and world == 1:
        res["meters_ms"] = meters_leg(base)
    # Operand statistics: one forward pass of model 0 on rank 0 -- no step, no collective.
    if operand_leg is not None and rank == 0:
        res["operand_stats"] = operand_leg()
    # Roofline leg: the SAME K steps once more with every launch bracketed by HIP events on its stream (kept out of the timed
    # region above: ~1200 event records per step would cost the step ~25 % and `value` must be the unperturbed rate).  Every rank
    # runs the steps (their all-reduces must meet); rank 0 alone records.
    if not args.no_kernel_events:
        if enter_event_leg is not None:
            enter_event_leg()
        record = rank == 0
        legs.prof_begin(record)
        for i in range(args.steps):
            one_step(base + i)
        legs.sync()
        res["prof"] = legs.prof_end(record)
    if ddp_on:
        dist.barrier()
    return res


def dry_launch(args, rank: int, world: int) -> None:
    """`--dry-launch`: the N-rank plumbing of this script on CPU -- gloo process group, FlatGradSync, and `measure()` itself (the
    barrier / max-over-ranks timing and EVERY leg behind the timed region, with CPU stand-ins for the device calls) -- around
    CoTrainer._run_step with oracle-injected networks (the product kernels are HIP-only; tests/test_ddp_cpu.py builds the same
    trainer).  Every collective a rank issues is counted; the line reports whether all ranks issued the same sequence.  Not a
    measurement: it exists so that the launch path is testable without a GPU."""
    import datetime
    import torch.distributed as dist
    from dct_amd import ddp
    from test_ddp_cpu import _build, _step
    torch.set_num_threads(2)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:        # a short timeout: mismatched collectives must fail the run, not park it for gloo's default 30 minutes
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.dry_timeout))
    ddp.init_from_env("gloo")
    issued = []
    originals = {name: getattr(dist, name) for name in ("all_reduce", "barrier", "broadcast", "all_gather", "reduce_scatter")}

    def counted(name):
        fn = originals[name]

        def call(*a, **k):
            issued.append(name)
            return fn(*a, **k)
        return call
    for name in originals:
        setattr(dist, name, counted(name))
    tr, lab, unl = _build(tempfile.mkdtemp(prefix=f"dct_dry_r{rank}_"), 100 * rank)
    ddp_on = world > 1
    if ddp_on:
        tr.grad_sync = ddp.FlatGradSync(tr.segmentators, measure=False)
    if args.dry_break_rank0_leg:        # the round-4 defect, kept reachable so that the test can show it is caught
        args.no_kernel_events = rank != 0
    res = measure(args, tr, lambda i: _step(tr, lab, unl), DryLegs(), rank, world, ddp_on)
    for name, fn in originals.items():
        setattr(dist, name, fn)
    elapsed, out = res["elapsed"], res["out"]
    sequences = [issued]
    if ddp_on:
        sequences = [None] * world
        dist.all_gather_object(sequences, issued)
    imgs = 3            # tests/test_ddp_cpu.py::_build: 2 models x 1 labeled + 1 unlabeled slice per rank
    line = {"metric": "co-train imgs/sec/node (lab+unlab), dry launch of the rank plumbing (CPU oracle networks over gloo)",
            "value": imgs * world / (elapsed / args.steps), "unit": "imgs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": {"workload": "dry launch: 2xUNet 176x176 C=2 on CPU, bs 1+1 per rank (not a measurement)",
                                            "global_batch": f"{world}+{world}", "parallelism": f"dp{world}"},
            "losses_last_step": {"sup": [float(v) for v in out["sup"]], "jsd": float(out["jsd"])},
            "collectives": {"per_rank": [len(s) for s in sequences], "same_sequence_on_every_rank": all(s == sequences[0] for s in sequences),
                            "behind_the_timed_region_legs": ["meters (world 1 only)", "operand stats (rank 0, no step)", "per-launch events"],
                            "event_leg_ran": res["prof"] is not None or rank != 0}}
    if ddp_on:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f16"])
    ap.add_argument("--grad-compress", default="none", choices=["none", "bf16"],
                    help="N > 1: gradients travel as bf16 on the xGMI ring (half the bytes); default fp32, as the reference's sums are")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE", help="dct_tune_set(KNOB, VALUE) before the run (A/B)")
    ap.add_argument("--attr", action="append", default=[], metavar="NAME=0|1", help="boolean CoTrainer switch (pass_streams, ...) (A/B)")
    ap.add_argument("--net-attr", action="append", default=[], metavar="NAME=0|1", help="boolean switch of every network (relu_bits, pool_codes, ...) (A/B)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="run the N > 1 code path (RCCL process group, gradient exchange, its report) even with one rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=0,
                    help="extra (untimed) training steps before the warm-up: the timed network is then that many Adam steps old "
                         "(operand statistics of a trained net; profiles/r03_cfg2_after_300_steps.json)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--data", default="blob", choices=["blob", "iid"],
                    help="synthetic inputs: blob-structured slices the nets learn (default: the timed network then carries a training net's "
                         "operand statistics), or iid = torch.rand images + torch.randint labels as SURVEY.md 8d words it")
    ap.add_argument("--allow-nan", action="store_true",
                    help="diagnostic runs that skip kernel families (--tune 1100=MASK: garbage results, timing only): do not insist on finite losses")
    ap.add_argument("--no-clock-probe", action="store_true", help="do not sample the shader clock beside the replayed step")
    ap.add_argument("--wgrad-stream", action="store_true", help="weight gradients on a second stream per model (eager only)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from the host instead of replaying the captured step")
    ap.add_argument("--single-stream", action="store_true",
                    help="queue all models on one stream (the mode the per-kernel roofline leg and rocprofv3 kernel durations use)")
    ap.add_argument("--global-batch", default="", metavar="L+U",
                    help="fixed GLOBAL batch (labeled+unlabeled, e.g. 64+64 for cfg4: BASELINE.json configs[3]) split evenly over the "
                         "ranks; the line then says \"scaling\": \"strong\".  Default: the configuration's per-GPU batch on every rank (weak)")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of K steps each; the median is reported (default 3)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="exercise the N-rank launch path on CPU (gloo, oracle-injected networks); prints one JSON line, measures nothing")
    ap.add_argument("--dry-timeout", type=int, default=180, help="--dry-launch: seconds before a stuck gloo collective fails the run")
    ap.add_argument("--dry-break-rank0-leg", action="store_true",
                    help="--dry-launch self-test: run the per-launch event leg on rank 0 only (the round-4 defect); the run must FAIL")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))       # (nothing above this line has touched the GPU)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_launch:
        return dry_launch(args, rank, world)
    if world != args.gpus:
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py measures the HIP path; it needs an MI355X"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    ddp_on = world > 1 or args.force_ddp
    if world > 1:
        from dct_amd import ddp
        ddp.init_from_env("nccl")
    elif args.force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)

    from dct_amd import _lib
    cfg = dict(CONFIGS[args.config])
    scaling = "weak"
    if args.global_batch:
        gl, gu = (int(v) for v in args.global_batch.split("+"))
        if gl % world or gu % world or gl < world or gu < world:
            raise SystemExit(f"--global-batch {args.global_batch} does not split evenly over {world} ranks")
        cfg["B_l"], cfg["B_u"] = gl // world, gu // world
        cfg["desc"] += f" -- fixed global batch {gl}+{gu} over {world} rank(s): {cfg['B_l']}+{cfg['B_u']} per GPU"
        scaling = "strong"
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[args.dtype]

    def sync_factory(segs):
        from dct_amd.ddp import FlatGradSync
        return FlatGradSync(segs, compress=None if args.grad_compress == "none" else args.grad_compress, measure=True)

    for kv in args.tune:
        k, v = kv.split("=")
        _lib.check(_lib.load().dct_tune_set(int(k), int(v)), f"dct_tune_set({kv})")
    tr, lab, unl = make_trainer(cfg, dtype, device, rank, 2 if (args.force_ddp and world == 1) else world, sync_factory, data=args.data)
    for kv in args.attr:
        k, v = kv.split("=")
        assert hasattr(tr, k), k
        setattr(tr, k, bool(int(v)) if isinstance(getattr(tr, k), bool) or getattr(tr, k) is None else int(v))
    for kv in args.net_attr:
        k, v = kv.split("=")
        for seg in tr.segmentators:
            assert hasattr(seg.torchnet, k), k
            setattr(seg.torchnet, k, bool(int(v)) if isinstance(getattr(seg.torchnet, k), bool) else int(v))
    S = cfg["S"]
    nb = len(unl)
    tr.model_streams = not args.single_stream

    def set_side_streams(flag):
        for seg in tr.segmentators:
            if hasattr(seg.torchnet, "wgrad_side_stream"):
                seg.torchnet.wgrad_side_stream = flag
    set_side_streams(args.wgrad_stream and not args.single_stream)
    tr.use_hip_graph = not args.no_graph

    def one_step(i):
        lb = [(lab[m][i % nb][0][0], lab[m][i % nb][0][1]) for m in range(S)]
        ub = (unl[i % nb][0][0], unl[i % nb][0][1])
        return tr._run_step(lb, ub, True, cfg["train_adv"], (0, 1) if cfg["train_adv"] else None)

    # Setup (not part of the W warm-up steps or of the timed region): first-touch allocations, kernel attribute setup
    # and the one-time HIP-graph capture of the step (two eager steps, the capture, and the first replays, which upload
    # the graph) -- the equivalent of a JIT / engine-build phase.  Then W untimed warm-up steps, then exactly K timed.
    for i in range(SETUP_STEPS + args.train_steps):
        one_step(i)
    torch.cuda.synchronize()

    def meters_leg(base):
        # what _train_loop adds around _run_step: DiceMeter.add on the labeled and unlabeled predictions, the loss meters, and the
        # progress read-out every 10 steps
        from dct_amd.metrics import AverageValueMeter, DiceMeter
        axes = list(range(1, cfg["C"]))
        dm = [DiceMeter(report_axises=axes, method='2d', C=cfg["C"]) for _ in range(2 * S)]
        lm = [AverageValueMeter() for _ in range(S)]
        nm = min(args.steps, 20)
        torch.cuda.synchronize()
        tm = time.perf_counter()
        for i in range(nm):
            o = one_step(base + i)
            lb_gt = [lab[m][(base + i) % nb][0][1] for m in range(S)]
            ub_gt = unl[(base + i) % nb][0][1]
            for m in range(S):
                dm[m].add(o["preds"][m], lb_gt[m])
                dm[S + m].add(o["unlab_probs"][m], ub_gt)
                lm[m].add(o["sup"][m])
            if i % 10 == 0 or i == nm - 1:
                [float(d.value()[0][0]) for d in dm[:S]]
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - tm) / nm

    def operand_leg():
        # Operand statistics of the timed network (MI355X_MICROARCH.md, DVFS give-back: zero / trivial operands clock higher):
        # fraction of exactly-zero activations at every conv output of model 0 on the next batch.
        if cfg["arch"] != "unet":
            return None
        net0 = tr.segmentators[0].torchnet
        keep_only, net0.pool_only = getattr(net0, "pool_only", False), False     # this pass wants every block's full-resolution output
        keep_dp, net0.fuse_drop_pool = getattr(net0, "fuse_drop_pool", False), False   # ... the dropped fourth level included
        _, tape = net0.plan_forward(torch.cat((lab[0][0][0][0], unl[0][0][0]), dim=0), True)
        torch.cuda.synchronize()
        net0.pool_only, net0.fuse_drop_pool = keep_only, keep_dp
        keys = ["a1", "d1", "a2", "d2", "a3", "d3", "a4", "d4", "c1", "c2", "e4a", "e4b", "e3a", "e3b", "e2a", "e2b", "e1a", "e1b"]
        stats = {"zero_fraction_of_bf16_activations": {k: round(float((tape[k] == 0).float().mean()), 4) for k in keys if k in tape and torch.is_tensor(tape[k])},
                 "note": "ReLU outputs (d4 / c2 include dropout p=0.5); ~0.5 is what a trained ReLU net carries"}
        del tape
        return stats

    def enter_event_leg():
        set_side_streams(False)
        tr.use_hip_graph = False      # event records are host calls around each launch
        tr.model_streams = False      # per-kernel durations are taken with one kernel on the device at a time

    m = measure(args, tr, one_step, HipLegs(device), rank, world, ddp_on, meters_leg=meters_leg,
                operand_leg=operand_leg, enter_event_leg=enter_event_leg)
    regions, elapsed, out = m["regions"], m["elapsed"], m["out"]
    exchange, meters_ms, operand_stats, prof = m["exchange"], m["meters_ms"], m["operand_stats"], m["prof"]
    clock = m["clock"]
    losses = dict(sup=[float(s) for s in out["sup"]], jsd=float(out["jsd"]))
    assert args.allow_nan or (all(v == v for v in losses["sup"]) and losses["jsd"] == losses["jsd"]), "NaN loss in the timed region"

    ms_per_step = 1e3 * elapsed / args.steps
    imgs_per_step = S * cfg["B_l"] + cfg["B_u"]
    value = imgs_per_step * world / (elapsed / args.steps)

    result = {
        "metric": "co-train imgs/sec/node (lab+unlab), 2xUNet ACDC 256x256" if args.config in ("cfg2", "cfg3")
        else f"co-train imgs/sec/node (lab+unlab), {cfg['S']}x{cfg['arch']} {cfg['H']}x{cfg['H']}",
        "value": value, "unit": "imgs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "timed_regions": {"count": len(regions), "steps_each": args.steps, "ms_per_step_each": [round(1e3 * r / args.steps, 4) for r in regions],
                          "reported": "median"},
        "config": {"workload": f"{args.config}: {cfg['desc']}", "imgs_per_step_per_gpu": imgs_per_step,
                   "global_batch": f"{cfg['B_l'] * world}+{cfg['B_u'] * world}", "parallelism": f"dp{world}",
                   "inputs": "blob-structured synthetic slices (tests/helpers.py::blob_batches)" if args.data == "blob"
                   else "iid: torch.rand images in [0, 1), torch.randint labels (tests/helpers.py::batches; SURVEY.md 8d)",
                   "weights": "random init (xavier_normal), reference architecture; trained for the setup + warm-up steps on these inputs",
                   "adam_steps_before_the_timed_region": SETUP_STEPS + args.train_steps + args.warmup},
        "losses_last_step": losses,
    }
    caps = list(tr._step_graphs._graphs.values()) if getattr(tr, "_step_graphs", None) is not None else []
    if caps and getattr(caps[0], "program", None) is not None:
        prog = caps[0].program
        from dct_amd.trainer.stream_sched import queue_probe_report
        probe = queue_probe_report(device)
        result["config"]["step_execution"] = {"mode": "program of per-stream HIP graphs (trainer/stream_sched.py)", "graphs": prog.n_graphs,
                                              "kernel_nodes": prog.n_nodes, "ops": len(prog.ops),
                                              "hardware_queue_groups_found": probe.get("sizes", []),
                                              # four groups are what the multi-queue layouts are dealt over; anything else means the
                                              # timing probe was disturbed and the streams were dealt blindly (slower, still correct)
                                              "hardware_queue_probe_ok": bool(probe.get("ok", probe.get("source") == "DCT_HW_QUEUES")),
                                              "hardware_queue_probe": probe}
    elif caps:
        result["config"]["step_execution"] = {"mode": "one HIP graph"}
    else:
        result["config"]["step_execution"] = {"mode": "eager launches"}
    if exchange is not None:
        result["gradient_exchange"] = exchange
    if meters_ms is not None:
        result["ms_per_step_with_meters"] = meters_ms
    if operand_stats is not None:
        result["operand_stats"] = operand_stats
    if rank == 0:
        if prof is not None and cfg["arch"] == "unet":
            gemm_f, small_f = unet_conv_flops(cfg["H"], cfg["H"], cfg["C"])
            passes = S * (cfg["B_l"] + cfg["B_u"])                      # image-passes with fwd+dgrad+wgrad
            fgsm_imgs = (cfg["B_l"] + cfg["B_u"]) if cfg["train_adv"] else 0
            # per step: igemm class runs fwd + dgrad (2F per image-pass); wgrad class runs F per image-pass;
            # FGSM adds one fwd+dgrad pass on model b (no wgrad) and one full pass on model a
            flops = {"igemm": (passes + 2 * fgsm_imgs) * 2 * gemm_f, "wgrad": (passes + fgsm_imgs) * gemm_f}
            # the timed steps of the three-queue adversarial layout take the FGSM generator's encoder from model b's joint pass (same input, same
            # weights): those convolutions are not launched a second time and are not counted as work done at step level.  (The per-kernel leg runs
            # on one stream, i.e. the sequential layout, which launches them: the kernel-class figures count them.)
            shared_f = fgsm_imgs * unet_encoder_gemm_flops(cfg["H"], cfg["H"]) if (fgsm_imgs and m.get("shared_fgsm_encoder")) else 0.0
            per = {k: {"ms_per_step": v["ms"] / args.steps, "launches_per_step": v["launches"] / args.steps}
                   for k, v in prof.items()}
            dom = max(("igemm", "wgrad"), key=lambda k: per[k]["ms_per_step"])
            ach = flops[dom] / (per[dom]["ms_per_step"] * 1e-3) / 1e12
            peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
            result["roofline"] = {
                "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": pmc_traffic(args.config),
                "kernel": {"igemm": "igemm2_kernel + igemm3m_kernel (conv fwd + data-grad implicit GEMM: per-tap and shared-halo tiles, incl. split-K epilogue)",
                           "wgrad": "wgrad2_kernel + wgrad3_kernel (weight-grad GEMM: per-tap and filter-row tiles, incl. fixed-order reduce)"}[dom],
                "algorithmic_flops_per_step": flops[dom], "kernel_ms_per_step": per[dom]["ms_per_step"],
                "avg_launch_us": 1e3 * per[dom]["ms_per_step"] / max(per[dom]["launches_per_step"], 1),
                "algorithmic_flops_per_launch": flops[dom] / max(per[dom]["launches_per_step"], 1),
                "conv_stack": {"achieved": (flops["igemm"] + flops["wgrad"]) /
                               ((per["igemm"]["ms_per_step"] + per["wgrad"]["ms_per_step"]) * 1e-3) / 1e12,
                               "unit": "TFLOP/s"},
                "per_class_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in per.items()},
                "measured": "HIP events around every launch on the launch stream, over the same K steps re-run right after the timed region",
                # the same FLOPs over the timed step (both model streams overlapping): what the job as a whole makes of the matrix peak
                "step_level": {"achieved": (flops["igemm"] + flops["wgrad"] - shared_f) / (ms_per_step * 1e-3) / 1e12, "unit": "TFLOP/s",
                               "frac": (flops["igemm"] + flops["wgrad"] - shared_f) / (ms_per_step * 1e-3) / 1e12 / peak,
                               **({"not_counted": "the FGSM generator's encoder, taken from model b's joint pass (%.0f GFLOP per step)" % (shared_f / 1e9)} if shared_f else {})},
                # counter evidence of an EARLIER run, read from the committed profile files (not measured by this run)
                "archived_counters": pmc_mfma_busy(args.config),
                # measured by THIS run, beside the replayed (captured) step
                "shader_clock_ghz_during_the_step": clock,
            }
        elif prof is not None:
            # Enet: HBM bound.  Algorithmic bytes = conv in+out activation elements x 2 B (bf16) x 3 (fwd, dgrad, wgrad)
            # per training image-pass; the FGSM generator pass counts fwd + dgrad (x 2).
            elems = ENET_ACT_ELEMS[cfg["H"]]
            esz = 4 if args.dtype == "f32" else 2
            passes = S * (cfg["B_l"] + cfg["B_u"])
            fgsm_imgs = (cfg["B_l"] + cfg["B_u"]) if cfg["train_adv"] else 0
            alg_bytes = elems * esz * (3 * passes + (2 + 3) * fgsm_imgs)
            ms = prof["other"]["ms"] / args.steps
            launches = prof["other"]["launches"] / args.steps
            ach = alg_bytes / (ms * 1e-3) / 1e9
            result["roofline"] = {
                "bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": pmc_traffic(args.config),
                "kernel": "enet_* kernel family (fused conv / BN / tail / wgrad; aggregate, the net is launch bound)",
                "algorithmic_bytes_per_step": alg_bytes, "kernel_ms_per_step": ms,
                "avg_launch_us": 1e3 * ms / max(launches, 1), "launches_per_step": launches,
                "per_class_ms_per_step": {k: round(v["ms"] / args.steps, 4) for k, v in prof.items()},
                "measured": "HIP events around every launch on the launch stream, over the same K steps re-run right after the timed region",
                # `frac` above divides by the SERIALISED kernel time of the eager single-stream leg; the timed step runs its chains on
                # four hardware queues, so the job-level figure is the same bytes over ms_per_step:
                "step_level": {"achieved": alg_bytes / (ms_per_step * 1e-3) / 1e9, "unit": "GB/s",
                               "frac": alg_bytes / (ms_per_step * 1e-3) / 1e9 / 8000.0},
            }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg)
    if ddp_on:
        # RCCL writes a version banner to stdout when a rank tears down: tear down first, then rank 0 prints the JSON line,
        # so that it stays the LAST line of the job's output; the ranks leave through os._exit (no further teardown output)
        dist.barrier()
        dist.destroy_process_group()
        sys.stdout.flush()
        if rank == 0:
            time.sleep(1.0 if world > 1 else 0.0)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if ddp_on:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()

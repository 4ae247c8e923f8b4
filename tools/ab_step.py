#!/usr/bin/env python3
"""In-process A/B of a dct_tune_set knob on the whole co-training step (eager launches, per-model streams): devices and
runs differ by +-3 %, so the two arms alternate inside one process.

    python tools/ab_step.py --knob 11 --a 0 --b 1 [--config cfg2] [--steps 15] [--rounds 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import bench  # noqa: E402
from dct_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", type=int, default=-1)
    ap.add_argument("--attr", default="", help="toggle a boolean CoTrainer attribute (a = False, b = True) instead of a library knob")
    ap.add_argument("--a", type=int, default=0)
    ap.add_argument("--b", type=int, default=1)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--single-stream", action="store_true")
    ap.add_argument("--pre", action="append", default=[], metavar="KNOB=VALUE", help="knobs set once before both arms")
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    dev = torch.device("cuda", 0)
    tr, lab, unl = bench.make_trainer(cfg, torch.bfloat16, dev, 0, 1, None)
    tr.use_hip_graph = False
    tr.model_streams = not args.single_stream
    S, nb = cfg["S"], len(unl)
    lib = _lib.load()
    for kv in args.pre:
        k, v = kv.split("=")
        assert lib.dct_tune_set(int(k), int(v)) == 0, kv

    def one_step(i):
        lb = [(lab[m][i % nb][0][0], lab[m][i % nb][0][1]) for m in range(S)]
        ub = (unl[i % nb][0][0], unl[i % nb][0][1])
        return tr._run_step(lb, ub, True, cfg["train_adv"], (0, 1) if cfg["train_adv"] else None)

    def arm(v):
        if args.attr:
            setattr(tr, args.attr, bool(v))
        else:
            lib.dct_tune_set(args.knob, v)
        for i in range(3):
            one_step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_step(i)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / args.steps

    for i in range(5):
        one_step(i)
    for r in range(args.rounds):
        ta, tb = arm(args.a), arm(args.b)
        print(f"round {r}: knob {args.knob} = {args.a}: {ta:.3f} ms/step   = {args.b}: {tb:.3f} ms/step   ({100 * (ta - tb) / ta:+.1f} %)")


if __name__ == "__main__":
    main()

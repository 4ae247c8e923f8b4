#!/usr/bin/env python3
"""When do the phases of the captured cfg2 step really happen?  A tracing profiler serialises the two model chains (profiles/
r05_traced_step_schedule_head.txt: depth 1 for 80 % of a 6.8 ms step); this tool dates them in the REAL replay: one-thread stamp kernels
(include/dct.h dct_stamp: the 100 MHz reference counter) queued on each model's stream at forward start / forward + loss done / backward
done / optimizer done, captured with the step and replayed with it.

    python tools/phase_stamps.py [--config cfg2] [--steps 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    dev = torch.device("cuda", 0)
    tr, lab, unl = bench.make_trainer(cfg, torch.bfloat16, dev, 0, 1, None)
    S, nb = cfg["S"], len(unl)
    R = tr.PHASE_RING
    tr.phase_stamps = torch.zeros(S, 4, 1 + R, dtype=torch.int64, device=dev)

    def one_step(i):
        lb = [(lab[m][i % nb][0][0], lab[m][i % nb][0][1]) for m in range(S)]
        ub = (unl[i % nb][0][0], unl[i % nb][0][1])
        return tr._run_step(lb, ub, True, cfg["train_adv"], (0, 1) if cfg["train_adv"] else None)

    for i in range(10):
        one_step(i)
    torch.cuda.synchronize()
    names = ["forward starts", "forward + loss done", "backward done", "optimizer done"]

    def timeline(K):
        """[K, S, 4] microseconds of the last K steps"""
        st = tr.phase_stamps.cpu()
        n = int(st[0, 0, 0])
        out = torch.zeros(K, S, 4, dtype=torch.float64)
        for j in range(K):
            idx = 1 + (n - K + j) % R
            out[j] = st[:, :, idx].to(torch.float64) / 100.0
        return out

    for mode in ("pipelined (no host synchronisation between replays)", "one replay at a time (device synchronise after every step)"):
        for i in range(args.steps):
            one_step(10 + i)
            if mode.startswith("one"):
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        tl = timeline(args.steps - 4)
        t0 = tl[:, :, 0].min(dim=1).values                                   # each step's first forward start
        rel = (tl - t0[:, None, None]).mean(0)
        period = float((t0[1:] - t0[:-1]).mean())
        gap = float((t0[1:] - tl[:-1, :, 3].max(dim=1).values).mean())
        print(f"{args.config}, {mode}: mean over {tl.shape[0]} replays, microseconds from the step's first forward start")
        if cfg["train_adv"]:
            # the three-queue adversarial step (CoTrainer._run_step_adv_chain) stamps other sites: model a's queue / the adversarial chain + model b
            rows = (("model a (its own queue)", ["forward starts", "forward + loss done", "backward done", "adversarial backward + optimizer done"]),
                    ("model b + adversarial chain", ["forward starts", "adversarial batch ready", "adversarial forward of model a done", "backward + optimizer done (third queue)"]))
            for m, (who, labels) in enumerate(rows):
                print(f"  {who}: " + ";  ".join(f"{lb} {float(rel[m][k]):.1f}" for k, lb in enumerate(labels)))
            print(f"  step period {period:.1f} us")
            continue
        for k, nme in enumerate(names):
            print(f"  {nme:22s} " + "  ".join(f"model {m}: {float(rel[m][k]):8.1f}" for m in range(S)))
        print(f"  step period {period:.1f} us;  last optimizer done -> next step's first forward start: {gap:.1f} us")


if __name__ == "__main__":
    main()

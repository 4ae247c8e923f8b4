#!/usr/bin/env python3
"""How much of a co-training step is host (Python + launch) time?  Runs K steps of a bench.py config and prints
the time to ENQUEUE them (no synchronisation) next to the time until the device has finished them.  When the
two are close the step is launch-bound and kernel work no longer sets the rate.

    python tools/host_time.py [--config cfg2] [--steps 20] [--single-stream]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--single-stream", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    dev = torch.device("cuda", 0)
    tr, lab, unl = bench.make_trainer(cfg, torch.bfloat16, dev, 0, 1, None)
    tr.model_streams = not args.single_stream
    tr.use_hip_graph = not args.no_graph
    for seg in tr.segmentators:
        if hasattr(seg.torchnet, "wgrad_side_stream"):
            seg.torchnet.wgrad_side_stream = False
    S, nb = cfg["S"], len(unl)

    def one_step(i):
        lb = [(lab[m][i % nb][0][0], lab[m][i % nb][0][1]) for m in range(S)]
        ub = (unl[i % nb][0][0], unl[i % nb][0][1])
        return tr._run_step(lb, ub, True, cfg["train_adv"], (0, 1) if cfg["train_adv"] else None)

    for i in range(5):
        one_step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{args.config}: enqueue {1e3 * (t1 - t0) / args.steps:.2f} ms/step, device done {1e3 * (t2 - t0) / args.steps:.2f} ms/step")


if __name__ == "__main__":
    main()

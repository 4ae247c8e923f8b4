"""The captured step PROGRAM of a bench configuration (trainer/stream_sched.py): its segments, and host vs device time of a replay.
    python tools/probe_step_program.py cfg4 [switch=0|1 ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
cfg = bench.CONFIGS[name]
dev = torch.device("cuda:0")
tr, lab, unl = bench.make_trainer(cfg, torch.bfloat16, dev, 0, 1, None)
for kv in sys.argv[2:]:                      # CoTrainer switches: name=0|1
    if "=" in kv:
        k_, v_ = kv.split("=")
        assert hasattr(tr, k_), k_
        setattr(tr, k_, bool(int(v_)))
S, nb = cfg["S"], len(unl)


def one_step(i):
    lb = [(lab[m][i % nb][0][0], lab[m][i % nb][0][1]) for m in range(S)]
    ub = (unl[i % nb][0][0], unl[i % nb][0][1])
    return tr._run_step(lb, ub, True, cfg["train_adv"], (0, 1) if cfg["train_adv"] else None)


for i in range(8):
    one_step(i)
torch.cuda.synchronize()
cap = next(iter(tr._step_graphs._graphs.values()))
prog = cap.program
names = {}


def nm(k):
    if k == 'main':
        return 'main'
    return names.setdefault(k.cuda_stream, "s%d" % len(names))


print("# %s: %d graph segments, %d kernel nodes, %d ops" % (name, prog.n_graphs, prog.n_nodes, len(prog.ops)))
for op in prog.ops:
    if op[0] == 'graph':
        print("  graph %-5s %5d nodes" % (nm(op[1]), op[3]))
    elif op[0] == 'wait':
        print("  wait  " + " ".join("%s<-%s" % (nm(d), nm(s)) for d, s, _ in op[1]))
    elif op[0] in ('record', 'wait_event'):
        print("  %s %s" % (op[0], nm(op[1])))
    else:
        print("  call  on %s" % nm(op[1]))
groups = __import__("dct_amd.trainer.stream_sched", fromlist=["x"]).queue_groups(dev)
print("# hardware-queue groups of the candidate streams:", [[nm(s) if s.cuda_stream in names else "-" for s in g] for g in groups])
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        one_step(i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    tt = time.perf_counter() - t0
    print("# 10 steps: host issue %.2f ms/step, until done %.2f ms/step" % (th * 100, tt * 100))
# each segment alone (synchronised after each op): the serial sum
main = torch.cuda.current_stream()
tot = {}
for rep in range(3):
    for op in prog.ops:
        if op[0] != 'graph':
            continue
        st = main if op[1] == 'main' else op[1]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            op[2].replay()
        torch.cuda.synchronize()
        tot[id(op)] = min(tot.get(id(op), 1e9), time.perf_counter() - t0)
print("# serial sum of the segments replayed one at a time: %.2f ms" % (sum(tot.values()) * 1e3))
for op in prog.ops:
    if op[0] == 'graph':
        print("  graph %-5s %5d nodes %8.3f ms  (%.2f us per node)" % (nm(op[1]), op[3], tot[id(op)] * 1e3, tot[id(op)] * 1e6 / op[3]))

# timeline of one concurrent replay: HIP events around every segment on its own stream
lbi = [(lab[m][0][0][0], lab[m][0][0][1]) for m in range(S)]
main = torch.cuda.current_stream()
for rep in range(2):
    marks = []
    torch.cuda.synchronize()
    e00 = torch.cuda.Event(enable_timing=True)
    e00.record(main)
    for op in prog.ops:
        if op[0] == 'graph':
            st = main if op[1] == 'main' else op[1]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.set_stream(st)
            a.record(st)
            op[2].replay()
            b.record(st)
            marks.append((nm(op[1]), op[3], a, b))
        elif op[0] == 'wait':
            for d, s_, ev in op[1]:
                ev.record(main if s_ == 'main' else s_)
                (main if d == 'main' else d).wait_event(ev)
        elif op[0] == 'record':
            op[2].record(main if op[1] == 'main' else op[1])
        elif op[0] == 'wait_event':
            (main if op[1] == 'main' else op[1]).wait_event(op[2])
        else:
            torch.cuda.set_stream(main if op[1] == 'main' else op[1])
            op[2]()
    torch.cuda.set_stream(main)
    torch.cuda.synchronize()
print("# timeline of one replay (ms from the start): stream, nodes, start, end")
for n_, k, a, b in marks:
    print("  %-5s %5d  %7.2f -> %7.2f   (%.2f ms)" % (n_, k, e00.elapsed_time(a), e00.elapsed_time(b), a.elapsed_time(b)))

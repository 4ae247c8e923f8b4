import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
mode = sys.argv[2] if len(sys.argv) > 2 else "multi"
multi = mode in ("multi", "models")
side = mode in ("multi", "side")
cfg = bench.CONFIGS[cfgname]
dev = torch.device("cuda", 0)
tr, lab, unl = bench.make_trainer(cfg, torch.bfloat16, dev, 0, 1, None)
tr.model_streams = multi
for seg in tr.segmentators:
    if hasattr(seg.torchnet, "wgrad_side_stream"):
        seg.torchnet.wgrad_side_stream = side
S, nb = cfg["S"], len(unl)
def one_step(i):
    lb = [(lab[m][i % nb][0][0], lab[m][i % nb][0][1]) for m in range(S)]
    ub = (unl[i % nb][0][0], unl[i % nb][0][1])
    return tr._run_step(lb, ub, True, cfg["train_adv"], (0, 1) if cfg["train_adv"] else None)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3): one_step(i)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10): one_step(i)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{cfgname} mode={mode} eager: {(t2-t0)*100:.2f} ms/step")
G = torch.cuda.CUDAGraph()
with torch.cuda.graph(G):
    out = one_step(0)
torch.cuda.synchronize()
for _ in range(3): G.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): G.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{cfgname} mode={mode} graph: enqueue {(t1-t0)*50:.2f} ms, done {(t2-t0)*50:.2f} ms/step; sup {[float(v) for v in out['sup']]}")

import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dct_amd
from dct_amd.arch import get_arch
dev = torch.device("cuda", 0)
arch = sys.argv[1] if len(sys.argv) > 1 else "unet"
H = 256 if arch == "unet" else 200
C = 4 if arch == "unet" else 2
torch.manual_seed(0)
net = get_arch(arch, {"num_classes": C, "compute_dtype": torch.bfloat16}).to(dev).train()
x = torch.rand(16, 1, H, H, device=dev)
g = torch.randn(16, C, H, H, device=dev).contiguous(memory_format=torch.channels_last) * 1e-3
def step():
    y = net(x)
    torch.autograd.backward([y], [g])
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"eager: enqueue {(t1-t0)*100:.2f} ms, done {(t2-t0)*100:.2f} ms per fwd+bwd")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
G = torch.cuda.CUDAGraph()
with torch.cuda.graph(G):
    step()
torch.cuda.synchronize()
for _ in range(3): G.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): G.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"graph: enqueue {(t1-t0)*100:.2f} ms, done {(t2-t0)*100:.2f} ms per fwd+bwd")

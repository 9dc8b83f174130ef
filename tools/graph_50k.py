"""Is the eager 50k step leaving the GPU idle between kernels?  Same step, eager vs one captured hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodalfusion_amd.graph import GraphedStep
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
model = bench.build_model(dev, False)
x = torch.randn(N, 1024, device=dev)
n_el = sum(p.numel() for p in model.parameters())
flat = torch.zeros(n_el, device=dev)
off = 0
for p in model.parameters():
    p.grad = flat[off:off + p.numel()].view_as(p); off += p.numel()
step = bench.make_step(model, x, dev, flat, 1)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"eager   {timeit(step):.4f} ms/step")
g = GraphedStep(step)
print(f"graphed {timeit(g):.4f} ms/step")

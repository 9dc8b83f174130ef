"""Throughput with two bags in flight on two HIP streams (bags of one accumulation window are independent)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dt = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
model = bench.build_model(dev, False)
xs = [torch.randn(N, 1024, device=dev).to(dt) for _ in range(2)]
Y, c = torch.tensor([1], device=dev), torch.tensor([0.0], device=dev)
loss_fn = NLLSurvLoss(alpha=0.0)
params = list(model.parameters())
def one(x):
    hz, S, _, _ = model(path_features=x)
    return torch.autograd.grad(loss_fn(hazards=hz, S=S, Y=Y, c=c), params)     # gradients returned, not accumulated
def run(nstreams, steps=200):
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    for s in streams: s.wait_stream(torch.cuda.current_stream())
    def loop(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % nstreams]):
                one(xs[i % 2])
    loop(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(steps); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for ns in (1, 2, 3):
    ms = run(ns)
    print(f"N={N} {dt}: {ns} stream(s): {ms:.4f} ms/bag = {1e3/ms:.0f} bags/s")

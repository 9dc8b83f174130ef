import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
model = bench.build_model(dev, False)
x = torch.randn(N, 1024, device=dev)
Y, c = torch.tensor([1], device=dev), torch.tensor([0.0], device=dev)
loss_fn = NLLSurvLoss(alpha=0.0)
acc = [0.0] * 5
n = 400
for it in range(n + 20):
    if it == 20:
        torch.cuda.synchronize(); acc = [0.0] * 5; T0 = time.perf_counter()
    t0 = time.perf_counter()
    for p in model.parameters(): p.grad = None
    t1 = time.perf_counter()
    hz, S, _, _ = model(path_features=x)
    t2 = time.perf_counter()
    loss = loss_fn(hazards=hz, S=S, Y=Y, c=c)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    del loss, hz, S
    t5 = time.perf_counter()
    for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[i] += d
torch.cuda.synchronize()
tot = (time.perf_counter() - T0) / n * 1e3
print(f"N={N}: wall {tot:.3f} ms/step; host: zero_grad {acc[0]/n*1e3:.3f}  forward {acc[1]/n*1e3:.3f}  loss {acc[2]/n*1e3:.3f}  backward {acc[3]/n*1e3:.3f}  del {acc[4]/n*1e3:.3f}")

"""Where the host time of a small-bag step goes (all numbers: wall ms per call, GPU queue drained each call)."""
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
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def fwd():
    return model(path_features=x)
def fwd_loss():
    hz, S, _, _ = model(path_features=x); return loss_fn(hazards=hz, S=S, Y=Y, c=c)
def full():
    for p in model.parameters(): p.grad = None
    fwd_loss().backward()
def nograd():
    with torch.no_grad(): model(path_features=x)
print(f"N={N}: no_grad fwd {t(nograd):.3f}  fwd {t(fwd):.3f}  fwd+loss {t(fwd_loss):.3f}  full step {t(full):.3f} ms")
import cProfile, pstats, io
hz, S, _, _ = model(path_features=x)
def loss_only():
    return loss_fn(hazards=hz, S=S, Y=Y, c=c)
print(f"loss only {t(loss_only):.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): loss_only()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:2500])

"""Host-side cost of one eager step at a small bag (the GPU is idle most of the time there)."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
model = bench.build_model(dev, False)
x = torch.randn(N, 1024, device=dev)
step = bench.make_step(model, x, dev)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print(f"N={N}: {(time.perf_counter()-t0)/200*1e3:.3f} ms/step eager")
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])

"""One bag size, the one-call step (model.nll_step) in a loop: the target of rocprofv3 --kernel-trace --stats runs.
usage: step_profile.py N [steps] [f32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
N = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"
dev = torch.device("cuda", 0)
model = bench.build_model(dev, False)
x = torch.randn(N, 1024, device=dev)
if bf16: x = x.to(torch.bfloat16)
step = bench.make_step(model, x, dev, None, 1)
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"N={N} {'bf16' if bf16 else 'f32'}: {1e3 * dt / steps:.4f} ms/step")

"""Debug: determinism and position independence of the bf16 fused forward's scores (forward-only and training step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
DEV = torch.device("cuda", 0)
torch.manual_seed(5)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
model = MIL_Attention_fc_surv_path(gate_path=True, model_size_wsi="small", dropout=False, n_classes=4).to(DEV).eval()
x = torch.randn(N, 1024, device=DEV).to(torch.bfloat16)
with torch.no_grad():
    A1 = model(path_features=x, attention_only=True)[0].clone()
    bad = 0
    for i in range(runs):
        A2 = model(path_features=x, attention_only=True)[0]
        d = (A1 != A2).nonzero().flatten()
        if d.numel():
            bad += 1
            print(f"run {i}: {d.numel()} rows differ, first {d[:4].tolist()} (tile {int(d[0]) // 128}, row in tile {int(d[0]) % 128}), max diff {(A1 - A2).abs().max().item():.3g}")
    print(f"forward-only: {bad} of {runs} runs differ from the first")
    perm = torch.randperm(N, device=DEV)
    Ap = model(path_features=x[perm].contiguous(), attention_only=True)[0]
    d = (Ap != A1[perm]).nonzero().flatten()
    print("permuted differing rows:", d.numel(), (Ap - A1[perm]).abs().max().item())

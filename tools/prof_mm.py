import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalfusion_amd import _lib
from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
dev = "cuda"; MODS = ["T1", "T2", "T1Gd", "FLAIR"]
torch.manual_seed(1)
mm = MM_MIL_Attention_fc_surv(input_dim=80, fusion="tensor", n_classes=4).to(dev).train()
kw = {m: torch.randn(512, 1024, device=dev) for m in MODS}
kw["path_features"] = torch.randn(int(os.environ.get("N", 2000)), 1024, device=dev); kw["genomic_features"] = torch.randn(80, device=dev)
Y, c = torch.tensor([1], device=dev), torch.tensor([0.0], device=dev); nll = NLLSurvLoss(alpha=0.0)
def step():
    for p in mm.parameters(): p.grad = None
    hz, S, _, _ = mm(**kw); nll(hazards=hz, S=S, Y=Y, c=c).backward()
for _ in range(3): step()
torch.cuda.synchronize(); _lib.profile_enable(True)
for _ in range(10): step()
torch.cuda.synchronize(); prof = _lib.profile_dump(); _lib.profile_enable(False)
tot = 0
for k, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:28s} launches/step {n/10:5.1f}  us/step {ms*1e3/10:8.1f}"); tot += ms*1e3/10
print("sum us/step", round(tot, 1))

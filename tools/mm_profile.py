"""mm_attention_mil (config 4) in a loop for rocprofv3: usage mm_profile.py [concat|tensor] [Np] [steps] [eager|graph|step]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
fusion = sys.argv[1] if len(sys.argv) > 1 else "tensor"
Np = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = "cuda"
torch.manual_seed(1)
mm = MM_MIL_Attention_fc_surv(input_dim=80, fusion=fusion, n_classes=4).to(dev).train()
kw = {m: torch.randn(512, 1024, device=dev) for m in ["T1", "T2", "T1Gd", "FLAIR"]}
kw["path_features"] = torch.randn(Np, 1024, device=dev); kw["genomic_features"] = torch.randn(80, device=dev)
if os.environ.get("MMF_MM_BF16") == "1": kw["path_features"] = kw["path_features"].to(torch.bfloat16)
if os.environ.get("MMF_MM_FORK_MIN"): mm.mmf_fork_min_one_call = int(os.environ["MMF_MM_FORK_MIN"])
Y, c = torch.tensor([1], device=dev), torch.tensor([0.0], device=dev); nll = NLLSurvLoss(alpha=0.0)
def step():
    for p in mm.parameters(): p.grad = None
    hz, S, _, _ = mm(**kw); nll(hazards=hz, S=S, Y=Y, c=c).backward()
mode = sys.argv[4] if len(sys.argv) > 4 else "eager"
if mode == "graph":
    from multimodalfusion_amd.graph import GraphedStep
    for p in mm.parameters(): p.grad = torch.zeros_like(p)
    def gstep():
        for p in mm.parameters(): p.grad.zero_()
        hz, S, _, _ = mm(**kw); nll(hazards=hz, S=S, Y=Y, c=c).backward()
    step = GraphedStep(gstep)
if mode == "step":                      # the one-call step (MM_MIL_Attention_fc_surv.nll_step), what the loop mirror runs
    def step():
        for p in mm.parameters(): p.grad = None
        mm.nll_step(Y, c, alpha=0.0, **kw)
for _ in range(10): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize(); print(f"host issue {t_issue / steps * 1e3:.4f} ms/step;", end=" "); print(f"mm {fusion} {mode} Np={Np}: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms/step")

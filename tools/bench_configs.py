"""Step times of the other BASELINE.json configs (parity-test cases, not the headline):
   config 2: radio_attention_mil 4 x (512 x 1024) + omic MaxNet (B=128, Cox);  config 3: mm_attention_mil (50k path bag);
   config 4: mm_attention_mil with a 100k x 1024 bf16 path bag (bf16-storage kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multimodalfusion_amd.models import MIL_Attention_fc_surv_radio, MaxNet, MM_MIL_Attention_fc_surv
from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss, CoxSurvLoss
dev = "cuda"
MODS = ["T1", "T2", "T1Gd", "FLAIR"]
def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
torch.manual_seed(1)
Y, c = torch.tensor([1], device=dev), torch.tensor([0.0], device=dev)
nll = NLLSurvLoss(alpha=0.0)
# radio
radio = MIL_Attention_fc_surv_radio(n_classes=4).to(dev).train()
rx = {m: torch.randn(512, 1024, device=dev) for m in MODS}
def radio_step():
    for p in radio.parameters(): p.grad = None
    hz, S, _, _ = radio(**rx); nll(hazards=hz, S=S, Y=Y, c=c).backward()
print(f"radio_attention_mil 4x512x1024 fwd+nll+bwd: {timeit(radio_step):.3f} ms/step")
# omic
omic = MaxNet(input_dim=36, bag_loss="cox_surv").to(dev).train()
ox = torch.randn(128, 36, device=dev); ot = torch.rand(128, dtype=torch.float64) * 100; oc = (torch.rand(128, device=dev) < 0.5).float()
cox = CoxSurvLoss()
def omic_step():
    for p in omic.parameters(): p.grad = None
    risk = omic(genomic_features=ox)[0]; cox(risks=risk, times=ot, c=oc).backward()
print(f"max_net B=128 G=36 Cox fwd+bwd, composable path: {timeit(omic_step):.3f} ms/step")
ot_dev = ot.to(dev) if torch.is_tensor(ot) else torch.as_tensor(ot, dtype=torch.float64).to(dev)
def omic_one_launch():
    for p in omic.parameters(): p.grad = None
    omic.cox_step(ox, ot_dev, oc)
print(f"max_net B=128 G=36 Cox fwd+bwd, one launch (MaxNet.cox_step): {timeit(omic_one_launch):.3f} ms/step")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
grads = [torch.zeros_like(p) for p in omic.parameters()]
torch.cuda.synchronize(); ev[0].record()
for _ in range(200): omic.cox_step(ox, ot_dev, oc, grad_out=grads, accumulate=False)
ev[1].record(); torch.cuda.synchronize()
print(f"   the same, 200 back-to-back steps by HIP events: {ev[0].elapsed_time(ev[1]) / 200:.4f} ms/step")
# multimodal
for fusion in ("concat", "tensor"):
    mm = MM_MIL_Attention_fc_surv(input_dim=80, fusion=fusion, n_classes=4).to(dev).train()
    kw = dict(rx); kw["path_features"] = torch.randn(50000, 1024, device=dev); kw["genomic_features"] = torch.randn(80, device=dev)
    def mm_step():
        for p in mm.parameters(): p.grad = None
        hz, S, _, _ = mm(**kw); nll(hazards=hz, S=S, Y=Y, c=c).backward()
    t = timeit(mm_step, n=20, w=3)
    print(f"mm_attention_mil ({fusion}) 50k path + 4x512 radio + omic[80]: {t:.3f} ms/step = {1e3/t:.0f} bags/s")
# config 5: multimodal with a 100k bf16 path bag
for fusion in ("concat", "tensor"):
    mm = MM_MIL_Attention_fc_surv(input_dim=80, fusion=fusion, n_classes=4).to(dev).train()
    kw = dict(rx); kw["path_features"] = torch.randn(100000, 1024, device=dev).to(torch.bfloat16); kw["genomic_features"] = torch.randn(80, device=dev)
    def mm_step():
        for p in mm.parameters(): p.grad = None
        hz, S, _, _ = mm(**kw); nll(hazards=hz, S=S, Y=Y, c=c).backward()
    t = timeit(mm_step, n=20, w=3)
    print(f"mm_attention_mil ({fusion}) 100k bf16 path + 4x512 radio + omic[80]: {t:.3f} ms/step = {1e3/t:.0f} bags/s")
# forward-only (inference consumers, infer.py): eval + no_grad takes the no-save kernels
from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
pm = MIL_Attention_fc_surv_path(n_classes=4).to(dev).eval()
for n, dt in ((50000, torch.float32), (100000, torch.bfloat16), (512, torch.float32)):
    xb = torch.randn(n, 1024, device=dev).to(dt)
    def fwd():
        with torch.no_grad(): pm(path_features=xb)
    t = timeit(fwd, n=30, w=5)
    print(f"path head forward-only {n} x 1024 {str(dt).split('.')[-1]}: {t:.3f} ms = {1e3/t:.0f} bags/s")

#!/bin/bash
# round 4: where the host time of the MM one-call step goes, and the bench leg next to the stand-alone loop
R=$GRAFT_REPO_ROOT
cd $R
python3 tools/mm_profile.py concat 50000 200 step 2>&1 | tail -1
python3 tools/mm_profile.py concat 50000 30 step 2>&1 | tail -1
python3 tools/mm_profile.py concat 8 300 step 2>&1 | tail -1
python3 - <<'PY'
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
dev = "cuda"
mm = MM_MIL_Attention_fc_surv(input_dim=80, fusion="concat", n_classes=4).to(dev).train()
kw = {m: torch.randn(512, 1024, device=dev) for m in ["T1", "T2", "T1Gd", "FLAIR"]}
kw["path_features"] = torch.randn(50000, 1024, device=dev); kw["genomic_features"] = torch.randn(80, device=dev)
Y, c = torch.tensor([1], device=dev), torch.tensor([0.0], device=dev)
def step():
    for p in mm.parameters(): p.grad = None
    mm.nll_step(Y, c, alpha=0.0, **kw)
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:6000])
PY

"""Does the MM step's side stream share a hardware queue with the main stream?  The k-th pool stream / a high-priority one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
dev = torch.device("cuda", 0)
Y, c = torch.tensor([1], device=dev), torch.tensor([0.0], device=dev)
mm = MM_MIL_Attention_fc_surv(input_dim=80, fusion="concat", n_classes=4).to(dev).train()
kw = {m: torch.randn(512, 1024, device=dev) for m in ["T1", "T2", "T1Gd", "FLAIR"]}
kw["path_features"] = torch.randn(50000, 1024, device=dev); kw["genomic_features"] = torch.randn(80, device=dev)
params = list(mm.parameters())
def timeit(fn, steps=60, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / steps
def one_call():
    for p in params: p.grad = None
    mm.nll_step(Y, c, alpha=0.0, **kw)
for k in range(10):
    st = torch.cuda.Stream(dev)
    mm.__dict__["_mmf_side"] = {torch.cuda.current_stream(dev).cuda_stream: st}
    print(f"pool stream {k} (handle {st.cuda_stream:#x}): {timeit(one_call):.4f} ms")
from multimodalfusion_amd.streams import stream_beside
for k in range(6):
    st = stream_beside([torch.cuda.current_stream(dev)], dev)
    mm.__dict__["_mmf_side"] = {torch.cuda.current_stream(dev).cuda_stream: st}
    print(f"probed stream {k} (handle {st.cuda_stream:#x}): {timeit(one_call):.4f} ms")
for k in range(2):
    st = torch.cuda.Stream(dev, priority=-1)
    mm.__dict__["_mmf_side"] = {torch.cuda.current_stream(dev).cuda_stream: st}
    print(f"high-priority stream {k}: {timeit(one_call):.4f} ms")

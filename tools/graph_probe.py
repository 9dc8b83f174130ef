"""Probe: capture forward + nll_surv + backward of the path head into a hipGraph (eval mode) and replay it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
for N in (1000, 10000, 50000):
    model = bench.build_model(dev, True)       # eval mode: no dropout seed inside the graph
    x = torch.randn(N, 1024, device=dev)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    step = bench.make_step(model, x, dev)
    def step_static():
        from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
        for p in model.parameters():
            p.grad.zero_()
        hz, S, Yh, _ = model(path_features=x)
        loss = NLLSurvLoss(alpha=0.0)(hazards=hz, S=S, Y=torch.tensor([1], device=dev), c=torch.tensor([0.], device=dev))
        loss.backward()
        return loss
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step_static()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): step_static()
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 50
    g = torch.cuda.CUDAGraph()
    Yt = torch.tensor([1], device=dev); ct = torch.tensor([0.], device=dev)
    def body():
        from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
        for p in model.parameters():
            p.grad.zero_()
        hz, S, Yh, _ = model(path_features=x)
        loss = NLLSurvLoss(alpha=0.0)(hazards=hz, S=S, Y=Yt, c=ct)
        loss.backward()
        return loss
    with torch.cuda.graph(g):
        l = body()
    g.replay(); torch.cuda.synchronize()
    ref = {k: p.grad.clone() for k, p in model.named_parameters()}
    body(); torch.cuda.synchronize()
    ok = all(torch.equal(ref[k], p.grad) for k, p in model.named_parameters())
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); graphed = (time.perf_counter() - t0) / 50
    print(f"N={N}: eager {eager*1e3:.3f} ms/step  graph {graphed*1e3:.3f} ms/step  grads identical={ok}")

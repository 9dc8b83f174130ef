"""One-bag step time, per-kernel times and the two-in-flight rate for a list of bag sizes, in ONE process
(boxes differ by a few percent, so sizes are compared inside one call).
usage: nsweep.py [N ...]   (default 1000 2000 4096 6000 8192 10000 12288 14000 16384 24000 40000 60000)
env:   NSWEEP_INFLIGHT=0 skips the in-flight leg, NSWEEP_INFLIGHT_N=2,3,4 the numbers of bags in flight (default 2),
       NSWEEP_STEPS (default 200)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

sizes = [int(a) for a in sys.argv[1:]] or [1000, 2000, 4096, 6000, 8192, 10000, 12288, 14000, 16384, 24000, 40000, 60000]
steps = int(os.environ.get("NSWEEP_STEPS", "200"))
dev = torch.device("cuda", 0)
model = bench.build_model(dev, False)
g = torch.Generator(device=dev); g.manual_seed(7)
rows = []
for n in sizes:
    x = torch.randn(n, 1024, device=dev, generator=g)
    k = max(20, min(steps, int(steps * 10000 / max(n, 2000))))
    one = bench.make_step(model, x, dev, None, 1)
    d1 = bench.time_steps(one, k, 10, 1, 5)
    prof = bench.kernel_profile(one, 10)
    ms = 1e3 * d1 / k
    row = {"N": n, "ms": round(ms, 4), "frac": round(bench.size_fractions(n, ms)["frac_fp32_mfma_peak"], 3),
           "kernels_us": {a.replace("_kernel", ""): round(v["avg_us"], 1) for a, v in sorted(prof.items())}}
    if os.environ.get("NSWEEP_INFLIGHT", "1") != "0":
        for nf in [int(v) for v in os.environ.get("NSWEEP_INFLIGHT_N", "2").split(",")]:
            d2 = bench.time_steps(bench.make_step_inflight(model, x, dev, 1, nf), k, 10, 1, 5)
            row[f"ms_inflight{nf}"] = round(1e3 * d2 / k, 4)
            row[f"frac_inflight{nf}"] = round(bench.size_fractions(n, 1e3 * d2 / k)["frac_fp32_mfma_peak"], 3)
    for p in model.parameters():
        p.grad = None
    rows.append(row)
    print(json.dumps(row), flush=True)
    del x

#!/usr/bin/env python3
"""bags/sec, forward + nll_surv + backward (all parameter grads materialised), path attention-MIL
`small` (1024 -> 256 -> 256, gated, K=4), one 50k x 1024 synthetic bag per GPU per step.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = forward + NLLSurvLoss + backward of one bag, every parameter gradient written to a flat buffer, bag resident
in HBM -- issued as the training-loop mirror issues it (multimodalfusion_amd/utils/core_utils.py): ONE C-ABI call,
`model.nll_step` -> mmf_amil_nll_step (the head and the loss run as the tail of the pooling merge kernel).  The same
step through the autograd surface (model(**bag) -> NLLSurvLoss -> .backward(), the calls utils/core_utils.py:200-243
of the reference makes) is timed beside it (`autograd_surface`; `--autograd` makes it the headline).  N > 1 adds ONE
RCCL all-reduce (SUM) of the flat gradient buffer per step (one bag per GPU == the reference's --gc N).
Train mode as `model.train()` with --drop_out off (one Dropout(0.25) mask, the headline mode of
BASELINE.md); rank 0 prints ONE JSON line.

Steps are issued round-robin on `--inflight` HIP streams (default 2 for fp32 bags, 3 for bf16 bags), each with its own
resident bag and its own flat gradient buffer
(multimodalfusion_amd/pipeline.py): bags are independent until the optimizer step (batch_size = 1 + gradient
accumulation in the reference), and one bag's kernels leave CUs idle (224 of 256 in the row-parallel GEMMs, every
kernel's tail, the latency-bound small kernels).  Every step is still one full forward + loss + backward of one bag
with all gradients materialised (+ one all-reduce when N > 1); `one_bag_in_flight` in the JSON is the strictly
sequential figure of the same run, and the roofline leg times kernels one bag at a time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md chip table (v_mfma_f32_32x32x2_f32, dense)
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 (v_mfma_f32_32x32x16_bf16); the bf16-storage path is HBM-bound, not MFMA-bound
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # a run pays ~1 ms once (clocks ramp up after the sync): keep it < 2 %
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--blocks", type=int, default=12,
                    help="the --steps block is timed this many times (each bracketed by barrier + synchronize); `value` is the "
                         "median block's rate, min / max are reported beside it (>= 10 keeps the timed region above 150 ms at --steps 20)")
    ap.add_argument("--bag", type=int, default=50000, help="instances per bag (BASELINE metric: 50000)")
    ap.add_argument("--eval-mode", action="store_true", help="no dropout (secondary figure)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-bags", type=int, default=8, help="timed bags of the CPU baseline sample")
    ap.add_argument("--no-extras", dest="extras", action="store_false", default=True,
                    help="skip the extra legs of the default N = 1 run (other bag sizes, hipGraph steps, BASELINE configs "
                         "3-5, PCIe-inclusive rate); they never change `value`")
    ap.add_argument("--extra-sizes", action="store_true", help="(kept for older command lines: the legs are on by default)")
    ap.add_argument("--graph", action="store_true", help="(kept for older command lines: the legs are on by default)")
    ap.add_argument("--h2d", action="store_true", default=True,
                    help="also report the PCIe-inclusive rate (extra key `pcie_inclusive`, never `value`); on by default at N = 1")
    ap.add_argument("--no-h2d", dest="h2d", action="store_false")
    ap.add_argument("--inflight", type=int, default=0,
                    help="bags in flight per GPU: steps are issued round-robin on this many HIP streams, each with its "
                         "own gradient buffer (pipeline.BagsInFlight); 1 = strictly one bag at a time; "
                         "0 = default: 2 (fp32: one 8-wave workgroup per CU, only kernel tails overlap; 1,372-1,379 bags/s with two, "
                         "1,351-1,357 with three), 3 (bf16: two 4-wave workgroups per CU leave room for a third bag)")
    ap.add_argument("--autograd", action="store_true",
                    help="time the step through the autograd surface (model -> loss -> backward) instead of the one-call step")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="storage type of the bag and saved activations; f32 is the BASELINE metric, bf16 is config 5 "
                         "(bf16 MFMA, fp32 accumulate/epilogues; HBM roofline)")
    ap.add_argument("--gemm", choices=["f32", "bf16x3"], default="f32",
                    help="f32 bags only (mmf_amil_desc::gemm): f32 = exact-fp32 MFMA (the BASELINE metric's arithmetic, default); "
                         "bf16x3 = every fp32 operand as the sum of three bf16 values on the bf16 matrix cores, fp32 accumulation "
                         "(fp32-equivalent error: tests/test_gpu_split.py).  The default run reports it as the extra leg `gemm_bf16x3`")
    return ap.parse_args()


def flops_per_bag(N, L=1024, H=256, D=256):
    """Algorithmic FLOPs: fwd 2LH + 2*2HD + ..., bwd dW1 + dWab + dh."""
    fwd = 2 * L * H + 2 * 2 * H * D + 2 * D + 2 * H
    bwd = 2 * H * L + 2 * 2 * D * H + 2 * 2 * D * H + 2048
    return (fwd + bwd) * N


def bytes_per_bag(N, bf16=False, L=1024):
    """Algorithmic bytes: x read once forward and once for dW1, A_raw, weights + grads."""
    return 2 * (2 if bf16 else 4) * L * N + 4 * N + 4.7e6


def build_model(dev, eval_mode):
    import torch
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    torch.manual_seed(1)
    model = MIL_Attention_fc_surv_path(gate_path=True, model_size_wsi="small", dropout=False, n_classes=4)
    model = model.to(dev)
    model.eval() if eval_mode else model.train()
    return model


def flat_size(model):
    from multimodalfusion_amd.dp import flat_layout
    return flat_layout(list(model.parameters()))[1]


def flat_views(model, flat):
    """Per-parameter views of a flat gradient buffer (multimodalfusion_amd.dp.flat_layout: 16-byte aligned tensors)."""
    from multimodalfusion_amd.dp import flat_layout
    params = list(model.parameters())
    return [flat[off:off + p.numel()].view_as(p) for p, off in zip(params, flat_layout(params)[0])]


def make_step(model, x, dev, flat=None, world=1, autograd=False):
    import torch
    import torch.distributed as dist
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    loss_fn = NLLSurvLoss(alpha=0.0)
    Y = torch.tensor([1], device=dev)
    c = torch.tensor([0.0], device=dev)
    inv = 1.0 / world
    params = list(model.parameters())
    if not autograd:
        # the one-call step: gradients of loss / world are WRITTEN (not accumulated) into the flat buffer
        if flat is None:
            flat = torch.empty(flat_size(model), device=dev)
        views = flat_views(model, flat)

        def fused():
            out = model.nll_step(x, Y, c, alpha=0.0, loss_scale=inv, grad_out=views, accumulate=False)
            if world > 1:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)   # RCCL over xGMI: one collective per step
            return out[4]

        return fused

    def step():
        if flat is not None:
            flat.zero_()
        else:
            for p in params:
                p.grad = None
        hazards, S, Y_hat, _ = model(path_features=x)
        loss = loss_fn(hazards=hazards, S=S, Y=Y, c=c)
        (loss * inv if world > 1 else loss).backward()
        if world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)   # RCCL over xGMI: one collective per step
        return loss

    return step


def make_step_inflight(model, x, dev, world, n_streams, autograd=False):
    """The same step (forward + nll_surv + backward, all parameter gradients materialised into a flat buffer, one
    all-reduce per bag when world > 1) with `n_streams` bags in flight: step i runs on stream i % n_streams, owns
    gradient slot i % n_streams and reads ITS OWN resident bag -- `x` is the first one, `distinct_bags` makes the others
    (same shape, different values): concurrent bags must not share L2 / Infinity-Cache lines of one tensor."""
    import torch
    from multimodalfusion_amd.pipeline import BagsInFlight
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    loss_fn = NLLSurvLoss(alpha=0.0)
    Y = torch.tensor([1], device=dev)
    c = torch.tensor([0.0], device=dev)
    inv = 1.0 / world
    pipe = BagsInFlight(model, n_streams, dev)
    xs = distinct_bags(x, n_streams)
    turn = [0]

    def step():
        xi = xs[turn[0] % n_streams]
        turn[0] += 1
        if autograd:
            def bag():
                hazards, S, Y_hat, _ = model(path_features=xi)
                loss = loss_fn(hazards=hazards, S=S, Y=Y, c=c)
                return loss * inv if world > 1 else loss
            loss = pipe.run(bag, accumulate=False)
        else:
            loss = pipe.run_fused(model, xi, Y, c, 0.0, loss_scale=inv, accumulate=False)[4]
        if world > 1:
            pipe.all_reduce_slot()       # RCCL over xGMI: one collective per bag, on the bag's stream
        return loss

    return step


def distinct_bags(x, n):
    """`n` resident bags of x's shape and dtype: x itself and n - 1 more N(0,1) bags of other seeds (x is never copied)."""
    import torch
    out = [x]
    for k in range(1, n):
        g = torch.Generator(device=x.device)
        g.manual_seed(977 + 31 * k)
        out.append(torch.randn(x.shape, device=x.device, generator=g).to(x.dtype))
    return out


def time_blocks(step, steps, warmup, world, blocks):
    """`warmup` untimed steps, then `blocks` timed blocks of EXACTLY `steps` steps each; every block is bracketed by a
    barrier + torch.cuda.synchronize() on both sides and its time is the MAX over ranks.  Returns the list of block times
    (seconds).  Why blocks: one block of the default 20-100 steps is 15-75 ms of GPU time, short enough for the clock ramp
    after an idle period and for one host hiccup to move the figure by several percent; the median over >= 10 blocks does
    not move, and min / max are reported beside it."""
    import torch
    import torch.distributed as dist
    for _ in range(warmup):
        step()
    out = []
    for _ in range(blocks):          # a fixed count: with N > 1 every step holds a collective, all ranks issue the same number
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        out.append(dt)
    return out


def median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def time_steps(step, steps, warmup, world, blocks=5):
    """Median block time (seconds per `steps` steps) of `blocks` blocks: the figure the secondary legs quote."""
    return median(time_blocks(step, steps, warmup, world, blocks))


def step_percentiles(step, steps):
    """Per-step device time from HIP events on the compute stream (median and p10 / p90)."""
    import torch
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    q = lambda f: ms[min(len(ms) - 1, int(f * len(ms)))]
    return {"p10": q(0.10), "p50": q(0.50), "p90": q(0.90)}


def kernel_profile(step, steps):
    """Per-kernel average duration from HIP events recorded on the launch stream, inside this process, over `steps`
    timed steps (include/mmf_amil.h "Kernel trace": the attention-stack entry points record an event pair around
    every kernel they launch while a trace is attached to their descriptor)."""
    import torch
    from multimodalfusion_amd import _lib
    torch.cuda.synchronize()
    with _lib.KernelTrace(capacity=64 * steps) as tr:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        prof = tr.dump()
    tr.close()
    return {k: dict(launches=n, avg_us=1e3 * ms / max(n, 1)) for k, (n, ms) in prof.items()}


def kernel_tables(N, L=1024, H=256, D=256):
    """Algorithmic FLOPs (fp32 kernels: MFMA-bound) and minimal HBM bytes (bf16-storage kernels: HBM-bound) per
    launch of each main kernel for one N x L bag of the `small` gated stack."""
    kflops = {
        "linear_nt_kernel": 2 * L * H * N,
        "gate_fwd_kernel": 2 * 2 * H * D * N,
        "bwd_dh_kernel": 2 * 2 * H * D * N,
        "tn_kernel": (2 * H * L + 2 * 2 * D * H) * N,
    }
    for k in ("linear_nt", "gate_fwd", "bwd_dh", "tn"):       # the same contractions on the split-operand core
        kflops[k + "_split_kernel"] = kflops[k + "_kernel"]
    kbytes = {
        "amil_fwd_fused_bf16_kernel": N * (L * 2 + H * 2 + 2 * D * 2 + 4),     # x read; h, a, b, A_raw written
        "linear_bf16_kernel": N * (L * 2 + H * 2),                             # x read, h written
        "gate_bf16_kernel": N * (H * 2 + 2 * D * 2 + 2 * 4),                   # h read; a, b, 2 score parts written
        "pool_partial_bf16_kernel": N * (H * 2 + 3 * 4),                       # h read; score parts read, A_raw written
        "dh_bf16_kernel": N * (2 * D * 2 + H * 2 + H * 2 + 2 * D * 2),         # a, b, h read; du, dP written
        "tn_bf16_kernel": N * (H * 2 + L * 2 + 2 * D * 2 + H * 2),             # du, x, dP, h read
    }
    return kflops, kbytes


def roofline_of(prof, N, bf16):
    """`roofline` block for the dominant kernel of a per-kernel profile (see kernel_profile)."""
    if not prof:
        return None
    dom = max(prof.items(), key=lambda kv: kv[1]["avg_us"] * kv[1]["launches"])[0]
    kflops, kbytes = kernel_tables(N)
    t_us = prof[dom]["avg_us"]
    traffic, source = recorded_traffic(dom, N, bf16)
    if bf16 and dom in kbytes:
        ach = kbytes[dom] / (t_us * 1e-6) / 1e9
        return {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source, "avg_launch_us": t_us,
                "bytes_per_launch": kbytes[dom]}
    if dom.endswith("_split_kernel"):
        # six bf16 MFMAs carry one fp32 product's worth of k: the roofline is the dense bf16 peak / 6, in fp32 FLOP
        ach = kflops[dom] / (t_us * 1e-6) / 1e12
        peak = BF16_MFMA_PEAK_TFLOPS / 6
        return {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "peak_note": "dense bf16 MFMA peak / 6 products per fp32 product (bf16x3 split operands)",
                "x_fp32_mfma_peak": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": source,
                "avg_launch_us": t_us, "flops_per_launch": kflops[dom]}
    if dom in kflops:
        ach = kflops[dom] / (t_us * 1e-6) / 1e12
        return {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": source,
                "avg_launch_us": t_us, "flops_per_launch": kflops[dom]}
    return None


def recorded_traffic(kernel, N, bf16):
    """HBM bytes per launch of `kernel` from the COMMITTED rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command): a recorded figure,
    not measured by this run -- PMC counters need rocprofv3 around the process -- and only for the workload it was
    recorded on; null otherwise."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        for rec in tj.get("workloads", [tj]):
            if rec.get("instances") == N and rec.get("dtype", "f32") == ("bf16" if bf16 else "f32") and kernel in rec["kernels"]:
                return rec["kernels"][kernel]["bytes"], "recorded: " + rec.get("source", "profiles/traffic.json")
    except Exception:
        pass
    return None, None


def size_fractions(N, ms, bf16=False):
    """Whole-step fractions of the two rooflines for one N x 1024 bag processed in `ms`."""
    tot = flops_per_bag(N)
    return {"tflops": tot / (ms * 1e-3) / 1e12,
            "frac_fp32_mfma_peak": tot / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "algorithmic_gbs": bytes_per_bag(N, bf16) / (ms * 1e-3) / 1e9,
            "frac_hbm_peak": bytes_per_bag(N, bf16) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def graph_leg(model, dev, steps, gen):
    """Small bags are host-bound in eager mode; the same train-mode step captured into ONE hipGraph
    (graph.GraphedStep: device-resident dropout seed, fresh masks per replay) shows the GPU-side rate."""
    import torch
    from multimodalfusion_amd.graph import GraphedStep
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    loss_fn = NLLSurvLoss(alpha=0.0)
    Y, c = torch.tensor([1], device=dev), torch.tensor([0.0], device=dev)
    res = {}
    for n in (1000, 10000):
        x = torch.randn(n, 1024, device=dev, generator=gen)
        fn = make_step(model, x, dev, None, 1)       # the one-call step: 7 kernel nodes, static gradient buffer

        gs = GraphedStep(fn)
        try:
            for _ in range(5):
                gs()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                gs()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            gs.close()
        res[str(n)] = {"bags_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps, **size_fractions(n, 1e3 * dt / steps)}
    for p in model.parameters():
        p.grad = None
    return res


def gemm_bf16x3_leg(model, x, dev, steps, warmup, inflight):
    """The default workload with mmf_amil_desc::gemm = MMF_GEMM_BF16X3: same bag, same model, same step; reported beside
    `value`, never as `value` (the BASELINE metric is quoted on exact-fp32 arithmetic)."""
    import torch
    from multimodalfusion_amd import ops
    N = x.shape[0]
    prev = ops.set_gemm(1)
    try:
        steps = max(steps, 100)
        one = make_step(model, x, dev, None, 1)
        d1 = time_steps(one, steps, warmup, 1)
        prof = kernel_profile(one, max(5, min(steps, 20)))
        dn = time_steps(make_step_inflight(model, x, dev, 1, inflight), steps, warmup, 1)
        sizes = {}
        if N == 50000:                  # north_star's other bag sizes in this mode (64-row split tiles below 16,384 instances)
            g = torch.Generator(device=dev)
            g.manual_seed(4321)
            for n2 in (1000, 10000):
                x2 = torch.randn(n2, 1024, device=dev, generator=g)
                d2 = time_steps(make_step(model, x2, dev, None, 1), max(steps, 200), warmup, 1)
                sizes[str(n2)] = {"ms_per_step": 1e3 * d2 / max(steps, 200), "bags_per_s": max(steps, 200) / d2}
    finally:
        ops.set_gemm(prev)
        for p in model.parameters():
            p.grad = None
    ms1 = 1e3 * d1 / steps
    return {"arithmetic": "fp32 operands split into three bf16 values each, six v_mfma_f32_32x32x16_bf16 products per fp32 "
                          "product, fp32 accumulation; inputs, outputs, saved activations and gradients fp32",
            "accuracy": "error against the fp64 oracle within 2x the exact-fp32 path's, output by output: tests/test_gpu_split.py",
            "value_one_bag_per_step": steps / d1, "ms_per_step_one_bag": ms1,
            "value": steps / dn, "bags_in_flight": inflight,
            "whole_step_x_fp32_mfma_peak": flops_per_bag(N) / (ms1 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            "roofline": roofline_of(prof, N, False), "other_sizes": sizes,
            "kernels_us": {k: round(v["avg_us"], 2) for k, v in sorted(prof.items())}}


def config5_leg(dev, steps, warmup, gen):
    """BASELINE config 5 on one GPU: bf16 storage, one 100,000 x 1024 bag, path head, train mode -- the HBM-roofline
    run.  Strictly one bag at a time, plus the rates with two and three bags in flight; `roofline` is for its dominant kernel."""
    import torch
    N = 100_000
    model = build_model(dev, False)
    x = torch.randn(N, 1024, device=dev, generator=gen).to(torch.bfloat16)
    step1 = make_step(model, x, dev, None, 1)
    d1 = time_steps(step1, steps, warmup, 1)
    ms = 1e3 * d1 / steps
    prof = kernel_profile(step1, max(5, min(steps, 20)))
    for p in model.parameters():
        p.grad = None
    d2 = time_steps(make_step_inflight(model, x, dev, 1, 2), steps, warmup, 1)
    for p in model.parameters():
        p.grad = None
    d3 = time_steps(make_step_inflight(model, x, dev, 1, 3), steps, warmup, 1)
    out = {"workload": "path_attention_mil small gated K=4, one 100000x1024 bf16 bag, train mode, fwd+nll_surv+bwd",
           "dtype": "bf16", "bags_per_s": steps / d1, "ms_per_step": ms, "two_bags_in_flight_bags_per_s": steps / d2,
           "three_bags_in_flight_bags_per_s": steps / d3,
           "roofline": roofline_of(prof, N, True), "kernels_us": {k: round(v["avg_us"], 2) for k, v in sorted(prof.items())},
           "whole_step": {"algorithmic_gbs": bytes_per_bag(N, True) / (ms * 1e-3) / 1e9,
                          "frac_hbm_peak": bytes_per_bag(N, True) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "tflops": flops_per_bag(N) / (ms * 1e-3) / 1e12}}
    del x, model
    return out


def other_configs_leg(dev, steps, warmup, gen):
    """BASELINE configs 3 and 4 (and 5's multimodal form) on one GPU, ms per fwd + loss + bwd step:
    radio 4 x 512 x 1024 + omic MaxNet (B = 128, Cox); mm_attention_mil concat / tensor with a 50k fp32 path bag and
    with a 100k bf16 path bag (+ 4 x 512 radio + omic[80])."""
    import torch
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_radio, MaxNet, MM_MIL_Attention_fc_surv
    from multimodalfusion_amd.utils.loss_utils import CoxSurvLoss, NLLSurvLoss
    MODS = ["T1", "T2", "T1Gd", "FLAIR"]
    Y, c = torch.tensor([1], device=dev), torch.tensor([0.0], device=dev)
    nll, cox = NLLSurvLoss(alpha=0.0), CoxSurvLoss()
    rn = lambda *shape: torch.randn(*shape, device=dev, generator=gen)

    def timeit(fn, n=None):
        """Median of three timed runs of n steps, the cyclic collector off while they run: these legs are 6-30 ms long, and one
        generation-2 collection of a process that has built a dozen models (tens of ms) inside a single run showed up as
        0.8 and 1.7 ms 'steps' of a 0.55 / 0.75 ms leg (two of eleven default runs at the end of round 4)."""
        import gc
        n = n or steps
        for _ in range(max(3, warmup)):
            fn()
        runs = []
        gc.collect()
        gc.disable()
        try:
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                runs.append(1e3 * (time.perf_counter() - t0) / n)
        finally:
            gc.enable()
        ms = sorted(runs)[1]
        return {"ms_per_step": ms, "bags_per_s": 1e3 / ms}

    def stepper(model, kw, loss_of, static_grads=False):
        params = list(model.parameters())
        if static_grads:                      # hipGraph capture needs the gradients at fixed addresses
            for p in params:
                p.grad = torch.zeros_like(p)

        def fn():
            for p in params:
                if static_grads:
                    p.grad.zero_()
                else:
                    p.grad = None
            loss_of(model(**kw)).backward()
        return fn

    def both(model, kw, loss_of, loss_of_graph=None):
        """Eager (Python + autograd + ~40-60 launches issued from the host per step: these steps are HOST-bound) and the same
        step captured into one hipGraph (graph.GraphedStep: device-resident dropout seed, fresh masks per replay), which
        takes the host off the critical path -- the figure that says what the kernels themselves cost."""
        from multimodalfusion_amd.graph import GraphedStep
        res = timeit(stepper(model, kw, loss_of))
        try:
            gs = GraphedStep(stepper(model, kw, loss_of_graph or loss_of, static_grads=True))
            g = timeit(gs)
            res["graphed_ms_per_step"] = g["ms_per_step"]
            res["graphed_bags_per_s"] = g["bags_per_s"]
            gs.close()
        except Exception as e:                # capture is an optimisation: report why it was not available
            res["graphed_error"] = f"{type(e).__name__}: {e}"[:200]
            torch.cuda.synchronize()
        for p in model.parameters():
            p.grad = None
        return res

    out = {}
    torch.manual_seed(1)
    rx = {m: rn(512, 1024) for m in MODS}
    radio = MIL_Attention_fc_surv_radio(n_classes=4).to(dev).train()
    res = both(radio, rx, lambda r: nll(hazards=r[0], S=r[1], Y=Y, c=c))
    # what the training-loop mirror runs for this model (MIL_Attention_fc_surv_radio.nll_step): reduce_dim, the stack + head +
    # loss + backward as one call, reduce_dim's backward -- no autograd graph; ms_per_step above is the autograd path
    rparams = list(radio.parameters())

    def radio_one_call():
        for p in rparams:
            p.grad = None
        radio.nll_step(Y, c, alpha=0.0, **rx)

    oc = timeit(radio_one_call, max(steps, 200))     # 0.2 ms steps: a 30-step run would mostly measure the clock ramp
    res["autograd_ms_per_step"] = res["ms_per_step"]
    res["one_call_step_ms_per_step"] = oc["ms_per_step"]
    res["ms_per_step"], res["bags_per_s"] = oc["ms_per_step"], oc["bags_per_s"]
    for p in rparams:
        p.grad = None
    out["config3_radio_4x512x1024"] = res
    omic = MaxNet(input_dim=36, bag_loss="cox_surv").to(dev).train()
    ot = torch.rand(128, dtype=torch.float64) * 100
    oc = (torch.rand(128, device=dev, generator=gen) < 0.5).float()
    # eager: event times arrive from the host every step (float64, pageable), as in the reference's loop; the captured step
    # reads them from a device-resident buffer (a copy from pageable memory cannot be captured)
    ot_dev = ot.to(dev)
    xo = rn(128, 36)
    composable = both(omic, {"genomic_features": xo}, lambda r: cox(risks=r[0], times=ot, c=oc),
                      lambda r: cox(risks=r[0], times=ot_dev, c=oc))
    # what the training-loop mirror runs for this model (utils/core_utils.py: MaxNet.cox_step): forward + Cox + backward in
    # ONE launch, event times device-resident; the composable path (3 dense + Cox + 3 dense-backward launches) beside it
    for p in omic.parameters():
        p.grad = torch.zeros_like(p)

    def one_launch():
        omic.cox_step(xo, ot_dev, oc, grad_out=[p.grad for p in omic.parameters()], accumulate=False)

    res = timeit(one_launch, max(steps, 300))        # 0.03 ms steps
    from multimodalfusion_amd import _lib
    torch.cuda.synchronize()
    with _lib.KernelTrace(capacity=256) as tr:
        for _ in range(20):
            one_launch()
        torch.cuda.synchronize()
        prof = tr.dump()
    tr.close()
    res["kernel_us"] = {k: round(1e3 * ms / max(n, 1), 2) for k, (n, ms) in prof.items()}
    res["composable"] = composable
    out["config3_omic_maxnet_B128_cox"] = res
    for p in omic.parameters():
        p.grad = None
    del radio, omic
    for tag, n, dt in (("config4_mm_50k_f32", 50_000, torch.float32), ("config5_mm_100k_bf16", 100_000, torch.bfloat16)):
        xp = rn(n, 1024).to(dt)
        for fusion in ("concat", "tensor"):
            mm = MM_MIL_Attention_fc_surv(input_dim=80, fusion=fusion, n_classes=4).to(dev).train()
            kw = dict(rx)
            kw["path_features"] = xp
            kw["genomic_features"] = rn(80)
            res = both(mm, kw, lambda r: nll(hazards=r[0], S=r[1], Y=Y, c=c))
            if True:
                # what the training-loop mirror runs for this model (utils/core_utils.py: MM_MIL_Attention_fc_surv.nll_step):
                # the same step as a fixed sequence of C-ABI calls, no autograd graph, one launch for head + loss + their
                # backward; ms_per_step above is model(**kw) + loss + backward() through autograd
                params = list(mm.parameters())

                def one_call():
                    for p in params:
                        p.grad = None
                    mm.nll_step(Y, c, alpha=0.0, **kw)

                oc = timeit(one_call)
                res["autograd_ms_per_step"] = res["ms_per_step"]
                res["one_call_step_ms_per_step"] = oc["ms_per_step"]
                res["ms_per_step"], res["bags_per_s"] = oc["ms_per_step"], oc["bags_per_s"]
                for p in params:
                    p.grad = None
            out[f"{tag}_{fusion}"] = res
            del mm
        del xp
    return out


def h2d_leg(model, N, dev, steps, warmup, bf16=False):
    """PCIe-inclusive rate (never `value`): every step takes a NEW bag from pinned host memory; (a) copied
    synchronously at the top of the step as the reference does (utils/core_utils.py:194-198), (b) staged by
    feed.DevicePrefetcher (side stream, 2 bags in flight) so the copy overlaps the previous bag's kernels."""
    import torch
    from multimodalfusion_amd.feed import DevicePrefetcher
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    # bf16 storage: the bags sit in host memory as bf16 (as a bf16 .pt file loads), half the PCIe bytes
    host = [(torch.randn(N, 1024).to(torch.bfloat16) if bf16 else torch.randn(N, 1024)).pin_memory() for _ in range(3)]
    esz = 2 if bf16 else 4
    loss_fn = NLLSurvLoss(alpha=0.0)
    Y, c = torch.tensor([1], device=dev), torch.tensor([0.0], device=dev)

    flat = torch.empty(flat_size(model), device=dev)
    views = flat_views(model, flat)

    def run(x):
        model.nll_step(x, Y, c, alpha=0.0, grad_out=views, accumulate=False)

    def batches(n):
        for i in range(n):
            yield ({}, host[i % 3], torch.zeros(1, 1), torch.tensor([1]), None, torch.tensor([0.0]))

    res = {}
    for name in ("sync_copy", "prefetch"):
        for phase, n in (("warm", warmup), ("timed", steps)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if name == "sync_copy":
                for b in batches(n):
                    run(b[1].to(dev, non_blocking=False))
            else:
                for b in DevicePrefetcher(batches(n), dev, depth=2):
                    run(b[1])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        res[name] = {"bags_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps}
    res["bytes_per_bag"] = N * 1024 * esz
    res["h2d_gbs_prefetch"] = N * 1024 * esz * res["prefetch"]["bags_per_s"] / 1e9
    return res


def cpu_baseline(N, n_bags, threads=None):
    """The oracle's torch-CPU port (kind 'port'), same op sequence as the reference's CPU path
    (models/model_attention_mil_path.py:50-72 + utils/loss_utils.py:22-39), train mode with the one always-on
    Dropout(0.25), fp32.  The intra-op thread count is the fastest of 8/16/32/64, each candidate timed on THREE bags after
    one warm-up bag (one timed bag per candidate picked 32 threads in one round and 8 in the next on the same CPU class);
    pass `threads` to reuse a choice.  Bounded sample: `n_bags` bags after one more warm-up."""
    import torch
    from oracle import inputs as gen
    from oracle import torch_port as tp
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    sd = tp.to_torch(gen.path_state_dict(seed=1, gated=True, size="small", n_classes=4), torch.float32)
    x = torch.as_tensor(gen.bag(1234, N))
    Y, c = torch.tensor([1]), torch.tensor([0.0])

    def one():
        for v in sd.values():
            v.grad = None
        hz, S, Yh, A, M = tp.path_forward(sd, x, True, False, masks="torch")
        loss = tp.nll_loss(hz, S, Y, c, alpha=0.0)
        loss.backward()

    probe = {}
    if threads is None:
        for t in sorted({min(avail, n) for n in (8, 16, 32, 64)}):
            torch.set_num_threads(t)
            one()
            t0 = time.perf_counter()
            for _ in range(3):
                one()
            probe[t] = (time.perf_counter() - t0) / 3
        threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    one()
    t0 = time.perf_counter()
    for _ in range(n_bags):
        one()
    dt = time.perf_counter() - t0
    # single-thread figure (SURVEY 8d asks for both): one timed bag after one warm-up bag at the small sizes
    torch.set_num_threads(1)
    if N <= 10000:
        one()
    t0 = time.perf_counter()
    one()
    dt1 = time.perf_counter() - t0
    torch.set_num_threads(threads)
    return dict(value=n_bags / dt, unit="bags/s", cores=threads, kind="port", value_1_thread=1.0 / dt1,
                thread_probe_s_per_bag={str(k): round(v, 4) for k, v in probe.items()},
                sample=f"{n_bags} bags of {N}x1024 fwd+nll_surv+bwd after warm-up, fp32, train mode "
                       f"(1 dropout mask), torch {torch.__version__} CPU, {threads} intra-op threads "
                       f"(fastest of 8/16/32/64, three timed bags per candidate, on a host showing {avail} cores); "
                       f"value_1_thread: one bag on one thread")


def self_launch(args):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves,
    as children (`python -m torch.distributed.run ... bench.py <same arguments>`), before this process has made any GPU
    call; relay rank 0's JSON line; a failed child is a non-zero exit.  Never exec: a parent that has touched the GPU must
    not be replaced (and this one has not touched it anyway)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    for l in r.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if r.returncode != 0 or len(lines) != 1:
        print(f"bench.py: the {args.gpus}-rank child run failed (exit {r.returncode}, {len(lines)} JSON lines)", file=sys.stderr)
        sys.exit(r.returncode or 1)
    if json.loads(lines[0]).get("n_gpus") != args.gpus:
        print(f"bench.py: child run reported n_gpus != {args.gpus}", file=sys.stderr)
        sys.exit(1)
    print(lines[0], flush=True)
    sys.exit(0)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)            # does not return
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # the line's n_gpus is the number of ranks that ran: it must be what --gpus asked for
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        # RCCL exchanges buffers between the ranks' processes through IPC handles; this pool's host driver supports the
        # dmabuf form only (with the legacy form hipIpcGetMemHandle fails with "invalid argument").  The image exports the
        # variable already; setdefault keeps a launcher's explicit choice.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("MMF_BENCH_COUNT_COLLECTIVES"):
            # test hook (tests/test_gpu_bench_dp.py): every collective this rank issues is counted and written out at exit
            _n = {"all_reduce": 0, "barrier": 0}
            _ar, _ba = dist.all_reduce, dist.barrier

            def _count_ar(*a, **k):
                _n["all_reduce"] += 1
                return _ar(*a, **k)

            def _count_ba(*a, **k):
                _n["barrier"] += 1
                return _ba(*a, **k)
            dist.all_reduce, dist.barrier = _count_ar, _count_ba
            import atexit
            atexit.register(lambda: open(os.environ["MMF_BENCH_COUNT_COLLECTIVES"] + f".rank{rank}", "w").write(json.dumps(_n)))
        if os.environ.get("MMF_BENCH_REHEARSAL") == "1":
            # rehearsal of the multi-process path on a 1-GPU box: every rank on device 0, gloo instead of RCCL
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local)
    N = args.bag

    model = build_model(dev, args.eval_mode)
    flat = None
    if world > 1:
        # flat fp32 gradient buffer; every p.grad is a view into it, so backward accumulates in place and the
        # step needs exactly one all-reduce
        flat = torch.zeros(flat_size(model), device=dev)
        for p, v in zip(model.parameters(), flat_views(model, flat)):
            p.grad = v
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x = torch.randn(N, 1024, device=dev, generator=g)      # synthetic bag, resident in HBM
    bf16 = args.dtype == "bf16"
    if bf16:
        x = x.to(torch.bfloat16)
    # bags in flight by default -- 50k fp32, same box, alternating: 2 -> 1,372-1,379 bags/s; 3 -> 1,351-1,357; 4 -> 1,315 (one 8-wave
    # workgroup per CU: only tails overlap); 100k bf16 (two 4-wave workgroups per CU): 2 -> 3,420-3,450; 3 -> 3,530-3,540
    inflight = args.inflight if args.inflight > 0 else (3 if bf16 else 2)
    if args.gemm == "bf16x3" and not bf16:
        from multimodalfusion_amd import ops
        ops.set_gemm(1)
    ag = args.autograd
    step = (make_step(model, x, dev, flat, world, ag) if inflight == 1
            else make_step_inflight(model, x, dev, world, inflight, ag))

    blocks = time_blocks(step, args.steps, args.warmup, world, max(1, args.blocks))
    dt = median(blocks)
    ms_per_step = 1e3 * dt / args.steps
    value = world * args.steps / dt

    out = {
        "metric": "bags/sec fwd+bwd, path-AMIL 50k x 1024 synthetic bag" if N == 50000 else f"bags/sec fwd+bwd, path-AMIL {N} x 1024 synthetic bag",
        "value": value, "unit": "bags/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "blocks": {"count": len(blocks), "steps_per_block": args.steps, "timed_region_ms": 1e3 * sum(blocks),
                   "value_median": value, "value_min": world * args.steps / max(blocks), "value_max": world * args.steps / min(blocks)},
        "dtype": args.dtype if args.gemm == "f32" or bf16 else "f32 (bf16x3 split operands on the bf16 MFMA, f32 accumulate)", "data": "synthetic",
        "config": {"workload": f"path_attention_mil small gated K=4, one {N}x1024 N(0,1) {'bf16 ' if bf16 else ''}bag per GPU per step, "
                               f"nll_surv alpha=0, {'eval' if args.eval_mode else 'train (1 dropout mask)'} mode, "
                               f"fwd+loss+bwd, grads materialised, "
                               + ("autograd surface (model -> loss -> backward)" if ag else "one C-ABI call per bag (model.nll_step)")
                               + (", 1 RCCL all-reduce/step" if world > 1 else "")
                               + (f", {inflight} bags in flight per GPU on {inflight} HIP streams" if inflight > 1 else ""),
                   "instances_per_bag": N, "parallelism": f"dp{world} (one bag per GPU per step)", "bags_in_flight": inflight},
    }

    if rank == 0:
        # ---- roofline of the dominant kernel, timed live with HIP events on the launch stream ----
        # rank 0 only: this leg must NOT contain the collective (the other ranks are already at the final barrier)
        local_step = make_step(model, x, dev, flat, 1, ag)
        if inflight > 1 and world == 1:      # the strictly sequential figure beside it (same run)
            b1 = time_blocks(local_step, args.steps, args.warmup, 1, max(1, args.blocks))
            d1 = median(b1)
            out["one_bag_in_flight"] = {"value": args.steps / d1, "ms_per_step": 1e3 * d1 / args.steps,
                                        "value_min": args.steps / max(b1), "value_max": args.steps / min(b1), "blocks": len(b1)}
            # SURVEY 8(d) defines the metric as ONE bag per GPU per step: that figure at top level, beside `value`
            out["value_one_bag_per_step"] = args.steps / d1
            out["ms_per_step_one_bag"] = 1e3 * d1 / args.steps
        out["step_ms_device"] = step_percentiles(local_step, max(10, min(args.steps, 50)))
        prof = kernel_profile(local_step, max(5, min(args.steps, 20)))
        rl = roofline_of(prof, N, bf16)
        if rl is not None:
            out["roofline"] = rl
        tot = flops_per_bag(N)
        mfma_peak = BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS
        out["whole_step"] = {"tflops": tot / (ms_per_step * 1e-3) / 1e12,
                             ("frac_bf16_mfma_peak" if bf16 else "frac_fp32_mfma_peak"):
                                 tot / (ms_per_step * 1e-3) / 1e12 / mfma_peak,
                             "algorithmic_gbs": bytes_per_bag(N, bf16) / (ms_per_step * 1e-3) / 1e9,
                             "frac_hbm_peak": bytes_per_bag(N, bf16) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
        out["kernels_us"] = {k: round(v["avg_us"], 2) for k, v in sorted(prof.items())}
        for p in model.parameters():
            p.grad = None
        if world == 1 and not ag:          # the same bag through the autograd surface, one at a time
            da = time_steps(make_step(model, x, dev, None, 1, True), args.steps, args.warmup, 1)
            out["autograd_surface"] = {"one_bag_value": args.steps / da, "ms_per_step": 1e3 * da / args.steps}
            for p in model.parameters():
                p.grad = None
        if args.extras and world == 1:
            # north_star's other bag sizes: eager (host-bound: Python + autograd + ~12 launches) and as one hipGraph
            extra = {}
            k2 = max(args.steps, 200)           # 0.1-0.25 ms steps: a short run would mostly measure the clock ramp
            for n2 in (1000, 10000):
                x2 = torch.randn(n2, 1024, device=dev, generator=g)
                d2 = time_steps(make_step(model, x2, dev, None, 1), k2, args.warmup, 1)
                ms2 = 1e3 * d2 / k2
                d3 = time_steps(make_step(model, x2, dev, None, 1, True), args.steps, args.warmup, 1)
                for p in model.parameters():
                    p.grad = None
                prof2 = kernel_profile(make_step(model, x2, dev, None, 1), 10)
                rl2 = roofline_of(prof2, n2, False)
                extra[str(n2)] = {"bags_per_s": k2 / d2, "ms_per_step": ms2, **size_fractions(n2, ms2),
                                  "autograd_surface_ms_per_step": 1e3 * d3 / args.steps, "roofline": rl2,
                                  "kernels_us": {k: round(v["avg_us"], 2) for k, v in sorted(prof2.items())}}
            out["other_sizes"] = extra
            # bag-size sweep, one bag per step: is there a cliff between the small-tile and the wide-tile plans?
            sweep = {}
            for n2 in (4000, 10000, 16000, 24000, 40000, 60000):
                x2 = torch.randn(n2, 1024, device=dev, generator=g)
                k3 = max(40, min(k2, int(k2 * 10000 / n2)))
                d2 = time_steps(make_step(model, x2, dev, None, 1), k3, args.warmup, 1)
                sweep[str(n2)] = {"ms_per_step": 1e3 * d2 / k3, "frac_fp32_mfma_peak": size_fractions(n2, 1e3 * d2 / k3)["frac_fp32_mfma_peak"]}
                if n2 <= 24000:
                    # mid-size bags leave CUs idle one at a time (39 rows per CU at 10k): the aggregate rate with four bags
                    # in flight on four streams (pipeline.BagsInFlight, a distinct resident bag per stream)
                    d4 = time_steps(make_step_inflight(model, x2, dev, 1, 4), k3, args.warmup, 1)
                    sweep[str(n2)]["four_in_flight_ms_per_bag"] = 1e3 * d4 / k3
                    sweep[str(n2)]["four_in_flight_frac_fp32_mfma_peak"] = size_fractions(n2, 1e3 * d4 / k3)["frac_fp32_mfma_peak"]
                    for p in model.parameters():
                        p.grad = None
                del x2
            out["n_sweep"] = sweep
            out["graphed_small_bags"] = graph_leg(model, dev, args.steps, g)
            if not bf16 and args.gemm == "f32":
                # two bags in flight: this mode keeps the package at its power limit by itself (1,907 bags/s with two,
                # 1,875 with three, 1,844 with four in the same run)
                out["gemm_bf16x3"] = gemm_bf16x3_leg(model, x, dev, args.steps, args.warmup, min(inflight, 2))
                # the same two headline figures in that mode, at top level beside `value` (never instead of it)
                out["value_gemm_bf16x3"] = out["gemm_bf16x3"]["value"]
                out["ms_per_step_one_bag_gemm_bf16x3"] = out["gemm_bf16x3"]["ms_per_step_one_bag"]
            if not bf16 and N == 50000:
                out["config5_bf16_100k"] = config5_leg(dev, max(10, args.steps), args.warmup, g)
                out["other_configs"] = other_configs_leg(dev, max(10, min(args.steps, 30)), args.warmup, g)
        if args.h2d and args.extras and world == 1:
            out["pcie_inclusive"] = h2d_leg(model, N, dev, max(10, min(args.steps, 30)), args.warmup, bf16)
        if world == 1 and not args.no_cpu_baseline and not bf16:
            # the CPU leg runs LAST (nothing on the GPU waits behind it): the metric's bag first, then north_star's other two
            # sizes (BASELINE config 1 is the 1000 x 1024 CPU run) beside the GPU figures of `other_sizes`
            out["cpu_baseline"] = cpu_baseline(N, args.cpu_bags)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            if args.extras and N == 50000:
                small = {}
                for n2, nb in ((10000, 20), (1000, 100)):
                    cb = cpu_baseline(n2, nb)
                    gpu = out.get("other_sizes", {}).get(str(n2), {}).get("bags_per_s")
                    small[str(n2)] = {"value": cb["value"], "unit": "bags/s", "cores": cb["cores"], "kind": "port",
                                      "value_1_thread": cb["value_1_thread"], "sample": cb["sample"],
                                      "thread_probe_s_per_bag": cb["thread_probe_s_per_bag"],
                                      "gpu_bags_per_s": gpu, "gpu_over_cpu": (gpu / cb["value"]) if gpu else None}
                out["cpu_baseline_other_sizes"] = small
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""CPU, world_size 2, gloo: the one-bag-per-GPU data-parallel step (flat gradient buffer + ONE all-reduce SUM)
equals the single-process gradient-accumulation step `gc = 2` of the reference (utils/core_utils.py:242-247),
including the L1 term that the reference adds un-divided on every micro-batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multimodalfusion_amd.dp import FlatGradBuffer, broadcast_parameters, dp_micro_step
from multimodalfusion_amd.utils.core_utils import concordance_index_censored
from multimodalfusion_amd.utils.utils import l1_reg_all


def _model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(16, 8), torch.nn.Tanh(), torch.nn.Linear(8, 4))


def _bags():
    g = torch.Generator().manual_seed(5)
    return [torch.randn(n, 16, generator=g) for n in (7, 11, 5, 9)]


def _loss(model, x):
    return torch.sigmoid(model(x)).mean(0).pow(2).sum()


LAM, LR = 1e-3, 1e-2


def _single_process(steps=2, gc=2):
    model = _model()
    opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=1e-5)
    bags = _bags()
    for s in range(steps):
        for i in range(gc):
            loss = _loss(model, bags[s * gc + i])
            (loss / gc + l1_reg_all(model) * LAM).backward()
        opt.step()
        opt.zero_grad()
    return [p.detach().clone() for p in model.parameters()]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _model()
    broadcast_parameters(model)
    buf = FlatGradBuffer(model)
    opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=1e-5)
    bags = _bags()
    for s in range(2):
        loss = _loss(model, bags[s * world + rank])          # rank r takes bag step*G + r
        dp_micro_step(loss, l1_reg_all(model) * LAM, world)
        buf.all_reduce()                                      # the ONE collective of the step
        opt.step()
        buf.zero()
    if rank == 0:
        q.put([p.detach().numpy().copy() for p in model.parameters()])
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_equals_gc2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _single_process()
    for a, b in zip(got, ref):
        np.testing.assert_allclose(a, b.numpy(), rtol=1e-6, atol=1e-7)


def test_flat_buffer_views():
    model = _model()
    buf = FlatGradBuffer(model)
    _loss(model, _bags()[0]).backward()
    off = 0
    for p in model.parameters():
        assert p.grad.data_ptr() == buf.flat.data_ptr() + 4 * off     # grads accumulate in place in the flat buffer
        assert torch.equal(buf.flat[off:off + p.numel()].view_as(p), p.grad)
        off += p.numel()
    assert float(buf.flat.abs().sum()) > 0
    buf.zero()
    assert all(float(p.grad.abs().sum()) == 0 for p in model.parameters())


def test_cindex_matches_bruteforce():
    rng = np.random.default_rng(0)
    n = 60
    t = np.floor(rng.uniform(0, 20, n))            # ties in time
    e = rng.uniform(size=n) < 0.6
    r = np.round(rng.normal(size=n), 1)            # ties in risk
    num = den = 0.0
    for i in range(n):
        for j in range(n):
            if i == j or not e[i]:
                continue
            if t[i] < t[j] or (t[i] == t[j] and not e[j]):
                den += 1
                num += 1.0 if r[i] > r[j] else (0.5 if r[i] == r[j] else 0.0)
    assert abs(concordance_index_censored(e, t, r)[0] - num / den) < 1e-12
    # perfectly ordered risks -> 1.0
    assert concordance_index_censored(np.ones(5, bool), np.arange(5.0), -np.arange(5.0))[0] == 1.0

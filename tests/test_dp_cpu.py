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
    from multimodalfusion_amd.dp import flat_layout
    torch.manual_seed(3)
    # odd-sized tensors (a [1 x 7] scorer and its 1-element bias, as attention_c has) must not misalign what follows
    model = torch.nn.Sequential(torch.nn.Linear(16, 7), torch.nn.Tanh(), torch.nn.Linear(7, 1), torch.nn.Linear(1, 4))
    buf = FlatGradBuffer(model)
    params = list(model.parameters())
    offs, total = flat_layout(params)
    assert total == buf.flat.numel() and all(o % 4 == 0 for o in offs)
    torch.sigmoid(model(_bags()[0])).mean(0).pow(2).sum().backward()
    for p, off in zip(params, offs):
        assert p.grad.data_ptr() == buf.flat.data_ptr() + 4 * off     # grads accumulate in place in the flat buffer
        assert p.grad.data_ptr() % 16 == 0
        assert torch.equal(buf.flat[off:off + p.numel()].view_as(p), p.grad)
    assert float(buf.flat.abs().sum()) > 0
    used = torch.zeros_like(buf.flat, dtype=torch.bool)
    for p, off in zip(params, offs):
        used[off:off + p.numel()] = True
    assert float(buf.flat[~used].abs().sum()) == 0                    # the padding stays zero
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


# ---- the training loop itself under DP: rank-sharded loader, skipped bags, odd bag count -------------
class _StubHead(torch.nn.Module):
    """CPU stand-in with the drop-in forward contract: forward(**kwargs) -> (hazards, S, Y_hat, A_raw)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(11)
        self.proj = torch.nn.Linear(16, 8)
        self.cls = torch.nn.Linear(8, 4)

    def forward(self, **kw):
        h = torch.tanh(self.proj(kw["path_features"])).mean(0, keepdim=True)
        logits = self.cls(h)
        hazards = torch.sigmoid(logits)
        return hazards, torch.cumprod(1 - hazards, dim=1), logits.argmax(1, keepdim=True), None


def _stub_loss():
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    from oracle import torch_port as tp

    class CpuNLL(NLLSurvLoss):           # the loop dispatches on the class; the arithmetic here is the CPU oracle's
        def __call__(self, hazards, S, Y, c, alpha=None):
            return tp.nll_loss(hazards, S, Y, c, alpha=self.alpha)

    return CpuNLL(alpha=0.0)


def _loop_loader():
    """7 positions (odd), batch tuples as collate_MIL_survival builds them; position 3 -- the LAST position of a window
    of 2 -- and position 4 hold the dataset's "pathology missing" sentinel."""
    g = torch.Generator().manual_seed(21)
    out = []
    for i in range(7):
        x = torch.zeros(1, 1) if i in (3, 4) else torch.randn(5 + i, 16, generator=g)
        out.append(({"T1": torch.zeros(1, 1)}, x, torch.zeros(1, 4), torch.tensor([i % 4]), np.array([float(i)]),
                    torch.tensor([float(i % 2)])))
    return out


def _run_loop(dp, epochs=2, gc=2):
    from multimodalfusion_amd.utils import core_utils
    model = _StubHead()
    opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=1e-5)
    steps = []
    step0 = opt.step
    opt.step = lambda *a, **k: (steps.append(1), step0(*a, **k))[1]
    for ep in range(epochs):
        core_utils.train_loop_survival(ep, model, _loop_loader(), opt, 4, "path", loss_fn=_stub_loss(),
                                       reg_fn=l1_reg_all, lambda_reg=LAM, gc=gc, dp=dp)
    return [p.detach().numpy().copy() for p in model.parameters()], len(steps)


def _loop_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params, nsteps = _run_loop(dp=True, gc=1)          # gc 1 x world 2 == the reference's --gc 2
    q.put((rank, params, nsteps))
    dist.barrier()
    dist.destroy_process_group()


def test_train_loop_dp2_with_skipped_bags_equals_gc2():
    """train_loop_survival(dp=True) on 2 ranks, a loader with an odd number of positions and two missing-modality
    sentinels (one of them on a window's last position, where the reference's `continue` also skips the optimizer
    step): terminates (same number of collectives on every rank), both ranks end with identical parameters, and they
    equal the single-process run with gc = 2 over the same loader -- including the trailing partial window whose
    gradients the reference carries into the next epoch."""
    ref, ref_steps = _run_loop(dp=False, gc=2)
    # 2 epochs x 7 positions, gc 2: boundaries after positions 1, 3, 5; position 3 is skipped => 2 steps per epoch
    assert ref_steps == 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        rank, params, nsteps = q.get(timeout=180)
        got[rank] = (params, nsteps)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] == ref_steps
    for a, b, r in zip(got[0][0], got[1][0], ref):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_allclose(a, r, rtol=1e-5, atol=1e-7)


def test_rank_shard_reads_only_its_own_bags():
    """feed.RankShard over a DataLoader: rank r's dataset sees exactly the indices at loader positions r, r + world, ..."""
    from multimodalfusion_amd.feed import RankShard

    class DS(torch.utils.data.Dataset):
        def __init__(self):
            self.seen = []

        def __len__(self):
            return 9

        def __getitem__(self, i):
            self.seen.append(i)
            return torch.tensor([i])

    ds = DS()
    loader = torch.utils.data.DataLoader(ds, batch_size=1, sampler=torch.utils.data.SequentialSampler(ds))
    sh = RankShard(loader, 1, 2)
    got = [int(b[0]) for b in sh]
    assert got == [1, 3, 5, 7] and ds.seen == got and sh.n_total == 9
    assert [sh.position(i) for i in range(4)] == got
    assert list(RankShard(list(range(9)), 0, 4)) == [0, 4, 8]


def test_out_of_range_label_raises_like_the_reference_gather():
    """labels reach the kernels on the device, where a bin outside [0, n_classes) would only produce a NaN loss;
    the reference's gather (utils/loss_utils.py:30-33) raises IndexError -- the loop checks while the label is on the host."""
    from multimodalfusion_amd.utils import core_utils
    model = _StubHead()
    opt = torch.optim.Adam(model.parameters(), lr=LR)
    good = ({"T1": torch.zeros(1, 1)}, torch.randn(5, 16), torch.zeros(1, 4), torch.tensor([3]), np.array([1.0]), torch.tensor([0.0]))
    for bad_label in (4, -1):
        bad = good[:3] + (torch.tensor([bad_label]),) + good[4:]
        with pytest.raises(IndexError):
            core_utils.train_loop_survival(0, model, [good, bad], opt, 4, "path", loss_fn=_stub_loss(), gc=1)


def test_one_call_step_is_taken_only_for_the_stock_head():
    """a subclass that overrides forward(), a hooked module or a non-stock loss must take the autograd path (the
    graph-free one-call step would silently train the base computation).  Checked on the predicate itself (no GPU needed:
    the tensor only has to claim to be a CUDA tensor)."""
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    from multimodalfusion_amd.utils import core_utils
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss

    class FakeCuda(torch.Tensor):
        @property
        def is_cuda(self):
            return True

    x = torch.randn(4, 1024).as_subclass(FakeCuda)
    feats = {"path_features": x}
    stock = MIL_Attention_fc_surv_path(n_classes=4)
    assert core_utils._fused_step_ok(stock, NLLSurvLoss(alpha=0.0), feats)

    class Tweaked(MIL_Attention_fc_surv_path):
        def forward(self, **kw):
            return super().forward(**kw)

    assert not core_utils._fused_step_ok(Tweaked(n_classes=4), NLLSurvLoss(alpha=0.0), feats)
    assert not core_utils._fused_step_ok(MIL_Attention_fc_surv_path(n_classes=40), NLLSurvLoss(alpha=0.0), feats)   # > 32 classes
    assert not core_utils._fused_step_ok(stock, _stub_loss(), feats)                                                 # not the stock loss
    hooked = MIL_Attention_fc_surv_path(n_classes=4)
    hooked.classifier.register_forward_hook(lambda m, i, o: None)                                                    # a SUB-module hook
    assert not core_utils._fused_step_ok(hooked, NLLSurvLoss(alpha=0.0), feats)
    assert not core_utils._fused_step_ok(stock, NLLSurvLoss(alpha=0.0), {"path_features": x.to(torch.float64).as_subclass(FakeCuda)})

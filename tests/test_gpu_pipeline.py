"""GPU: several bags in flight on separate HIP streams (pipeline.BagsInFlight) == the same bags one after the other."""
import numpy as np
import pytest
import torch

from oracle import inputs as gen
from test_gpu_path import DEV, _load

pytestmark = pytest.mark.gpu


def _model():
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    sd = gen.path_state_dict(seed=21, gated=True, size="small", n_classes=4, dropout=False, bias_std=0.02)
    return _load(MIL_Attention_fc_surv_path(gate_path=True, model_size_wsi="small", dropout=False, n_classes=4), sd).eval()


@pytest.mark.parametrize("n_streams", [2, 3])
def test_bags_in_flight_match_sequential(n_streams):
    from multimodalfusion_amd.pipeline import BagsInFlight
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    model = _model()
    loss_fn = NLLSurvLoss(alpha=0.0)
    sizes = [3000, 700, 9000, 1, 4100, 2500, 33]
    bags = [torch.as_tensor(gen.bag(100 + i, n)).to(DEV) for i, n in enumerate(sizes)]
    ys = [torch.tensor([i % 4], device=DEV) for i in range(len(bags))]
    c = torch.tensor([0.0], device=DEV)
    params = [p for p in model.parameters()]
    from multimodalfusion_amd.dp import flat_layout
    offs, total_len = flat_layout(params)       # the slots' layout: every tensor on a 16-byte boundary, zero padding

    def pack(grads):
        flat = torch.zeros(total_len, device=DEV)
        for g, p, off in zip(grads, params, offs):
            flat[off:off + p.numel()] = g.reshape(-1)
        return flat

    def loss_of(i):
        hz, S, _, _ = model(path_features=bags[i])
        return loss_fn(hazards=hz, S=S, Y=ys[i], c=c)

    # sequential reference: per-bag flat gradients on the default stream
    want, want_loss = [], []
    for i in range(len(bags)):
        l = loss_of(i)
        want.append(pack(torch.autograd.grad(l, params)))
        want_loss.append(l.detach())
    torch.cuda.synchronize()

    pipe = BagsInFlight(model, n_streams)
    losses = [pipe.run(lambda i=i: loss_of(i)) for i in range(len(bags))]
    total = pipe.reduce().clone()
    torch.cuda.synchronize()
    for a, b in zip(losses, want_loss):
        assert torch.equal(a.detach(), b)
    # slot s accumulated bags s, s+n, ... in order; the slots are summed 0, 1, ...: replay that order exactly
    ref = None
    for s in range(n_streams):
        acc = None
        for i in range(s, len(bags), n_streams):
            acc = want[i].clone() if acc is None else acc.add_(want[i])
        ref = acc if ref is None else ref.add_(acc)
    assert torch.equal(total, ref)
    # a second window starts clean
    pipe.run(lambda: loss_of(0))
    again = pipe.reduce()
    torch.cuda.synchronize()
    assert torch.equal(again, want[0])


def test_assign_grads_feeds_an_optimizer():
    from multimodalfusion_amd.pipeline import BagsInFlight
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    model = _model()
    loss_fn = NLLSurvLoss(alpha=0.0)
    x = torch.as_tensor(gen.bag(7, 800)).to(DEV)
    pipe = BagsInFlight(model, 2)
    for _ in range(4):
        pipe.run(lambda: loss_fn(hazards=model(path_features=x)[0], S=model(path_features=x)[1], Y=torch.tensor([1], device=DEV),
                                 c=torch.tensor([0.0], device=DEV)))
    flat = pipe.reduce()
    pipe.assign_grads(flat)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in model.parameters()]
    opt.step()
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, model.parameters()))
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())

"""GPU: device-resident dropout seed and hipGraph capture of a train-mode step."""
import numpy as np
import pytest
import torch

from oracle import cases
from test_gpu_path import DEV, _load, _t

pytestmark = pytest.mark.gpu


def _model_and_bag(N=3000):
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    m = dict(N=N, gated=True, size="small", K=4, dropout=True, y=1, c=0, alpha=0.0, bias_std=0.05,
             train=True, seed=21, x_seed=22, mask_seed=0)
    sd, x, _ = cases.path_inputs(m)
    model = _load(MIL_Attention_fc_surv_path(gate_path=True, dropout=True, n_classes=4), sd).train()
    return m, model, _t(x)


def _step(model, x, host_seed, monkeypatch):
    from multimodalfusion_amd import ops
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    monkeypatch.setattr(ops, "next_dropout_seed", lambda: host_seed)
    for p in model.parameters():
        p.grad = None
    hz, S, Yh, A = model(path_features=x)
    loss = NLLSurvLoss(alpha=0.0)(hazards=hz, S=S, Y=torch.tensor([1], device=DEV), c=torch.tensor([0.], device=DEV))
    loss.backward()
    return loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}


def test_device_seed_adds_to_host_seed(monkeypatch):
    """effective seed = host seed + device word (uint32 wrap): all three dropout sites, forward and backward."""
    from multimodalfusion_amd.graph import DeviceSeed
    m, model, x = _model_and_bag()
    with DeviceSeed(0xFFFFFF00) as ds:
        l1, g1 = _step(model, x, 0x00000345, monkeypatch)
    l2, g2 = _step(model, x, (0x00000345 + 0xFFFFFF00) & 0xFFFFFFFF, monkeypatch)
    l3, _ = _step(model, x, 0x00000345, monkeypatch)
    assert torch.equal(l1, l2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k
    assert not torch.equal(l1, l3)          # and the device word really changed the masks


def test_graphed_train_step_replays_with_fresh_masks(monkeypatch):
    from multimodalfusion_amd import ops
    from multimodalfusion_amd.graph import SEED_BUMP, GraphedStep
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    m, model, x = _model_and_bag(2000)
    monkeypatch.setattr(ops, "next_dropout_seed", lambda: 77)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    Y, c = torch.tensor([1], device=DEV), torch.tensor([0.], device=DEV)
    loss_fn = NLLSurvLoss(alpha=0.0)

    def fn():
        for p in model.parameters():
            p.grad.zero_()
        hz, S, Yh, A = model(path_features=x)
        loss = loss_fn(hazards=hz, S=S, Y=Y, c=c)
        loss.backward()
        return loss

    gs = GraphedStep(fn, warmup=2, seed0=1000)
    try:
        losses, grads = [], []
        for _ in range(3):
            out = gs()
            torch.cuda.synchronize()
            losses.append(float(out))
            grads.append(model.classifier.weight.grad.clone())
        seed_after = gs.seed.value()
    finally:
        gs.close()
    assert len({round(l, 7) for l in losses}) == 3            # a new mask on every replay
    # replay r used device word seed0 + (warmup + 1 + r) * BUMP; check the last one against an eager step
    assert seed_after == (1000 + (2 + 3) * SEED_BUMP) & 0xFFFFFFFF
    l_ref, g_ref = _step(model, x, (77 + seed_after) & 0xFFFFFFFF, monkeypatch)
    assert abs(float(l_ref) - losses[-1]) < 1e-6
    assert torch.allclose(g_ref["classifier.weight"], grads[-1], atol=1e-7)

"""The radiology head's step without an autograd graph (MIL_Attention_fc_surv_radio.nll_step: reduce_dim, then stack + head +
nll_surv + backward in one call with d loss / d bag, then reduce_dim's backward) against the reference fixtures, the fp64
oracle and the autograd path it replaces (models/model_attention_mil_radio.py:73-115 + utils/loss_utils.py:22-39)."""
import numpy as np
import pytest
import torch

from conftest import check_summary
from oracle import cases
from test_gpu_path import DEV, _grads, _load, _t, compare

pytestmark = pytest.mark.gpu


def _model(m, sd):
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_radio
    return _load(MIL_Attention_fc_surv_radio(radio_fusion="concat", gate_radio=m["gated"], dropout=m["dropout"],
                                             n_classes=m["K"], modalities=cases.MODS[:m["n_mod"]]), sd)


def test_radio_step_golden_cases(golden, monkeypatch):
    from multimodalfusion_amd import ops
    g = golden("radio")
    for name, m in g.meta.items():
        sd, xs, masks = cases.radio_inputs(m)
        model = _model(m, sd)
        if m["train"]:
            model.train()
            monkeypatch.setattr(ops, "next_dropout_seed", lambda: m["mask_seed"])
        else:
            model.eval()
        kw = {k: _t(x) for k, x in zip(cases.MODS, xs)}
        hz, S, Yh, A_raw, loss, risk = model.nll_step(torch.tensor([m["y"]], device=DEV), torch.tensor([float(m["c"])], device=DEV),
                                                      alpha=m["alpha"], **kw)
        res = dict(hazards=hz.cpu().numpy(), S=S.cpu().numpy(), Y_hat=Yh.cpu().numpy(), A_raw=A_raw.cpu().numpy(),
                   loss=float(loss), M=None, grads=_grads(model))
        compare(res, cases.run_radio(m), name)
        tag = name + "/f64"
        assert abs(res["loss"] - float(g[tag + "/loss"])) <= 1e-5
        for k, gr in res["grads"].items():
            check_summary(g, f"{tag}/grad/{k}", gr, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,n_mod", [(512, 4), (77, 2), (300, 1), (16389, 4)])
def test_radio_step_equals_autograd_path(n, n_mod):
    from multimodalfusion_amd import ops
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    m = dict(n=n, n_mod=n_mod, gated=True, K=4, dropout=True, y=3, c=0, alpha=0.2, bias_std=0.05, train=True, seed=70 + n_mod,
             x_seed=80 + n_mod, mask_seed=5)
    sd, xs, _ = cases.radio_inputs(m)
    model = _model(m, sd).train()
    kw = {k: _t(x) for k, x in zip(cases.MODS, xs)}
    Y, c = torch.tensor([3], device=DEV), torch.tensor([0.0], device=DEV)
    torch.manual_seed(2)
    ops._drop_calls = 0
    hz, S, Yh, A_raw = model(**kw)
    loss = NLLSurvLoss(alpha=0.2)(hazards=hz, S=S, Y=Y, c=c)
    (loss * 0.5).backward()
    ref = {k: p.grad.clone() for k, p in model.named_parameters()}
    for p in model.parameters():
        p.grad = None
    ops._drop_calls = 0
    hz2, S2, Yh2, A2, loss2, _ = model.nll_step(Y, c, alpha=0.2, loss_scale=0.5, **kw)
    assert torch.allclose(hz2, hz.detach(), rtol=0, atol=1e-6) and torch.equal(A2, A_raw.detach()) and torch.equal(Yh2, Yh)
    assert abs(float(loss2) - float(loss.detach())) <= 1e-6 * max(1.0, abs(float(loss.detach())))
    for k, p in model.named_parameters():
        tol = 1e-6 + 2e-5 * float(ref[k].abs().max())
        assert float((p.grad - ref[k]).abs().max()) <= tol, k
    # .grad present: added to; grad_out: written / added, .grad untouched
    g1 = [p.grad.clone() for p in model.parameters()]
    ops._drop_calls = 0
    model.nll_step(Y, c, alpha=0.2, loss_scale=0.5, **kw)
    for a, p in zip(g1, model.parameters()):
        assert torch.allclose(p.grad, 2 * a, rtol=1e-5, atol=1e-9)
    views = [torch.full_like(p, 3.0) for p in model.parameters()]
    for p in model.parameters():
        p.grad = None
    ops._drop_calls = 0
    model.nll_step(Y, c, alpha=0.2, loss_scale=0.5, grad_out=views, accumulate=False, **kw)
    assert all(p.grad is None for p in model.parameters())
    for a, v in zip(g1, views):
        assert torch.equal(a, v)
    ops._drop_calls = 0
    model.nll_step(Y, c, alpha=0.2, loss_scale=0.5, grad_out=views, accumulate=True, **kw)
    for a, v in zip(g1, views):
        assert torch.allclose(v, 2 * a, rtol=1e-5, atol=1e-9)

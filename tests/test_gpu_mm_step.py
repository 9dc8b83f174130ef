"""The multimodal head's (concat and tensor fusion) one-call training step (MM_MIL_Attention_fc_surv.nll_step; mmf_surv_head_nll_step) against the
fp64 oracle / the reference-generated fixtures and against the autograd path it replaces (models/model_mm_attention_mil.py:
128-200 + utils/loss_utils.py:22-39 of the reference)."""
import numpy as np
import pytest
import torch

from conftest import check_summary
from oracle import cases
from test_gpu_path import DEV, _grads, _load, _t, compare

pytestmark = pytest.mark.gpu


def _model(m, sd=None, **kw):
    from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
    model = MM_MIL_Attention_fc_surv(input_dim=m["G"], radio_fusion="concat", fusion=m["fusion"], gate=True,
                                     gate_path=m.get("gate_path", True), gate_omic=True, gate_radio=m.get("gate_radio", True),
                                     dropout=m.get("dropout", False), n_classes=m["K"], mode=m["mode"], **kw)
    return _load(model, sd) if sd is not None else model.to(DEV)


def _inputs(m):
    sd, xs, xp, xo = cases.mm_inputs(m)
    kw = {k: _t(x) for k, x in zip(cases.MODS, xs)}
    kw["path_features"] = _t(xp)
    kw["genomic_features"] = _t(xo)
    return sd, kw


def _step(model, kw, m, **extra):
    Y, c = torch.tensor([m["y"]], device=DEV), torch.tensor([float(m["c"])], device=DEV)
    return model.nll_step(Y, c, alpha=m["alpha"], **extra, **kw)


def test_mm_step_golden_cases(golden):
    g = golden("mm")
    n = 0
    for name, m in g.meta.items():
        n += 1
        sd, kw = _inputs(m)
        model = _model(m, sd).eval()
        hz, S, Yh, A_raw, loss, risk = _step(model, kw, m)
        res = dict(hazards=hz.cpu().numpy(), S=S.cpu().numpy(), Y_hat=Yh.cpu().numpy(),
                   A_raw={k: v.cpu().numpy() for k, v in A_raw.items()}, loss=float(loss), M=None, grads=_grads(model))
        compare(res, cases.run_mm(m), name)
        tag = name + "/f32"
        assert abs(res["loss"] - float(g[tag + "/loss"])) <= 2e-5
        np.testing.assert_allclose(res["hazards"], g[tag + "/hazards"], rtol=0, atol=1e-4)
        for k, gr in res["grads"].items():
            check_summary(g, f"{tag}/grad/{k}", gr, rtol=2e-4, atol=2e-5)
        assert abs(float(risk) + float(S.sum())) <= 1e-6
    assert n >= 1


@pytest.mark.parametrize("fusion", ["concat", "tensor"])
@pytest.mark.parametrize("mode,Np,dropout", [("radio_path_omic", 700, True), ("radio_path_omic", 31000, True),
                                             ("path_omic", 31000, False), ("radio_path", 900, True), ("radio_omic", 0, True)])
def test_mm_step_equals_autograd_path(mode, Np, dropout, fusion):
    """Same seeds -> same dropout draws: outputs and every gradient agree with model(**kw) + loss + backward to fp32 rounding
    of the head (one fused launch instead of three), in train mode, with and without the side stream (31000 rows fork)."""
    from multimodalfusion_amd import ops
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    m = dict(fusion=fusion, mode=mode, Np=max(Np, 8), nr=64, G=80, gate_path=True, gate_radio=True, K=4, seed=11, x_seed=12,
             y=2, c=0, alpha=0.3, bias_std=0.02, dropout=dropout)
    sd, kw = _inputs(m)
    model = _model(m, sd).train()
    Y, c = torch.tensor([2], device=DEV), torch.tensor([0.0], device=DEV)
    torch.manual_seed(5)
    ops._drop_calls = 0
    hz, S, Yh, A_raw = model(**kw)
    loss = NLLSurvLoss(alpha=0.3)(hazards=hz, S=S, Y=Y, c=c)
    (loss * 0.25).backward()
    ref = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    none = {k for k, p in model.named_parameters() if p.grad is None}
    for p in model.parameters():
        p.grad = None
    ops._drop_calls = 0
    hz2, S2, Yh2, A2, loss2, risk2 = model.nll_step(Y, c, alpha=0.3, loss_scale=0.25, **kw)
    torch.cuda.synchronize()
    assert torch.allclose(hz2, hz, rtol=0, atol=1e-6) and torch.allclose(S2, S, rtol=0, atol=1e-6) and torch.equal(Yh2, Yh)
    assert abs(float(loss2) - float(loss)) <= 1e-6 * max(1.0, abs(float(loss)))
    for k in A_raw:
        assert torch.equal(A2[k], A_raw[k]), k
    got = {k: p.grad for k, p in model.named_parameters()}
    assert {k for k, v in got.items() if v is None} == none
    for k, r in ref.items():
        tol = 1e-6 + 2e-5 * float(r.abs().max())
        assert float((got[k] - r).abs().max()) <= tol, (k, float((got[k] - r).abs().max()), tol)


@pytest.mark.parametrize("fusion", ["concat", "tensor"])
def test_mm_step_accumulates_and_fills_grad_out(fusion):
    m = dict(fusion=fusion, mode="radio_path_omic", Np=500, nr=48, G=80, gate_path=True, gate_radio=False, K=4, seed=3,
             x_seed=4, y=0, c=1, alpha=0.0, bias_std=0.02)
    sd, kw = _inputs(m)
    model = _model(m, sd).eval()
    _step(model, kw, m)
    g1 = [p.grad.clone() for p in model.parameters()]
    _step(model, kw, m)                                   # .grad present: added to
    for a, p in zip(g1, model.parameters()):
        assert torch.allclose(p.grad, 2 * a, rtol=1e-6, atol=1e-9)
    views = [torch.full_like(p, 7.0) for p in model.parameters()]
    for p in model.parameters():
        p.grad = None
    _step(model, kw, m, grad_out=views, accumulate=False)
    assert all(p.grad is None for p in model.parameters())
    for a, v in zip(g1, views):
        assert torch.equal(a, v)
    _step(model, kw, m, grad_out=views, accumulate=True)
    for a, v in zip(g1, views):
        assert torch.allclose(v, 2 * a, rtol=1e-6, atol=1e-9)


def test_mm_step_bf16_path_bag_and_bit_reproducible():
    from multimodalfusion_amd import ops
    m = dict(fusion="concat", mode="radio_path_omic", Np=4096, nr=64, G=80, gate_path=True, gate_radio=True, K=4, seed=8,
             x_seed=9, y=1, c=0, alpha=0.0, bias_std=0.02, dropout=True)
    sd, kw = _inputs(m)
    kw["path_features"] = kw["path_features"].to(torch.bfloat16)
    model = _model(m, sd).train()
    outs = []
    for _ in range(2):
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(1)
        ops._drop_calls = 0
        r = _step(model, kw, m)
        torch.cuda.synchronize()
        outs.append((r[4].clone(), [p.grad.clone() for p in model.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.isfinite(outs[0][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)


def test_mm_step_rejects_what_it_does_not_cover():
    from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
    model = MM_MIL_Attention_fc_surv(input_dim=80, fusion="concat", n_classes=4).to(DEV)
    model.classifier.bias.requires_grad_(False)
    with pytest.raises(RuntimeError):
        model.nll_step(torch.tensor([1]), torch.tensor([0.0]), path_features=torch.zeros(8, 1024, device=DEV))


def test_surv_head_nll_step_shapes():
    """The head launch alone, F not a multiple of 32 and K = 1 / 32, against torch autograd on the same formulas."""
    from multimodalfusion_amd import ops
    from oracle import torch_port as tp
    gen = torch.Generator().manual_seed(0)
    for F, K, y, cc, alpha in ((768, 4, 1, 0.0, 0.0), (100, 1, 0, 1.0, 0.4), (1024, 32, 31, 0.0, 0.2), (7, 3, 2, 1.0, 0.0)):
        feat = torch.randn(1, F, generator=gen)
        Wk, bk = torch.randn(K, F, generator=gen) * 0.05, torch.randn(K, generator=gen) * 0.1
        f64, W64, b64 = (t.double().requires_grad_(True) for t in (feat, Wk, bk))
        hz = torch.sigmoid(f64 @ W64.T + b64)
        S = torch.cumprod(1 - hz, dim=1)
        loss = tp.nll_loss(hz, S, torch.tensor([[y]]), torch.tensor([[cc]], dtype=torch.float64), alpha=alpha)
        (loss * 0.5).backward()
        dWk, dbk = torch.empty(K, F, device=DEV), torch.empty(K, device=DEV)
        out = ops.surv_head_nll_step(feat.to(DEV), Wk.to(DEV), bk.to(DEV), torch.tensor([y]), torch.tensor([cc]), alpha,
                                     dWk, dbk, loss_scale=0.5)
        assert abs(float(out[3]) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
        np.testing.assert_allclose(out[0].cpu().numpy(), hz.detach().numpy(), rtol=0, atol=1e-6)
        np.testing.assert_allclose(out[5].cpu().numpy(), f64.grad.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(dWk.cpu().numpy(), W64.grad.numpy(), rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(dbk.cpu().numpy(), b64.grad.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("which", ["mm_concat", "mm_tensor", "radio"])
@pytest.mark.parametrize("flat", [False, True])
def test_loop_takes_the_one_call_step_and_matches_the_autograd_route(which, flat):
    """utils/core_utils.train_loop_survival (gc = 2, four patients, eval-mode masks off) with the one-call step against the
    same loop with `model.mmf_one_call_step = False` (model(**feats) + loss + backward through autograd): same losses, same
    parameters after the two optimizer steps, with torch.optim.Adam + autograd L1 and with FlatAdam (fused L1 + Adam tail)."""
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_radio, MM_MIL_Attention_fc_surv
    from multimodalfusion_amd.optim import FlatAdam
    from multimodalfusion_amd.utils import core_utils
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    from multimodalfusion_amd.utils.utils import l1_reg_all
    rng = np.random.default_rng(3)
    loader = []
    for i in range(4):
        radio = {m: torch.as_tensor(rng.standard_normal((40 + 8 * i, 1024)).astype(np.float32)) for m in cases.MODS}
        path = torch.as_tensor(rng.standard_normal((300 + 50 * i, 1024)).astype(np.float32))
        omic = torch.as_tensor(rng.standard_normal((1, 80)).astype(np.float32))
        loader.append((radio, path, omic, torch.tensor([i % 4]), np.array([10.0 + i]), torch.tensor([float(i % 2)])))
    outs = []
    for one_call in (True, False):
        torch.manual_seed(11)
        if which == "radio":
            model, mode = MIL_Attention_fc_surv_radio(n_classes=4, dropout=False), "radio"
        else:
            model, mode = MM_MIL_Attention_fc_surv(input_dim=80, fusion=which[3:], n_classes=4), "radio_path_omic"
        model = model.to(DEV)
        model.eval()
        model.train = lambda mode=True, _m=model: _m          # dropout off: the two routes draw their seeds alike anyway
        model.mmf_one_call_step = one_call
        called = []
        orig = model.nll_step
        model.nll_step = lambda *a, **k: (called.append(1), orig(*a, **k))[1]
        if flat:
            opt = FlatAdam(model, lr=1e-3, weight_decay=1e-4, lambda_l1=1e-5)
        else:
            opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        out = core_utils.train_loop_survival(0, model, loader, opt, 4, mode, loss_fn=NLLSurvLoss(alpha=0.1), reg_fn=l1_reg_all,
                                             lambda_reg=1e-5, gc=2)
        assert bool(called) == one_call
        outs.append((out["losses"].copy(), {k: v.detach().clone() for k, v in model.state_dict().items()}))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6)
    for k, v in outs[0][1].items():
        ref = outs[1][1][k]
        # Adam divides by sqrt(v): an element whose gradient is a near-cancellation moves by lr * O(relative gradient error),
        # so the bar is a few percent of one normalised step (lr = 1e-3); gradient-level equality is tested above
        d = float((v - ref).abs().max())
        assert d <= 5e-5 + 2e-5 * float(ref.abs().max()), (k, d)

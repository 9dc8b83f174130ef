"""GPU parity at the FULL sizes BASELINE.json's configs 4 and 5 name, against the CPU oracle (not only properties,
not HIP-vs-HIP):

  config 4  mm_attention_mil (path + radio + omic fused), 50,000 x 1024 fp32 path bag, 4 x 512 x 1024 radio, omic[80],
            concat and tensor fusion, vs the fp64 autograd oracle (oracle/torch_port.py);
  config 5  bf16 storage, 100,000 x 1024 path bag: the path head vs oracle/bf16_port.py (the kernels' rounding
            points, hand-derived backward) and the multimodal head vs its MM extension (mm_step_bf16: pathology
            branch rounded, radio / omic / fusion as the fp32 reference).

Bars as everywhere else: fp32 -- scores / hazards 1e-4, loss 1e-5, gradients 1e-5 + 1e-4 max|g|; bf16 vs the bf16
oracle -- the bars of tests/test_gpu_bf16.py (scores 5e-3 with the 99th percentile below 2e-4, hazards 2e-3,
loss 1e-3, gradients 1 % in norm).  Each oracle run is 10-40 s of CPU.
"""
import numpy as np
import pytest
import torch

from oracle import bf16_port, cases
from test_gpu_bf16 import ZERO_GRADS, compare_bf16, run_path_hip_bf16
from test_gpu_omic_mm import run_mm_hip
from test_gpu_path import DEV, _grads, _load, _t, compare, relu_kink_units

pytestmark = pytest.mark.gpu


def _config4(fusion):
    return dict(fusion=fusion, mode="radio_path_omic", Np=50_000, nr=512, G=80, gate_path=True, gate_radio=True, K=4,
                seed=404, x_seed=405, y=1, c=0, alpha=0.0, bias_std=0.02)


_config4_ref = {}


def _config4_oracle(fusion):
    """The fp64 oracle of config 4 and the ReLU-kink allowance, computed once per fusion (10-40 s of CPU each)."""
    if fusion not in _config4_ref:
        m = _config4(fusion)
        sd, xs, xp, xo = cases.mm_inputs(m)
        _config4_ref[fusion] = (cases.run_mm(m), relu_kink_units(sd, xp))
    return _config4_ref[fusion]


@pytest.mark.parametrize("fusion", ["concat", "tensor"])
def test_config4_mm_with_50k_fp32_path_bag_vs_fp64_oracle(fusion):
    res = run_mm_hip(_config4(fusion))
    ref, kink = _config4_oracle(fusion)
    compare(res, ref, f"config4 {fusion}", kink_units=kink)


@pytest.mark.parametrize("fusion", ["concat", "tensor"])
def test_config4_mm_one_call_step_50k_vs_fp64_oracle(fusion):
    """The same configuration through the step the loop mirror runs (MM_MIL_Attention_fc_surv.nll_step: branch calls by hand on
    two streams, head + loss + backward in one launch), against the same oracle."""
    from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
    m = _config4(fusion)
    sd, xs, xp, xo = cases.mm_inputs(m)
    model = _load(MM_MIL_Attention_fc_surv(input_dim=80, radio_fusion="concat", fusion=fusion, gate=True, gate_path=True,
                                           gate_omic=True, gate_radio=True, n_classes=4, mode=m["mode"]), sd).eval()
    kw = {k: _t(x) for k, x in zip(cases.MODS, xs)}
    kw["path_features"] = _t(xp)
    kw["genomic_features"] = _t(xo)
    hz, S, Yh, A_raw, loss, risk = model.nll_step(torch.tensor([m["y"]], device=DEV), torch.tensor([float(m["c"])], device=DEV),
                                                  alpha=m["alpha"], **kw)
    torch.cuda.synchronize()
    res = dict(hazards=hz.cpu().numpy(), S=S.cpu().numpy(), Y_hat=Yh.cpu().numpy(),
               A_raw={k: v.cpu().numpy() for k, v in A_raw.items()}, loss=float(loss), M=None, grads=_grads(model))
    ref, kink = _config4_oracle(fusion)
    compare(res, ref, f"config4 one-call {fusion}", kink_units=kink)


def test_config5_bf16_path_head_100k_vs_bf16_oracle(monkeypatch):
    m = dict(seed=505, gated=True, size="small", K=4, dropout=False, bias_std=0.02, x_seed=506, N=100_000, train=True,
             mask_seed=5050, y=1, c=0, alpha=0.0)
    sd, x, masks = cases.path_inputs(m)
    xq = bf16_port.rb(bf16_port._t(x)).numpy()
    del x
    res = run_path_hip_bf16(m, monkeypatch, xq)
    ref = bf16_port.path_step_bf16(sd, xq, m["y"], m["c"], m["alpha"], gated=True, dropout=False, masks=masks)
    compare_bf16(res, ref, "config5 path 100k", a_tol=5e-3, h_tol=2e-3, l_tol=1e-3, g_rel=1e-2)


_config5_ref = {}


@pytest.mark.parametrize("route", ["autograd", "one_call"])
@pytest.mark.parametrize("fusion", ["concat", "tensor"])
def test_config5_mm_with_100k_bf16_path_bag_vs_bf16_oracle(fusion, route):
    """route: model(**kw) + loss + backward through autograd, or the step the loop mirror runs (nll_step); one oracle run per
    fusion serves both."""
    from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    m = dict(fusion=fusion, mode="radio_path_omic", Np=100_000, nr=512, G=80, gate_path=True, gate_radio=True, K=4,
             seed=515, x_seed=516, y=2, c=0, alpha=0.0, bias_std=0.02)
    sd, xs, xp, xo = cases.mm_inputs(m)
    xq = bf16_port.rb(bf16_port._t(xp)).numpy()
    del xp
    model = _load(MM_MIL_Attention_fc_surv(input_dim=80, radio_fusion="concat", fusion=fusion, gate=True, gate_path=True,
                                           gate_omic=True, gate_radio=True, n_classes=4, mode=m["mode"]), sd).eval()
    kw = {k: _t(x) for k, x in zip(cases.MODS, xs)}
    kw["path_features"] = torch.as_tensor(xq).to(torch.float32).to(torch.bfloat16).to(DEV)
    kw["genomic_features"] = _t(xo)
    if route == "autograd":
        hz, S, Yh, A_raw = model(**kw)
        loss = NLLSurvLoss(alpha=0.0)(hazards=hz, S=S, Y=torch.tensor([m["y"]], device=DEV), c=torch.tensor([0.0], device=DEV))
        loss.backward()
    else:
        hz, S, Yh, A_raw, loss, _ = model.nll_step(torch.tensor([m["y"]], device=DEV), torch.tensor([0.0], device=DEV), alpha=0.0, **kw)
        torch.cuda.synchronize()
    res = dict(hazards=hz.detach().cpu().numpy(), loss=float(loss.detach()), grads=_grads(model),
               A={k: v.detach().cpu().numpy() for k, v in A_raw.items()})
    if fusion not in _config5_ref:
        _config5_ref[fusion] = bf16_port.mm_step_bf16(sd, xs, xq, xo, m["y"], m["c"], m["alpha"], fusion=fusion, gate_path=True,
                                                      gate_radio=True, mode=m["mode"])
    ref = _config5_ref[fusion]
    tag = f"config5 mm {fusion} {route}"
    assert abs(res["loss"] - ref["loss"]) <= 1e-3, (tag, res["loss"], ref["loss"])
    np.testing.assert_allclose(res["hazards"], ref["hazards"], rtol=0, atol=2e-3, err_msg=tag)
    # the radiology branch never sees bf16: the fp32 bar
    np.testing.assert_allclose(res["A"]["radiology"], ref["A_raw"]["radiology"], rtol=0, atol=1e-4, err_msg=tag)
    dA = np.abs(res["A"]["pathology"] - ref["A_raw"]["pathology"])
    assert float(dA.max()) <= 5e-3 and float(np.quantile(dA, 0.99)) <= 2e-4, (tag, float(dA.max()))
    for k, g in ref["grads"].items():
        if k.endswith(ZERO_GRADS):
            assert float(np.abs(res["grads"][k]).max()) <= 1e-4, (tag, k)
            continue
        err, nrm = float(np.linalg.norm(res["grads"][k] - g)), float(np.linalg.norm(g))
        if k.startswith("attention_net_WSI."):
            assert err <= 1e-2 * nrm + 1e-6, f"{tag} grad {k}: |err| {err:.3e} vs |g| {nrm:.3e}"
        else:   # parameters outside the rounded branch see it only through M_path (1 x 256) and d(M_path)
            assert err <= 2e-3 * nrm + 1e-6, f"{tag} grad {k}: |err| {err:.3e} vs |g| {nrm:.3e}"

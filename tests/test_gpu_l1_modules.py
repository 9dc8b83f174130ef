"""GPU: FlatAdam with an L1 mask (`l1_modules=[model.fc_omic, model.mm]`) == torch Adam + the reference's
`l1_reg_modules` through autograd (utils/utils.py:259-268: only the omic SNN and the fusion block are regularised),
two optimizer steps of two micro-batches each on the multimodal model; and the loop refuses a mismatched pair."""
import numpy as np
import pytest
import torch

from oracle import cases
from test_gpu_path import DEV, _load, _t

pytestmark = pytest.mark.gpu

LR, WD, LAM = 1e-3, 1e-5, 1e-3


def _model_and_batches():
    from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
    m = dict(fusion="tensor", mode="radio_path_omic", Np=300, nr=40, G=80, gate_path=True, gate_radio=True, K=4,
             seed=61, x_seed=62, y=1, c=0, alpha=0.0, bias_std=0.02)
    sd, xs, xp, xo = cases.mm_inputs(m)
    model = _load(MM_MIL_Attention_fc_surv(input_dim=80, radio_fusion="concat", fusion="tensor", gate=True, gate_path=True,
                                           gate_omic=True, gate_radio=True, n_classes=4, mode=m["mode"]), sd).eval()
    kw = {k: _t(x) for k, x in zip(cases.MODS, xs)}
    kw["path_features"] = _t(xp)
    kw["genomic_features"] = _t(xo)
    return model, kw


def _run(fused):
    from multimodalfusion_amd.optim import FlatAdam
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    from multimodalfusion_amd.utils.utils import l1_reg_modules
    model, kw = _model_and_batches()
    loss_fn = NLLSurvLoss(alpha=0.0)
    Y, c = torch.tensor([1], device=DEV), torch.tensor([0.0], device=DEV)
    if fused:
        opt = FlatAdam(model, lr=LR, weight_decay=WD, lambda_l1=LAM, l1_modules=[model.fc_omic, model.mm])
        want_l1 = LAM * float(l1_reg_modules(model))
        assert abs(float(opt.l1_value()) - want_l1) <= 1e-5 * want_l1
    else:
        opt = torch.optim.Adam(model.parameters(), lr=LR, weight_decay=WD)
    for step in range(2):
        for micro in range(2):
            hz, S, _, _ = model(**kw)
            loss = loss_fn(hazards=hz, S=S, Y=Y, c=c) / 2
            if not fused:
                loss = loss + l1_reg_modules(model) * LAM
            loss.backward()
        if fused:
            opt.step(l1_micro_batches=2)
        else:
            opt.step()
        opt.zero_grad()
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def test_flat_adam_l1_mask_matches_l1_reg_modules():
    a, b = _run(True), _run(False)
    for k in b:
        if k.endswith("attention_c.bias"):
            continue      # its gradient is analytically zero: Adam normalises pure rounding noise to +-lr
        np.testing.assert_allclose(a[k], b[k], rtol=2e-5, atol=2e-7, err_msg=k)


def test_loop_refuses_a_mismatched_regulariser():
    from multimodalfusion_amd.optim import FlatAdam
    from multimodalfusion_amd.utils import core_utils
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    from multimodalfusion_amd.utils.utils import l1_reg_modules
    model, kw = _model_and_batches()
    opt = FlatAdam(model, lr=LR)          # no mask: regularises everything
    with pytest.raises(ValueError):
        core_utils.train_loop_survival(0, model, [], opt, 4, "radio_path_omic", loss_fn=NLLSurvLoss(alpha=0.0),
                                       reg_fn=l1_reg_modules, lambda_reg=LAM, gc=2)

"""GPU parity: omic MaxNet (SNN + nll / Cox) and the multimodal model (concat and Kronecker fusion) against
the golden fixtures made from the reference and the live fp64 oracle."""
import numpy as np
import pytest
import torch

from conftest import check_summary
from oracle import cases
from oracle import inputs as gen
from oracle import torch_port as tp
from test_gpu_path import DEV, _grads, _load, _t, compare

pytestmark = pytest.mark.gpu


def run_omic_hip(m, monkeypatch):
    from multimodalfusion_amd import ops
    from multimodalfusion_amd.models import MaxNet
    from multimodalfusion_amd.utils.loss_utils import CoxSurvLoss, NLLSurvLoss
    sd, x, t, c, keeps = cases.omic_inputs(m)
    model = _load(MaxNet(input_dim=m["G"], model_size_omic="small",
                         bag_loss="nll_surv" if m["nll"] else "cox_surv", n_classes=m["K"]), sd)
    if m["train"]:
        model.train()
        monkeypatch.setattr(ops, "next_dropout_seed", lambda: m["mask_seed"])
    else:
        model.eval()
    xt = _t(x)
    res = model(genomic_features=xt)
    if m["nll"]:
        hz, S, Yh, _ = res
        loss = NLLSurvLoss(alpha=m["alpha"])(hazards=hz[0], S=S[0], Y=torch.tensor([m["y"]], device=DEV),
                                             c=_t(c[:1]))
        out = dict(hazards=hz.detach().cpu().numpy(), S=S.detach().cpu().numpy(), Y_hat=Yh.cpu().numpy())
    else:
        risk = res[0]
        loss = CoxSurvLoss()(risks=risk, times=torch.as_tensor(t), c=_t(c))
        out = dict(hazards=risk.detach().reshape(-1).cpu().numpy())
    loss.backward()
    out["loss"] = float(loss.detach())
    out["grads"] = _grads(model)
    if not m["train"]:
        with torch.no_grad():
            out["M"] = model(genomic_features=xt, return_features=True).cpu().numpy()
    return out


def test_omic_golden_cases(golden, monkeypatch):
    g = golden("omic")
    for name, m in g.meta.items():
        res = run_omic_hip(m, monkeypatch)
        ref = cases.run_omic(m)
        assert abs(res["loss"] - float(ref["loss"])) <= 1e-5 * max(1.0, abs(float(ref["loss"]))), name
        np.testing.assert_allclose(res["hazards"], ref["hazards"], atol=1e-4, err_msg=name)
        if "S" in res:
            np.testing.assert_allclose(res["S"], ref["S"], atol=1e-4, err_msg=name)
            assert np.array_equal(res["Y_hat"], ref["Y_hat"])
        if "M" in res:
            np.testing.assert_allclose(res["M"], ref["M"], atol=1e-4, err_msg=name)
        for k, gr in ref["grads"].items():
            tol = 1e-5 + 1e-4 * float(np.abs(gr).max())
            assert float(np.abs(res["grads"][k] - gr).max()) <= tol, (name, k)
        tag = name + "/f64"
        assert abs(res["loss"] - float(g[tag + "/loss"])) <= 1e-5 * max(1.0, abs(float(g[tag + "/loss"])))
        for k, gr in res["grads"].items():
            check_summary(g, f"{tag}/grad/{k}", gr, rtol=1e-4, atol=1e-5)


def run_mm_hip(m, train_seeds=None, monkeypatch=None):
    from multimodalfusion_amd import ops
    from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    sd, xs, xp, xo = cases.mm_inputs(m)
    model = _load(MM_MIL_Attention_fc_surv(input_dim=m["G"], radio_fusion="concat", fusion=m["fusion"], gate=True,
                                           gate_path=m["gate_path"], gate_omic=True, gate_radio=m["gate_radio"],
                                           dropout=m.get("dropout", False), n_classes=m["K"], mode=m["mode"]), sd)
    if train_seeds:
        model.train()
        it = iter(train_seeds)
        monkeypatch.setattr(ops, "next_dropout_seed", lambda: next(it))
    else:
        model.eval()
    kw = {k: _t(x) for k, x in zip(cases.MODS, xs)}
    kw["path_features"] = _t(xp)
    kw["genomic_features"] = _t(xo)
    hz, S, Yh, A_raw = model(**kw)
    loss = NLLSurvLoss(alpha=m["alpha"])(hazards=hz, S=S, Y=torch.tensor([m["y"]], device=DEV),
                                         c=torch.tensor([float(m["c"])], device=DEV))
    loss.backward()
    return dict(hazards=hz.detach().cpu().numpy(), S=S.detach().cpu().numpy(), Y_hat=Yh.cpu().numpy(),
                A_raw={k: v.detach().cpu().numpy() for k, v in A_raw.items()}, loss=float(loss.detach()),
                M=None, grads=_grads(model))


def test_mm_golden_cases(golden):
    g = golden("mm")
    for name, m in g.meta.items():
        res = run_mm_hip(m)
        ref = cases.run_mm(m)
        compare(res, ref, name)
        tag = name + "/f32"     # the reference itself ran fp32 for the tensor cases (FloatTensor shim)
        assert abs(res["loss"] - float(g[tag + "/loss"])) <= 2e-5
        np.testing.assert_allclose(res["hazards"], g[tag + "/hazards"], rtol=0, atol=1e-4)
        for k, A in res["A_raw"].items():
            check_summary(g, f"{tag}/A_raw_{k}", A, rtol=0, atol=1e-4)
        for k, gr in res["grads"].items():
            check_summary(g, f"{tag}/grad/{k}", gr, rtol=2e-4, atol=2e-5)


def test_mm_tensor_train_mode_masks(monkeypatch):
    """Train mode, every dropout site active (AMIL x2, AlphaDropout x2, fusion x6, classifier): the oracle
    rebuilds each mask from the device's integer hash with the same (seed, site)."""
    m = dict(fusion="tensor", mode="radio_path_omic", Np=500, nr=40, G=80, gate_path=True, gate_radio=True, K=4,
             seed=1234, x_seed=4321, y=1, c=0, alpha=0.0, bias_std=0.05, dropout=True)
    seeds = [11, 22, 33, 44]          # radio stack, path stack, omic stack, fusion (+ classifier)
    res = run_mm_hip(m, train_seeds=seeds, monkeypatch=monkeypatch)
    sd_np, xs, xp, xo = cases.mm_inputs(m)
    sd_np = gen.mm_state_dict(seed=m["seed"], input_dim=m["G"], fusion="tensor", gate_path=True, gate_radio=True,
                              dropout=True, n_classes=4, mode=m["mode"], n_mod=4, bias_std=0.05)
    sd = tp.to_torch(sd_np, torch.float64)
    T = lambda a: torch.as_tensor(np.asarray(a)).double()
    mk = lambda seed, site, r, c: T(gen.drop_scale_mask(seed, site, r, c, 0.25, np.float64))
    masks = {
        "radio": {"h": mk(11, 0, 40, 256), "a": mk(11, 1, 40, 256), "b": mk(11, 2, 40, 256)},
        "path": {"h": mk(22, 0, 500, 256), "a": mk(22, 1, 500, 256), "b": mk(22, 2, 500, 256)},
        "omic_keeps": [T(gen.keep_mask(33, i, 1, 256, 0.25).astype(np.float64)) for i in range(2)],
        "mm": {"o0": mk(44, 0, 1, 16), "o1": mk(44, 1, 1, 16), "o2": mk(44, 2, 1, 16),
               "post": mk(44, 8, 1, 17 ** 3), "enc1": mk(44, 9, 1, 512), "enc2": mk(44, 10, 1, 512)},
        "cls": mk(44, 11, 1, 256),
    }
    hz, S, Yh, A_raw, MM = tp.mm_forward(sd, [T(x) for x in xs], T(xp), T(xo), fusion="tensor", gate_path=True,
                                         gate_radio=True, dropout=True, mode=m["mode"], masks=masks)
    loss = tp.nll_loss(hz, S, torch.tensor([1]), torch.tensor([0.0]), alpha=0.0)
    gr = tp.grads_of(loss, sd)
    ref = dict(hazards=hz.detach().numpy(), S=S.detach().numpy(), loss=float(loss.detach()),
               A_raw={k: v.detach().numpy() for k, v in A_raw.items()}, M=None,
               grads={k: v.detach().numpy() for k, v in gr.items()})
    compare(res, ref, "mm tensor train")


def test_dense_odd_shapes():
    """The small dense kernels on awkward shapes (K = 186, 4913; B = 1, 3, 128) against torch fp64."""
    from multimodalfusion_amd import ops
    for (B, K, N, act) in ((1, 4913, 512, "relu"), (3, 186, 256, "selu"), (128, 36, 256, "selu"),
                           (1, 1280, 512, "relu"), (2, 768, 16, "none"), (5, 16, 16, "sigmoid")):
        x = gen.normal(1, (B, K), stream=K)
        W = gen.normal(2, (N, K), stream=N, std=1.0 / np.sqrt(K))
        b = gen.normal(3, (N,), stream=1, std=0.1)
        gy = gen.normal(4, (B, N), stream=2)
        tx, tW, tb = (_t(a).requires_grad_(True) for a in (x, W, b))
        y = ops.dense(tx, tW, tb, act=act)
        y.backward(_t(gy))
        rx, rW, rb = (torch.as_tensor(a).double().requires_grad_(True) for a in (x, W, b))
        pre = torch.nn.functional.linear(rx, rW, rb)
        ry = {"relu": torch.relu, "selu": torch.selu, "none": lambda v: v, "sigmoid": torch.sigmoid}[act](pre)
        ry.backward(torch.as_tensor(gy).double())
        np.testing.assert_allclose(y.detach().cpu().numpy(), ry.detach().numpy(), atol=2e-5, rtol=1e-5)
        for a, r in ((tx, rx), (tW, rW), (tb, rb)):
            np.testing.assert_allclose(a.grad.cpu().numpy(), r.grad.numpy(), atol=2e-5, rtol=1e-4)


def test_mm_side_stream_branches_give_identical_results():
    """Big fp32 path bags run the radio / omic branches on a second HIP stream (model_mm_attention_mil.py: _side_stream);
    outputs and every gradient must equal the single-stream run (to fp32 rounding where the tile plan differs, see below)."""
    from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    m = dict(fusion="tensor", mode="radio_path_omic", Np=31000, nr=96, G=80, gate_path=True, gate_radio=True, K=4,
             seed=5, x_seed=6, y=1, c=0, alpha=0.0, bias_std=0.02)
    sd, xs, xp, xo = cases.mm_inputs(m)
    outs = []
    for side in (True, False):
        model = _load(MM_MIL_Attention_fc_surv(input_dim=80, radio_fusion="concat", fusion="tensor", gate=True,
                                               gate_path=True, gate_omic=True, gate_radio=True, n_classes=4,
                                               mode=m["mode"]), sd).eval()
        model.mmf_side_stream = side
        kw = {k: _t(x) for k, x in zip(cases.MODS, xs)}
        kw["path_features"] = _t(xp)
        kw["genomic_features"] = _t(xo)
        for _ in range(2):                      # second pass: the side stream already exists, buffers get recycled
            for p in model.parameters():
                p.grad = None
            hz, S, Yh, A_raw = model(**kw)
            loss = NLLSurvLoss(alpha=0.0)(hazards=hz, S=S, Y=torch.tensor([1], device=DEV), c=torch.tensor([0.0], device=DEV))
            loss.backward()
        torch.cuda.synchronize()
        outs.append((hz.detach().clone(), {k: v.detach().clone() for k, v in A_raw.items()},
                     {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    # the side-stream run plans the pathology stack's wide tiles for 224 CUs (mmf_amil_desc::concurrent): a row that lands
    # in a 16-row half block (v_mfma_f32_16x16x4_f32: four k per instruction) in one plan and in a 32-row block (32x32x2) in
    # the other is summed in the same k order but rounded per instruction -- equal to fp32 rounding, not to the bit; the
    # radio / omic branches, whose plan does not change, stay bit-identical
    (h1, a1, g1), (h2, a2, g2) = outs
    assert torch.allclose(h1, h2, rtol=0, atol=1e-6)
    assert torch.equal(a1["radiology"], a2["radiology"])
    assert torch.allclose(a1["pathology"], a2["pathology"], rtol=0, atol=2e-5)
    for k in g1:
        tol = 1e-7 + 2e-5 * float(g2[k].abs().max())
        assert float((g1[k] - g2[k]).abs().max()) <= tol, k

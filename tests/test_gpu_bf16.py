"""GPU parity of the bf16-storage path (mmf_amil_bf16_*; BASELINE config 5) through the drop-in head.

Two references:
  * oracle/bf16_port.py with the kernels' rounding points (pinned on CPU by tests/test_oracle_bf16.py).  HIP and oracle
    then differ only by fp32-vs-fp64 accumulation, which can flip a bf16 rounding of a saved activation in a few
    elements (1 bf16 ulp = 2^-8 relative: a flipped h or a / b element moves one score by up to ~2.5e-3, measured);
    tolerances: scores 5e-3 abs with the 99th percentile below 2e-4, hazards 2e-3, loss 1e-3, gradients 1 % in norm;
  * the fp32/fp64 reference fixtures: bf16 quantisation only (scores 3e-2, hazards 1e-2, gradients 15 % in norm).
"""
import numpy as np
import pytest
import torch

from oracle import bf16_port, cases
from oracle import inputs as gen
from test_gpu_path import DEV, _grads, _load

pytestmark = pytest.mark.gpu

ZERO_GRADS = ("attention_c.bias", "module.2.bias", "module.3.bias")     # analytically zero


def run_path_hip_bf16(m, monkeypatch, x_np):
    from multimodalfusion_amd import ops
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    sd, _, _ = cases.path_inputs(m)
    model = _load(MIL_Attention_fc_surv_path(gate_path=m["gated"], model_size_wsi=m["size"],
                                             dropout=m["dropout"], n_classes=m["K"]), sd)
    if m["train"]:
        model.train()
        monkeypatch.setattr(ops, "next_dropout_seed", lambda: m["mask_seed"])
    else:
        model.eval()
    xt = torch.as_tensor(x_np).to(torch.float32).to(torch.bfloat16).to(DEV)
    hz, S, Yh, A_raw = model(path_features=xt)
    loss = NLLSurvLoss(alpha=m["alpha"])(hazards=hz, S=S, Y=torch.tensor([m["y"]], device=DEV),
                                         c=torch.tensor([float(m["c"])], device=DEV))
    loss.backward()
    with torch.no_grad():
        M = None if m["train"] else model(path_features=xt, return_features=True).cpu().numpy()
    torch.cuda.synchronize()
    return dict(hazards=hz.detach().cpu().numpy(), S=S.detach().cpu().numpy(), Y_hat=Yh.cpu().numpy(),
                A_raw=A_raw.detach().cpu().numpy(), loss=float(loss), M=M, grads=_grads(model))


def compare_bf16(res, ref, tag, a_tol, h_tol, l_tol, g_rel):
    assert abs(res["loss"] - float(ref["loss"])) <= l_tol, (tag, res["loss"], float(ref["loss"]))
    np.testing.assert_allclose(res["hazards"], ref["hazards"], rtol=0, atol=h_tol, err_msg=tag)
    np.testing.assert_allclose(res["A_raw"], ref["A_raw"], rtol=0, atol=a_tol, err_msg=tag)
    if a_tol < 1e-2 and res["A_raw"].size >= 100:        # vs the bf16 oracle: outliers are rare rounding flips
        assert float(np.quantile(np.abs(res["A_raw"] - ref["A_raw"]), 0.99)) <= 2e-4, tag
    if res.get("M") is not None:
        np.testing.assert_allclose(res["M"], ref["M"], rtol=0, atol=a_tol, err_msg=tag)
    for k, g in ref["grads"].items():
        if k.endswith(ZERO_GRADS):
            assert float(np.abs(res["grads"][k]).max()) <= 1e-4, (tag, k)
            continue
        err, nrm = float(np.linalg.norm(res["grads"][k] - g)), float(np.linalg.norm(g))
        assert err <= g_rel * nrm + 1e-6, f"{tag} grad {k}: |err| {err:.3e} vs |g| {nrm:.3e}"


def _oracle(m, x_q):
    sd, _, masks = cases.path_inputs(m)
    return bf16_port.path_step_bf16(sd, x_q, m["y"], m["c"], m["alpha"], gated=m["gated"], dropout=m["dropout"],
                                    masks=masks)


def _xq(m):
    _, x, _ = cases.path_inputs(m)
    return bf16_port.rb(bf16_port._t(x)).numpy()


def test_bf16_path_golden_cases(golden, monkeypatch):
    """Every small path fixture (gated/ungated, small/big, eval/train masks, ragged N) in bf16 storage."""
    g = golden("path")
    n = 0
    for name, m in g.meta.items():
        if m["N"] > 2000:
            continue
        xq = _xq(m)
        res = run_path_hip_bf16(m, monkeypatch, xq)
        compare_bf16(res, _oracle(m, xq), name + "/bf16-oracle", a_tol=5e-3, h_tol=2e-3, l_tol=1e-3, g_rel=1e-2)
        compare_bf16(res, cases.run_path(m), name + "/fp64-reference", a_tol=3e-2, h_tol=1e-2, l_tol=3e-2, g_rel=0.15)
        n += 1
    assert n >= 8


@pytest.mark.parametrize("N", [4099, 20000])
def test_bf16_path_mid_sizes(monkeypatch, N):
    """Sizes that use several row tiles and several K splits (ragged last tile and last split)."""
    m = dict(seed=3, gated=True, size="small", K=4, dropout=True, bias_std=0.02, x_seed=77, N=N, train=True,
             mask_seed=4242, y=1, c=0, alpha=0.0)
    xq = _xq(m)
    res = run_path_hip_bf16(m, monkeypatch, xq)
    compare_bf16(res, _oracle(m, xq), f"N={N}", a_tol=5e-3, h_tol=2e-3, l_tol=1e-3, g_rel=1e-2)


def test_bf16_100k_properties():
    """BASELINE config-5 size (100k x 1024, bf16): size-independent properties of the path."""
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    torch.manual_seed(5)
    N = 100_000
    model = MIL_Attention_fc_surv_path(gate_path=True, model_size_wsi="small", dropout=False, n_classes=4).to(DEV).eval()
    x = torch.randn(N, 1024, device=DEV).to(torch.bfloat16)
    with torch.no_grad():
        A = model(path_features=x, attention_only=True)
        M = model(path_features=x, return_features=True)
        perm = torch.randperm(N, device=DEV)
        A_p = model(path_features=x[perm].contiguous(), attention_only=True)
        M_p = model(path_features=x[perm].contiguous(), return_features=True)
        # instance scores do not depend on the other instances: bit-identical under a permutation of the bag
        assert torch.equal(A_p[0], A[0][perm])
        # the pooled embedding is permutation invariant up to fp32 summation order
        assert float((M_p - M).abs().max()) <= 1e-4
        # duplicating the bag leaves softmax pooling unchanged
        M2 = model(path_features=torch.cat([x[:50_000], x[:50_000]]).contiguous(), return_features=True)
        M1 = model(path_features=x[:50_000].contiguous(), return_features=True)
        assert float((M2 - M1).abs().max()) <= 1e-4
    # gradients: finite, and db1 equals the column sums of du implied by dW1 on a constant-1 probe column
    model.train()
    hz, S, Yh, A_raw = model(path_features=x)
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    loss = NLLSurvLoss(alpha=0.0)(hazards=hz, S=S, Y=torch.tensor([1], device=DEV), c=torch.tensor([0.0], device=DEV))
    loss.backward()
    for k, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    assert torch.isfinite(loss)


def test_bf16_constant_column_gives_bias_gradient(monkeypatch):
    """dW1[:, j] for a feature column that is identically 1 must equal db1 (both are column sums of du)."""
    m = dict(seed=9, gated=True, size="small", K=4, dropout=False, bias_std=0.02, x_seed=5, N=3000, train=False,
             mask_seed=0, y=2, c=0, alpha=0.0)
    xq = _xq(m)
    xq[:, 17] = 1.0
    res = run_path_hip_bf16(m, monkeypatch, xq)
    dW1, db1 = res["grads"]["attention_net_WSI.0.weight"], res["grads"]["attention_net_WSI.0.bias"]
    np.testing.assert_allclose(dW1[:, 17], db1, rtol=1e-4, atol=1e-7)


def test_mm_with_bf16_path_bag_tracks_the_fp32_run():
    """BASELINE config 5 shape: the multimodal head with the path bag in bf16 storage (radio / omic stay fp32).
    Same weights and inputs through the fp32 kernels and through the bf16 path kernels: the difference is bf16
    quantisation of the path branch only."""
    from multimodalfusion_amd.models import MM_MIL_Attention_fc_surv
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    from test_gpu_path import _t
    m = dict(fusion="concat", mode="radio_path_omic", Np=3000, nr=64, G=80, gate_path=True, gate_radio=True, K=4,
             seed=77, x_seed=78, y=2, c=0, alpha=0.0, bias_std=0.02)
    sd, xs, xp, xo = cases.mm_inputs(m)
    out = {}
    for tag in ("f32", "bf16"):
        model = _load(MM_MIL_Attention_fc_surv(input_dim=80, radio_fusion="concat", fusion="concat", gate=True,
                                               gate_path=True, gate_omic=True, gate_radio=True, n_classes=4,
                                               mode=m["mode"]), sd).eval()
        kw = {k: _t(x) for k, x in zip(cases.MODS, xs)}
        xpq = torch.as_tensor(xp).to(torch.bfloat16)
        kw["path_features"] = xpq.to(DEV) if tag == "bf16" else xpq.to(torch.float32).to(DEV)
        kw["genomic_features"] = _t(xo)
        hz, S, Yh, A_raw = model(**kw)
        loss = NLLSurvLoss(alpha=0.0)(hazards=hz, S=S, Y=torch.tensor([m["y"]], device=DEV), c=torch.tensor([0.0], device=DEV))
        loss.backward()
        out[tag] = dict(hz=hz.detach().cpu().numpy(), A={k: v.detach().cpu().numpy() for k, v in A_raw.items()},
                        loss=float(loss.detach()), grads=_grads(model))
    a, b = out["f32"], out["bf16"]
    np.testing.assert_allclose(b["hz"], a["hz"], rtol=0, atol=1e-2)
    assert abs(b["loss"] - a["loss"]) <= 3e-2
    np.testing.assert_allclose(b["A"]["pathology"], a["A"]["pathology"], rtol=0, atol=3e-2)
    np.testing.assert_allclose(b["A"]["radiology"], a["A"]["radiology"], rtol=0, atol=1e-5)   # radio branch is untouched
    for k, g in a["grads"].items():
        if k.endswith(ZERO_GRADS):
            continue
        err, nrm = float(np.linalg.norm(b["grads"][k] - g)), float(np.linalg.norm(g))
        assert err <= 0.15 * nrm + 1e-6, (k, err, nrm)


@pytest.mark.parametrize("mode", ["eval", "train", "train_attention_dropout"])
@pytest.mark.parametrize("N", [100_000, 33_333])
def test_bf16_step_is_bit_reproducible(N, mode, monkeypatch):
    """Two 4-wave workgroups share a CU in the fused forward and in K-dh (second forms): any cross-wave race or missed hazard
    shows as a handful of differing elements in a few tiles of a few launches (round 3 found one that way: packed-fp32
    instructions in the fused forward, tools/f2_debug.py).  Scores and every gradient must be bit-identical over repeated
    forward + backward passes -- in eval mode (the <false> instantiations of the two kernels) and in train mode with a pinned
    dropout seed, without and with attention dropout: the <true> instantiations that training and bench.py run (keep-bits
    hashed inside the fused forward's main loop; K-dh's attention-dropout variant), alone and with two bags in flight."""
    from multimodalfusion_amd import ops
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    torch.manual_seed(5)
    model = MIL_Attention_fc_surv_path(gate_path=True, model_size_wsi="small", dropout=(mode == "train_attention_dropout"),
                                       n_classes=4).to(DEV)
    if mode == "eval":
        model.eval()
    else:
        model.train()
        monkeypatch.setattr(ops, "next_dropout_seed", lambda: 4242)      # the same masks in every pass
    x = torch.randn(N, 1024, device=DEV).to(torch.bfloat16)
    Y, c = torch.tensor([1], device=DEV), torch.tensor([0.0], device=DEV)

    def step():
        for p in model.parameters():
            p.grad = None
        hz, S, Yh, A = model(path_features=x)
        NLLSurvLoss(alpha=0.0)(hazards=hz, S=S, Y=Y, c=c).backward()
        torch.cuda.synchronize()
        return [A.detach().clone()] + [p.grad.clone() for p in model.parameters()]

    ref = step()
    for i in range(10):
        cur = step()
        for k, (a, b) in enumerate(zip(cur, ref)):
            assert torch.equal(a, b), (i, k, int((a != b).sum()))
    if mode == "train_attention_dropout":
        return
    # the same bag on two streams at once (a third and fourth workgroup compete for every CU): each stream's result must
    # still be the single-stream result, bit for bit
    streams = [torch.cuda.Stream(DEV) for _ in range(2)]
    models = [model, MIL_Attention_fc_surv_path(gate_path=True, model_size_wsi="small", dropout=False, n_classes=4).to(DEV)]
    models[1].load_state_dict(model.state_dict())
    models[1].train(model.training)
    for i in range(4):
        outs = []
        for st, mdl in zip(streams, models):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                for p in mdl.parameters():
                    p.grad = None
                hz, S, Yh, A = mdl(path_features=x)
                NLLSurvLoss(alpha=0.0)(hazards=hz, S=S, Y=Y, c=c).backward()
                outs.append((A, mdl))
        torch.cuda.synchronize()
        for A, mdl in outs:
            cur = [A.detach()] + [p.grad for p in mdl.parameters()]
            for k, (a, b) in enumerate(zip(cur, ref)):
                assert torch.equal(a, b), ("two streams", i, k, int((a != b).sum()))

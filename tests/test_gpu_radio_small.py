"""GPU parity: radiology head (cat + reduce_dim on MFMA without materialising the concat), the survival
head, nll_surv and Cox kernels -- against the golden fixtures and the live fp64 oracle."""
import numpy as np
import pytest
import torch

from conftest import check_summary
from oracle import cases
from oracle import inputs as gen
from oracle import torch_port as tp
from test_gpu_path import relu_kink_units, DEV, _grads, _load, _t, compare

pytestmark = pytest.mark.gpu


def run_radio_hip(m, monkeypatch):
    from multimodalfusion_amd import ops
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_radio
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    sd, xs, masks = cases.radio_inputs(m)
    model = _load(MIL_Attention_fc_surv_radio(radio_fusion="concat", gate_radio=m["gated"], dropout=m["dropout"],
                                              n_classes=m["K"], modalities=cases.MODS[:m["n_mod"]]), sd)
    if m["train"]:
        model.train()
        monkeypatch.setattr(ops, "next_dropout_seed", lambda: m["mask_seed"])
    else:
        model.eval()
    kw = {k: _t(x) for k, x in zip(cases.MODS, xs)}
    hz, S, Yh, A_raw = model(**kw)
    loss = NLLSurvLoss(alpha=m["alpha"])(hazards=hz, S=S, Y=torch.tensor([m["y"]], device=DEV),
                                         c=torch.tensor([float(m["c"])], device=DEV))
    loss.backward()
    M = None
    if not m["train"]:
        with torch.no_grad():
            M = model(return_features=True, **kw).cpu().numpy()
    return dict(hazards=hz.detach().cpu().numpy(), S=S.detach().cpu().numpy(), Y_hat=Yh.cpu().numpy(),
                A_raw=A_raw.detach().cpu().numpy(), loss=float(loss), M=M, grads=_grads(model))


def test_radio_golden_cases(golden, monkeypatch):
    g = golden("radio")
    for name, m in g.meta.items():
        res = run_radio_hip(m, monkeypatch)
        compare(res, cases.run_radio(m), name)
        tag = name + "/f64"
        assert abs(res["loss"] - float(g[tag + "/loss"])) <= 1e-5
        np.testing.assert_allclose(res["hazards"], g[tag + "/hazards"], rtol=0, atol=1e-4)
        check_summary(g, tag + "/A_raw", res["A_raw"], rtol=0, atol=1e-4)
        for k, gr in res["grads"].items():
            check_summary(g, f"{tag}/grad/{k}", gr, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,n_mod,train", [(16389, 4, True), (16389, 2, False)])
def test_radio_large_bags(n, n_mod, train, monkeypatch):
    """Large multi-modality bags: the projection runs its wide tiles over a K range that is the concatenation of
    separate modality buffers (segment switch inside the staged loads), the backward its large-bag kernels plus the
    plain NN GEMM for d(reduce_dim out); ragged last tile; against the live fp64 oracle."""
    m = dict(n=n, n_mod=n_mod, gated=True, K=4, dropout=False, y=2, c=0, alpha=0.1, bias_std=0.05, train=train,
             seed=4300 + n_mod, x_seed=5300 + n_mod, mask_seed=99)
    sd, xs, _ = cases.radio_inputs(m)
    # the stack's input is itself a 4096-term fp32 contraction here (error ~1e-5), hence the wider kink threshold
    h0 = np.concatenate([np.asarray(x, np.float64) for x in xs], axis=1) @ np.asarray(sd["reduce_dim.weight"], np.float64).T \
        + np.asarray(sd["reduce_dim.bias"], np.float64)
    compare(run_radio_hip(m, monkeypatch), cases.run_radio(m), f"radio n={n} n_mod={n_mod} train={train}",
            kink_units=relu_kink_units(sd, h0, "attention_net_radio", thr=4e-5), kink_prefix="attention_net_radio")


@pytest.mark.parametrize("act", ["none", "relu", "tanh", "sigmoid", "selu"])
@pytest.mark.parametrize("drop_p", [0.0, 0.25])
def test_linear_forward_abi_all_activations_large(act, drop_p):
    """mmf_linear_forward through the C ABI at a size that takes the wide projection tiles, every activation the ABI
    names (the kernel compiles ReLU / none in and decides the others per element) with and without dropout, ragged
    row count; against float64 numpy with the oracle's mask."""
    import ctypes as C
    from multimodalfusion_amd import _lib
    l = _lib.lib()
    M, K, N, seed, site = 17003, 64, 256, 321, 1
    x = gen.normal(21, (M, K), stream=1)
    W = gen.normal(22, (N, K), stream=2, std=0.2)
    b = gen.normal(23, (N,), stream=3, std=0.3)
    tx, tW, tb = _t(x), _t(W), _t(b)
    y = torch.empty((M, N), dtype=torch.float32, device=DEV)
    segs = (C.c_void_p * 1)(tx.data_ptr())
    code = {"none": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "selu": 4}[act]
    rc = l.mmf_linear_forward(segs, 1, K, M, C.c_void_p(tW.data_ptr()), C.c_void_p(tb.data_ptr()), N, code,
                              C.c_float(drop_p), seed, site, None, C.c_void_p(y.data_ptr()), None, 0, None, 0,
                              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    pre = x.astype(np.float64) @ W.astype(np.float64).T + b.astype(np.float64)
    ref = {"none": lambda v: v, "relu": lambda v: np.maximum(v, 0), "tanh": np.tanh,
           "sigmoid": lambda v: 1 / (1 + np.exp(-v)),
           "selu": lambda v: 1.0507009873554805 * np.where(v > 0, v, 1.6732632423543772 * (np.exp(v) - 1))}[act](pre)
    if drop_p > 0:
        ref = np.where(gen.keep_mask(seed, site, M, N, drop_p), ref / (1 - drop_p), 0.0)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("B,K", [(1, 4), (1, 8), (5, 4), (16, 8)])
def test_surv_head_and_nll(B, K):
    from multimodalfusion_amd import ops
    F = 256
    feat = gen.normal(11, (B, F), stream=B)
    Wk = gen.normal(12, (K, F), stream=K, std=0.2)
    bk = gen.normal(13, (K,), stream=1, std=0.3)
    Y = (np.arange(B) * 3) % K
    c = (np.arange(B) % 2).astype(np.float32)
    for alpha in (0.0, 0.4):
        tf, tW, tb = (_t(a).requires_grad_(True) for a in (feat, Wk, bk))
        hz, S, Yh = ops.surv_head(tf, tW, tb)
        loss = ops.nll_surv(hz, S, _t(Y, torch.int64), _t(c), alpha=alpha)
        loss.backward()
        rf, rW, rb = (torch.as_tensor(a).double().requires_grad_(True) for a in (feat, Wk, bk))
        rhz, rS, rYh = tp.surv_head(torch.nn.functional.linear(rf, rW, rb))
        rloss = tp.nll_loss(rhz, rS, torch.as_tensor(Y), torch.as_tensor(c), alpha=alpha)
        rloss.backward()
        assert abs(float(loss) - float(rloss)) < 1e-5
        np.testing.assert_allclose(hz.detach().cpu().numpy(), rhz.detach().numpy(), atol=1e-5)
        np.testing.assert_allclose(S.detach().cpu().numpy(), rS.detach().numpy(), atol=1e-5)
        assert np.array_equal(Yh.cpu().numpy(), rYh.numpy())
        for a, b in ((tf, rf), (tW, rW), (tb, rb)):
            np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), atol=1e-5, rtol=1e-4)


def test_nll_clamp_edges():
    """hazards / S below eps: the clamp(min=eps) branch has zero gradient (torch semantics)."""
    from multimodalfusion_amd import ops
    hz = np.array([[1e-9, 0.5, 0.3, 0.9]], np.float32)
    S = np.array([[1.0 - 1e-9, 1e-9, 1e-10, 0.0]], np.float32)
    for y, c in ((0, 0.0), (1, 1.0), (2, 0.0), (3, 1.0)):
        th, tS = _t(hz).requires_grad_(True), _t(S).requires_grad_(True)
        loss = ops.nll_surv(th, tS, _t([y], torch.int64), _t([c]), alpha=0.2)
        loss.backward()
        rh, rS = torch.as_tensor(hz).double().requires_grad_(True), torch.as_tensor(S).double().requires_grad_(True)
        rl = tp.nll_loss(rh, rS, torch.tensor([y]), torch.tensor([c]), alpha=0.2)
        rl.backward()
        assert abs(float(loss) - float(rl)) < 1e-4 * max(1.0, abs(float(rl)))
        np.testing.assert_allclose(th.grad.cpu().numpy(), rh.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(tS.grad.cpu().numpy(), rS.grad.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B", [1, 2, 128, 300])
def test_cox(B):
    from multimodalfusion_amd import ops
    from oracle.gen_golden import omic_batch
    _, t, c = omic_batch(77 + B, B, 4, ties=True)
    risks = gen.normal(5, (B,), stream=B, std=0.7)
    tr = _t(risks).requires_grad_(True)
    loss = ops.cox_surv(tr, torch.as_tensor(t), _t(c))
    loss.backward()
    rr = torch.as_tensor(risks).double().requires_grad_(True)
    rl = tp.cox_loss(rr, t, torch.as_tensor(c))
    rl.backward()
    assert abs(float(loss) - float(rl)) < 1e-5
    np.testing.assert_allclose(tr.grad.cpu().numpy(), rr.grad.numpy(), atol=1e-6, rtol=1e-4)


def test_linear_cat_matches_torch():
    """cat + Linear on MFMA vs torch fp64, including a non-tile-multiple row count."""
    from multimodalfusion_amd import ops
    for M, nseg in ((77, 4), (512, 4), (300, 2), (1, 1)):
        xs = [gen.normal(3, (M, 1024), stream=i) for i in range(nseg)]
        W = gen.normal(4, (1024, 1024 * nseg), stream=9, std=0.02)
        b = gen.normal(4, (1024,), stream=10, std=0.1)
        tW, tb = _t(W).requires_grad_(True), _t(b).requires_grad_(True)
        y = ops.linear_cat([_t(x) for x in xs], tW, tb)
        gy = gen.normal(6, (M, 1024), stream=1)
        y.backward(_t(gy))
        rW, rb = torch.as_tensor(W).double().requires_grad_(True), torch.as_tensor(b).double().requires_grad_(True)
        ry = torch.nn.functional.linear(torch.cat([torch.as_tensor(x).double() for x in xs], 1), rW, rb)
        ry.backward(torch.as_tensor(gy).double())
        np.testing.assert_allclose(y.detach().cpu().numpy(), ry.detach().numpy(), atol=2e-5, rtol=1e-5)
        sc = float(rW.grad.abs().max())
        np.testing.assert_allclose(tW.grad.cpu().numpy(), rW.grad.numpy(), atol=1e-5 * max(sc, 1), rtol=1e-4)
        np.testing.assert_allclose(tb.grad.cpu().numpy(), rb.grad.numpy(), atol=1e-4, rtol=1e-4)

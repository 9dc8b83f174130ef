"""GPU: the forward-only consumers (multimodalfusion_amd/infer.py) on the mmf_amil[_bf16]_infer entry points."""
import numpy as np
import pytest
import torch

from oracle import cases
from oracle import inputs as gen
from oracle import torch_port as tp
from test_gpu_path import DEV, _load, _t

pytestmark = pytest.mark.gpu


def _path_model(gated=True, size="small", K=4, seed=3):
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    sd = gen.path_state_dict(seed=seed, gated=gated, size=size, n_classes=K, dropout=False, bias_std=0.02)
    return _load(MIL_Attention_fc_surv_path(gate_path=gated, model_size_wsi=size, dropout=False, n_classes=K), sd).eval(), sd


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N", [1, 700, 20000])
def test_no_grad_forward_is_bit_identical_to_the_training_forward(dtype, N):
    """Same kernels, minus the stores of what only a backward needs."""
    model, _ = _path_model()
    x = torch.as_tensor(gen.bag(11, N)).to(dtype).to(DEV)
    hz, S, Yh, A = model(path_features=x)                      # autograd on: mmf_amil_forward
    with torch.no_grad():
        hz2, S2, Yh2, A2 = model(path_features=x)              # mmf_amil_infer
        M2 = model(path_features=x, return_features=True)
    M = model(path_features=x, return_features=True)
    assert torch.equal(hz, hz2) and torch.equal(S, S2) and torch.equal(Yh, Yh2) and torch.equal(A, A2) and torch.equal(M, M2)


def test_infer_workspace_is_the_small_one():
    from multimodalfusion_amd._lib import lib
    l = lib()
    full, small = l.mmf_amil_workspace_bytes(50000, 1024, 256, 256, 1), l.mmf_amil_infer_workspace_bytes(50000, 1024, 256, 256, 1)
    assert small < full / 4
    assert l.mmf_amil_bf16_infer_workspace_bytes(50000, 1024, 256, 256, 1) < l.mmf_amil_bf16_workspace_bytes(50000, 1024, 256, 256, 1) / 4


def test_infer_patient_matches_the_oracle():
    from multimodalfusion_amd.infer import infer_patient
    model, sd = _path_model(K=4, seed=5)
    x = gen.bag(21, 1500)
    Y_hat_model, risk, A = infer_patient(model, torch.as_tensor(x), bins=[-3.0, -2.0, -1.0])
    hz, S, Yh, A_raw, M = tp.path_forward(tp.to_torch(sd, torch.float64, False), torch.as_tensor(x).double(), True, False, None)
    np.testing.assert_allclose(risk, -S.sum(dim=1).numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(A, A_raw.numpy().reshape(-1, 1), rtol=0, atol=1e-4)
    assert A.shape == (1500, 1) and int(Y_hat_model) == int(Yh.numpy()[0][0])


def test_infer_patient_radio_head():
    from multimodalfusion_amd.infer import infer_patient
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_radio
    sd = gen.radio_state_dict(seed=2, gated=True, n_classes=4, dropout=True, n_mod=4, bias_std=0.02)
    model = _load(MIL_Attention_fc_surv_radio(n_classes=4), sd).eval()
    xs = {m: torch.as_tensor(gen.bag(30 + i, 96)) for i, m in enumerate(cases.MODS)}
    Y_hat_model, risk, A = infer_patient(model, xs)
    hz, S, Yh, A_raw, M = tp.radio_forward(tp.to_torch(sd, torch.float64, False), [v.double() for v in xs.values()], True, True, None)
    np.testing.assert_allclose(risk, -S.sum(dim=1).numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(A, A_raw.numpy().reshape(-1, 1), rtol=0, atol=1e-4)


def test_patch_batches_score_like_the_whole_bag_and_percentiles_match_scipy():
    """An instance's raw score does not depend on the rest of its bag, so scoring 512-patch batches
    (heatmap_utils.py:129-141) must give exactly the scores of the concatenated bag."""
    from scipy.stats import percentileofscore
    from multimodalfusion_amd.infer import score_patch_batches
    model, _ = _path_model(seed=8)
    x = torch.as_tensor(gen.bag(41, 512 * 3 + 77))
    batches = [x[i:i + 512] for i in range(0, x.shape[0], 512)]
    got = np.concatenate(list(score_patch_batches(model, batches)), axis=0)
    with torch.no_grad():
        whole = model(path_features=x.to(DEV), attention_only=True).view(-1, 1).cpu().numpy()
    assert got.shape == whole.shape and np.array_equal(got, whole)
    ref = np.random.RandomState(0).normal(size=300).astype(np.float32)
    ref[:10] = whole[:10, 0]                                     # ties with the scored values
    pct = np.concatenate(list(score_patch_batches(model, batches[:1], ref_scores=ref)), axis=0)
    want = np.array([percentileofscore(ref, s) for s in whole[:512, 0]]).reshape(-1, 1)
    np.testing.assert_allclose(pct, want, rtol=0, atol=1e-4)


def test_feature_export_loop_skips_missing_modalities():
    from multimodalfusion_amd.infer import extract_features_for_subjects
    from multimodalfusion_amd.models import MaxNet
    model, sd = _path_model(seed=4)
    omic_sd = gen.maxnet_state_dict(seed=6, input_dim=36, nll=True, n_classes=4, bias_std=0.02)
    omic = _load(MaxNet(input_dim=36, model_size_omic="small", bag_loss="nll_surv", n_classes=4), omic_sd).eval()
    missing = torch.zeros((1, 1))
    subjects = [("s1", {}, torch.as_tensor(gen.bag(1, 300)), torch.as_tensor(gen.normal(2, (1, 36)))),
                ("s2", {}, missing, torch.as_tensor(gen.normal(3, (1, 36)))),
                ("s3", {}, torch.as_tensor(gen.bag(4, 50)), missing)]
    out = list(extract_features_for_subjects({"path": model, "omic": omic}, subjects))
    assert [(s, m) for s, m, _ in out] == [("s1", "path"), ("s1", "omic"), ("s2", "omic"), ("s3", "path")]
    for _, m, f in out:
        assert f.device.type == "cpu" and f.shape[-1] == 256 and bool(torch.isfinite(f).all())
    # the path embedding equals the oracle's pooled M
    hz, S, Yh, A_raw, M = tp.path_forward(tp.to_torch(sd, torch.float64, False), torch.as_tensor(gen.bag(1, 300)).double(), True, False, None)
    np.testing.assert_allclose(out[0][2].numpy(), M.numpy(), rtol=0, atol=1e-4)


def test_prefetcher_delivers_bf16_bags():
    """feed.DevicePrefetcher(path_dtype=bf16): fp32 host bags are narrowed on the device, bf16 host bags cross as they
    are; either way the head sees exactly x.to(bf16) and takes the bf16-storage kernels."""
    from multimodalfusion_amd.feed import DevicePrefetcher
    model, _ = _path_model(seed=12)
    x = torch.as_tensor(gen.bag(51, 900))
    with torch.no_grad():
        want = model(path_features=x.to(torch.bfloat16).to(DEV), attention_only=True)
        for host in (x, x.to(torch.bfloat16)):
            batches = [({}, host, torch.zeros(1, 1), torch.tensor([1]), None, torch.tensor([0.0]))] * 2
            for b in DevicePrefetcher(batches, DEV, depth=2, path_dtype=torch.bfloat16):
                assert b[1].dtype == torch.bfloat16 and b[1].is_cuda
                assert torch.equal(model(path_features=b[1], attention_only=True), want)

"""GPU parity of mmf_amil_desc::gemm = MMF_GEMM_BF16X3 (csrc/mmf_gemm_split.h): the stack's four large contractions on
the bf16 matrix cores with every fp32 operand split into three bf16 values (six products, fp32 accumulation).

Two bars:
  * the same bars as the exact-fp32 path (tests/test_gpu_path.py) against the live fp64 oracle, on bags that take the
    split tiles (wide row tiles + 256x256 split-K tile from 20,480 instances, 64-row tiles + 128x128 split-K tile
    below), ragged, with and without the dropout sites;
  * the claim the mode rests on -- fp32-equivalent accuracy: its error against the fp64 oracle is compared with the
    exact-fp32 path's error on the same inputs, output by output (it must not exceed 2 x that error + rounding noise).
"""
import numpy as np
import pytest
import torch

from oracle import cases
from test_gpu_path import DEV, _load, _t, compare, relu_kink_units, run_path_hip

pytestmark = pytest.mark.gpu


@pytest.fixture
def split_mode():
    from multimodalfusion_amd import ops
    prev = ops.set_gemm(1)
    yield
    ops.set_gemm(prev)


def _case(N, gated=True, dropout=False, train=True, size="small"):
    return dict(N=N, gated=gated, size=size, K=4, dropout=dropout, y=N % 4, c=N % 2, alpha=0.1, bias_std=0.05,
                train=train, seed=4200 + N, x_seed=5200 + N, mask_seed=4321)


@pytest.mark.parametrize("N,gated,dropout,train,size", [
    (16421, True, False, True, "small"),     # smallest bags on the split tiles, ragged last tile, one dropout site
    (23333, True, True, True, "small"),      # + attention dropout: the (gated, dropout) instantiations of K-dh and K-tn
    (20011, True, False, False, "small"),    # eval mode
    (17011, True, False, True, "big"),       # 1024 / 512 / 384: two column tiles per row tile in K-dh, three gate tiles in K-tn
    (77, True, True, True, "small"),         # two ragged 64-row tiles
    (1100, True, True, True, "small"),       # 18 row tiles of 64, ragged
    (4099, True, True, True, "small"),       # below the wide tiles: 64-row split GEMM tiles, 128 x 128 split TN tile, fused K-prep
    (10000, True, False, True, "small"),     # BASELINE's 10k bag: K-prep as its own launch beside the 64-row split K-dh
    (5003, False, True, True, "big"),        # small tiles, ungated, big model: 8 column tiles in K-dh, 64-dim TN gate tiles
    (16999, False, False, True, "small"),    # Attn_Net (ungated: the reference's CLI default): one-part K-dh ring, 256-dim TN gate tiles
    (19999, False, True, True, "small"),     # ungated + attention dropout
])
def test_split_against_fp64_oracle(N, gated, dropout, train, size, split_mode, monkeypatch):
    m = _case(N, gated=gated, dropout=dropout, train=train, size=size)
    sd, x, _ = cases.path_inputs(m)
    compare(run_path_hip(m, monkeypatch), cases.run_path(m), f"bf16x3 N={N} gated={gated} dropout={dropout} train={train} {size}",
            kink_units=relu_kink_units(sd, x))


def test_split_config4_mm_with_50k_path_bag(split_mode):
    """BASELINE config 4 at its full size in this mode: the multimodal head (concat fusion) with a 50,000 x 1024 fp32 path
    bag (the path stack on the split kernels; the 512-instance radio bags stay on the exact-fp32 small tiles) against the
    fp64 autograd oracle."""
    from test_gpu_omic_mm import run_mm_hip
    m = dict(fusion="concat", mode="radio_path_omic", Np=50_000, nr=512, G=80, gate_path=True, gate_radio=True, K=4,
             seed=404, x_seed=405, y=1, c=0, alpha=0.0, bias_std=0.02)
    sd, xs, xp, xo = cases.mm_inputs(m)
    compare(run_mm_hip(m), cases.run_mm(m), "bf16x3 config4 concat", kink_units=relu_kink_units(sd, xp))


def test_split_scores_do_not_depend_on_the_bag_size(split_mode):
    """Every bag size takes the split tiles (split_min_rows() = 1), so the heat-map property holds in this mode too, bit
    for bit: 512-patch batches score exactly like the concatenated bag (tests/test_gpu_infer.py)."""
    import test_gpu_infer as I
    I.test_patch_batches_score_like_the_whole_bag_and_percentiles_match_scipy()


def test_split_kernels_are_the_ones_that_ran(split_mode, monkeypatch):
    """The mode must not silently fall back: the kernel trace of a step on a large gated bag names the split kernels."""
    from multimodalfusion_amd._lib import KernelTrace
    m = _case(16421)
    with KernelTrace() as tr:
        run_path_hip(m, monkeypatch)
    names = set(tr.dump().keys())
    assert {"linear_nt_split_kernel", "gate_fwd_split_kernel", "bwd_dh_split_kernel", "tn_split_kernel"} <= names, names
    assert not {"linear_nt_kernel", "gate_fwd_kernel", "bwd_dh_kernel", "tn_kernel"} & names, names


def test_split_error_matches_exact_fp32(monkeypatch):
    """fp32-equivalent accuracy, measured: per output, |bf16x3 - fp64| <= 2 |fp32 MFMA - fp64| + a rounding floor."""
    from multimodalfusion_amd import ops
    m = _case(32768 + 77, dropout=True)
    ref = cases.run_path(m)
    sd, x, _ = cases.path_inputs(m)
    kink = relu_kink_units(sd, x)
    res = {}
    for mode in (0, 1):
        prev = ops.set_gemm(mode)
        try:
            res[mode] = run_path_hip(m, monkeypatch)
        finally:
            ops.set_gemm(prev)

    def err(r, key):
        return float(np.abs(np.asarray(r[key], np.float64) - np.asarray(ref[key], np.float64)).max())

    report = {}
    for key in ("A_raw", "hazards", "S"):
        e0, e1 = err(res[0], key), err(res[1], key)
        report[key] = (e0, e1)
        assert e1 <= 2 * e0 + 2e-7 * max(1.0, float(np.abs(ref[key]).max())), (key, e0, e1)
    for k, g in ref["grads"].items():
        g = np.asarray(g, np.float64)
        d0 = np.abs(res[0]["grads"][k] - g)
        d1 = np.abs(res[1]["grads"][k] - g)
        if k in ("attention_net_WSI.0.weight", "attention_net_WSI.0.bias"):   # rows on the ReLU kink differ legitimately (see compare)
            keep = np.ones(g.shape[0], bool)
            keep[list(kink)] = False
            d0, d1 = d0[keep], d1[keep]
        e0, e1 = float(d0.max()), float(d1.max())
        report[k] = (e0, e1)
        assert e1 <= 2 * e0 + 2e-7 * max(float(np.abs(g).max()), 1e-30), (k, e0, e1)
    print("max abs error vs fp64 (exact fp32, bf16x3):", {k: (f"{a:.2e}", f"{b:.2e}") for k, (a, b) in report.items()})


@pytest.mark.parametrize("x_kind", ["relu", "lognormal"])
def test_split_error_on_non_gaussian_bags(x_kind, monkeypatch):
    """The accuracy claim beyond N(0, 1) bags: non-negative, half-sparse features (post-ReLU ResNet pooling) and
    log-normal magnitudes spanning 1e-6 .. 1e4 in one bag.  Output by output the split mode's error against the fp64 oracle
    stays within 4 x the exact-fp32 path's or within 1e-5 of the tensor's max -- a tenth of the parity bar `compare` holds
    both modes to -- whichever is larger.  Measured on the non-negative bag: gate weight gradient 8.7e-9 against 2.8e-9 on a
    tensor of max 9.4e-3 (9e-7 relative); its bias gradient, a 20,813-term column sum taken in a different order by the two
    loaders, 5.7e-9 against 5.4e-10 on a max of 1.5e-3 (3.7e-6 relative, ordinary fp32 summation noise).  Same-sign data
    leaves the exact path's accumulation unusually accurate; the Gaussian bag stays within 2 x."""
    from multimodalfusion_amd import ops
    m = dict(_case(20480 + 333, dropout=False), x_kind=x_kind)
    ref = cases.run_path(m)
    sd, x, _ = cases.path_inputs(m)
    kink = relu_kink_units(sd, x, thr=4e-6 * max(1.0, float(np.abs(x).max())))
    res = {}
    for mode in (0, 1):
        prev = ops.set_gemm(mode)
        try:
            res[mode] = run_path_hip(m, monkeypatch)
        finally:
            ops.set_gemm(prev)
    err = lambda r, key: float(np.abs(np.asarray(r[key], np.float64) - np.asarray(ref[key], np.float64)).max())
    for key in ("A_raw", "hazards", "S"):
        e0, e1 = err(res[0], key), err(res[1], key)
        assert np.isfinite(res[1][key]).all()
        assert e1 <= 4 * e0 + 4e-7 * max(1.0, float(np.abs(ref[key]).max())), (x_kind, key, e0, e1)
    if x_kind == "relu":          # the suite's absolute bars assume O(1) activations: scores of the log-normal bag reach 1e3
        compare(res[1], ref, f"bf16x3 {x_kind}", kink_units=kink)
    worst = (0.0, None)
    for k, g in ref["grads"].items():
        g = np.asarray(g, np.float64)
        d0, d1 = np.abs(res[0]["grads"][k] - g), np.abs(res[1]["grads"][k] - g)
        if k in ("attention_net_WSI.0.weight", "attention_net_WSI.0.bias"):
            keep = np.ones(g.shape[0], bool)
            keep[list(kink)] = False
            d0, d1 = d0[keep], d1[keep]
        if d1.size:
            ratio = float(d1.max()) / max(float(d0.max()), 1e-300)
            if ratio > worst[0]:
                worst = (ratio, k)
            # 1e-6 absolute: a tenth of `compare`'s absolute bar (attention_c.bias: the gradient is analytically zero, both
            # arithmetics return rounding noise of the 20,813-term sum of ds)
            assert float(d1.max()) <= max(4 * float(d0.max()), 1e-5 * float(np.abs(g).max()), 1e-6), \
                (x_kind, k, float(d0.max()), float(d1.max()))
    print(f"{x_kind}: worst bf16x3 / exact-fp32 gradient error ratio {worst[0]:.2f} ({worst[1]})")


def test_split_denormals_and_values_beyond_bf16_max_stay_finite(monkeypatch):
    """A bag with fp32 denormals, zeros, and -- in a feature column whose first-layer weights are zero -- values between the
    largest bf16 and FLT_MAX.  The exact-fp32 path is finite there (x * 0 = 0); nearest rounding of the first bf16 plane
    would make it inf * 0 = NaN.  The split clamps that plane (mmf_gemm_split.h: split_pair): every output and gradient must
    be finite and equal to the exact path's within the usual bars."""
    from multimodalfusion_amd import ops
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    m = dict(_case(20480 + 77, dropout=False, train=False), x_kind="edge")
    sd, x, _ = cases.path_inputs(m)
    sd = dict(sd)
    w1 = np.array(sd["attention_net_WSI.0.weight"], np.float32, copy=True)
    w1[:, 0] = 0.0
    sd["attention_net_WSI.0.weight"] = w1
    out = {}
    for mode in (0, 1):
        prev = ops.set_gemm(mode)
        try:
            model = _load(MIL_Attention_fc_surv_path(gate_path=True, n_classes=4), sd).eval()
            hz, S, Yh, A = model(path_features=_t(x))
            loss = NLLSurvLoss(alpha=0.0)(hazards=hz, S=S, Y=torch.tensor([1], device=DEV), c=torch.tensor([0.0], device=DEV))
            loss.backward()
            torch.cuda.synchronize()
            out[mode] = dict(A=A.detach().cpu().numpy(), hz=hz.detach().cpu().numpy(), loss=float(loss),
                             grads={k: p.grad.cpu().numpy() for k, p in model.named_parameters()})
        finally:
            ops.set_gemm(prev)
    for mode in (0, 1):
        assert np.isfinite(out[mode]["A"]).all() and np.isfinite(out[mode]["hz"]).all() and np.isfinite(out[mode]["loss"]), mode
        for k, g in out[mode]["grads"].items():
            if k == "attention_net_WSI.0.weight":
                g = g[:, 1:]                  # column 0 of dW1 = du^T x[:, 0] overflows in BOTH arithmetics (sum of 1e38-sized terms)
            assert np.isfinite(g).all(), (mode, k)
    np.testing.assert_allclose(out[1]["A"], out[0]["A"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(out[1]["hz"], out[0]["hz"], rtol=0, atol=1e-6)
    for k, g in out[0]["grads"].items():
        a, b = out[1]["grads"][k], g
        if k == "attention_net_WSI.0.weight":
            a, b = a[:, 1:], b[:, 1:]
        assert float(np.abs(a - b).max()) <= 1e-6 + 1e-4 * float(np.abs(b).max()), k


def test_split_size_independent_properties_at_full_size(split_mode):
    """The oracle-free checks of tests/test_gpu_path.py in this mode: permutation / shift invariance, the softmax spike,
    and the 50,000 x 1024 BASELINE bag's linearity / finiteness / softmax-mass properties."""
    import test_gpu_path as P
    P.test_permutation_and_shift_invariance()
    P.test_softmax_spike_forces_rescale()
    P.test_full_size_50k_properties()


def test_unknown_gemm_mode_is_refused():
    """mmf_amil_desc::gemm takes 0 or 1; anything else is an argument error, not a silent default."""
    from multimodalfusion_amd import ops
    from multimodalfusion_amd._lib import MmfError
    prev = ops.set_gemm(7)
    try:
        with pytest.raises(MmfError):
            run_path_hip(_case(300, train=False))
    finally:
        ops.set_gemm(prev)

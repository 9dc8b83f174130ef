"""GPU: the training-loop mirror (utils/core_utils.py:173-264 semantics) reproduces the golden 2-step Adam
trajectory captured from the reference modules (gc = 2, reg_type = all, dropout disabled)."""
import numpy as np
import pytest
import torch

from conftest import check_summary
from oracle import inputs as gen

pytestmark = pytest.mark.gpu


def _setup(golden):
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    g = golden("trajectory")
    meta = g.meta
    sd = gen.path_state_dict(seed=meta["seed"], gated=True, size="small", n_classes=meta["K"], bias_std=0.05)
    model = MIL_Attention_fc_surv_path(gate_path=True, model_size_wsi="small", dropout=False, n_classes=meta["K"])
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model.relocate()
    model.eval()
    model.train = lambda mode=True: model      # dropout disabled as in the fixture
    loader = []
    for b in meta["bags"]:
        x = torch.as_tensor(gen.bag(b["x_seed"], b["n"]))
        loader.append(({"T1": torch.zeros(1, 1)}, x, torch.zeros(1, 4), torch.tensor([b["y"]]),
                       np.array([10.0]), torch.tensor([float(b["c"])])))
    return g, meta, sd, model, loader


def test_fused_l1_adam_tail_matches_reference(golden):
    """Row N2: FlatAdam (L1 gradient + Adam + L2 in one kernel over the flat buffer) reproduces the trajectory the
    reference gets with autograd-L1 + torch.optim.Adam, and its L1 value equals lambda * sum|W|."""
    from multimodalfusion_amd.optim import FlatAdam
    from multimodalfusion_amd.utils import core_utils
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    from multimodalfusion_amd.utils.utils import l1_reg_all
    g, meta, sd, model, loader = _setup(golden)
    opt = FlatAdam(model, lr=meta["lr"], weight_decay=meta["reg"], lambda_l1=meta["lambda_reg"])
    l1_ref = meta["lambda_reg"] * sum(float(np.abs(v).sum()) for v in sd.values())
    assert abs(float(opt.l1_value()) - l1_ref) <= 1e-5 * l1_ref
    snaps = []
    step0 = opt.step

    def step(**kw):
        step0(**kw)
        snaps.append({k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()})

    opt.step = step
    out = core_utils.train_loop_survival(0, model, loader, opt, meta["K"], "path", loss_fn=NLLSurvLoss(alpha=0.0),
                                         reg_fn=l1_reg_all, lambda_reg=meta["lambda_reg"], gc=meta["gc"])
    np.testing.assert_allclose(out["losses"], g["f64/losses"], atol=1e-5)
    assert len(snaps) == 2
    for si, snap in enumerate(snaps, start=1):
        for k, v in snap.items():
            check_summary(g, f"f64/step{si}/{k}", v, rtol=2e-5, atol=2e-6)


def test_prefetched_feed_gives_identical_results(golden):
    """Row N1: bags staged through pinned memory on a side stream give bit-identical losses and parameters."""
    from multimodalfusion_amd.feed import DevicePrefetcher
    from multimodalfusion_amd.optim import FlatAdam
    from multimodalfusion_amd.utils import core_utils
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    outs = []
    for use_feed in (False, True):
        g, meta, sd, model, loader = _setup(golden)
        opt = FlatAdam(model, lr=meta["lr"], weight_decay=meta["reg"])
        feed = DevicePrefetcher(loader, depth=2) if use_feed else loader
        out = core_utils.train_loop_survival(0, model, feed, opt, meta["K"], "path", loss_fn=NLLSurvLoss(alpha=0.0),
                                             reg_fn=None, lambda_reg=0.0, gc=meta["gc"])
        outs.append((out["losses"].copy(), opt.flat_w.detach().cpu().numpy().copy()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])


def test_trajectory_matches_reference(golden):
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_path
    from multimodalfusion_amd.utils import core_utils
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    from multimodalfusion_amd.utils.utils import l1_reg_all
    g = golden("trajectory")
    meta = g.meta
    sd = gen.path_state_dict(seed=meta["seed"], gated=True, size="small", n_classes=meta["K"], bias_std=0.05)
    model = MIL_Attention_fc_surv_path(gate_path=True, model_size_wsi="small", dropout=False, n_classes=meta["K"])
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model.relocate()
    # dropout disabled as in the fixture: keep the loop's model.train() from enabling it
    model.eval()
    model.train = lambda mode=True: model
    opt = torch.optim.Adam(model.parameters(), lr=meta["lr"], weight_decay=meta["reg"])
    loader = []
    for b in meta["bags"]:
        x = torch.as_tensor(gen.bag(b["x_seed"], b["n"]))
        loader.append(({"T1": torch.zeros(1, 1)}, x, torch.zeros(1, 4), torch.tensor([b["y"]]),
                       np.array([10.0]), torch.tensor([float(b["c"])])))
    snaps = []

    class Opt:   # records the parameters after every optimizer step
        def step(self):
            opt.step()
            snaps.append({k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()})

        def zero_grad(self):
            opt.zero_grad()

    out = core_utils.train_loop_survival(0, model, loader, Opt(), meta["K"], "path", loss_fn=NLLSurvLoss(alpha=0.0),
                                         reg_fn=l1_reg_all, lambda_reg=meta["lambda_reg"], gc=meta["gc"])
    np.testing.assert_allclose(out["losses"], g["f64/losses"], atol=1e-5)
    np.testing.assert_allclose(out["risks"], g["f64/risks"], atol=1e-4)
    assert len(snaps) == 2
    for si, snap in enumerate(snaps, start=1):
        for k, v in snap.items():
            check_summary(g, f"f64/step{si}/{k}", v, rtol=2e-5, atol=2e-6)


def test_summary_survival_matches_per_bag_forward(golden):
    """utils/core_utils.py:358-430: per-subject risk = -sum(S) and the c-index over the loader; missing modality skipped."""
    from multimodalfusion_amd.utils.core_utils import concordance_index_censored, summary_survival
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    g, meta, sd, model, loader = _setup(golden)
    loader = list(loader)
    times = [10.0, 3.0, 7.0, 1.0]
    loader = [(b[0], b[1], b[2], b[3], np.array([times[i % 4]]), b[5]) for i, b in enumerate(loader)]
    loader.insert(1, ({"T1": torch.zeros(1, 1)}, torch.zeros(1, 1), torch.zeros(1, 4), torch.tensor([0]),
                      np.array([5.0]), torch.tensor([0.0])))             # pathology missing -> skipped
    res, cidx = summary_survival(model, loader, meta["K"], "path", loss_fn=NLLSurvLoss(alpha=0.0))
    n = len(loader) - 1
    assert len(res["risk"]) == n and list(res["subject_id"]) == [0] + list(range(2, n + 1))
    want = []
    with torch.no_grad():
        for b in loader:
            if b[1].shape == (1, 1):
                continue
            hz, S, _, _ = model(path_features=b[1].cuda())
            want.append(float(-S.sum()))
    np.testing.assert_allclose(res["risk"], np.array(want), rtol=0, atol=1e-6)
    t = np.concatenate([b[4] for b in loader if b[1].shape != (1, 1)])
    c = np.array([float(b[5]) for b in loader if b[1].shape != (1, 1)])
    assert abs(cidx - concordance_index_censored((1 - c).astype(bool), t, np.array(want))[0]) < 1e-12


def test_bags_in_flight_reproduce_the_reference_trajectory(golden):
    """inflight=2: the two bags of each accumulation window run on two HIP streams into two gradient slots; the
    parameters after each optimizer step are still the reference's (gc = 2 fixture)."""
    from multimodalfusion_amd.optim import FlatAdam
    from multimodalfusion_amd.utils import core_utils
    from multimodalfusion_amd.utils.loss_utils import NLLSurvLoss
    from multimodalfusion_amd.utils.utils import l1_reg_all
    g, meta, sd, model, loader = _setup(golden)
    opt = FlatAdam(model, lr=meta["lr"], weight_decay=meta["reg"], lambda_l1=meta["lambda_reg"])
    snaps = []
    step0 = opt.step

    def step(**kw):
        step0(**kw)
        snaps.append({k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()})

    opt.step = step
    out = core_utils.train_loop_survival(0, model, loader, opt, meta["K"], "path", loss_fn=NLLSurvLoss(alpha=0.0),
                                         reg_fn=l1_reg_all, lambda_reg=meta["lambda_reg"], gc=meta["gc"], inflight=2)
    np.testing.assert_allclose(out["losses"], g["f64/losses"], atol=1e-5)
    assert len(snaps) == 2
    for si, snap in enumerate(snaps, start=1):
        for k, v in snap.items():
            check_summary(g, f"f64/step{si}/{k}", v, rtol=2e-5, atol=2e-6)

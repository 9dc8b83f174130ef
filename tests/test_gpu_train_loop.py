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

"""CPU: stage-2 oracle pinned against fixtures from the imported reference; drop-in surface of our modules."""
import numpy as np
import pytest
import torch

from conftest import check_summary
from oracle import stage2_port as s2


def _model(m):
    from multimodalfusion_amd.models import coxranking_models_pretrained as cm
    from multimodalfusion_amd.models import nll_models_pretrained as nm
    mod = nm if m["family"] == "nll" else cm
    cls = mod.unimonal_pretrained if m["kind"] == "uni" else mod.multimodal_pretrained
    return cls(n_classes=m["K"], mode=m["mode"], train_type=m["train_type"], n_layers=m["n_layers"])


def test_oracle_matches_reference_fixtures(golden):
    g = golden("stage2")
    cases = g.meta["cases"]
    assert len(cases) >= 15
    for name, m in cases.items():
        res = s2.run_case(m, {k: tuple(v) for k, v in m["shapes"].items()})
        tag = name + "/f64"
        assert abs(res["loss"] - float(g[tag + "/loss"])) <= 1e-10, name
        np.testing.assert_allclose(res["risk"].reshape(-1), g[tag + "/risk"].reshape(-1), rtol=0, atol=1e-10, err_msg=name)
        if "hazards" in res:
            np.testing.assert_allclose(res["hazards"], g[tag + "/hazards"], rtol=0, atol=1e-11)
            np.testing.assert_allclose(res["S"], g[tag + "/S"], rtol=0, atol=1e-11)
        for k, gr in res["grads"].items():
            check_summary(g, f"{tag}/grad/{k}", gr, rtol=1e-9, atol=1e-12)
        for k, b in res["buffers"].items():
            check_summary(g, f"{tag}/buf/{k}", b, rtol=1e-9, atol=1e-12)


def test_state_dict_keys_and_shapes_match_the_reference(golden):
    g = golden("stage2")
    for name, m in g.meta["cases"].items():
        sd = _model(m).state_dict()
        assert {k: list(v.shape) for k, v in sd.items()} == m["shapes"], name


@pytest.mark.parametrize("tag,family,cls,kw", [
    ("init_nll_mm_late_highway", "nll", "multimodal_pretrained", dict(train_type="late-highway", mode="radio_path_omic", n_layers=2)),
    ("init_cox_uni_fcnn", "cox", "unimonal_pretrained", dict(train_type="fcnn", mode="path")),
])
def test_same_seed_initial_state(golden, tag, family, cls, kw):
    """initialize_weights (utils/utils_pretrained.py:145-154) consumes torch's RNG in the reference's module order."""
    from multimodalfusion_amd.models import coxranking_models_pretrained as cm
    from multimodalfusion_amd.models import nll_models_pretrained as nm
    g = golden("stage2")
    torch.manual_seed(1234)
    mdl = getattr(nm if family == "nll" else cm, cls)(**kw)
    for k, v in mdl.state_dict().items():
        a = v.detach().double().numpy().reshape(-1)
        assert abs(float(a.sum()) - float(g[f"{tag}/{k}/sum"])) <= 1e-9 * max(1.0, abs(float(a.sum()))), k
        assert abs(float(np.sqrt((a * a).sum())) - float(g[f"{tag}/{k}/l2"])) <= 1e-9 * max(1.0, float(g[f"{tag}/{k}/l2"])), k


def test_ranking_loss_restatement_edge_cases():
    r = torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64, requires_grad=True)
    # nobody has an event: no comparable pair -> zero (utils/loss_utils.py:84-85)
    assert float(s2.ranking_loss(r, torch.tensor([1.0, 2.0, 3.0]), torch.ones(3), "sigmoid", "mean")) == 0.0
    # all tied times: no comparable pair either
    assert float(s2.ranking_loss(r, torch.tensor([2.0, 2.0, 2.0]), torch.zeros(3), "relu", "sum")) == 0.0

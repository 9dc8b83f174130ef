"""CPU: bag assembly of the feed against the reference's recipe (datasets/dataset_survival.py:354-366)."""
import torch

from multimodalfusion_amd.feed import load_slide_bags


def test_multi_slide_bag_equals_torch_cat(tmp_path):
    torch.manual_seed(0)
    bags = [torch.randn(n, 1024) for n in (5, 1, 17)]
    paths = []
    for i, b in enumerate(bags):
        p = tmp_path / f"slide_{i}.pt"
        torch.save(b, p)
        paths.append(str(p))
    got = load_slide_bags(paths, pin=False)
    assert torch.equal(got, torch.cat(bags, dim=0))                 # what the reference's __getitem__ returns
    assert load_slide_bags(paths, pin=False, dtype=torch.bfloat16).dtype == torch.bfloat16
    assert torch.equal(load_slide_bags([], pin=False), torch.zeros((1, 1)))      # "pathology missing" sentinel


def test_bf16_on_disk_round_trip(tmp_path):
    b = torch.randn(9, 1024).to(torch.bfloat16)
    torch.save(b, tmp_path / "s.pt")
    got = load_slide_bags([str(tmp_path / "s.pt")], pin=False)
    assert got.dtype == torch.bfloat16 and torch.equal(got, b)


# ---- radiology bags: slices common to every modality (datasets/dataset_survival.py:346-348) --------------------------
def _feed_fixture():
    import json
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "feed.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def test_intersect_modalities_matches_reference_fixture():
    """tests/golden/feed.npz holds what the reference's own three lines returned (oracle/gen_golden_feed.py executes them,
    read as text from the reference tree, on seeded inputs): every case must be reproduced bit for bit."""
    import numpy as np
    import pytest
    from multimodalfusion_amd.feed import intersect_modalities
    z, meta = _feed_fixture()
    assert meta["reference_lines"] == "datasets/dataset_survival.py:346-348"
    for case in meta["cases"]:
        name, mods = case["name"], case["modalities"]
        feats = {m: z[f"{name}/in/{m}/features"] for m in mods}
        idx = {m: z[f"{name}/in/{m}/slice_index"] for m in mods}
        want = {m: z[f"{name}/out/{m}"] for m in mods}
        ragged = len({w.shape[0] for w in want.values()}) > 1
        if ragged:      # a slice id repeated inside one modality: the reference keeps unequal bags (and fails later, in torch.cat)
            with pytest.raises(ValueError):
                intersect_modalities(feats, idx, mods, pin=False)
        got = intersect_modalities(feats, idx, mods, pin=False, require_equal=False)
        assert list(got.keys()) == mods
        for m in mods:
            assert got[m].dtype == torch.float32 and tuple(got[m].shape) == want[m].shape, (name, m)
            assert np.array_equal(got[m].numpy(), want[m]), (name, m)          # bit for bit


def test_intersect_modalities_inputs_and_types():
    import numpy as np
    from multimodalfusion_amd.feed import intersect_modalities
    f = {"a": torch.arange(12.0).reshape(4, 3), "b": np.arange(9.0, dtype=np.float32).reshape(3, 3)}
    i = {"a": [5, 7, 9, 11], "b": np.array([11, 5, 6])}
    got = intersect_modalities(f, i, pin=False)
    assert torch.equal(got["a"], f["a"][[0, 3]]) and torch.equal(got["b"], torch.as_tensor(f["b"])[[0, 1]])   # stored order kept
    assert intersect_modalities(f, i, pin=False, dtype=torch.bfloat16)["a"].dtype == torch.bfloat16
    assert intersect_modalities({}, {}) == {}
    import pytest
    with pytest.raises(ValueError):
        intersect_modalities({"a": torch.zeros(3, 2)}, {"a": [1, 2]}, pin=False)


def test_load_radio_bags_fails_loudly_without_h5py():
    import importlib.util
    import pytest
    from multimodalfusion_amd.feed import load_radio_bags
    if importlib.util.find_spec("h5py") is None:
        with pytest.raises(ImportError):
            load_radio_bags({"T1": "/nonexistent.h5"})

"""GPU: the radiology feed end to end (datasets/dataset_survival.py:339-348 -> utils/core_utils.py:194-200).

feed.intersect_modalities(pin=True) -> feed.DevicePrefetcher -> MIL_Attention_fc_surv_radio.  The index work is pinned by
tests/golden/feed.npz (the reference's own three lines, executed by oracle/gen_golden_feed.py): for every fixture case the
slice ids are the fixture's, the kept ROWS are the ones the reference kept (recovered from the fixture's in / out
features), and the features are 1024 wide (the fixture's are 8-24 wide: the head needs 1024).  The rows that reach the
GPU must be those rows bit for bit; the head's outputs on them must equal the fp64 oracle's on the same rows."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import inputs as gen
from oracle import torch_port as tp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fixture():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "feed.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def _reference_rows(z, name, m):
    """Row numbers the reference kept for modality m: the fixture's output rows located in its input rows."""
    fin, fout = z[f"{name}/in/{m}/features"], z[f"{name}/out/{m}"]
    rows = []
    for r in fout:
        hit = np.nonzero((fin == r).all(axis=1))[0]
        assert hit.size >= 1
        rows.append(int(hit[0]) if not rows or hit[hit > rows[-1]].size == 0 else int(hit[hit > rows[-1]][0]))
    return rows


@pytest.mark.parametrize("case", ["four_mods_partial_overlap", "unsorted_ids", "single_modality", "identical_ids_full_width"])
def test_intersected_radiology_bags_through_the_prefetcher_match_the_oracle(case):
    from multimodalfusion_amd.feed import DevicePrefetcher, intersect_modalities
    from multimodalfusion_amd.models import MIL_Attention_fc_surv_radio
    z, meta = _fixture()
    c = [k for k in meta["cases"] if k["name"] == case][0]
    mods = c["modalities"]
    idx = {m: z[f"{case}/in/{m}/slice_index"] for m in mods}
    feats = {m: gen.bag(8800 + i, idx[m].shape[0], stream=3 * i) for i, m in enumerate(mods)}    # 1024 wide
    want_rows = {m: _reference_rows(z, case, m) for m in mods}

    bags = intersect_modalities(feats, idx, mods, pin=True)
    assert all(t.is_pinned() for t in bags.values())
    for m in mods:                                   # the rows the reference's lines keep, in their stored order, bit for bit
        assert np.array_equal(bags[m].numpy(), feats[m][want_rows[m]]), m
    n = bags[mods[0]].shape[0]
    assert n > 0 and all(t.shape == (n, 1024) for t in bags.values())

    sd = gen.radio_state_dict(seed=61, gated=True, n_classes=4, dropout=False, n_mod=len(mods), bias_std=0.05)
    model = MIL_Attention_fc_surv_radio(radio_fusion="concat", gate_radio=True, dropout=False, n_classes=4, modalities=mods)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    model = model.to(DEV).eval()
    batch = (bags, torch.zeros(1, 1), torch.zeros(1, 1), torch.tensor([1]), 3.0, torch.tensor([0.0]))
    got = None
    for radio, path, genomic, label, event_time, cens in DevicePrefetcher([batch], DEV, depth=2):
        for m in mods:                               # what arrived in HBM is what was gathered
            assert radio[m].is_cuda and torch.equal(radio[m].cpu(), bags[m])
        with torch.no_grad():
            got = model(**radio)
    hz, S, Yh, A_raw = got
    ref = tp.radio_forward(tp.to_torch(sd, torch.float64, False), [torch.as_tensor(feats[m][want_rows[m]]).double() for m in mods],
                           True, False, None)
    np.testing.assert_allclose(hz.cpu().numpy(), ref[0].numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(S.cpu().numpy(), ref[1].numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(A_raw.cpu().numpy(), ref[3].numpy(), rtol=0, atol=1e-4)
    assert int(Yh) == int(ref[2])

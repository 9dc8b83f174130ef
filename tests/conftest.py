import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Read access to one tests/golden/*.npz fixture (data only: inputs are regenerated from meta)."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(str(self.z["meta"])) if "meta" in self.z.files else {}

    def __getitem__(self, k):
        return self.z[k]

    def has(self, k):
        return k in self.z.files

    def keys(self, prefix):
        return [k for k in self.z.files if k.startswith(prefix)]


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


def check_summary(g, prefix, arr, rtol, atol, kink_rows=None):
    """Compare an array against a fixture entry written by oracle.gen_golden.summarize().

    kink_rows (first-layer gradients only; tests/test_gpu_path.py: relu_kink_units): rows of a [H x L] / [H] gradient whose
    hidden unit has, for some instance of the bag, an fp64 pre-activation within fp32 rounding of the ReLU kink.  relu'(u)
    there legitimately differs between two fp32 arithmetics (and between fp32 and fp64), so entries of THOSE rows may miss
    the bar by at most 1 % of the tensor's max -- the same allowance every live-oracle comparison of the suite makes."""
    from oracle.gen_golden import sample_idx
    a = np.asarray(arr, dtype=np.float64).reshape(-1)
    scale = max(float(g[prefix + "/absmax"]), 1e-30)
    tol = atol + rtol * scale
    rows = np.asarray(arr).shape[0] if np.asarray(arr).ndim >= 1 else 1
    per_row = max(a.size // max(rows, 1), 1)
    kink = np.zeros(a.size, dtype=bool)
    for r in (kink_rows or ()):
        kink[r * per_row:(r + 1) * per_row] = True

    def judge(err, mask, what):
        plain = err[~mask].max() if (~mask).any() else 0.0
        assert plain <= tol, f"{prefix}: {what} max abs err {plain:.3e} > {tol:.3e}"
        if mask.any():
            assert err[mask].max() <= max(tol, 1e-2 * scale), f"{prefix}: {what} kink-row err {err[mask].max():.3e} > 1 % of {scale:.3e}"

    if g.has(prefix + "/full"):
        ref = g[prefix + "/full"]
        assert ref.shape == a.shape, (prefix, ref.shape, a.shape)
        if a.size:
            judge(np.abs(a - ref), kink, "")
    else:
        ref = g[prefix + "/sample"]
        idx = sample_idx(a.size)
        judge(np.abs(a[idx] - ref), kink[idx], "sampled")
    l2 = float(g[prefix + "/l2"])
    got_l2 = float(np.sqrt((a * a).sum()))
    slack = 1e-2 * scale * np.sqrt(float(kink.sum())) if kink.any() else 0.0
    assert abs(got_l2 - l2) <= atol * np.sqrt(max(a.size, 1)) + rtol * max(l2, 1e-30) * 10 + slack, \
        f"{prefix}: l2 {got_l2} vs {l2}"

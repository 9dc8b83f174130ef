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


def check_summary(g, prefix, arr, rtol, atol):
    """Compare an array against a fixture entry written by oracle.gen_golden.summarize()."""
    from oracle.gen_golden import sample_idx
    a = np.asarray(arr, dtype=np.float64).reshape(-1)
    scale = max(float(g[prefix + "/absmax"]), 1e-30)
    tol = atol + rtol * scale
    if g.has(prefix + "/full"):
        ref = g[prefix + "/full"]
        assert ref.shape == a.shape, (prefix, ref.shape, a.shape)
        err = np.abs(a - ref).max() if a.size else 0.0
        assert err <= tol, f"{prefix}: max abs err {err:.3e} > {tol:.3e}"
    else:
        ref = g[prefix + "/sample"]
        got = a[sample_idx(a.size)]
        err = np.abs(got - ref).max()
        assert err <= tol, f"{prefix}: sampled max abs err {err:.3e} > {tol:.3e}"
    l2 = float(g[prefix + "/l2"])
    got_l2 = float(np.sqrt((a * a).sum()))
    assert abs(got_l2 - l2) <= atol * np.sqrt(max(a.size, 1)) + rtol * max(l2, 1e-30) * 10, \
        f"{prefix}: l2 {got_l2} vs {l2}"

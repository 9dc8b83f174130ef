"""CPU: pin the oracle (oracle/torch_port.py) against fixtures made from the imported reference.

Tolerances: fp64 oracle vs fp64 reference run -> 1e-9 relative (same math, same ATen ops);
fp64 oracle vs fp32 reference run -> the north_star bars (1e-4 outputs, 1e-5 loss).
"""
import numpy as np
import pytest

from conftest import check_summary
from oracle import cases


def _check_case(g, tag_root, res, n_a_raw=True):
    for tag, rtol, atol, ltol in ((tag_root + "/f64", 1e-9, 1e-11, 1e-10), (tag_root + "/f32", 2e-4, 2e-5, 1e-5)):
        if not g.has(tag + "/loss"):
            continue
        assert abs(float(res["loss"]) - float(g[tag + "/loss"])) <= ltol, (tag, res["loss"], g[tag + "/loss"])
        np.testing.assert_allclose(res["hazards"], g[tag + "/hazards"], rtol=0, atol=max(atol, 1e-4 if "f32" in tag else 0))
        if "S" in res and g.has(tag + "/S"):
            np.testing.assert_allclose(res["S"], g[tag + "/S"], rtol=0, atol=max(atol, 1e-4 if "f32" in tag else 0))
        if "Y_hat" in res and g.has(tag + "/Y_hat"):
            assert np.array_equal(res["Y_hat"], g[tag + "/Y_hat"])
        if "M" in res and g.has(tag + "/M"):
            np.testing.assert_allclose(res["M"], g[tag + "/M"], rtol=0, atol=max(atol, 1e-4 if "f32" in tag else 0))
        if n_a_raw and "A_raw" in res:
            items = res["A_raw"].items() if isinstance(res["A_raw"], dict) else [("", res["A_raw"])]
            for name, A in items:
                key = f"{tag}/A_raw{('_' + name) if name else ''}"
                check_summary(g, key, A, rtol=0, atol=1e-10 if "f64" in tag else 1e-4)
                a = np.asarray(A).reshape(-1)
                assert int(a.argmax()) == int(g[key + "/argmax"]) or "f32" in tag
        for k, gr in res["grads"].items():
            check_summary(g, f"{tag}/grad/{k}", gr, rtol=rtol, atol=atol if "f64" in tag else 1e-6)


def test_path_cases(golden):
    g = golden("path")
    assert len(g.meta) >= 12
    for name, m in g.meta.items():
        res = cases.run_path(m)
        _check_case(g, name, res)


def test_radio_cases(golden):
    g = golden("radio")
    for name, m in g.meta.items():
        _check_case(g, name, cases.run_radio(m))


def test_omic_cases(golden):
    g = golden("omic")
    for name, m in g.meta.items():
        _check_case(g, name, cases.run_omic(m))


def test_mm_cases(golden):
    g = golden("mm")
    for name, m in g.meta.items():
        _check_case(g, name, cases.run_mm(m))


def test_trajectory(golden):
    g = golden("trajectory")
    res = cases.run_trajectory(g.meta)
    np.testing.assert_allclose(res["losses"], g["f64/losses"], rtol=1e-10)
    np.testing.assert_allclose(res["risks"], g["f64/risks"], rtol=1e-10)
    np.testing.assert_allclose(res["losses"], g["f32/losses"], atol=1e-5)
    for si, sd in enumerate(res["steps"], start=1):
        for k, v in sd.items():
            check_summary(g, f"f64/step{si}/{k}", v, rtol=1e-9, atol=1e-12)
            check_summary(g, f"f32/step{si}/{k}", v, rtol=1e-5, atol=1e-6)


def test_port_fp32_meets_north_star_bars(golden):
    """The fp32 port (what bench.py times as cpu_baseline) is itself within the bars of the fp64 reference."""
    import torch
    g = golden("path")
    for name in ("g_small_k4_n1000", "g_small_k4_n10000"):
        m = g.meta[name]
        res = cases.run_path(m, dtype=torch.float32)
        tag = name + "/f64"
        assert abs(float(res["loss"]) - float(g[tag + "/loss"])) <= 1e-5
        np.testing.assert_allclose(res["hazards"], g[tag + "/hazards"], atol=1e-4)
        check_summary(g, tag + "/A_raw", res["A_raw"], rtol=0, atol=1e-4)


def test_keep_mask_rate_and_determinism():
    from oracle import inputs as gen
    k1 = gen.keep_mask(7, 0, 500, 256, 0.25)
    k2 = gen.keep_mask(7, 0, 500, 256, 0.25)
    assert np.array_equal(k1, k2)
    assert abs(k1.mean() - 0.75) < 0.01
    assert not np.array_equal(k1, gen.keep_mask(7, 1, 500, 256, 0.25))
    # row-to-row and col-to-col decorrelation (hash quality smoke check)
    assert abs(np.corrcoef(k1[:-1].ravel(), k1[1:].ravel())[0, 1]) < 0.02
    assert abs(np.corrcoef(k1[:, :-1].ravel(), k1[:, 1:].ravel())[0, 1]) < 0.02

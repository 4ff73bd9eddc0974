"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerances (north_star): fp32 trajectories are sensitive to summation order, so the bar is
on the reconstruction and residual relative to ||X||_F (<= 1e-5) and on VAF (<= 1e-5 absolute); W and H
themselves are compared tightly only at small iteration counts.
"""
import numpy as np
import pytest

from oracle import nmf_mu_oracle as orc
from muscle_synergies_amd.synth import emg_matrix, random_init

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _fit(X, W0, H0, **kw):
    import muscle_synergies_amd as ms

    return ms.fit_batched(X, W0, H0, **kw)


def _compare(X, res_W, res_H, res_err, ref, tol=TOL):
    xn = np.linalg.norm(X.astype(np.float64))
    wh = res_W.astype(np.float64) @ res_H.astype(np.float64)
    wh_ref = ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)
    d_wh = np.linalg.norm(wh - wh_ref) / xn
    d_err = abs(float(res_err) - float(ref["reconstruction_err"])) / xn
    assert d_wh <= tol, f"rel |WH - WH_ref| = {d_wh:.3e}"
    assert d_err <= tol, f"rel |err - err_ref| = {d_err:.3e}"
    return d_wh, d_err


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n_iter", [1, 2, 10, 100])
def test_small_loop_vs_oracle_and_golden(g2_small, dtype, n_iter):
    dt = np.dtype(dtype).name
    X = np.asfortranarray(g2_small[f"X_{dt}"])
    W0, H0 = g2_small[f"W0_{dt}"], g2_small[f"H0_{dt}"]
    res = _fit(X, W0, H0, max_iter=n_iter, tol=0.0)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=n_iter, tol=0.0)
    _compare(X, res.W[0], res.H[0], res.reconstruction_err[0], ref)
    gold = {"W": g2_small[f"W_{dt}_{n_iter}"], "H": g2_small[f"H_{dt}_{n_iter}"],
            "reconstruction_err": g2_small[f"err_{dt}_{n_iter}"]}
    _compare(X, res.W[0], res.H[0], res.reconstruction_err[0], gold)
    assert int(res.n_iter[0]) == n_iter
    if n_iter <= 10:
        rt = 2e-5 if dtype == np.float32 else 1e-11
        np.testing.assert_allclose(res.W[0], gold["W"], rtol=rt, atol=rt * 1e-2)
        np.testing.assert_allclose(res.H[0], gold["H"], rtol=rt, atol=rt * 1e-2)


@pytest.mark.parametrize("layout", ["channel_major", "row_major"])
def test_config2_500_iterations(g2_full, layout):
    """Config #2: one 16 x 10 000 matrix, k = 5, fp32, 500 mu iterations vs sklearn's recorded output."""
    c = g2_full["cases"][0]
    assert c["dtype"] == "float32" and c["init"] == "random"
    X = emg_matrix(c["seed"], dtype=np.float32)
    if layout == "row_major":
        X = np.ascontiguousarray(X)
    W0, H0 = random_init(X, 5, c["seed"])
    res = _fit(X, W0, H0, max_iter=500, tol=0.0)
    g = c["iters"]["500"]
    xn = c["X_fro"]
    WH = res.W[0].astype(np.float64) @ res.H[0].astype(np.float64)
    rows = g2_full["rows"]
    assert abs(float(res.reconstruction_err[0]) - g["reconstruction_err"]) / xn <= TOL
    assert abs(np.sqrt((WH ** 2).sum()) - g["WH_fro"]) / xn <= TOL
    assert np.linalg.norm(WH.sum(axis=0) - np.array(g["WH_colsum"])) / np.linalg.norm(g["WH_colsum"]) <= TOL
    assert np.abs(WH[rows] - np.array(g["WH_rows"])).max() <= 2e-4
    assert abs(float(res.vaf[0, 0]) - g["vaf_all"]) <= TOL
    np.testing.assert_allclose(res.vaf[0, 1:], g["vaf_col"], atol=TOL)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=500, tol=0.0)
    _compare(X, res.W[0], res.H[0], res.reconstruction_err[0], ref)


def test_batch_matches_per_matrix_oracle():
    B, T = 6, 1000
    Xs = [emg_matrix(100 + b, T=T, dtype=np.float32) for b in range(B)]
    inits = [random_init(Xs[b], 5, b) for b in range(B)]
    X = np.stack([np.ascontiguousarray(x.T) for x in Xs]).transpose(0, 2, 1)  # [B, T, m] view, channel-major
    W0 = np.stack([i[0] for i in inits])
    H0 = np.stack([i[1] for i in inits])
    res = _fit(X, W0, H0, max_iter=60, tol=0.0)
    for b in range(B):
        ref = orc.nmf_mu_fit(Xs[b], inits[b][0], inits[b][1], max_iter=60, tol=0.0)
        _compare(Xs[b], res.W[b], res.H[b], res.reconstruction_err[b], ref)
        va, vc = orc.vaf(Xs[b].astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
        assert abs(res.vaf[b, 0] - va) <= TOL
        np.testing.assert_allclose(res.vaf[b, 1:], vc, atol=TOL)


def test_stop_rule_matches_sklearn_trace(g3):
    for c in g3["cases"]:
        dt = np.dtype(c["dtype"])
        X = emg_matrix(c["seed"], T=c["T"], dtype=dt)
        from muscle_synergies_amd.init import initialize_nmf

        W0, H0 = initialize_nmf(X, 5, init="nndsvda", random_state=0)
        res = _fit(X, W0, H0, max_iter=c["max_iter"], tol=c["tol"])
        assert int(res.n_iter[0]) == c["n_iter"], (c["dtype"], c["T"], c["tol"])
        xn = np.linalg.norm(X.astype(np.float64))
        assert abs(float(res.reconstruction_err[0]) - c["reconstruction_err"]) / xn <= TOL


def test_find_synergies_abridged_fixture(g1):
    """Config #1 replayed from the fixture through the reference-shaped entry point."""
    import pandas as pd

    import muscle_synergies_amd as ms

    V = pd.DataFrame(np.array(g1["V"]), columns=g1["columns"])
    c = g1["single_k4"]
    res = ms.find_synergies(V, 4, **c["kwargs"])
    assert isinstance(res.model, ms.HipNMF)
    assert res.model.n_iter_ == c["n_iter"]
    np.testing.assert_allclose(res.model.reconstruction_err_, c["reconstruction_err"], rtol=1e-9)
    np.testing.assert_allclose(res.components.to_numpy(), np.array(c["components"]), rtol=1e-9, atol=1e-13)
    assert list(res.vaf_values.columns) == c["vaf_columns"]
    np.testing.assert_allclose(res.vaf_values.to_numpy()[0], c["vaf_values"], rtol=1e-10)
    r = g1["range_2_4"]
    res = ms.find_synergies(V, 2, 4, **r["kwargs"])
    assert list(res.components.keys()) == r["keys"] == [2, 3, 4]
    assert list(res.vaf_values.index) == r["vaf_index"]
    np.testing.assert_allclose(res.vaf_values.to_numpy(), np.array(r["vaf_values"]), rtol=1e-9)
    for k in r["keys"]:
        np.testing.assert_allclose(res.components[k].to_numpy(), np.array(r["components"][str(k)]), rtol=1e-8, atol=1e-12)
        assert res.model[k].n_iter_ == r["n_iter"][str(k)]

"""Pin the NumPy oracle against the golden vectors captured from the reference
(``tests/golden/make_golden.py``: reference ``find_synergies`` + sklearn 1.7.2).

On the machine that generated the fixtures the oracle is bit-identical to
sklearn; the tolerances below only leave room for a different host BLAS
(summation order inside ``np.dot``) on other CPUs.
"""
import numpy as np
import pytest

from oracle import nmf_mu_oracle as orc
from muscle_synergies_amd.synth import emg_matrix, random_init

RT = {"float32": 2e-5, "float64": 1e-11}


def test_epsilon():
    assert orc.EPSILON == np.finfo(np.float32).eps == np.float32(1.1920929e-07)


def test_g1_abridged_single_k4(g1):
    c = g1["single_k4"]
    V = np.asfortranarray(np.array(g1["V"]))
    r = orc.nmf_mu_fit(V, np.array(c["W0"]), np.array(c["H0"]), max_iter=200, tol=c["tol"])
    assert r["n_iter"] == c["n_iter"] == 200
    np.testing.assert_allclose(r["reconstruction_err"], c["reconstruction_err"], rtol=1e-10)
    np.testing.assert_allclose(r["H"], np.array(c["components"]), rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(r["W"], np.array(c["transformed"]), rtol=1e-10, atol=1e-14)
    va, vc = orc.vaf(V, r["W"], r["H"])
    np.testing.assert_allclose(np.r_[va, vc], np.array(c["vaf_values"]), rtol=1e-12)
    # the published numbers of the default-init call
    d = g1["single_k4_default_init"]
    assert d["n_iter"] == 200
    np.testing.assert_allclose(d["reconstruction_err"], 0.001624623973129921, rtol=1e-9)
    np.testing.assert_allclose(d["vaf_values"][0], 0.9967912687038506, rtol=1e-9)


@pytest.mark.parametrize("dt", ["float32", "float64"])
@pytest.mark.parametrize("n", [1, 2, 10, 100])
def test_g2_small_loop(g2_small, dt, n):
    X = np.asfortranarray(g2_small[f"X_{dt}"])
    r = orc.nmf_mu_fit(X, g2_small[f"W0_{dt}"], g2_small[f"H0_{dt}"], max_iter=n, tol=0)
    scale = 1 if n <= 10 else 50  # fp32 trajectories drift with summation order
    np.testing.assert_allclose(r["W"], g2_small[f"W_{dt}_{n}"], rtol=RT[dt] * scale, atol=RT[dt] * scale * 1e-2)
    np.testing.assert_allclose(r["H"], g2_small[f"H_{dt}_{n}"], rtol=RT[dt] * scale, atol=RT[dt] * scale * 1e-2)
    np.testing.assert_allclose(r["reconstruction_err"], g2_small[f"err_{dt}_{n}"], rtol=RT[dt] * scale)
    assert r["n_iter"] == n


@pytest.mark.parametrize("idx", range(6))
def test_g2_full_config2(g2_full, idx):
    """Config #2 shape: inputs rebuilt from the recipe, outputs vs sklearn checksums."""
    c = g2_full["cases"][idx]
    dt = np.dtype(c["dtype"])
    X = emg_matrix(c["seed"], dtype=dt)
    np.testing.assert_allclose(X.astype(np.float64).sum(), c["X_sum"], rtol=1e-12)
    if c["init"] == "random":
        W0, H0 = random_init(X, 5, c["seed"])
    else:
        pytest.importorskip("sklearn")
        from sklearn.decomposition._nmf import _initialize_nmf
        W0, H0 = _initialize_nmf(X, 5, init="nndsvda", random_state=0)
        np.testing.assert_allclose(H0, np.array(c["H0"], dtype=dt), rtol=1e-4 if dt == np.float32 else 1e-9)
    np.testing.assert_allclose(W0.astype(np.float64).sum(), c["W0_sum"], rtol=1e-6 if dt == np.float32 else 1e-12)
    xfro = c["X_fro"]
    for n in (1, 10, 100):
        g = c["iters"][str(n)]
        r = orc.nmf_mu_fit(X, W0, H0, max_iter=n, tol=0)
        WH = r["W"].astype(np.float64) @ r["H"].astype(np.float64)
        rows = g2_full["rows"]
        assert np.linalg.norm(WH[rows] - np.array(g["WH_rows"])) / np.linalg.norm(np.array(g["WH_rows"])) < 1e-5
        assert abs(np.sqrt((WH ** 2).sum()) - g["WH_fro"]) / xfro < 1e-5
        assert abs(float(r["reconstruction_err"]) - g["reconstruction_err"]) / xfro < 1e-5
        va, vc = orc.vaf(X.astype(np.float64), r["W"].astype(np.float64), r["H"].astype(np.float64))
        assert abs(va - g["vaf_all"]) < 1e-5
        np.testing.assert_allclose(vc, g["vaf_col"], atol=1e-5)


def test_g3_stop_rule(g3):
    for c in g3["cases"]:
        dt = np.dtype(c["dtype"])
        X = emg_matrix(c["seed"], T=c["T"], dtype=dt)
        pytest.importorskip("sklearn")
        from sklearn.decomposition._nmf import _initialize_nmf
        W0, H0 = _initialize_nmf(X, 5, init="nndsvda", random_state=0)
        trace = []
        W, H = W0.copy(), H0.copy()
        W, H, n_iter = orc.fit_multiplicative_update(X, W, H, c["max_iter"], c["tol"], err_trace=trace)
        assert n_iter == c["n_iter"]
        assert n_iter % 10 == 0
        np.testing.assert_allclose(trace, c["err_trace"], rtol=1e-4 if dt == np.float32 else 1e-10)


@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_g5_transform_and_regularisation(g5, g2_small, dt):
    X2 = np.asfortranarray(g5[f"X2_{dt}"])
    r = orc.nmf_mu_transform(X2, g5[f"H_fit_{dt}"], max_iter=40, tol=0)
    np.testing.assert_allclose(r["W"], g5[f"W_transform_{dt}"], rtol=RT[dt] * 20, atol=RT[dt])
    X = np.asfortranarray(g2_small[f"X_{dt}"])
    r = orc.nmf_mu_fit(X, g2_small[f"W0_{dt}"], g2_small[f"H0_{dt}"], max_iter=60, tol=0,
                       alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3)
    np.testing.assert_allclose(r["W"], g5[f"W_reg_{dt}"], rtol=RT[dt] * 50, atol=RT[dt])
    np.testing.assert_allclose(r["H"], g5[f"H_reg_{dt}"], rtol=RT[dt] * 50, atol=RT[dt])
    np.testing.assert_allclose(r["reconstruction_err"], g5[f"err_reg_{dt}"], rtol=RT[dt] * 20)


def test_sharded_restatement_equals_unsharded():
    X = emg_matrix(5, T=640, dtype=np.float64)
    W0, H0 = random_init(X, 4, 5)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=25, tol=0)
    W, H = W0.copy(), H0.copy()
    bounds = [0, 200, 410, 640]
    for _ in range(25):
        sums = [orc.shard_pass(X[a:b], W[a:b], H) for a, b in zip(bounds[:-1], bounds[1:])]
        orc.h_update_from_sums(sum(s[0] for s in sums), sum(s[1] for s in sums), H)
    np.testing.assert_allclose(W, ref["W"], rtol=1e-9)
    np.testing.assert_allclose(H, ref["H"], rtol=1e-9)


@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_g7_kullback_leibler_oracle(dt):
    from conftest import load_npz

    g7 = load_npz("g7_kl.npz")
    X = np.asfortranarray(g7[f"X_{dt}"])
    for n in (1, 2, 10, 100):
        r = orc.nmf_mu_fit_kl(X, g7[f"W0_{dt}"], g7[f"H0_{dt}"], max_iter=n, tol=0)
        scale = 1 if n <= 10 else 50
        np.testing.assert_allclose(r["W"], g7[f"W_{dt}_{n}"], rtol=RT[dt] * scale, atol=RT[dt] * scale * 1e-2)
        np.testing.assert_allclose(r["H"], g7[f"H_{dt}_{n}"], rtol=RT[dt] * scale, atol=RT[dt] * scale * 1e-2)
        np.testing.assert_allclose(r["reconstruction_err"], g7[f"err_{dt}_{n}"], rtol=RT[dt] * scale * 10)
    r = orc.nmf_mu_fit_kl(X, g7[f"W0_{dt}"], g7[f"H0_{dt}"], max_iter=2000, tol=1e-4)
    assert r["n_iter"] == int(g7[f"stop_n_iter_{dt}"])
    r = orc.nmf_mu_fit_kl(X, g7[f"W0_{dt}"], g7[f"H0_{dt}"], max_iter=40, tol=0, alpha_W=0.002, alpha_H=0.001,
                          l1_ratio=0.3)
    np.testing.assert_allclose(r["H"], g7[f"H_reg_{dt}"], rtol=RT[dt] * 50, atol=RT[dt])

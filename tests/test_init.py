"""Host-side initialisation (`muscle_synergies_amd.init`) against vectors captured from sklearn 1.7.2."""
import numpy as np
import pytest

from muscle_synergies_amd.init import initialize_nmf
from muscle_synergies_amd.synth import emg_matrix, random_init


@pytest.mark.parametrize("backend", ["builtin", "auto"])
def test_matches_sklearn_fixtures(g4, backend):
    """Both paths -- the in-repo restatement and the delegation to scikit-learn -- against fixture G4."""
    arrays, meta = g4
    for c in meta:
        dt, T, m, k, init = c["dtype"], c["T"], c["m"], c["k"], c["init"]
        X = np.asfortranarray(arrays[f"X_{dt}_{T}_{m}"])
        W0, H0 = initialize_nmf(X, k, init=init, random_state=c["random_state"], backend=backend)
        Wg, Hg = arrays[f"W0_{dt}_{T}_{m}_{k}_{init}"], arrays[f"H0_{dt}_{T}_{m}_{k}_{init}"]
        assert W0.dtype == X.dtype and H0.dtype == X.dtype
        assert W0.shape == (T, k) and H0.shape == (k, m)
        if init == "random":
            assert np.array_equal(W0, Wg) and np.array_equal(H0, Hg)
        else:  # SVD based: allow for a different LAPACK/BLAS rounding on another host
            tol = 5e-4 if dt == "float32" else 1e-8
            np.testing.assert_allclose(W0, Wg, rtol=tol, atol=tol * np.abs(Wg).max())
            np.testing.assert_allclose(H0, Hg, rtol=tol, atol=tol * np.abs(Hg).max())
        assert (W0 >= 0).all() and (H0 >= 0).all()


@pytest.mark.parametrize("backend", ["builtin", "auto"])
def test_default_init_choice_and_errors(backend):
    X = emg_matrix(1, T=30, m=6, k_true=3, dtype=np.float64)
    Wd, Hd = initialize_nmf(X, 3, init=None, random_state=0, backend=backend)
    Wa, Ha = initialize_nmf(X, 3, init="nndsvda", random_state=0, backend=backend)
    assert np.array_equal(Wd, Wa) and np.array_equal(Hd, Ha)
    assert (Wa > 0).all()  # nndsvda fills zeros with the mean
    Wr, Hr = initialize_nmf(X, 9, init=None, random_state=0, backend=backend)  # k > min(T, m) falls back to 'random'
    assert Wr.shape == (30, 9)
    with pytest.raises(ValueError, match="can only be used when n_components <= min"):
        initialize_nmf(X, 9, init="nndsvd", backend=backend)
    with pytest.raises(ValueError, match="Negative values in data passed to NMF initialization."):
        initialize_nmf(-X, 2, backend=backend)
    with pytest.raises(ValueError, match="Invalid init parameter"):
        initialize_nmf(X, 2, init="bogus", backend=backend)


def test_against_live_sklearn_when_available():
    sk = pytest.importorskip("sklearn.decomposition._nmf")
    for dt in (np.float32, np.float64):
        X = emg_matrix(4, T=200, m=8, k_true=4, dtype=dt)
        for init in ("random", "nndsvd", "nndsvda", "nndsvdar"):
            W0, H0 = initialize_nmf(X, 4, init=init, random_state=11, backend="builtin")
            Ws, Hs = sk._initialize_nmf(X, 4, init=init, random_state=11)
            Wd, Hd = initialize_nmf(X, 4, init=init, random_state=11)  # delegates: bit-identical
            assert np.array_equal(Wd, Ws) and np.array_equal(Hd, Hs)
            tol = 1e-4 if dt == np.float32 else 1e-10
            np.testing.assert_allclose(W0, Ws, rtol=tol, atol=tol)
            np.testing.assert_allclose(H0, Hs, rtol=tol, atol=tol)


def test_synthetic_workload_is_reproducible(g2_full):
    for c in g2_full["cases"][:2]:
        X = emg_matrix(c["seed"], dtype=np.dtype(c["dtype"]))
        assert X.shape == (10000, 16) and X.flags["F_CONTIGUOUS"]
        assert X.min() >= 0 and X.max() == 1.0
        np.testing.assert_allclose(X.astype(np.float64).sum(), c["X_sum"], rtol=1e-12)
        W0, H0 = random_init(X, 5, c["seed"])
        np.testing.assert_allclose(W0.astype(np.float64).sum(), c["W0_sum"], rtol=1e-7)
        np.testing.assert_allclose(H0.astype(np.float64).sum(), c["H0_sum"], rtol=1e-7)


def test_exact_svd_solver_agrees_with_randomized_when_that_is_exact():
    X = emg_matrix(9, T=300, m=8, k_true=4, dtype=np.float64)
    a = initialize_nmf(X, 4, init="nndsvda", random_state=0)
    b = initialize_nmf(X, 4, init="nndsvda", svd_solver="exact")
    np.testing.assert_allclose(a[0], b[0], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-9, atol=1e-12)

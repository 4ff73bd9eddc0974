"""GPU parity of the Kullback-Leibler loss (row f-4: ``NMF(solver='mu', beta_loss='kullback-leibler')``)
against sklearn's recorded outputs (``tests/golden/g7_kl.npz``) and the oracle restatement.

Tolerances: the reconstruction WH relative to ||X||_F (<= 1e-5, fp32 and fp64); the reported error
``sqrt(2 KL)`` relative to its own size (sklearn evaluates ``sum(x log(x/wh)) - sum(x) + sum(wh)`` as three
large fp32 sums that cancel, this engine sums the non-negative per-element terms -- both are compared with
the fp64 value, and against each other at 2e-3 in fp32 / 1e-9 in fp64).
"""
import warnings

import numpy as np
import pytest

from conftest import load_npz
from oracle import nmf_mu_oracle as orc
from muscle_synergies_amd.synth import emg_matrix, random_init

pytestmark = pytest.mark.gpu

TOL = 1e-5
ERR_RT = {"float32": 2e-3, "float64": 1e-9}


def _fit(X, W0, H0, **kw):
    import muscle_synergies_amd as ms

    return ms.fit_batched(X, W0, H0, beta_loss="kullback-leibler", **kw)


def _rel_wh(X, W, H, Wr, Hr):
    xn = np.linalg.norm(X.astype(np.float64))
    return np.linalg.norm(W.astype(np.float64) @ H.astype(np.float64) - Wr.astype(np.float64) @ Hr.astype(np.float64)) / xn


@pytest.mark.parametrize("dt", ["float32", "float64"])
@pytest.mark.parametrize("n_iter", [1, 2, 10, 100])
def test_kl_loop_vs_golden_and_oracle(dt, n_iter):
    g7 = load_npz("g7_kl.npz")
    X = np.asfortranarray(g7[f"X_{dt}"])
    W0, H0 = g7[f"W0_{dt}"], g7[f"H0_{dt}"]
    res = _fit(X, W0, H0, max_iter=n_iter, tol=0.0)
    assert int(res.n_iter[0]) == n_iter
    assert _rel_wh(X, res.W[0], res.H[0], g7[f"W_{dt}_{n_iter}"], g7[f"H_{dt}_{n_iter}"]) <= TOL
    ref = orc.nmf_mu_fit_kl(X, W0, H0, max_iter=n_iter, tol=0)
    assert _rel_wh(X, res.W[0], res.H[0], ref["W"], ref["H"]) <= TOL
    np.testing.assert_allclose(float(res.reconstruction_err[0]), float(g7[f"err_{dt}_{n_iter}"]), rtol=ERR_RT[dt])
    # the error the engine reports is the divergence of the factors it returns (checked in fp64)
    e64 = orc.kl_divergence(X.astype(np.float64), res.W[0].astype(np.float64), res.H[0].astype(np.float64), True)
    np.testing.assert_allclose(float(res.reconstruction_err[0]), e64, rtol=1e-4 if dt == "float32" else 1e-11)
    if n_iter <= 10:
        rt = 2e-5 if dt == "float32" else 1e-11
        np.testing.assert_allclose(res.W[0], g7[f"W_{dt}_{n_iter}"], rtol=rt, atol=rt * 1e-2)
        np.testing.assert_allclose(res.H[0], g7[f"H_{dt}_{n_iter}"], rtol=rt, atol=rt * 1e-2)
    # the squared-error VAF columns are still produced for the KL fit
    err2 = ((X.astype(np.float64) - res.W[0].astype(np.float64) @ res.H[0].astype(np.float64)) ** 2).sum(axis=0)
    vaf_col = 1 - err2 / (X.astype(np.float64) ** 2).sum(axis=0)
    np.testing.assert_allclose(res.vaf[0][1:], vaf_col, atol=1e-5)


def test_kl_stop_rule_and_regularisation():
    g7 = load_npz("g7_kl.npz")
    X = np.asfortranarray(g7["X_float64"])
    W0, H0 = g7["W0_float64"], g7["H0_float64"]
    res = _fit(X, W0, H0, max_iter=2000, tol=1e-4)
    assert int(res.n_iter[0]) == int(g7["stop_n_iter_float64"])
    np.testing.assert_allclose(float(res.reconstruction_err[0]), float(g7["stop_err_float64"]), rtol=1e-9)
    # fp32: the stop rule fires on the same check or a neighbouring one (the fp32 divergence is noisy)
    X32 = np.asfortranarray(g7["X_float32"])
    res32 = _fit(X32, g7["W0_float32"], g7["H0_float32"], max_iter=2000, tol=1e-4)
    assert abs(int(res32.n_iter[0]) - int(g7["stop_n_iter_float32"])) <= 20 and int(res32.n_iter[0]) % 10 == 0
    for dt in ("float32", "float64"):
        Xd = np.asfortranarray(g7[f"X_{dt}"])
        T, m = Xd.shape
        l1w, l1h, l2w, l2h = orc.compute_regularization(T, m, 0.002, 0.001, 0.3)
        r = _fit(Xd, g7[f"W0_{dt}"], g7[f"H0_{dt}"], max_iter=40, tol=0.0, l1_reg_W=l1w, l1_reg_H=l1h,
                 l2_reg_W=l2w, l2_reg_H=l2h)
        assert _rel_wh(Xd, r.W[0], r.H[0], g7[f"W_reg_{dt}"], g7[f"H_reg_{dt}"]) <= TOL


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("T,m,k", [(1, 1, 1), (63, 3, 2), (1001, 8, 3), (777, 16, 8), (3000, 16, 5), (9000, 16, 5),
                                    (2000, 5, 1), (1500, 12, 7), (640, 20, 4), (500, 32, 6)])
def test_kl_shapes_vs_oracle(dtype, T, m, k):
    """Every lane mapping (m <= 4, 8, 16, 32), ragged tails, the LDS-resident / streamed split of W."""
    X = emg_matrix(900 + T + m, T=T, m=m, k_true=min(5, m), dtype=dtype)
    if T > 10:
        X[::7, 0] = 0
    W0, H0 = random_init(X, k, seed=3)
    res = _fit(X, W0, H0, max_iter=30, tol=0.0)
    ref = orc.nmf_mu_fit_kl(X, W0, H0, max_iter=30, tol=0)
    assert _rel_wh(X, res.W[0], res.H[0], ref["W"], ref["H"]) <= TOL
    e64 = orc.kl_divergence(X.astype(np.float64), res.W[0].astype(np.float64), res.H[0].astype(np.float64), True)
    # In float32 every term x log(x / wh) - x + wh cancels to ~eps32 * x (sklearn's own float32 evaluation does too):
    # the divergence carries an absolute error of a few eps32 * ||X||_2, which matters where it is ~0 (the 1 x 1 case)
    atol = 1e-6
    if dtype == np.float32:
        delta = 4 * np.finfo(np.float32).eps * float(np.linalg.norm(X.astype(np.float64)))
        atol = max(atol, min(np.sqrt(2 * delta), delta / max(e64, 1e-30)))
    np.testing.assert_allclose(float(res.reconstruction_err[0]), e64, rtol=2e-4 if dtype == np.float32 else 1e-10,
                               atol=atol)


def test_kl_batch_transform_estimator_and_unsupported_paths():
    sk = pytest.importorskip("sklearn.decomposition")
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    # a batch: every matrix equals its single-matrix fit bit for bit (one workgroup per matrix, fixed-order sums)
    Xs = np.stack([np.ascontiguousarray(emg_matrix(40 + b, T=2500, dtype=np.float32)) for b in range(6)])
    inits = [random_init(Xs[b], 4, seed=b) for b in range(6)]
    W0 = np.stack([w for w, _ in inits])
    H0 = np.stack([h for _, h in inits])
    rb = _fit(Xs, W0, H0, max_iter=50, tol=0.0)
    for b in (0, 5):
        r1 = _fit(Xs[b], W0[b], H0[b], max_iter=50, tol=0.0)
        assert np.array_equal(r1.W[0], rb.W[b]) and np.array_equal(r1.H[0], rb.H[b])

    # the estimator against live sklearn: fit (NNDSVDa start) and transform
    X = np.asfortranarray(emg_matrix(77, T=1500, dtype=np.float64))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mine = ms.HipNMF(5, solver="mu", beta_loss="kullback-leibler", init="nndsvda", max_iter=150, tol=1e-4,
                         random_state=0)
        Wm = mine.fit_transform(X)
        ref = sk.NMF(5, solver="mu", beta_loss="kullback-leibler", init="nndsvda", max_iter=150, tol=1e-4, random_state=0)
        Wr = ref.fit_transform(X)
    assert mine.n_iter_ == ref.n_iter_
    assert _rel_wh(X, Wm, mine.components_, Wr, ref.components_) <= TOL
    np.testing.assert_allclose(mine.reconstruction_err_, ref.reconstruction_err_, rtol=1e-7)
    Xn = np.asfortranarray(emg_matrix(78, T=700, dtype=np.float64))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Tm, Tr = mine.transform(Xn), ref.transform(Xn)
    assert _rel_wh(Xn, Tm, mine.components_, Tr, ref.components_) <= TOL
    assert ms.HipNMF.supports(solver="mu", beta_loss="kullback-leibler")
    assert not ms.HipNMF.supports(solver="mu", beta_loss="itakura-saito")

    # the time-shard building blocks are Frobenius only and say so
    import ctypes

    import torch

    from muscle_synergies_amd.engine import make_problem

    p = make_problem(1, 128, 4, 2, x_layout=_lib.X_CHANNEL_MAJOR, ldx=128, x_batch_stride=512, loss=_lib.LOSS_KL)
    h = _lib.get_handle(0)
    buf = torch.zeros(4096, device="cuda")
    rc = _lib.load().hipnmf_shard_pass_f32(h.ptr, ctypes.byref(p), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(),
                                           buf.data_ptr())
    assert rc == _lib.HIPNMF_ERR_UNSUPPORTED
    with pytest.raises(NotImplementedError):
        ms.fit_batched(X, Wm, mine.components_, beta_loss="itakura-saito")


def test_kl_ragged_trials_and_rank_sweep():
    """Trials of unequal length in one launch, and the batched rank sweep, under the KL loss."""
    import pandas as pd
    import torch

    import muscle_synergies_amd as ms

    lens = [700, 1234, 64, 2999]
    Xs = [np.ascontiguousarray(emg_matrix(300 + i, T=t, m=8, k_true=3, dtype=np.float64)) for i, t in enumerate(lens)]
    inits = [random_init(x, 3, seed=i) for i, x in enumerate(Xs)]
    res = ms.fit_ragged(Xs, [w for w, _ in inits], [h for _, h in inits], max_iter=60, tol=0.0,
                        beta_loss="kullback-leibler")
    for b, x in enumerate(Xs):
        ref = orc.nmf_mu_fit_kl(x, inits[b][0], inits[b][1], max_iter=60, tol=0)
        assert _rel_wh(x, res.W[b].cpu().numpy(), res.H[b].cpu().numpy(), ref["W"], ref["H"]) <= 1e-9
        np.testing.assert_allclose(float(res.reconstruction_err[b]), float(ref["reconstruction_err"]), rtol=1e-9)
    cols = [f"m{j}" for j in range(8)]
    dfs = [pd.DataFrame(x, columns=cols) for x in Xs[:2]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = ms.find_synergies_batched(dfs, 2, 3, max_iter=100, tol=0.0, beta_loss="kullback-leibler", random_state=0)
        one = [ms.find_synergies(df, 2, 3, solver="mu", beta_loss="kullback-leibler", max_iter=100, tol=0.0,
                                 random_state=0) for df in dfs]
    for g, r in zip(got, one):
        np.testing.assert_allclose(g.vaf_values.to_numpy(), r.vaf_values.to_numpy(), atol=1e-9)
        assert g.model[3].beta_loss == "kullback-leibler"
    Xb = torch.from_numpy(np.stack([emg_matrix(400 + b, T=1500, dtype=np.float32) for b in range(5)])).cuda()
    sw = ms.rank_sweep_batched(Xb, 2, 5, max_iter=100, tol=0.0, beta_loss="kullback-leibler")
    assert tuple(sw.vaf_all.shape) == (5, 4) and bool(torch.isfinite(sw.vaf_all).all())
    assert bool((sw.vaf_all[:, 1:] >= sw.vaf_all[:, :-1] - 1e-2).all())


def test_kl_float64_wide_with_stop_rule():
    """Regression (tests/fuzz_gpu.py seed 12 case 69, tools/repro/case69.py): float64, 17-32 features, KL loss, stop rule
    live, short matrices -- fit_persistent_kernel<double,4,8,K,1> never returned while the residual's logarithm sat
    inside a divergent branch (the instance spills ~1 KB per lane; a spill placed inside that region lost the
    inactive lanes of a buffer descriptor).  Branch-free now; checked against the oracle incl. the iteration count."""
    import muscle_synergies_amd as ms

    for (T, m, k, B) in [(64, 24, 7, 7), (64, 24, 4, 1), (128, 17, 7, 3), (63, 32, 8, 2), (300, 24, 2, 2)]:
        Xs = [emg_matrix(6900 + b, T=T, m=m, k_true=5, dtype=np.float64) for b in range(B)]
        inits = [random_init(x, k, seed=69 + b) for b, x in enumerate(Xs)]
        W0, H0 = np.stack([w for w, _ in inits]), np.stack([h for _, h in inits])
        got = ms.fit_batched(np.stack(Xs), W0, H0, max_iter=120, tol=1e-3, beta_loss="kullback-leibler")
        for b in range(B):
            Wo, Ho, n_it = orc.fit_multiplicative_update_kl(Xs[b], W0[b].copy(), H0[b].copy(), 120, 1e-3)
            assert int(got.n_iter[b]) == n_it, (T, m, k, b)
            np.testing.assert_allclose(np.asarray(got.W[b]) @ np.asarray(got.H[b]), Wo @ Ho, rtol=1e-9, atol=1e-12)
            assert abs(float(got.reconstruction_err[b]) - orc.kl_divergence(Xs[b], Wo, Ho, square_root=True)) <= 1e-9 * np.linalg.norm(Xs[b])


def test_unsupported_configuration_falls_back_to_sklearn_loudly(monkeypatch):
    """HipNMF runs scikit-learn's mu solver from the same starting point, with a RuntimeWarning, when the library answers
    HIPNMF_ERR_UNSUPPORTED (exercised by making the engine refuse)."""
    import warnings

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib, engine

    def refuse(*a, **kw):
        raise _lib.HipNmfError(_lib.HIPNMF_ERR_UNSUPPORTED, "not in this build")

    monkeypatch.setattr(engine, "fit_batched", refuse)
    X = emg_matrix(4242, T=300, m=12, k_true=5, dtype=np.float64)
    W0, H0 = random_init(X, 4, seed=1)
    model = ms.HipNMF(n_components=4, init="custom", solver="mu", beta_loss="kullback-leibler", tol=1e-4, max_iter=300)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        W = model.fit_transform(X, W=W0.copy(), H=H0.copy())
    assert any("running scikit-learn on the CPU instead" in str(w.message) for w in rec)
    Ws, Hs, n_it = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 300, 1e-4)
    assert model.n_iter_ == n_it
    np.testing.assert_allclose(W @ model.components_, Ws @ Hs, rtol=1e-8, atol=1e-11)
    assert model.vaf_.shape == (13,)


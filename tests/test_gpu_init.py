"""On-device NNDSVD / NNDSVDa (row f-2) against sklearn's _initialize_nmf and the host restatement."""
import numpy as np
import pytest

from muscle_synergies_amd.synth import emg_matrix

pytestmark = pytest.mark.gpu


# The device path takes the singular triplets from the exact Gram-matrix SVD; sklearn's randomized SVD
# (k + 10 = 15 random vectors, 4 power iterations) is itself only accurate to ~1e-5 on the tightly clustered
# trailing singular vectors of a 16-channel matrix (measured: 8.7e-6), and exact when k + 10 >= m.
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-4), (np.float32, 2e-3)])
@pytest.mark.parametrize("init", ["nndsvd", "nndsvda"])
def test_device_nndsvd_matches_host(dtype, tol, init):
    from muscle_synergies_amd.init import initialize_nmf, nndsvd_init_batched

    shapes = [(2000, 16, 5), (2000, 16, 1), (500, 8, 3), (77, 6, 6), (4000, 20, 4)]
    for T, m, k in shapes:
        Xs = np.stack([np.ascontiguousarray(emg_matrix(500 + b, T=T, m=m, k_true=min(5, m), dtype=dtype)) for b in range(3)])
        W0, H0 = nndsvd_init_batched(Xs, k, init=init)
        W0, H0 = W0.cpu().numpy(), H0.cpu().numpy()
        assert W0.dtype == dtype and W0.shape == (3, T, k) and H0.shape == (3, k, m)
        for b in range(3):
            # (1) against the host algorithm fed with an exact LAPACK SVD: same triplets, tight tolerance
            We, He = initialize_nmf(Xs[b], k, init=init, svd_solver="exact")
            scale = max(np.abs(We).max(), np.abs(He).max())
            te = 1e-8 if dtype == np.float64 else tol
            # entries that sit at the 1e-6 truncation threshold may fall on either side of it
            close_w = np.isclose(W0[b], We, rtol=te, atol=te * scale)
            assert close_w.mean() > 0.999, (T, m, k, b, (~close_w).sum())
            assert np.isclose(H0[b], He, rtol=te, atol=te * scale).all(), (T, m, k, b)
            # (2) against sklearn's randomized SVD where that one is exact too (k + 10 >= m)
            if k + 10 >= m:
                Wr, Hr = initialize_nmf(Xs[b], k, init=init, random_state=0)
                tl = 1e-9 if dtype == np.float64 else tol
                assert np.isclose(W0[b], Wr, rtol=tl, atol=tl * scale).mean() > 0.999, (T, m, k, b)
                assert np.isclose(H0[b], Hr, rtol=tl, atol=tl * scale).all(), (T, m, k, b)
            assert (W0[b] >= 0).all() and (H0[b] >= 0).all()


def test_device_nndsvda_against_live_sklearn_and_as_a_starting_point():
    sk = pytest.importorskip("sklearn.decomposition._nmf")
    import muscle_synergies_amd as ms
    from muscle_synergies_amd.init import nndsvd_init_batched
    from oracle import nmf_mu_oracle as orc

    Xs = np.stack([np.ascontiguousarray(emg_matrix(600 + b, T=3000, dtype=np.float64)) for b in range(4)])
    W0, H0 = nndsvd_init_batched(Xs, 5, init="nndsvda")
    for b in range(4):
        Ws, Hs = sk._initialize_nmf(Xs[b], 5, init="nndsvda", random_state=0)
        # sklearn's randomized SVD (15 random vectors for 16 channels) is itself only ~1e-4 accurate here
        np.testing.assert_allclose(H0[b].cpu().numpy(), Hs, rtol=1e-3, atol=1e-3)
        assert np.isclose(W0[b].cpu().numpy(), Ws, rtol=1e-3, atol=1e-3).mean() > 0.999  # threshold straddlers
    res = ms.fit_batched(Xs, W0, H0, max_iter=100, tol=0.0)
    for b in range(4):  # the solver itself, from the device-made starting point, against the oracle
        ref = orc.nmf_mu_fit(Xs[b], W0[b].cpu().numpy(), H0[b].cpu().numpy(), max_iter=100, tol=0.0)
        assert abs(float(res.reconstruction_err[b]) - float(ref["reconstruction_err"])) / np.linalg.norm(Xs[b]) <= 1e-9
        Ws, Hs = sk._initialize_nmf(Xs[b], 5, init="nndsvda", random_state=0)
        ref_sk = orc.nmf_mu_fit(Xs[b], Ws, Hs, max_iter=100, tol=0.0)  # and it lands where sklearn's start lands
        assert abs(float(res.reconstruction_err[b]) - float(ref_sk["reconstruction_err"])) / np.linalg.norm(Xs[b]) <= 1e-4
    with pytest.raises(ValueError, match="Negative values"):
        nndsvd_init_batched(-Xs, 5)
    with pytest.raises(ValueError, match="can only be used when"):
        nndsvd_init_batched(Xs, 17)


def test_rank_sweep_and_batched_synergies_with_device_nndsvda():
    import warnings

    import pandas as pd
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd.synth import emg_batch

    Xb = emg_batch(range(70, 76), T=1200, m=16, k_true=4)
    X = torch.from_numpy(Xb).cuda().transpose(1, 2)
    a = ms.rank_sweep_batched(X, 2, 5, max_iter=100, tol=0.0, init="nndsvda")
    assert tuple(a.vaf_all.shape) == (6, 4) and bool(torch.isfinite(a.vaf_all).all())
    assert float(a.vaf_all[:, -1].min()) > 0.9
    cols = [f"m{j}" for j in range(16)]
    dfs = [pd.DataFrame(Xb[b].T.astype(np.float64), columns=cols) for b in range(3)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = ms.find_synergies_batched(dfs, 3, max_iter=200, tol=0.0)           # device NNDSVDa (equal lengths)
        ref = [ms.find_synergies(df, 3, solver="mu", max_iter=200, tol=0.0, random_state=0) for df in dfs]
    for g, r in zip(got, ref):  # same optimum up to sklearn's randomized-SVD error in the starting point
        np.testing.assert_allclose(g.vaf_values.to_numpy(), r.vaf_values.to_numpy(), atol=1e-4)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("T,m,k", [(1500, 64, 8), (900, 40, 12), (1200, 128, 16), (700, 100, 24), (300, 33, 9)])
def test_device_nndsvd_wide_shapes_match_the_host_algorithm(dtype, T, m, k):
    """Round 4: the Gram / projection kernels take any number of channels and components (blocks of 32 x 32 entries / 8
    components); the default init of find_synergies (_nmf.py:296-300) for 33..128 channels no longer runs a randomized SVD per
    matrix on the host."""
    from muscle_synergies_amd.init import initialize_nmf, nndsvd_init_batched

    Xs = np.stack([np.ascontiguousarray(emg_matrix(900 + b, T=T, m=m, k_true=min(k, 6), dtype=dtype)) for b in range(2)])
    for init in ("nndsvd", "nndsvda"):
        W0, H0 = nndsvd_init_batched(Xs, k, init=init)
        W0, H0 = W0.cpu().numpy(), H0.cpu().numpy()
        assert W0.dtype == dtype and W0.shape == (2, T, k) and H0.shape == (2, k, m)
        assert (W0 >= 0).all() and (H0 >= 0).all() and np.isfinite(W0).all()
        for b in range(2):
            We, He = initialize_nmf(Xs[b], k, init=init, svd_solver="exact")
            # the leading triplets (well separated) to the Gram route's accuracy; what matters downstream is the product
            te = 1e-6 if dtype == np.float64 else 5e-3
            scale = max(np.abs(We).max(), np.abs(He).max())
            rel = np.linalg.norm(W0[b].astype(np.float64) @ H0[b] - We.astype(np.float64) @ He) / np.linalg.norm(We.astype(np.float64) @ He)
            # (float32: LAPACK's own float32 SVD of the host path is only ~1e-3 accurate on the noise-level trailing components)
            assert rel <= (1e-6 if dtype == np.float64 else 1e-2), (init, b, rel)
            assert np.isclose(H0[b][:3], He[:3], rtol=te, atol=te * scale).all(), (init, b)


def test_find_synergies_batched_on_wide_frames():
    """analysis.py took the device-NNDSVD branch for equal-length trials without looking at the channel count and the init
    kernels stopped at 32 channels (round 3: HipNmfError for 33..128 channels).  64 channels, ranks 6..9, equal and unequal lengths,
    against per-trial find_synergies calls and -- for the device-made starting point -- the oracle."""
    import warnings

    import pandas as pd

    import muscle_synergies_amd as ms
    from muscle_synergies_amd.init import nndsvd_init_batched
    from oracle import nmf_mu_oracle as orc

    m = 64
    cols = [f"ch{j}" for j in range(m)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for lengths in ((800, 800, 800), (700, 950, 820)):
            dfs = [pd.DataFrame(emg_matrix(40 + i, T=T, m=m, k_true=6, dtype=np.float64), columns=cols) for i, T in enumerate(lengths)]
            got = ms.find_synergies_batched(dfs, 6, 9, max_iter=60, tol=0.0, random_state=0)
            assert len(got) == 3
            for df, res in zip(dfs, got):
                assert sorted(res.components) == [6, 7, 8, 9] and res.components[9].shape == (9, m)
                assert list(res.vaf_values.columns[:1]) == ["All signals"] and res.vaf_values.shape == (4, 1 + m)
                assert np.isfinite(res.vaf_values.to_numpy()).all()
                if len(set(lengths)) > 1:  # host initialisation per trial (sklearn's randomized SVD, seeded): the single-frame call's numbers
                    one = ms.find_synergies(df, 6, 9, solver="mu", max_iter=60, tol=0.0, random_state=0)
                    for r in (6, 9):
                        np.testing.assert_allclose(res.components[r].to_numpy(), one.components[r].to_numpy(), rtol=1e-7, atol=1e-10)
            if len(set(lengths)) == 1:  # device NNDSVDa: the fit from that starting point against the oracle
                X = np.stack([df.to_numpy() for df in dfs])
                W0, H0 = nndsvd_init_batched(X, 8, init="nndsvda")
                ref = orc.nmf_mu_fit(X[1], W0[1].cpu().numpy(), H0[1].cpu().numpy(), max_iter=60, tol=0.0)
                np.testing.assert_allclose(got[1].components[8].to_numpy(), ref["H"], rtol=1e-7, atol=1e-10)

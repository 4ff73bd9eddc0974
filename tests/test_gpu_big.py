"""GPU parity of the general-shape kernels (csrc/nmf_big.hpp): more than 128 channels or 32 components, and float64 with more
than 16 components on more than 64 channels -- every shape the reference's validation accepts (analysis.py:829-846) up to
512 x 64 -- against the NumPy restatement of sklearn's loop (_nmf.py:540-554, 638-640, 827-884), through the same host API."""
import numpy as np
import pytest

from oracle import nmf_mu_oracle as orc
from muscle_synergies_amd.synth import emg_matrix, random_init

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rel(X, W, H, ref):
    xn = np.linalg.norm(X.astype(np.float64))
    return np.linalg.norm(W.astype(np.float64) @ H.astype(np.float64) - ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)) / xn


def _case(T, m, k, dtype, seed=0):
    X = emg_matrix(seed, T=T, m=m, k_true=min(8, m), dtype=dtype)
    W0, H0 = random_init(X, k, seed)
    return X, W0, H0


def _last_kernel():
    from muscle_synergies_amd import _lib

    return _lib.get_handle(0).last_kernel()


SHAPES = [(200, 12), (256, 16), (129, 3), (512, 64), (300, 40), (40, 33), (130, 17), (384, 7), (144, 48), (500, 1)]


@pytest.mark.parametrize("m,k", SHAPES)
@pytest.mark.parametrize("T", [7, 64, 333, 1100])
def test_big_shape_sweep_fp32(m, k, T):
    import muscle_synergies_amd as ms

    if k > min(T, m):
        pytest.skip("n_components > min(n_samples, n_features)")
    X, W0, H0 = _case(T, m, k, np.float32, seed=m + k)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=15, tol=0.0)
    for layout in ("F", "C"):
        Xl = np.asfortranarray(X) if layout == "F" else np.ascontiguousarray(X)
        res = ms.fit_batched(Xl, W0, H0, max_iter=15, tol=0.0)
        # 129..256 channels with at most 16 components keep the one-wave one-pass kernel (inst_wide_f32_xl.hip), the rest is the
        # workgroup-cooperative one-pass kernel of nmf_big1.hpp (round 5; the two-pass pair of nmf_big.hpp before)
        xl = 128 < m <= 256 and k <= 16
        assert _last_kernel().startswith("fit_wide_kernel<float,%d,16,4" % (160 if m <= 160 else 192 if m <= 192 else 256) if xl
                                         else "big1_pass_kernel<float,%d,%d," % ((k + 15) // 16 * 16, 1 if m <= 128 else 2 if m <= 256 else 4)), _last_kernel()
        assert int(res.n_iter[0]) == 15
        assert _rel(X, res.W[0], res.H[0], ref) <= TOL, (layout, m, k, T)
        assert abs(float(res.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL
        np.testing.assert_allclose(res.W[0], ref["W"], rtol=1e-3, atol=1e-6)
        np.testing.assert_allclose(res.H[0], ref["H"], rtol=1e-3, atol=1e-6)
        assert (res.W[0] >= 0).all() and (res.H[0] >= 0).all()
        va, vc = orc.vaf(X.astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
        assert abs(res.vaf[0, 0] - va) <= TOL
        np.testing.assert_allclose(res.vaf[0, 1:], vc, atol=5e-5)


@pytest.mark.parametrize("m,k,T", [(200, 12, 700), (256, 16, 500), (128, 24, 900), (96, 24, 640), (512, 64, 300), (65, 17, 130), (130, 33, 257)])
def test_big_shape_sweep_fp64(m, k, T):
    """Includes float64 with more than 16 components on 65..128 channels: the shapes whose 16x16x4 instance does not fit LDS."""
    import muscle_synergies_amd as ms

    X, W0, H0 = _case(T, m, k, np.float64, seed=9)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=40, tol=0.0)
    for layout in ("F", "C"):
        Xl = np.asfortranarray(X) if layout == "F" else np.ascontiguousarray(X)
        res = ms.fit_batched(Xl, W0, H0, max_iter=40, tol=0.0)
        # float64: the one-pass kernel up to 256 channels x 32 components (round 5), the two-pass pair beyond
        one_pass = m <= 256 and k <= 32
        assert _last_kernel().startswith("big1_pass_kernel<double" if one_pass else "big_pass_w_kernel<double"), _last_kernel()
        np.testing.assert_allclose(res.W[0], ref["W"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(res.H[0], ref["H"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(res.reconstruction_err[0], ref["reconstruction_err"], rtol=1e-9)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_big_stop_rule_batch_regularisation_and_transform(dtype):
    import muscle_synergies_amd as ms

    m, k = 200, 12
    Xs, Ws, Hs, refs = [], [], [], []
    for s in range(4):
        X, W0, H0 = _case(520, m, k, dtype, seed=60 + s)
        Xs.append(X), Ws.append(W0), Hs.append(H0)
        refs.append(orc.nmf_mu_fit(X, W0, H0, max_iter=300, tol=1e-3 if s % 2 else 3e-4))
    for tol, sel in ((1e-3, [1, 3]), (3e-4, [0, 2])):  # sklearn's stop rule per matrix of a batch (_nmf.py:872-884)
        res = ms.fit_batched(np.stack([Xs[i] for i in sel]), np.stack([Ws[i] for i in sel]), np.stack([Hs[i] for i in sel]),
                             max_iter=300, tol=tol)
        for q, i in enumerate(sel):
            assert int(res.n_iter[q]) == refs[i]["n_iter"], (i, int(res.n_iter[q]), refs[i]["n_iter"])
            assert _rel(Xs[i], res.W[q], res.H[q], refs[i]) <= (TOL if dtype == np.float32 else 1e-9)
    X, W0, H0 = Xs[0], Ws[0], Hs[0]
    regs = dict(l1_reg_W=0.02, l1_reg_H=0.03, l2_reg_W=0.05, l2_reg_H=0.01)
    Wr, Hr, _ = orc.fit_multiplicative_update(X, W0.copy(), H0.copy(), max_iter=30, tol=0.0, **regs)
    res = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0, **regs)
    assert _rel(X, res.W[0], res.H[0], {"W": Wr, "H": Hr}) <= (TOL if dtype == np.float32 else 1e-9)
    Wt = np.full_like(W0, np.sqrt(X.mean() / k))
    Wt_ref, _, _ = orc.fit_multiplicative_update(X, Wt.copy(), Hr.copy(), max_iter=25, tol=0.0, update_H=False)
    res_t = ms.fit_batched(X, Wt, Hr, max_iter=25, tol=0.0, update_H=False)
    np.testing.assert_array_equal(res_t.H[0], Hr)
    np.testing.assert_allclose(res_t.W[0], Wt_ref, rtol=1e-3 if dtype == np.float32 else 1e-9, atol=1e-7)


def test_big_long_frame_is_sliced_over_the_chip_and_deterministic():
    """One HD-EMG sized frame: 320 channels x 20 000 samples, 20 synergies, 200 iterations; twice, bitwise equal."""
    import muscle_synergies_amd as ms

    X, W0, H0 = _case(20_000, 320, 20, np.float32, seed=3)
    r1 = ms.fit_batched(np.ascontiguousarray(X), W0, H0, max_iter=200, tol=0.0)
    r2 = ms.fit_batched(np.ascontiguousarray(X), W0, H0, max_iter=200, tol=0.0)
    assert np.array_equal(r1.W, r2.W) and np.array_equal(r1.H, r2.H)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=200, tol=0.0)
    assert _rel(X, r1.W[0], r1.H[0], ref) <= TOL
    assert abs(float(r1.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL


@pytest.mark.parametrize("m,k,T,B", [(256, 16, 5000, 1), (256, 8, 1200, 300), (192, 12, 900, 40), (140, 5, 3000, 2)])
def test_hd_emg_grids_up_to_256_channels_on_the_one_pass_kernel(m, k, T, B):
    """fp32, 129..256 channels, at most 16 components: fit_wide_kernel<float, 160 / 192 / 256, 16, 4> (one workgroup per matrix
    for batches, rows sliced over the chip for few long frames), both losses, against the oracle at 60 iterations."""
    import muscle_synergies_amd as ms

    Xs, Ws, Hs = [], [], []
    for b in range(B):
        X, W0, H0 = _case(T, m, k, np.float32, seed=500 + b % 5)
        Xs.append(np.ascontiguousarray(X)), Ws.append(W0), Hs.append(H0)
    res = ms.fit_batched(np.stack(Xs), np.stack(Ws), np.stack(Hs), max_iter=60, tol=0.0)
    assert _last_kernel().startswith("fit_wide_kernel<float,%d,16,4" % (160 if m <= 160 else 192 if m <= 192 else 256)), _last_kernel()
    for b in (0, B - 1):
        ref = orc.nmf_mu_fit(Xs[b], Ws[b], Hs[b], max_iter=60, tol=0.0)
        assert _rel(Xs[b], res.W[b], res.H[b], ref) <= TOL
        assert abs(float(res.reconstruction_err[b]) - float(ref["reconstruction_err"])) / np.linalg.norm(Xs[b]) <= TOL
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(Xs[0], Ws[0].copy(), Hs[0].copy(), 15, 0.0)
    rk = ms.fit_batched(Xs[0], Ws[0], Hs[0], max_iter=15, tol=0.0, beta_loss="kullback-leibler")
    assert _rel(Xs[0], rk.W[0], rk.H[0], {"W": Wr, "H": Hr}) <= 3e-5


def test_find_synergies_on_an_hd_emg_grid_stays_on_the_gpu():
    """find_synergies(df, 4, 6, solver='mu') on 256 channels: HipNMF models, device NNDSVDa, no scikit-learn fallback warning."""
    import warnings

    import pandas as pd

    import muscle_synergies_amd as ms

    X = emg_matrix(12, T=1500, m=256, k_true=5, dtype=np.float64)
    df = pd.DataFrame(X, columns=[f"e{j}" for j in range(256)])
    with warnings.catch_warnings():
        warnings.simplefilter("error", category=RuntimeWarning)
        res = ms.find_synergies(df, 4, 6, solver="mu", max_iter=80, tol=0.0, init="random", random_state=1)
    assert all(isinstance(mdl, ms.HipNMF) for mdl in res.model.values())
    assert res.components[6].shape == (6, 256) and res.vaf_values.shape == (3, 257)
    W0, H0 = random_init(X, 5, 1)  # the same call by hand against the oracle (random_state=1 draws H then W per rank: use custom)
    r = ms.fit_batched(X, W0, H0, max_iter=80, tol=0.0)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=80, tol=0.0)
    np.testing.assert_allclose(r.H[0], ref["H"], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("dtype,m,k", [(np.float32, 64, 8), (np.float64, 64, 12), (np.float32, 200, 20)])
def test_time_sharded_fit_of_a_wide_recording_is_shard_count_invariant(dtype, m, k):
    """Row e-tshard for wide recordings (round 4: hipnmf_shard_* on the general-shape kernels): one 64-channel recording fitted
    unsharded, as 1, 2 and 3 time shards (Python-driven loop, sums added across shards on the device) and through the library's own
    hipnmf_fit_tsharded_* -- against the oracle and against each other."""
    import torch

    from muscle_synergies_amd.tsharded import HipShardOps, fit_tsharded

    T = 6000
    X = emg_matrix(31, T=T, m=m, k_true=6, dtype=dtype)
    W0, H0 = random_init(X, k, 2)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=40, tol=0.0)
    tol = TOL if dtype == np.float32 else 1e-9
    results = []
    for nshards in (1, 2, 3):
        bounds = np.linspace(0, T, nshards + 1).astype(int)
        shards = [HipShardOps(np.ascontiguousarray(X[lo:hi]), W0[lo:hi], H0) for lo, hi in zip(bounds[:-1], bounds[1:])]
        assert all(sh.wide for sh in shards)

        class Multi:  # the shards of ONE process standing in for ranks: sums added in shard order, H replicated by hand
            def shard_pass(self):
                tot = shards[0].shard_pass().clone()
                for sh in shards[1:]:
                    tot += sh.shard_pass()
                return tot

            def h_update(self, sums):
                for sh in shards:
                    sh.h_update(sums)

            def residual(self):
                sse, xsq = (t.clone() for t in shards[0].residual())
                for sh in shards[1:]:
                    a, b = sh.residual()
                    sse += a
                    xsq += b
                return sse, xsq

            def result_W(self):
                return torch.cat([sh.result_W() for sh in shards], dim=1)

            def result_H(self):
                return shards[0].H

        r = fit_tsharded(Multi(), max_iter=40, tol=0.0)
        W, H = r.W_local[0].cpu().numpy(), r.H[0].cpu().numpy()
        assert W.shape == (T, k) and _rel(X, W, H, ref) <= tol, nshards
        assert abs(float(r.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL
        results.append(W @ H)
    assert np.linalg.norm(results[0] - results[2]) / np.linalg.norm(X) <= (2e-6 if dtype == np.float32 else 1e-12)
    # the whole sharded fit as one library call, stop rule live
    ref_s = orc.nmf_mu_fit(X, W0, H0, max_iter=200, tol=1e-3)
    ops = HipShardOps(np.ascontiguousarray(X), W0, H0)
    rn = ops.fit_native(max_iter=200, tol=1e-3)
    assert rn.n_iter == ref_s["n_iter"]
    assert _rel(X, rn.W_local[0].cpu().numpy(), rn.H[0].cpu().numpy(), ref_s) <= tol


@pytest.mark.parametrize("dtype,m,k", [(np.float64, 200, 12), (np.float32, 300, 20), (np.float64, 100, 24)])
def test_ragged_batch_on_the_general_shape_kernels(dtype, m, k):
    """Trials of unequal length beyond the one-pass instances (round 4: fitted trial by trial inside hipnmf_fit_ragged_*, each a
    chip-filling row-sliced fit; the zero padding rows of the packed layout are inert): per-trial parity with the oracle, stop
    rule per trial, and `find_synergies_batched` on a 200-channel float64 recording without leaving the GPU."""
    import warnings

    import pandas as pd

    import muscle_synergies_amd as ms

    Ts = [333, 1000, 77, 2049]
    Xs, Ws, Hs = [], [], []
    for s, T in enumerate(Ts):
        X, W0, H0 = _case(T, m, k, dtype, seed=50 + s)
        Xs.append(X), Ws.append(W0), Hs.append(H0)
    res = ms.fit_ragged(Xs, Ws, Hs, max_iter=30, tol=0.0)
    assert _last_kernel().startswith("big1_pass_kernel<float" if dtype == np.float32 else "big1_pass_kernel<double"), _last_kernel()
    tol = TOL if dtype == np.float32 else 1e-9
    for b, T in enumerate(Ts):
        ref = orc.nmf_mu_fit(Xs[b], Ws[b], Hs[b], max_iter=30, tol=0.0)
        W, H = res.W[b].cpu().numpy(), res.H[b].cpu().numpy()
        assert W.shape == (T, k) and _rel(Xs[b], W, H, ref) <= tol, b
        assert abs(float(res.reconstruction_err[b]) - float(ref["reconstruction_err"])) / np.linalg.norm(Xs[b]) <= TOL
    res = ms.fit_ragged(Xs, Ws, Hs, max_iter=300, tol=1e-3)
    for b in range(len(Ts)):
        ref = orc.nmf_mu_fit(Xs[b], Ws[b], Hs[b], max_iter=300, tol=1e-3)
        assert abs(int(res.n_iter[b]) - ref["n_iter"]) <= (10 if dtype == np.float32 else 0), b
    if dtype == np.float64 and m == 200:
        cols = [f"ch{j}" for j in range(m)]
        dfs = [pd.DataFrame(x, columns=cols) for x in Xs[:3]]
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)  # a fallback to scikit-learn warns
            warnings.simplefilter("ignore", category=UserWarning)
            try:
                from sklearn.exceptions import ConvergenceWarning
                warnings.simplefilter("ignore", category=ConvergenceWarning)
            except ImportError:
                pass
            got = ms.find_synergies_batched(dfs, 3, 4, max_iter=60, tol=0.0, random_state=0)
        assert _last_kernel().startswith("big1_pass_kernel<double"), _last_kernel()
        for g in got:
            assert g.vaf_values.shape[0] == 2 and np.isfinite(g.vaf_values.to_numpy()).all()


@pytest.mark.parametrize("dtype,m,k,T", [(np.float64, 200, 12, 700), (np.float32, 300, 20, 1000), (np.float64, 100, 24, 333),
                                         (np.float32, 512, 64, 150), (np.float32, 129, 17, 2100), (np.float64, 136, 3, 4097)])
def test_kullback_leibler_on_the_general_shape_kernels(dtype, m, k, T):
    """beta_loss='kullback-leibler' beyond 128 channels / 32 components (round 4: Q = X / WH from the pipe inside
    big_pass_w_kernel / big_records_kernel, colsum(W) through a column of ones, the divergence per column in the residual
    kernel) against the oracle's restatement of _nmf.py:556-591, 642-684: fixed iterations, both layouts, the error, stop rule,
    regularisation, transform; the estimator on a 200-channel frame stays on the GPU."""
    import warnings

    import muscle_synergies_amd as ms

    X, W0, H0 = _case(T, m, k, dtype, seed=2 * m + k)
    tol = 3e-5 if dtype == np.float32 else 1e-9
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 20, 0.0)
    for layout in ("F", "C"):
        Xl = np.asfortranarray(X) if layout == "F" else np.ascontiguousarray(X)
        res = ms.fit_batched(Xl, W0, H0, max_iter=20, tol=0.0, beta_loss="kullback-leibler")
        # fp32: the one-pass kernel's Kullback-Leibler flavour (round 5) wherever both operand layouts of H fit LDS; float64 and
        # 48 / 64 padded components on more than 256 channels: the two-pass pair
        kp = (k + 15) // 16 * 16
        one_pass = (not (kp >= 48 and m > 256)) if dtype == np.float32 else (kp <= 32 and m <= 256 and not (kp == 32 and m > 128))
        assert _last_kernel().startswith("big1_pass_kernel<" if one_pass else "big_pass_w_kernel"), _last_kernel()
        assert (",1>[sliced]" in _last_kernel()) == one_pass, _last_kernel()
        assert _rel(X, res.W[0], res.H[0], {"W": Wr, "H": Hr}) <= tol, layout
        err = orc.kl_divergence(X, Wr, Hr, square_root=True)
        assert abs(float(res.reconstruction_err[0]) - err) <= (5e-3 if dtype == np.float32 else 1e-9) * max(err, 1e-30)
        assert (res.W[0] >= 0).all() and (res.H[0] >= 0).all()
    Ws, Hs, n_it = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 150, 1e-3, 0.01, 0.02, 0.03, 0.01)
    res = ms.fit_batched(X, W0, H0, max_iter=150, tol=1e-3, beta_loss="kullback-leibler", l1_reg_W=0.01, l1_reg_H=0.02,
                         l2_reg_W=0.03, l2_reg_H=0.01)
    if dtype == np.float64:
        assert int(res.n_iter[0]) == n_it
        assert _rel(X, res.W[0], res.H[0], {"W": Ws, "H": Hs}) <= 1e-9
    else:
        assert abs(int(res.n_iter[0]) - n_it) <= 10
    Wt = np.full_like(W0, np.sqrt(X.mean() / k))
    Wt_ref, _, _ = orc.fit_multiplicative_update_kl(X, Wt.copy(), Hr.copy(), 15, 0.0, update_H=False)
    rt = ms.fit_batched(X, Wt, Hr, max_iter=15, tol=0.0, beta_loss="kullback-leibler", update_H=False)
    np.testing.assert_array_equal(rt.H[0], Hr)
    np.testing.assert_allclose(rt.W[0], Wt_ref, rtol=2e-3 if dtype == np.float32 else 1e-8, atol=1e-6 if dtype == np.float32 else 1e-12)
    if m == 200:
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)  # a fallback to scikit-learn warns
            try:
                from sklearn.exceptions import ConvergenceWarning
                warnings.simplefilter("ignore", category=ConvergenceWarning)
            except ImportError:
                pass
            model = ms.HipNMF(n_components=k, init="custom", solver="mu", beta_loss="kullback-leibler", max_iter=20, tol=0.0)
            Wm = model.fit_transform(X, W=W0.copy(), H=H0.copy())
        assert _last_kernel().startswith(("big_pass_w_kernel", "big1_pass_kernel")), _last_kernel()
        assert _rel(X, Wm, model.components_, {"W": Wr, "H": Hr}) <= tol


@pytest.mark.parametrize("dtype,m,k", [(np.float64, 16, 5), (np.float32, 64, 8), (np.float64, 200, 12)])
def test_time_sharded_kullback_leibler_fit(dtype, m, k):
    """hipnmf_shard_* / hipnmf_fit_tsharded_* with HIPNMF_LOSS_KL (round 4: always on the general-shape kernels, whatever the
    shape): one recording as 1 and 3 time shards -- the Python-driven loop with the sums added across shards on the device,
    and the library's own loop -- against the oracle: factors, sqrt(2 KL), the squared-error VAF, the stop rule."""
    import torch

    from muscle_synergies_amd.tsharded import HipShardOps, fit_tsharded

    T = 3000
    X = emg_matrix(41, T=T, m=m, k_true=min(6, m), dtype=dtype)
    W0, H0 = random_init(X, k, 4)
    ref = orc.nmf_mu_fit_kl(X, W0, H0, max_iter=30, tol=0.0)
    tol = 3e-5 if dtype == np.float32 else 1e-9
    va, vc = orc.vaf(X.astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
    for nshards in (1, 3):
        bounds = np.linspace(0, T, nshards + 1).astype(int)
        shards = [HipShardOps(X[a:b], W0[a:b], H0, beta_loss="kullback-leibler") for a, b in zip(bounds[:-1], bounds[1:])]

        class Multi:
            kl = True

            def shard_pass(self):
                tot = shards[0].shard_pass().clone()
                for sh in shards[1:]:
                    tot += sh.shard_pass()
                return tot

            def h_update(self, sums):
                for sh in shards:
                    sh.h_update(sums)

            def _sum(self, which):
                sse, xsq = (t.clone() for t in getattr(shards[0], which)())
                for sh in shards[1:]:
                    a, b = getattr(sh, which)()
                    sse += a
                    xsq += b
                return sse, xsq

            def residual(self):
                return self._sum("residual")

            def residual_squared(self):
                return self._sum("residual_squared")

            def result_W(self):
                return torch.cat([sh.result_W() for sh in shards], dim=1)

            def result_H(self):
                return shards[0].H

        r = fit_tsharded(Multi(), max_iter=30, tol=0.0)
        W, H = r.W_local[0].cpu().numpy(), r.H[0].cpu().numpy()
        assert W.shape == (T, k) and _rel(X, W, H, ref) <= tol, nshards
        assert abs(float(r.reconstruction_err[0]) - float(ref["reconstruction_err"])) <= (5e-3 if dtype == np.float32 else 1e-9) * float(ref["reconstruction_err"])
        np.testing.assert_allclose(r.vaf[0].cpu().numpy(), np.r_[va, vc], atol=5e-5 if dtype == np.float32 else 1e-9)
    # the whole sharded fit as one library call, stop rule live
    ref_s = orc.nmf_mu_fit_kl(X, W0, H0, max_iter=200, tol=1e-3)
    ops = HipShardOps(np.ascontiguousarray(X), W0, H0, beta_loss="kullback-leibler")
    rn = ops.fit_native(max_iter=200, tol=1e-3)
    assert abs(rn.n_iter - ref_s["n_iter"]) <= (10 if dtype == np.float32 else 0)
    if dtype == np.float64:
        assert _rel(X, rn.W_local[0].cpu().numpy(), rn.H[0].cpu().numpy(), ref_s) <= 1e-9
        np.testing.assert_allclose(float(rn.reconstruction_err[0]), float(ref_s["reconstruction_err"]), rtol=1e-9)
        va_s, vc_s = orc.vaf(X, ref_s["W"], ref_s["H"])
        np.testing.assert_allclose(rn.vaf[0].cpu().numpy(), np.r_[va_s, vc_s], atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("m,k,T,B", [(9, 3, 60001, 1), (12, 6, 30000, 1), (16, 5, 40000, 1), (24, 8, 20000, 3), (33, 8, 9000, 2), (64, 8, 12000, 1),
                                     (100, 12, 6000, 2), (128, 6, 5000, 4)])
def test_kullback_leibler_few_long_matrices_take_the_row_sliced_kernel(dtype, m, k, T, B):
    """Round 5: the Kullback-Leibler loss has one workgroup per matrix on every other family; a few long matrices go to the one-pass
    general-shape kernel, which is row-sliced by construction (hipnmf_kl_row_sliced_wins, hipnmf_wide.hip) -- whatever their width.
    Route, parity with the oracle (fixed iteration count, then the stop rule with regularisation), transform."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    f32 = dtype == np.float32
    tol = 3e-5 if f32 else 1e-10
    Xs = [emg_matrix(13 * m + b, T=T, m=m, k_true=min(5, m), dtype=dtype) for b in range(B)]
    inits = [random_init(x, k, b) for b, x in enumerate(Xs)]
    W0, H0 = np.stack([w for w, _ in inits]), np.stack([h for _, h in inits])
    res = ms.fit_batched(np.stack(Xs), W0, H0, max_iter=20, tol=0.0, beta_loss="kullback-leibler")
    name = _lib.get_handle(0).last_kernel()
    assert name.startswith("big1_pass_kernel<%s,16," % ("float" if f32 else "double")) and name.endswith(",1>[sliced]"), name
    for b in range(B):
        Wr, Hr, _ = orc.fit_multiplicative_update_kl(Xs[b], W0[b].copy(), H0[b].copy(), 20, 0.0)
        xn = np.linalg.norm(Xs[b].astype(np.float64))
        d = np.linalg.norm(res.W[b].astype(np.float64) @ res.H[b].astype(np.float64) - Wr.astype(np.float64) @ Hr.astype(np.float64)) / xn
        assert d <= tol, (b, d)
        ref_err = np.sqrt(2 * max(orc.kl_divergence(Xs[b].astype(np.float64), Wr.astype(np.float64), Hr.astype(np.float64)), 0.0))
        assert abs(float(res.reconstruction_err[b]) - ref_err) <= (2e-3 if f32 else 1e-9) * max(ref_err, 1.0)
    regs = dict(l1_reg_W=0.01, l1_reg_H=0.02, l2_reg_W=0.03, l2_reg_H=0.01)
    Wr, Hr, n_it = orc.fit_multiplicative_update_kl(Xs[0], W0[0].copy(), H0[0].copy(), 100, 1e-3, *regs.values())
    r = ms.fit_batched(Xs[0], W0[0], H0[0], max_iter=100, tol=1e-3, beta_loss="kullback-leibler", **regs)
    assert _lib.get_handle(0).last_kernel().startswith("big1_pass_kernel<"), _lib.get_handle(0).last_kernel()
    if not f32:
        assert int(r.n_iter[0]) == n_it
        assert np.linalg.norm(r.W[0] @ r.H[0] - Wr @ Hr) / np.linalg.norm(Xs[0]) <= 1e-9
    else:
        assert abs(int(r.n_iter[0]) - n_it) <= 10
    Wt = np.full_like(W0[0], np.sqrt(Xs[0].mean() / k))
    Wt_ref, _, _ = orc.fit_multiplicative_update_kl(Xs[0], Wt.copy(), Hr.copy(), 15, 0.0, update_H=False)
    rt = ms.fit_batched(Xs[0], Wt, Hr, max_iter=15, tol=0.0, beta_loss="kullback-leibler", update_H=False)
    np.testing.assert_array_equal(rt.H[0], Hr)
    np.testing.assert_allclose(rt.W[0], Wt_ref, rtol=2e-3 if f32 else 1e-9, atol=1e-6 if f32 else 1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,m,k,T,B,kernel", [
    (np.float32, 2, 1, 30000, 1, "fit_persistent_kernel<float"),    # few channels: the lane mappings' one workgroup is faster than 256 CUs of padding
    (np.float32, 16, 5, 4000, 1, "fit_persistent_kernel<float"),
    (np.float64, 4, 2, 10000, 1, "fit_persistent_kernel<double"),
    (np.float32, 64, 8, 2500, 200, "fit_wide4_kernel<64,2"),         # a batch near one matrix per CU
    (np.float64, 64, 8, 200, 2, "fit_wide4d_kernel<64,2"),           # short matrices
])
def test_kullback_leibler_cost_model_leaves_these_on_one_workgroup_per_matrix(dtype, m, k, T, B, kernel):
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.synth import emg_batch

    X = np.ascontiguousarray(emg_batch(range(7, 7 + B), T=T, m=m, k_true=min(4, m)).astype(dtype).transpose(0, 2, 1))  # [B, T, m]
    inits = [random_init(X[b], k, b) for b in range(B)]
    res = ms.fit_batched(X, np.stack([w for w, _ in inits]), np.stack([h for _, h in inits]), max_iter=10, tol=0.0, beta_loss="kullback-leibler")
    assert _lib.get_handle(0).last_kernel().startswith(kernel), _lib.get_handle(0).last_kernel()
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(X[0], inits[0][0].copy(), inits[0][1].copy(), 10, 0.0)
    d = np.linalg.norm(res.W[0].astype(np.float64) @ res.H[0].astype(np.float64) - Wr.astype(np.float64) @ Hr.astype(np.float64)) / np.linalg.norm(X[0].astype(np.float64))
    assert d <= (3e-5 if dtype == np.float32 else 1e-10), d

"""GPU parity over shapes, layouts, dtypes and solver paths (persistent vs row-sliced), plus the
transform / regularisation branches and the time-shard building blocks."""
import numpy as np
import pytest

from oracle import nmf_mu_oracle as orc
from muscle_synergies_amd.synth import emg_matrix, random_init

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rel(X, W, H, ref):
    xn = np.linalg.norm(X.astype(np.float64))
    wh = W.astype(np.float64) @ H.astype(np.float64)
    wr = ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)
    return np.linalg.norm(wh - wr) / xn


def _case(T, m, k, dtype, seed=0):
    X = emg_matrix(seed, T=T, m=m, k_true=min(5, m), dtype=dtype)
    W0, H0 = random_init(X, k, seed)
    return X, W0, H0


@pytest.mark.parametrize("m,k", [(1, 1), (3, 2), (4, 4), (6, 3), (8, 8), (12, 5), (16, 1), (16, 5), (16, 8),
                                 (20, 4), (32, 6)])
@pytest.mark.parametrize("T", [7, 64, 1001])
def test_shape_sweep_fp32(m, k, T):
    import muscle_synergies_amd as ms

    X, W0, H0 = _case(T, m, k, np.float32, seed=m * 100 + k)
    for layout in ("F", "C"):
        Xl = np.asfortranarray(X) if layout == "F" else np.ascontiguousarray(X)
        res = ms.fit_batched(Xl, W0, H0, max_iter=20, tol=0.0)
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=20, tol=0.0)
        assert _rel(X, res.W[0], res.H[0], ref) <= TOL, (layout, m, k, T)
        assert abs(float(res.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL
        np.testing.assert_allclose(res.W[0], ref["W"], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(res.H[0], ref["H"], rtol=2e-4, atol=1e-6)
        assert (res.W[0] >= 0).all() and (res.H[0] >= 0).all()


@pytest.mark.parametrize("m,k,T", [(5, 2, 33), (8, 3, 200), (16, 5, 777), (24, 7, 130)])
def test_shape_sweep_fp64(m, k, T):
    import muscle_synergies_amd as ms

    X, W0, H0 = _case(T, m, k, np.float64, seed=7)
    res = ms.fit_batched(X, W0, H0, max_iter=40, tol=0.0)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=40, tol=0.0)
    np.testing.assert_allclose(res.W[0], ref["W"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(res.H[0], ref["H"], rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(res.reconstruction_err[0], ref["reconstruction_err"], rtol=1e-10)


def test_padded_row_major_leading_dimension():
    import torch

    import muscle_synergies_amd as ms

    X, W0, H0 = _case(500, 6, 3, np.float32, seed=3)
    big = torch.zeros((1, 500, 16), dtype=torch.float32, device="cuda")
    big[0, :, :6] = torch.from_numpy(np.ascontiguousarray(X)).cuda()
    view = big[:, :, :6]  # row-major with ldx = 16
    res = ms.fit_batched(view, torch.from_numpy(W0).cuda()[None], torch.from_numpy(H0).cuda()[None], max_iter=30, tol=0.0)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=30, tol=0.0)
    assert _rel(X, res.W[0].cpu().numpy(), res.H[0].cpu().numpy(), ref) <= TOL


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("tol", [0.0, 1e-3])
def test_sliced_path_equals_persistent_path(dtype, tol):
    """variant 2 = row-sliced launches (few matrices / long T), variant 1 = one workgroup per matrix."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    X, W0, H0 = _case(5000, 16, 5, dtype, seed=5)
    h = _lib.get_handle(0)
    out = {}
    try:
        for variant in (1, 2):
            h.set_tuning(0, 0, variant)
            out[variant] = ms.fit_batched(np.stack([X, X[::-1]]), np.stack([W0, W0[::-1]]), np.stack([H0, H0]),
                                          max_iter=200 if tol else 60, tol=tol)
    finally:
        h.set_tuning(0, 0, 0)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=200 if tol else 60, tol=tol)
    for variant in (1, 2):
        r = out[variant]
        assert int(r.n_iter[0]) == ref["n_iter"], variant
        assert _rel(X, r.W[0], r.H[0], ref) <= TOL
        assert abs(float(r.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL
        va, vc = orc.vaf(X.astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
        assert abs(r.vaf[0, 0] - va) <= TOL
    np.testing.assert_allclose(out[1].H[0], out[2].H[0], rtol=5e-4 if dtype == np.float32 else 1e-10)


@pytest.mark.parametrize("dt", ["float32", "float64"])
def test_transform_and_regularisation_vs_sklearn_fixtures(g5, g2_small, dt):
    import muscle_synergies_amd as ms

    X2 = np.asfortranarray(g5[f"X2_{dt}"])
    H = g5[f"H_fit_{dt}"]
    k = H.shape[0]
    W0 = np.full((X2.shape[0], k), np.sqrt(X2.mean() / k), dtype=X2.dtype)
    res = ms.fit_batched(X2, W0, H, max_iter=40, tol=0.0, update_H=False)
    rt = 5e-4 if dt == "float32" else 1e-9
    np.testing.assert_allclose(res.W[0], g5[f"W_transform_{dt}"], rtol=rt, atol=rt * 1e-2)
    np.testing.assert_array_equal(res.H[0], H)  # H untouched
    # estimator-level transform
    est = ms.HipNMF(k, init="custom", max_iter=40, tol=0.0)
    est.components_, est.n_components_, est.n_features_in_ = H, k, H.shape[1]
    np.testing.assert_allclose(est.transform(X2), g5[f"W_transform_{dt}"], rtol=rt, atol=rt * 1e-2)
    # L1 / L2 regularised fit (alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3)
    X = np.asfortranarray(g2_small[f"X_{dt}"])
    est = ms.HipNMF(5, init="custom", max_iter=60, tol=0.0, alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3)
    W = est.fit_transform(X, W=g2_small[f"W0_{dt}"], H=g2_small[f"H0_{dt}"])
    xn = np.linalg.norm(X.astype(np.float64))
    wh = W.astype(np.float64) @ est.components_.astype(np.float64)
    wh_ref = g5[f"W_reg_{dt}"].astype(np.float64) @ g5[f"H_reg_{dt}"].astype(np.float64)
    assert np.linalg.norm(wh - wh_ref) / xn <= TOL
    assert abs(float(est.reconstruction_err_) - float(g5[f"err_reg_{dt}"])) / xn <= TOL


def test_estimator_default_init_and_inverse_transform():
    import muscle_synergies_amd as ms

    X = emg_matrix(2, T=400, m=8, k_true=3, dtype=np.float64)
    est = ms.HipNMF(3, random_state=0, max_iter=100, tol=0.0)
    W = est.fit_transform(X)
    from muscle_synergies_amd.init import initialize_nmf

    W0, H0 = initialize_nmf(X, 3, init=None, random_state=0)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=100, tol=0.0)
    np.testing.assert_allclose(W, ref["W"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(est.components_, ref["H"], rtol=1e-9, atol=1e-13)
    assert est.n_iter_ == 100 and est.n_components_ == 3 and est.n_features_in_ == 8
    np.testing.assert_allclose(est.inverse_transform(W), W @ est.components_)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_shard_building_blocks_single_rank(dtype):
    """fit_tsharded driven by the hipnmf_shard_* entry points (one rank, no process group)."""
    from muscle_synergies_amd.tsharded import HipShardOps, fit_tsharded

    T = 3001  # padded to 3004 inside HipShardOps
    X, W0, H0 = _case(T, 16, 5, dtype, seed=9)
    ops = HipShardOps(np.ascontiguousarray(X), W0, H0)
    res = fit_tsharded(ops, max_iter=30, tol=0.0)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=30, tol=0.0)
    W = res.W_local.cpu().numpy()[0]
    H = res.H.cpu().numpy()[0]
    assert W.shape == (T, 5)
    assert _rel(X, W, H, ref) <= TOL
    assert abs(float(res.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL
    ops = HipShardOps(np.ascontiguousarray(X), W0, H0)
    res2 = fit_tsharded(ops, max_iter=300, tol=1e-3)
    ref2 = orc.nmf_mu_fit(X, W0, H0, max_iter=300, tol=1e-3)
    assert res2.n_iter == ref2["n_iter"]


def test_two_shards_on_one_gpu_sum_to_the_unsharded_result():
    """Emulates two ranks on one device: the all-reduce is replaced by an explicit sum of both shards."""
    import torch

    from muscle_synergies_amd.tsharded import HipShardOps, shard_bounds

    T = 4000
    X, W0, H0 = _case(T, 16, 5, np.float64, seed=13)
    (a0, a1), (b0, b1) = shard_bounds(T, 2)
    A = HipShardOps(np.ascontiguousarray(X[a0:a1]), W0[a0:a1], H0)
    B = HipShardOps(np.ascontiguousarray(X[b0:b1]), W0[b0:b1], H0)
    for _ in range(20):
        s = A.shard_pass().clone() + B.shard_pass()
        A.h_update(s)
        B.h_update(s)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=20, tol=0.0)
    W = torch.cat([A.result_W(), B.result_W()], dim=1).cpu().numpy()[0]
    np.testing.assert_allclose(W, ref["W"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(A.result_H().cpu().numpy()[0], ref["H"], rtol=1e-9, atol=1e-13)
    np.testing.assert_array_equal(A.result_H().cpu().numpy(), B.result_H().cpu().numpy())


def test_rank_sweep_matches_per_rank_fits():
    """Config #4 in miniature: k = 2..6 on a batch of trials, VAF threshold selection."""
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd.synth import emg_batch

    Xb = emg_batch(range(40, 52), T=1500, m=16, k_true=4)  # [B, m, T]
    X = torch.from_numpy(Xb).cuda().transpose(1, 2)
    res = ms.rank_sweep_batched(X, 2, 6, vaf_threshold=0.9, max_iter=150, tol=0.0, seed=3)
    assert res.ranks == [2, 3, 4, 5, 6] and tuple(res.vaf_all.shape) == (12, 5)
    v = res.vaf_all.cpu().numpy()
    assert (np.diff(v, axis=1) > -5e-3).all()  # VAF grows with the rank (up to local-minimum noise)
    sel = res.selected.cpu().numpy()
    for b in range(12):
        ok = np.nonzero(v[b] >= 0.9)[0]
        assert sel[b] == (res.ranks[ok[0]] if len(ok) else -1)
    # one (trial, rank) cell against the oracle from the same device-drawn initial factors
    k = 4
    W0, H0 = ms.random_init_batched(X, k, seed=3 + k)
    ref = orc.nmf_mu_fit(np.ascontiguousarray(Xb[5].T), W0[5].cpu().numpy(), H0[5].cpu().numpy(), max_iter=150, tol=0.0)
    va, _ = orc.vaf(Xb[5].T.astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
    assert abs(v[5, res.ranks.index(k)] - va) <= TOL
    with pytest.raises(ValueError, match="invalid number of components"):
        ms.rank_sweep_batched(X, 3, 17)


def test_oversize_matrix_is_rejected_not_mangled():
    """A single matrix >= 2 GiB cannot be addressed by one buffer resource: loud error, pointing at sharding."""
    import ctypes

    import torch

    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.engine import make_problem

    T = 40_000_000  # 16 x 4e7 x 4 B = 2.56 GB
    p = make_problem(1, T, 16, 5, x_layout=_lib.X_CHANNEL_MAJOR, ldx=T, x_batch_stride=16 * T)
    dummy = torch.zeros(16, device="cuda")
    h = _lib.get_handle(0)
    rc = _lib.load().hipnmf_fit_batched_f32(h.ptr, ctypes.byref(p), dummy.data_ptr(), dummy.data_ptr(), dummy.data_ptr(),
                                            None, None, None, None)
    assert rc == _lib.HIPNMF_ERR_UNSUPPORTED
    assert b"shard the time axis" in _lib.load().hipnmf_last_error()


def test_bad_arguments_are_reported():
    import ctypes

    import torch

    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.engine import make_problem

    lib, h = _lib.load(), _lib.get_handle(0)
    d = torch.zeros(1024, device="cuda")
    ok = dict(x_layout=_lib.X_ROW_MAJOR, ldx=8, x_batch_stride=80)
    for kwargs, code in ((dict(ok, max_iter=0), _lib.HIPNMF_ERR_BAD_ARG), (dict(ok, tol=-1.0), _lib.HIPNMF_ERR_BAD_ARG),
                         (dict(ok, ldx=4), _lib.HIPNMF_ERR_BAD_ARG)):
        p = make_problem(1, 10, 8, 2, **kwargs)
        assert lib.hipnmf_fit_batched_f32(h.ptr, ctypes.byref(p), d.data_ptr(), d.data_ptr(), d.data_ptr(), None, None,
                                          None, None) == code
    big = torch.zeros(10 * 600, device="cuda")
    for m, k in ((513, 2), (80, 65)):  # beyond the general-shape kernels (512 channels, 64 components)
        p = make_problem(1, 10, m, k, x_layout=_lib.X_ROW_MAJOR, ldx=m, x_batch_stride=10 * m)
        assert lib.hipnmf_fit_batched_f32(h.ptr, ctypes.byref(p), big.data_ptr(), big.data_ptr(), big.data_ptr(), None, None, None,
                                          None) == _lib.HIPNMF_ERR_UNSUPPORTED
    p = make_problem(1, 10, 40, 2, x_layout=_lib.X_CHANNEL_MAJOR, ldx=12, x_batch_stride=480)  # time-shard entries: narrow shapes only
    assert lib.hipnmf_shard_pass_f32(h.ptr, ctypes.byref(p), big.data_ptr(), big.data_ptr(), big.data_ptr(), big.data_ptr()) == \
        _lib.HIPNMF_ERR_UNSUPPORTED
    p = make_problem(1, 10, 8, 2, **ok)
    p.struct_size = 8
    assert lib.hipnmf_fit_batched_f32(h.ptr, ctypes.byref(p), d.data_ptr(), d.data_ptr(), d.data_ptr(), None, None, None,
                                      None) == _lib.HIPNMF_ERR_BAD_ARG
    assert lib.hipnmf_fit_batched_f32(h.ptr, ctypes.byref(make_problem(1, 10, 8, 2, **ok)), None, d.data_ptr(), d.data_ptr(),
                                      None, None, None, None) == _lib.HIPNMF_ERR_BAD_ARG


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ragged_batch_matches_per_matrix_oracle(dtype):
    """Row f-3: trials of unequal length in one launch (lengths deliberately not multiples of 4 / 64 / 512)."""
    import muscle_synergies_amd as ms

    Ts = [7, 64, 333, 1001, 2500, 9999, 10000, 13001]
    Xs = [emg_matrix(200 + i, T=t, m=16, dtype=dtype) for i, t in enumerate(Ts)]
    inits = [random_init(x, 5, i) for i, x in enumerate(Xs)]
    res = ms.fit_ragged(Xs, [w for w, _ in inits], [h for _, h in inits], max_iter=40, tol=0.0)
    assert len(res.W) == len(Ts)
    for b, t in enumerate(Ts):
        ref = orc.nmf_mu_fit(Xs[b], inits[b][0], inits[b][1], max_iter=40, tol=0.0)
        W, H = res.W[b].cpu().numpy(), res.H[b].cpu().numpy()
        assert W.shape == (t, 5)
        assert _rel(Xs[b], W, H, ref) <= TOL, t
        assert abs(float(res.reconstruction_err[b]) - float(ref["reconstruction_err"])) / np.linalg.norm(Xs[b]) <= TOL
        va, _ = orc.vaf(Xs[b].astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
        assert abs(float(res.vaf[b, 0]) - va) <= TOL
    # per-matrix stop rule with mixed lengths
    res2 = ms.fit_ragged(Xs[2:5], [w for w, _ in inits[2:5]], [h for _, h in inits[2:5]], max_iter=500, tol=1e-3)
    for i, b in enumerate(range(2, 5)):
        ref = orc.nmf_mu_fit(Xs[b], inits[b][0], inits[b][1], max_iter=500, tol=1e-3)
        assert int(res2.n_iter[i]) == ref["n_iter"]


def test_find_synergies_batched_equals_per_trial_calls():
    """A list of trials of unequal length through one launch per rank == find_synergies trial by trial."""
    import warnings

    import pandas as pd

    import muscle_synergies_amd as ms

    cols = [f"m{j}" for j in range(8)]
    dfs = [pd.DataFrame(emg_matrix(300 + i, T=t, m=8, k_true=3, dtype=np.float64), columns=cols)
           for i, t in enumerate((180, 200, 257))]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        batched = ms.find_synergies_batched(dfs, 2, 3, max_iter=300, tol=1e-5, random_state=0)
        single = [ms.find_synergies(df, 2, 3, solver="mu", max_iter=300, tol=1e-5, random_state=0) for df in dfs]
    assert len(batched) == 3
    for b, s in zip(batched, single):
        assert list(b.components.keys()) == [2, 3] and list(b.vaf_values.index) == [2, 3]
        np.testing.assert_allclose(b.vaf_values.to_numpy(), s.vaf_values.to_numpy(), rtol=1e-9)
        for k in (2, 3):
            np.testing.assert_allclose(b.components[k].to_numpy(), s.components[k].to_numpy(), rtol=1e-8, atol=1e-12)
            assert b.model[k].n_iter_ == s.model[k].n_iter_
            assert list(b.components[k].columns) == cols
    one = ms.find_synergies_batched(dfs, 2, max_iter=50, tol=0.0, random_state=0)
    assert isinstance(one[0].components, pd.DataFrame) and one[1].model.n_iter_ == 50
    with pytest.raises(ValueError, match="invalid number of components"):
        ms.find_synergies_batched(dfs, 9)


def test_multi_gpu_scatter_on_the_visible_devices():
    """Host-thread scatter (one handle per device, contiguous slices, no collective); with one visible GPU the
    batch goes to it in one slice, and a device listed twice exercises the two-thread path."""
    import torch

    import muscle_synergies_amd as ms

    B, T = 10, 800
    Xs = np.stack([np.ascontiguousarray(emg_matrix(400 + b, T=T, dtype=np.float32)) for b in range(B)])
    inits = [random_init(Xs[b], 5, b) for b in range(B)]
    W0, H0 = np.stack([i[0] for i in inits]), np.stack([i[1] for i in inits])
    ref = ms.fit_batched(Xs, W0, H0, max_iter=30, tol=0.0)
    out = ms.fit_batched_multi_gpu(Xs, W0, H0, max_iter=30, tol=0.0)
    np.testing.assert_array_equal(out.W, ref.W)
    if torch.cuda.device_count() == 1:
        two = ms.fit_batched_multi_gpu(Xs, W0, H0, devices=[0, 0], max_iter=30, tol=0.0)
        assert two.W.shape == ref.W.shape
        np.testing.assert_allclose(two.H, ref.H, rtol=2e-4, atol=1e-6)  # a 5-matrix slice may use another path
        np.testing.assert_array_equal(two.n_iter, ref.n_iter)


def test_multi_restart_best_of_r_matches_single_fits():
    """Row f-2: R random starts per trial in one launch (X shared through the per-matrix descriptors)."""
    import torch

    import muscle_synergies_amd as ms
    from oracle import nmf_mu_oracle as orc

    for T in (1000, 1001):  # leading dimension already a multiple of 4 (in place) / padded copy
        Xs = np.stack([np.ascontiguousarray(emg_matrix(800 + b, T=T, m=8, k_true=3, dtype=np.float64)) for b in range(3)])
        res = ms.fit_restarts(Xs, 3, n_restarts=5, seed=11, max_iter=80, tol=0.0)
        assert tuple(res.restart_err.shape) == (3, 5) and tuple(res.best.W.shape) == (3, T, 3)
        err = res.restart_err.cpu().numpy()
        assert np.array_equal(res.chosen.cpu().numpy(), err.argmin(axis=1))
        assert len(np.unique(np.round(err[0], 9))) > 1  # the restarts really start from different points
        for b in range(3):
            W, H = res.best.W[b].cpu().numpy(), res.best.H[b].cpu().numpy()
            assert (W >= 0).all() and (H >= 0).all()
            resid = np.linalg.norm(Xs[b] - W @ H)
            np.testing.assert_allclose(resid, err[b].min(), rtol=1e-9)
            np.testing.assert_allclose(float(res.best.reconstruction_err[b]), err[b].min(), rtol=0, atol=0)
            np.testing.assert_allclose(res.best.vaf[b, 0].item(), 1 - resid**2 / (Xs[b] ** 2).sum(), atol=1e-9)
            assert int(res.best.n_iter[b]) == 80
    # same seed, same answer; and one restart of the batch equals the plain batched fit from the same start
    again = ms.fit_restarts(Xs, 3, n_restarts=5, seed=11, max_iter=80, tol=0.0)
    assert torch.equal(again.restart_err, res.restart_err)
    kl = ms.fit_restarts(Xs.astype(np.float32), 2, n_restarts=3, max_iter=30, tol=0.0, beta_loss="kullback-leibler")
    assert bool(torch.isfinite(kl.restart_err).all())


@pytest.fixture
def tuned_handle():
    from muscle_synergies_amd import _lib

    h = _lib.get_handle(0)
    yield h
    h.set_tuning(0, 0, 0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("B,T,m,k", [(1, 10000, 16, 5), (1, 1001, 8, 3), (3, 5000, 12, 4), (2, 2500, 32, 6), (1, 130, 4, 2),
                                     (5, 20000, 16, 8), (1, 70000, 16, 5)])
def test_cooperative_path_matches_oracle_and_persistent(tuned_handle, dtype, B, T, m, k):
    """Kernel 1b: S workgroups per matrix with a grid barrier per iteration (few long matrices)."""
    import muscle_synergies_amd as ms
    from oracle import nmf_mu_oracle as orc

    Xs = np.stack([np.ascontiguousarray(emg_matrix(60 + b, T=T, m=m, k_true=min(5, m), dtype=dtype)) for b in range(B)])
    inits = [random_init(Xs[b], k, seed=b) for b in range(B)]
    W0, H0 = np.stack([w for w, _ in inits]), np.stack([h for _, h in inits])
    tuned_handle.set_tuning(0, 0, 3)
    got = ms.fit_batched(Xs, W0, H0, max_iter=40, tol=0.0)
    tuned_handle.set_tuning(0, 0, 1)
    per = ms.fit_batched(Xs, W0, H0, max_iter=40, tol=0.0)
    tol = 1e-5 if dtype == np.float32 else 1e-10
    for b in range(B):
        xn = np.linalg.norm(Xs[b].astype(np.float64))
        wh = got.W[b].astype(np.float64) @ got.H[b].astype(np.float64)
        assert np.linalg.norm(wh - per.W[b].astype(np.float64) @ per.H[b].astype(np.float64)) / xn <= tol
        if T <= 20000:
            ref = orc.nmf_mu_fit(Xs[b], W0[b], H0[b], max_iter=40, tol=0.0)
            assert np.linalg.norm(wh - ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)) / xn <= tol
            assert abs(float(got.reconstruction_err[b]) - float(ref["reconstruction_err"])) / xn <= tol
        np.testing.assert_allclose(got.vaf[b], per.vaf[b], atol=1e-5 if dtype == np.float32 else 1e-10)
        assert int(got.n_iter[b]) == 40
    # run-to-run determinism: fixed summation order across workgroups
    tuned_handle.set_tuning(0, 0, 3)
    again = ms.fit_batched(Xs, W0, H0, max_iter=40, tol=0.0)
    assert np.array_equal(again.W, got.W) and np.array_equal(again.H, got.H)


def test_cooperative_path_stop_rule_transform_and_auto_selection(tuned_handle):
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib
    from oracle import nmf_mu_oracle as orc

    X = np.ascontiguousarray(emg_matrix(5, T=8000, m=16, dtype=np.float64))
    W0, H0 = random_init(X, 5, seed=1)
    tuned_handle.set_tuning(0, 0, 3)
    got = ms.fit_batched(X, W0, H0, max_iter=600, tol=1e-4)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=600, tol=1e-4)
    assert int(got.n_iter[0]) == ref["n_iter"] and ref["n_iter"] < 600
    np.testing.assert_allclose(float(got.reconstruction_err[0]), float(ref["reconstruction_err"]), rtol=1e-9)
    # transform: H fixed
    tr = ms.fit_batched(X, np.full_like(W0, np.sqrt(X.mean() / 5)), got.H[0], max_iter=50, tol=0.0, update_H=False)
    Wt = np.full_like(W0, np.sqrt(X.mean() / 5))
    for _ in range(50):
        Wt = orc.multiplicative_update_w(X, Wt, got.H[0].copy())
    np.testing.assert_allclose(tr.W[0], Wt, rtol=1e-9, atol=1e-12)
    assert np.array_equal(tr.H[0], got.H[0])
    # regularised fit
    reg = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0, l1_reg_W=0.02, l1_reg_H=0.5, l2_reg_W=0.01, l2_reg_H=0.3)
    W, H, _ = orc.fit_multiplicative_update(X, W0.copy(), H0.copy(), 30, 0.0, 0.02, 0.5, 0.01, 0.3)
    np.testing.assert_allclose(reg.W[0] @ reg.H[0], W @ H, rtol=1e-8, atol=1e-10)
    # the library's own choice for one long matrix is this path (bitwise the same answer); the row-sliced
    # launches land on it too; where the cooperative path does not apply, forcing it is an error rather than a
    # silent substitution
    tuned_handle.set_tuning(0, 0, 0)
    auto = ms.fit_batched(X, W0, H0, max_iter=600, tol=1e-4)
    assert np.array_equal(auto.W, got.W) and int(auto.n_iter[0]) == ref["n_iter"]
    tuned_handle.set_tuning(0, 0, 2)
    sliced = ms.fit_batched(X, W0, H0, max_iter=600, tol=1e-4)
    np.testing.assert_allclose(sliced.W, got.W, rtol=1e-7, atol=1e-10)
    assert int(sliced.n_iter[0]) == ref["n_iter"]
    tuned_handle.set_tuning(0, 0, 3)
    Xb = np.stack([X[:640]] * 200)
    with pytest.raises(_lib.HipNmfError, match="cooperative path not applicable"):
        ms.fit_batched(Xb, np.stack([W0[:640]] * 200), np.stack([H0] * 200), max_iter=5, tol=0.0)


def test_cooperative_path_with_slices_larger_than_lds(tuned_handle):
    """One very long matrix: the slice of a workgroup exceeds LDS, part of its W streams from global memory."""
    import muscle_synergies_amd as ms

    T = 3_000_000
    X = np.ascontiguousarray(emg_matrix(9, T=T, m=8, k_true=3, dtype=np.float64))
    W0, H0 = random_init(X, 5, seed=2)
    tuned_handle.set_tuning(0, 0, 3)
    got = ms.fit_batched(X, W0, H0, max_iter=12, tol=0.0)
    tuned_handle.set_tuning(0, 0, 2)
    ref = ms.fit_batched(X, W0, H0, max_iter=12, tol=0.0)
    np.testing.assert_allclose(got.H[0], ref.H[0], rtol=1e-9)
    np.testing.assert_allclose(got.W[0][::997], ref.W[0][::997], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(float(got.reconstruction_err[0]), float(ref.reconstruction_err[0]), rtol=1e-10)
    # and a fully reduced check of the last rows (tail of the last slice) against the update rule itself
    from oracle import nmf_mu_oracle as orc

    tail = slice(T - 1000, T)
    Wt = orc.multiplicative_update_w(X[tail], ref.W[0][tail].copy(), ref.H[0].copy())
    one_more = ms.fit_batched(X, ref.W[0], ref.H[0], max_iter=1, tol=0.0, update_H=False)
    np.testing.assert_allclose(one_more.W[0][tail], Wt, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("T,m,k", [(1, 16, 1), (63, 9, 2), (65, 12, 3), (1001, 16, 5), (4096, 13, 4), (7777, 16, 5)])
def test_row_per_lane_instance_layouts_and_paths(tuned_handle, variant, T, m, k):
    """fp32 with 9..16 channels and k <= 5 runs on the row-per-lane kernels (row-major X, channels padded to 16):
    C-order input is streamed in place when m == 16, F-order / narrower input is converted once; every solver path."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib
    from oracle import nmf_mu_oracle as orc

    if variant == 3 and T < 128:
        pytest.skip("cooperative path needs at least two workgroup steps of rows")
    X = emg_matrix(70 + T + m, T=T, m=m, k_true=min(5, m), dtype=np.float32)  # F-contiguous
    W0, H0 = random_init(X, k, seed=5)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=25, tol=0.0)
    xn = max(np.linalg.norm(X.astype(np.float64)), 1e-30)
    tuned_handle.set_tuning(0, 0, variant)
    outs = []
    for arr in (X, np.ascontiguousarray(X)):  # channel-major and row-major memory order
        try:
            got = ms.fit_batched(arr, W0, H0, max_iter=25, tol=0.0)
        except _lib.HipNmfError as e:
            assert variant == 3 and "not applicable" in str(e)
            return
        wh = got.W[0].astype(np.float64) @ got.H[0].astype(np.float64)
        assert np.linalg.norm(wh - ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)) / xn <= 1e-5
        assert abs(float(got.reconstruction_err[0]) - float(ref["reconstruction_err"])) / xn <= 1e-5
        outs.append(got)
    assert np.array_equal(outs[0].W, outs[1].W) and np.array_equal(outs[0].H, outs[1].H)  # same kernel, same bits


def test_row_per_lane_instance_batch_with_padded_rows_and_strides():
    """A batch whose row-major rows are wider than m (ldx = 20 > 16: a view into a larger array) and a stop rule."""
    import torch

    import muscle_synergies_amd as ms
    from oracle import nmf_mu_oracle as orc

    B, T, m, k = 5, 3000, 16, 4
    big = np.zeros((B, T, 20), dtype=np.float32)
    for b in range(B):
        big[b, :, :m] = emg_matrix(900 + b, T=T, m=m, dtype=np.float32)
    Xv = torch.from_numpy(big).cuda()[:, :, :m]  # row stride 20 elements, in place (a multiple of 4)
    inits = [random_init(big[b, :, :m], k, seed=b) for b in range(B)]
    W0, H0 = np.stack([w for w, _ in inits]), np.stack([h for _, h in inits])
    got = ms.fit_batched(Xv, W0, H0, max_iter=400, tol=1e-4)
    for b in (0, 4):
        ref = orc.nmf_mu_fit(np.ascontiguousarray(big[b, :, :m]), W0[b], H0[b], max_iter=400, tol=1e-4)
        assert abs(int(got.n_iter[b]) - ref["n_iter"]) <= 10  # fp32 stop rule: same check or the neighbouring one
        xn = np.linalg.norm(big[b, :, :m].astype(np.float64))
        if int(got.n_iter[b]) == ref["n_iter"]:
            wh = got.W[b].cpu().numpy().astype(np.float64) @ got.H[b].cpu().numpy().astype(np.float64)
            assert np.linalg.norm(wh - ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)) / xn <= 1e-5

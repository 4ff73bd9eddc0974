"""GPU parity of the wide-shape matrix-pipe kernels (nmf_wide.hpp): n_features up to 128, n_components up to 16 --
every shape the reference's validation accepts (analysis.py:829-846) beyond the narrow lane mappings -- against the
NumPy oracle, through the same host API the narrow shapes use."""
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
    X = emg_matrix(seed, T=T, m=m, k_true=min(6, m), dtype=dtype)
    W0, H0 = random_init(X, k, seed)
    return X, W0, H0


def _last_kernel():
    from muscle_synergies_amd import _lib

    return _lib.get_handle(0).last_kernel()


SHAPES = [(33, 8), (40, 3), (48, 12), (64, 8), (64, 16), (65, 9), (96, 16), (100, 5), (128, 16), (128, 1),
          (16, 12), (12, 9), (24, 16), (32, 10), (9, 9)]


@pytest.mark.parametrize("m,k", SHAPES)
@pytest.mark.parametrize("T", [5, 16, 250, 1003])
def test_wide_shape_sweep_fp32(m, k, T):
    import muscle_synergies_amd as ms

    X, W0, H0 = _case(T, m, k, np.float32, seed=m * 100 + k)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=20, tol=0.0)
    for layout in ("F", "C"):
        Xl = np.asfortranarray(X) if layout == "F" else np.ascontiguousarray(X)
        res = ms.fit_batched(Xl, W0, H0, max_iter=20, tol=0.0)
        assert _last_kernel().startswith(("fit_wide_kernel<float", "fit_wide4_kernel<")), _last_kernel()
        assert int(res.n_iter[0]) == 20
        assert _rel(X, res.W[0], res.H[0], ref) <= TOL, (layout, m, k, T)
        assert abs(float(res.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL
        np.testing.assert_allclose(res.W[0], ref["W"], rtol=5e-4, atol=1e-6)
        np.testing.assert_allclose(res.H[0], ref["H"], rtol=5e-4, atol=1e-6)
        assert (res.W[0] >= 0).all() and (res.H[0] >= 0).all()
        va, vc = orc.vaf(X.astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
        assert abs(res.vaf[0, 0] - va) <= TOL
        np.testing.assert_allclose(res.vaf[0, 1:], vc, atol=5e-5)


@pytest.mark.parametrize("m,k,T", [(33, 8, 130), (64, 16, 777), (128, 16, 300), (100, 7, 64), (16, 12, 500), (48, 4, 33)])
def test_wide_shape_sweep_fp64(m, k, T):
    import muscle_synergies_amd as ms

    X, W0, H0 = _case(T, m, k, np.float64, seed=7)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=40, tol=0.0)
    for layout in ("F", "C"):
        Xl = np.asfortranarray(X) if layout == "F" else np.ascontiguousarray(X)
        res = ms.fit_batched(Xl, W0, H0, max_iter=40, tol=0.0)
        assert _last_kernel().startswith(("fit_wide_kernel<double", "fit_wide4d_kernel<")), _last_kernel()
        np.testing.assert_allclose(res.W[0], ref["W"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(res.H[0], ref["H"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(res.reconstruction_err[0], ref["reconstruction_err"], rtol=1e-9)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("m,k", [(64, 8), (40, 12)])
def test_wide_stop_rule_and_batch(dtype, m, k):
    """sklearn's stop rule (every 10th iteration, _nmf.py:872-884) per matrix of a batch whose members converge
    at different iterations."""
    import muscle_synergies_amd as ms

    Xs, Ws, Hs, refs = [], [], [], []
    for s in range(5):
        X, W0, H0 = _case(600, m, k, dtype, seed=40 + s)
        Xs.append(X), Ws.append(W0), Hs.append(H0)
        refs.append(orc.nmf_mu_fit(X, W0, H0, max_iter=300, tol=1e-3 if s % 2 else 3e-4))
    for tol, sel in ((1e-3, [1, 3]), (3e-4, [0, 2, 4])):
        res = ms.fit_batched(np.stack([Xs[i] for i in sel]), np.stack([Ws[i] for i in sel]), np.stack([Hs[i] for i in sel]),
                             max_iter=300, tol=tol)
        for q, i in enumerate(sel):
            assert int(res.n_iter[q]) == refs[i]["n_iter"], (i, int(res.n_iter[q]), refs[i]["n_iter"])
            assert _rel(Xs[i], res.W[q], res.H[q], refs[i]) <= (TOL if dtype == np.float32 else 1e-9)
            assert abs(float(res.reconstruction_err[q]) - float(refs[i]["reconstruction_err"])) / np.linalg.norm(Xs[i]) <= TOL


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_wide_transform_and_regularisation(dtype):
    import muscle_synergies_amd as ms

    X, W0, H0 = _case(400, 64, 10, dtype, seed=11)
    regs = dict(l1_reg_W=0.02, l1_reg_H=0.03, l2_reg_W=0.05, l2_reg_H=0.01)
    Wr, Hr, _ = orc.fit_multiplicative_update(X, W0.copy(), H0.copy(), max_iter=30, tol=0.0, **regs)
    res = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0, **regs)
    assert _rel(X, res.W[0], res.H[0], {"W": Wr, "H": Hr}) <= (TOL if dtype == np.float32 else 1e-9)
    Wt = np.full_like(W0, np.sqrt(X.mean() / 10))
    Wt_ref, _, _ = orc.fit_multiplicative_update(X, Wt.copy(), Hr.copy(), max_iter=25, tol=0.0, update_H=False)
    res_t = ms.fit_batched(X, Wt, Hr, max_iter=25, tol=0.0, update_H=False)
    np.testing.assert_array_equal(res_t.H[0], Hr)
    np.testing.assert_allclose(res_t.W[0], Wt_ref, rtol=5e-4 if dtype == np.float32 else 1e-9, atol=1e-7)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_wide_ragged_batch(dtype):
    """Trials of unequal length through hipnmf_fit_ragged_* (packed channel-major X, component-major W)."""
    import muscle_synergies_amd as ms

    m, k = 64, 9
    Ts = [17, 300, 64, 1025, 5]
    Xs, Ws, Hs = [], [], []
    for s, T in enumerate(Ts):
        X, W0, H0 = _case(T, m, k, dtype, seed=70 + s)
        Xs.append(X), Ws.append(W0), Hs.append(H0)
    res = ms.fit_ragged(Xs, Ws, Hs, max_iter=25, tol=0.0)
    assert _last_kernel().startswith(("fit_wide_kernel", "fit_wide4_kernel"))
    for b, T in enumerate(Ts):
        ref = orc.nmf_mu_fit(Xs[b], Ws[b], Hs[b], max_iter=25, tol=0.0)
        W = res.W[b].cpu().numpy()
        H = res.H[b].cpu().numpy()
        assert W.shape == (T, k)
        assert _rel(Xs[b], W, H, ref) <= (TOL if dtype == np.float32 else 1e-9), b
        assert abs(float(res.reconstruction_err[b]) - float(ref["reconstruction_err"])) / np.linalg.norm(Xs[b]) <= TOL


def test_wide_padded_leading_dimension_and_in_place_layouts():
    """A row-major X with ldx > m (a column slice of a wider array) and a W whose k is a multiple of 4 are used in
    place; other layouts are converted -- same numbers either way."""
    import torch

    import muscle_synergies_amd as ms

    X, W0, H0 = _case(700, 64, 8, np.float32, seed=3)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=30, tol=0.0)
    big = torch.zeros((1, 700, 80), dtype=torch.float32, device="cuda")
    big[0, :, :64] = torch.from_numpy(np.ascontiguousarray(X)).cuda()
    big[0, :, 64:] = 7.0  # must never be read
    res = ms.fit_batched(big[:, :, :64], torch.from_numpy(W0).cuda()[None], torch.from_numpy(H0).cuda()[None], max_iter=30, tol=0.0)
    assert _rel(X, res.W[0].cpu().numpy(), res.H[0].cpu().numpy(), ref) <= TOL
    res2 = ms.fit_batched(np.asfortranarray(X), W0, H0, max_iter=30, tol=0.0)
    np.testing.assert_array_equal(res.W[0].cpu().numpy(), res2.W[0])
    np.testing.assert_array_equal(res.H[0].cpu().numpy(), res2.H[0])


def test_wide_find_synergies_does_not_fall_back():
    """find_synergies(df 64 channels, solver='mu') stays on the GPU (round 2 handed it to scikit-learn)."""
    import warnings

    import pandas as pd

    import muscle_synergies_amd as ms
    from muscle_synergies_amd.hip_nmf import HipNMF

    X = emg_matrix(5, T=500, m=64, k_true=6, dtype=np.float64)
    df = pd.DataFrame(X, columns=[f"ch{i}" for i in range(64)])
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)  # the fallback announces itself with a RuntimeWarning
        res = ms.find_synergies(df, 10, solver="mu", max_iter=60, tol=0.0, init="nndsvda", random_state=0)
    assert isinstance(res.model, HipNMF)
    assert _last_kernel().startswith(("fit_wide_kernel<double", "fit_wide4d_kernel<"))
    assert res.components.shape == (10, 64)
    from muscle_synergies_amd.init import initialize_nmf

    W0, H0 = initialize_nmf(X, 10, init="nndsvda", random_state=0)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=60, tol=0.0)
    np.testing.assert_allclose(res.model.components_, ref["H"], rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("m,k,T", [(33, 8, 130), (64, 16, 400), (128, 5, 77), (16, 12, 300), (100, 9, 1003)])
def test_wide_kullback_leibler(dtype, m, k, T):
    """beta_loss='kullback-leibler' on the wide shapes (both W H reconstructions and both products on the matrix pipe)
    against the oracle's restatement of _nmf.py:556-591, 642-684: fixed iteration count, stop rule, regularisation."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    h1 = _lib.Handle(0)
    h1.set_tuning(0, 1, 0)  # one workgroup per matrix (a lone matrix of 1 000 rows would take the row-sliced one-pass kernel: test_gpu_big.py)
    X, W0, H0 = _case(T, m, k, dtype, seed=m + k)
    tol = 3e-5 if dtype == np.float32 else 1e-9
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 25, 0.0)
    for layout in ("F", "C"):
        Xl = np.asfortranarray(X) if layout == "F" else np.ascontiguousarray(X)
        res = ms.fit_batched(Xl, W0, H0, max_iter=25, tol=0.0, beta_loss="kullback-leibler", handle=h1)
        # (round 4: fp32 with at most 8 components on 33..128 channels takes the 4x4x1 kernel's KL flavour)
        # (round 5: float64 with at most 8 components takes the 4x4x4 kernel's)
        want = ("fit_wide4_kernel" if dtype == np.float32 else "fit_wide4d_kernel") if k <= 8 and m > 32 else "fit_wide_kernel"
        assert h1.last_kernel().endswith(",1>") and h1.last_kernel().startswith(want + "<"), h1.last_kernel()
        assert _rel(X, res.W[0], res.H[0], {"W": Wr, "H": Hr}) <= tol, layout
        err = orc.kl_divergence(X, Wr, Hr, square_root=True)
        assert abs(float(res.reconstruction_err[0]) - err) <= (5e-3 if dtype == np.float32 else 1e-9) * max(err, 1e-30)
    Ws, Hs, n_it = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 150, 1e-3, 0.01, 0.02, 0.03, 0.01)
    res = ms.fit_batched(X, W0, H0, max_iter=150, tol=1e-3, beta_loss="kullback-leibler", l1_reg_W=0.01, l1_reg_H=0.02,
                         l2_reg_W=0.03, l2_reg_H=0.01, handle=h1)
    if dtype == np.float64:
        assert int(res.n_iter[0]) == n_it
        assert _rel(X, res.W[0], res.H[0], {"W": Ws, "H": Hs}) <= 1e-9
    else:
        assert abs(int(res.n_iter[0]) - n_it) <= 10


def test_wide_kl_through_the_estimator():
    """HipNMF(beta_loss='kullback-leibler') on a 64-channel frame stays on the GPU."""
    import warnings

    import muscle_synergies_amd as ms

    X = emg_matrix(9, T=300, m=64, k_true=6, dtype=np.float64)
    W0, H0 = random_init(X, 10, seed=2)
    model = ms.HipNMF(n_components=10, init="custom", solver="mu", beta_loss="kullback-leibler", tol=0.0, max_iter=40)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        W = model.fit_transform(X, W=W0.copy(), H=H0.copy())
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 40, 0.0)
    np.testing.assert_allclose(W @ model.components_, Wr @ Hr, rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("m,k,T", [(24, 20, 200), (40, 32, 333), (64, 17, 1001), (128, 32, 150), (48, 24, 64)])
def test_wide_17_to_32_components(dtype, m, k, T):
    """17..32 components (two 16-component blocks on the matrix pipe), Frobenius and Kullback-Leibler, both layouts;
    float64 beyond 64 channels does not fit LDS in this configuration: since round 4 the general-shape kernels (nmf_big.hpp)
    take it (both losses)."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    X, W0, H0 = _case(T, m, k, dtype, seed=3 * m + k)
    if dtype == np.float64 and m > 64:
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=25, tol=0.0)
        res = ms.fit_batched(X, W0, H0, max_iter=25, tol=0.0)
        assert _last_kernel().startswith("big1_pass_kernel<double,32,1,2"), _last_kernel()  # (round 5: the one-pass kernel in float64)
        assert _rel(X, res.W[0], res.H[0], ref) <= 1e-9
        # (round 4, later: the Kullback-Leibler loss runs there too)
        Wk, Hk, _ = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 5, 0.0)
        rk = ms.fit_batched(X, W0, H0, max_iter=5, tol=0.0, beta_loss="kullback-leibler")
        assert _last_kernel().startswith("big1_pass_kernel<double,32,1,2,true,2,1>"), _last_kernel()
        assert _rel(X, rk.W[0], rk.H[0], {"W": Wk, "H": Hk}) <= 1e-9
        return
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=25, tol=0.0)
    tol = TOL if dtype == np.float32 else 1e-9
    for layout in ("F", "C"):
        Xl = np.asfortranarray(X) if layout == "F" else np.ascontiguousarray(X)
        res = ms.fit_batched(Xl, W0, H0, max_iter=25, tol=0.0)
        assert ",32,4," in _last_kernel(), _last_kernel()
        assert _rel(X, res.W[0], res.H[0], ref) <= tol, layout
        assert abs(float(res.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 15, 0.0)
    res = ms.fit_batched(X, W0, H0, max_iter=15, tol=0.0, beta_loss="kullback-leibler")
    assert _rel(X, res.W[0], res.H[0], {"W": Wr, "H": Hr}) <= (3e-5 if dtype == np.float32 else 1e-9)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("m,k,T", [(64, 8, 5000), (40, 12, 1003), (128, 16, 777), (33, 20, 2048)])
def test_wide_row_sliced_path(dtype, m, k, T):
    """Few long matrices (the reference's own single-DataFrame call): rows sliced over the whole chip, one pass + one H
    update per iteration (graph replay), stop rule with per-matrix done flags -- forced with variant 2 and compared with
    the one-workgroup-per-matrix path (variant 1) and the oracle; the library picks it by itself for B = 1."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    if dtype == np.float64 and k > 16 and m > 64:
        pytest.skip("outside the compiled kernel set")
    h = _lib.Handle(0)
    Xs, Ws, Hs = [], [], []
    for s in range(2):
        X, W0, H0 = _case(T - 37 * s if s == 0 else T, m, k, dtype, seed=90 + s)
        Xs.append(X), Ws.append(W0), Hs.append(H0)
    X3, W3, H3 = np.stack(Xs), np.stack(Ws), np.stack(Hs)
    tol = TOL if dtype == np.float32 else 1e-9
    out = {}
    for variant in (1, 2):
        h.set_tuning(0, 0, variant)
        out[variant] = ms.fit_batched(X3, W3, H3, max_iter=45, tol=0.0, handle=h)
        assert ("[sliced]" in h.last_kernel()) == (variant == 2), h.last_kernel()
    for b in range(2):
        ref = orc.nmf_mu_fit(Xs[b], Ws[b], Hs[b], max_iter=45, tol=0.0)
        for variant in (1, 2):
            r = out[variant]
            assert int(r.n_iter[b]) == 45
            assert _rel(Xs[b], r.W[b], r.H[b], ref) <= tol, (variant, b)
            assert abs(float(r.reconstruction_err[b]) - float(ref["reconstruction_err"])) / np.linalg.norm(Xs[b]) <= TOL
            va, vc = orc.vaf(Xs[b].astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
            assert abs(r.vaf[b, 0] - va) <= TOL
    # stop rule: the two matrices converge at different checks
    h.set_tuning(0, 0, 2)
    r = ms.fit_batched(X3, W3, H3, max_iter=400, tol=2e-4, handle=h)
    for b in range(2):
        ref = orc.nmf_mu_fit(Xs[b], Ws[b], Hs[b], max_iter=400, tol=2e-4)
        if dtype == np.float64:
            assert int(r.n_iter[b]) == ref["n_iter"]
            assert _rel(Xs[b], r.W[b], r.H[b], ref) <= 1e-9
        else:
            assert abs(int(r.n_iter[b]) - ref["n_iter"]) <= 10
    # transform (H fixed) through the sliced path
    rt = ms.fit_batched(X3, W3, out[1].H, max_iter=20, tol=0.0, update_H=False, handle=h)
    h.set_tuning(0, 0, 1)
    rt1 = ms.fit_batched(X3, W3, out[1].H, max_iter=20, tol=0.0, update_H=False, handle=h)
    # (fp32, at most 8 components: the two paths are different formulations -- 16x16x4 slices vs fit_wide4_kernel -- equal to rounding)
    np.testing.assert_allclose(rt.W, rt1.W, rtol=1e-4 if dtype == np.float32 else 1e-12, atol=1e-7 if dtype == np.float32 else 1e-9)
    # the library's own choice for one long matrix
    h.set_tuning(0, 0, 0)
    ms.fit_batched(X3[:1], W3[:1], H3[:1], max_iter=3, tol=0.0, handle=h)
    if T >= 2000:  # (a short matrix is as cheap on one workgroup as through two launches per iteration)
        assert "[sliced]" in h.last_kernel(), h.last_kernel()

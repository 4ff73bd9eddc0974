"""Host-side initialisation of W and H (not part of the hot loop).

``find_synergies`` never passes ``init`` unless the user does, so the reference's default path is
scikit-learn's ``_initialize_nmf`` with ``init=None`` -> NNDSVDa (``sklearn/decomposition/_nmf.py:221-373``)
on top of a randomized SVD (``sklearn/utils/extmath.py:287-372, 531-605, 895-953``).

:func:`initialize_nmf` therefore **delegates to scikit-learn's own function whenever scikit-learn is
importable** ("reuse sklearn's own ``_initialize_nmf`` on the host") -- the reference
depends on scikit-learn anyway, and that is the only way to be bit-identical with it for a given
``random_state``.  Without scikit-learn (a GPU node with the library only) the built-in restatement below
is used.

Attribution: the built-in path restates algorithms published in scikit-learn (BSD-3-Clause, Copyright (c)
2007-2024 The scikit-learn developers): NNDSVD / NNDSVDa / NNDSVDar after C. Boutsidis and E. Gallopoulos,
"SVD based initialization: A head start for nonnegative matrix factorization", Pattern Recognition 41
(2008), and the randomized range finder of N. Halko, P.-G. Martinsson and J. Tropp, "Finding structure
with randomness", SIAM Review 53 (2011), in the configuration ``_initialize_nmf`` uses.  The sequence of
``RandomState`` draws and LAPACK calls is fixed by that algorithm (drawing in another order would give a
different, equally valid but non-matching start); ``tests/test_init.py`` pins both paths to vectors captured
from scikit-learn 1.7.2 (fixture G4).
"""

from __future__ import annotations

import numbers

import numpy as np
from scipy import linalg


def check_random_state(seed):
    """None -> the global RandomState, int -> a fresh one, RandomState -> itself."""
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, numbers.Integral):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError(f"{seed!r} cannot be used to seed a numpy.random.RandomState instance")


def _sklearn_initialize():
    """scikit-learn's ``_initialize_nmf`` or None when scikit-learn (or that private name) is unavailable."""
    try:
        from sklearn.decomposition._nmf import _initialize_nmf
    except Exception:  # not installed, or a future release moved the helper
        return None
    return _initialize_nmf


def _fix_signs(left, right_t, by_left):
    """Deterministic signs of singular vector pairs (sklearn ``svd_flip``): the entry of largest magnitude of
    each left vector (``by_left``) or of each right vector is made positive."""
    ref = left.T if by_left else right_t
    pick = np.argmax(np.abs(ref), axis=1)
    sign = np.sign(ref[np.arange(ref.shape[0]), pick])
    left *= sign[np.newaxis, :]
    right_t *= sign[:, np.newaxis]
    return left, right_t


def randomized_svd(M, n_components, random_state, n_oversamples=10):
    """Leading singular triplets by a randomized range finder with LU-normalised power iterations, configured
    like ``_initialize_nmf`` configures it (k + 10 Gaussian probes, 7 iterations when k < 0.1 min(shape) else 4,
    the wide case handled on the transpose).  Draws ``rng.normal`` once, in the same shape as sklearn."""
    rng = check_random_state(random_state)
    rows, cols = M.shape
    passes = 7 if n_components < 0.1 * min(rows, cols) else 4
    wide = rows < cols
    A = M.T if wide else M
    basis = rng.normal(size=(A.shape[1], n_components + n_oversamples)).astype(A.dtype, copy=False)
    for _ in range(passes):
        basis, _ = linalg.lu(A @ basis, permute_l=True, check_finite=False)
        basis, _ = linalg.lu(A.T @ basis, permute_l=True, check_finite=False)
    basis, _ = linalg.qr(A @ basis, mode="economic", check_finite=False)
    small_u, sing, vt = linalg.svd(basis.T @ A, full_matrices=False, lapack_driver="gesdd")
    u = basis @ small_u
    u, vt = _fix_signs(u, vt, by_left=not wide)
    k = n_components
    if wide:  # triplets of M from those of M^T
        return vt[:k].T, sing[:k], u[:, :k].T
    return u[:, :k], sing[:k], vt[:k]


def exact_svd(M, n_components):
    """Leading singular triplets from LAPACK (gesdd) with the same sign convention."""
    u, sing, vt = linalg.svd(M, full_matrices=False, lapack_driver="gesdd")
    u, vt = _fix_signs(u, vt, by_left=M.shape[0] >= M.shape[1])
    return u[:, :n_components], sing[:n_components], vt[:n_components]


def _nndsvd_factors(U, S, Vt, eps):
    """Boutsidis-Gallopoulos non-negative factors from singular triplets, all components at once.

    Component 0: ``sqrt(s0) |u0|``, ``sqrt(s0) |v0|``.  Component j > 0: of the positive parts (u+, v+) and the
    negative parts (u-, v-) keep the pair with the larger ``|u.||v.|``, normalised, scaled by
    ``sqrt(s_j |u.||v.|)``.  Entries below ``eps`` become 0."""
    k = S.shape[0]
    up, un = np.maximum(U, 0), np.maximum(-U, 0)          # T x k
    vp, vn = np.maximum(Vt, 0), np.maximum(-Vt, 0)        # k x m
    nrm_up = np.sqrt(np.einsum("tj,tj->j", up, up))
    nrm_un = np.sqrt(np.einsum("tj,tj->j", un, un))
    nrm_vp = np.sqrt(np.einsum("jm,jm->j", vp, vp))
    nrm_vn = np.sqrt(np.einsum("jm,jm->j", vn, vn))
    mass_p, mass_n = nrm_up * nrm_vp, nrm_un * nrm_vn
    take_p = mass_p > mass_n
    W = np.empty_like(U)
    H = np.empty_like(Vt)
    for j in range(k):
        if j == 0:
            W[:, 0] = np.sqrt(S[0]) * np.abs(U[:, 0])
            H[0] = np.sqrt(S[0]) * np.abs(Vt[0])
            continue
        if take_p[j]:
            scale = np.sqrt(S[j] * mass_p[j])
            W[:, j] = scale * (up[:, j] / nrm_up[j])
            H[j] = scale * (vp[j] / nrm_vp[j])
        else:
            scale = np.sqrt(S[j] * mass_n[j])
            W[:, j] = scale * (un[:, j] / nrm_un[j])
            H[j] = scale * (vn[j] / nrm_vn[j])
    W[W < eps] = 0
    H[H < eps] = 0
    return W, H


_INITS = (None, "random", "nndsvd", "nndsvda", "nndsvdar")


def initialize_nmf(X, n_components, init=None, eps=1e-6, random_state=None, svd_solver="randomized",
                   backend="auto"):
    """W0 (T x k), H0 (k x m) in ``X.dtype``; same options and error messages as sklearn's ``_initialize_nmf``.

    ``backend='auto'`` calls scikit-learn's function when it is importable and the request is one it serves
    (``svd_solver='randomized'``), ``'builtin'`` forces the restatement in this module, ``'sklearn'`` insists on
    scikit-learn.  ``svd_solver='exact'`` (built-in only) takes the triplets from a full LAPACK SVD -- what the
    on-device :func:`nndsvd_init_batched` reproduces through the Gram matrix."""
    if backend not in ("auto", "builtin", "sklearn"):
        raise ValueError(f"backend must be 'auto', 'builtin' or 'sklearn' (got {backend!r})")
    X = np.asarray(X)
    if backend != "builtin" and svd_solver == "randomized" and init in _INITS:
        sk_init = _sklearn_initialize()
        if sk_init is not None:
            return sk_init(X, n_components, init=init, eps=eps, random_state=random_state)
        if backend == "sklearn":
            raise ImportError("scikit-learn's _initialize_nmf is not importable")
    elif backend == "sklearn" and svd_solver != "randomized":
        raise ValueError("backend='sklearn' implements svd_solver='randomized' only")

    if (X < 0).any():
        raise ValueError("Negative values in data passed to NMF initialization.")
    n_samples, n_features = X.shape
    full_rank_ok = n_components <= min(n_samples, n_features)
    if init is not None and init != "random" and not full_rank_ok:
        raise ValueError(
            "init = '{}' can only be used when n_components <= min(n_samples, n_features)".format(init)
        )
    if init is None:
        init = "nndsvda" if full_rank_ok else "random"

    if init == "random":  # sqrt(mean / k) |N(0, 1)|, H drawn before W (the order fixes the stream)
        scale = np.sqrt(X.mean() / n_components)
        rng = check_random_state(random_state)
        H = np.abs(scale * rng.standard_normal(size=(n_components, n_features)).astype(X.dtype, copy=False))
        W = np.abs(scale * rng.standard_normal(size=(n_samples, n_components)).astype(X.dtype, copy=False))
        return W, H

    if init not in ("nndsvd", "nndsvda", "nndsvdar"):
        raise ValueError(
            "Invalid init parameter: got %r instead of one of %r"
            % (init, (None, "random", "nndsvd", "nndsvda", "nndsvdar"))
        )

    if svd_solver == "exact":
        U, S, Vt = exact_svd(X, n_components)
    else:
        U, S, Vt = randomized_svd(X, n_components, random_state)
    W, H = _nndsvd_factors(U, S, Vt, eps)
    if init == "nndsvda":  # zeros -> mean of X
        fill = X.mean()
        W[W == 0] = fill
        H[H == 0] = fill
    elif init == "nndsvdar":  # zeros -> small random values, W's drawn first
        rng = check_random_state(random_state)
        fill = X.mean()
        zw, zh = W == 0, H == 0
        W[zw] = np.abs(fill * rng.standard_normal(size=int(zw.sum())) / 100)
        H[zh] = np.abs(fill * rng.standard_normal(size=int(zh.sum())) / 100)
    return W, H


# ------------------------------------------------------------------------------------------------
# Batched NNDSVD / NNDSVDa on the device
def nndsvd_init_batched(X, n_components: int, init: str = "nndsvda", eps: float = 1e-6, device=None):
    """``_initialize_nmf(X_b, k, init='nndsvd'|'nndsvda')`` for every matrix of ``X [B, T, m]`` (``T >= m``).

    The two T-long passes run on the GPU (``hipnmf_gram_*``, ``hipnmf_nndsvd_stats_*`` / ``_write_*``); the
    ``m x m`` symmetric eigen-problem and the ``k x m`` algebra of Boutsidis & Gallopoulos run on the host in
    fp64.  The singular triplets come from the exact Gram-matrix SVD instead of sklearn's randomized SVD;
    they are identical when ``k + 10 >= m`` (sklearn's random range then spans everything) and otherwise differ
    by sklearn's own approximation error (about 1e-5 on clustered trailing singular vectors;
    ``tests/test_gpu_init.py``).  Returns device tensors ``(W0 [B, T, k], H0 [B, k, m])`` in ``X``'s dtype.
    """
    import ctypes

    import torch

    from . import _lib
    from .engine import _as_device_tensor, _x_layout, make_problem, resolve_device

    if init not in ("nndsvd", "nndsvda"):
        raise ValueError("nndsvd_init_batched implements init='nndsvd' and 'nndsvda'")
    dev = resolve_device(device)
    Xt = _as_device_tensor(X, dev)
    if Xt.dim() == 2:
        Xt = Xt.unsqueeze(0)
    if Xt.dtype not in (torch.float32, torch.float64):
        Xt = Xt.to(torch.float64)
    B, T, m = Xt.shape
    k = int(n_components)
    if k > min(T, m):
        raise ValueError("init = '{}' can only be used when n_components <= min(n_samples, n_features)".format(init))
    if bool((Xt < 0).any()):
        raise ValueError("Negative values in data passed to NMF initialization.")
    layout, ldx, xbs, Xt = _x_layout(Xt)
    p = make_problem(B, T, m, k, x_layout=layout, ldx=ldx, x_batch_stride=xbs)
    sfx = "f32" if Xt.dtype == torch.float32 else "f64"
    lib, h = _lib.load(), _lib.get_handle(dev.index)
    f64 = dict(dtype=torch.float64, device=dev)
    gram, colsum = torch.empty((B, m, m), **f64), torch.empty((B, m), **f64)
    torch.cuda.synchronize(dev)
    _lib.check(getattr(lib, f"hipnmf_gram_{sfx}")(h.ptr, ctypes.byref(p), Xt.data_ptr(), gram.data_ptr(),
                                                  colsum.data_ptr()))
    G = gram.cpu().numpy()
    evals, Q = np.linalg.eigh(0.5 * (G + G.transpose(0, 2, 1)))  # ascending
    order = np.argsort(-evals, axis=1)[:, :k]
    S = np.sqrt(np.maximum(np.take_along_axis(evals, order, axis=1), 0.0))  # [B, k]
    V = np.stack([Q[b][:, order[b]].T for b in range(B)])  # [B, k, m]
    inv_s = np.where(S > 0, 1.0 / np.where(S > 0, S, 1.0), 0.0)
    Vd, isd = torch.from_numpy(V).to(dev), torch.from_numpy(inv_s).to(dev)
    stats = torch.empty((B, k, 4), **f64)
    _lib.check(getattr(lib, f"hipnmf_nndsvd_stats_{sfx}")(h.ptr, ctypes.byref(p), Xt.data_ptr(), Vd.data_ptr(),
                                                          isd.data_ptr(), stats.data_ptr()))
    st = stats.cpu().numpy()
    # svd_flip (extmath.py:895-953): the entry of largest magnitude of every left vector becomes positive
    sign = np.where(st[:, :, 2] < 0, -1.0, 1.0)
    V = V * sign[:, :, None]
    sp = np.where(sign > 0, st[:, :, 0], st[:, :, 1])  # ||u_+||^2 after the flip
    sn = np.where(sign > 0, st[:, :, 1], st[:, :, 0])
    H0 = np.zeros((B, k, m))
    coef = np.zeros((B, k, 2))
    H0[:, 0, :] = np.sqrt(S[:, 0])[:, None] * np.abs(V[:, 0, :])  # leading triplet is non-negative (_nmf.py:323-326)
    coef[:, 0, 0] = np.sqrt(S[:, 0])
    for j in range(1, k):
        y = V[:, j, :]
        y_p, y_n = np.maximum(y, 0), np.abs(np.minimum(y, 0))
        x_p_nrm, x_n_nrm = np.sqrt(sp[:, j]), np.sqrt(sn[:, j])
        y_p_nrm, y_n_nrm = np.linalg.norm(y_p, axis=1), np.linalg.norm(y_n, axis=1)
        m_p, m_n = x_p_nrm * y_p_nrm, x_n_nrm * y_n_nrm
        pos = m_p > m_n
        sigma = np.where(pos, m_p, m_n)
        lbd = np.sqrt(S[:, j] * sigma)
        with np.errstate(divide="ignore", invalid="ignore"):
            coef[:, j, 0] = lbd / np.where(pos, x_p_nrm, x_n_nrm)
            H0[:, j, :] = lbd[:, None] * np.where(pos[:, None], y_p / y_p_nrm[:, None], y_n / y_n_nrm[:, None])
        coef[:, j, 1] = np.where(pos, 1.0, -1.0)
    avg = colsum.cpu().numpy().sum(axis=1) / (T * m)
    np_dtype = np.float32 if Xt.dtype == torch.float32 else np.float64
    H0 = H0.astype(np_dtype)
    H0[H0 < eps] = 0
    fill = avg if init == "nndsvda" else np.zeros(B)
    if init == "nndsvda":
        H0 = np.where(H0 == 0, avg[:, None, None].astype(np_dtype), H0)
    W0 = torch.empty((B, T, k), dtype=Xt.dtype, device=dev)
    Vd = torch.from_numpy(np.ascontiguousarray(V)).to(dev)
    cd, fd = torch.from_numpy(coef).to(dev), torch.from_numpy(np.ascontiguousarray(fill, dtype=np.float64)).to(dev)
    _lib.check(getattr(lib, f"hipnmf_nndsvd_write_{sfx}")(h.ptr, ctypes.byref(p), Xt.data_ptr(), Vd.data_ptr(),
                                                          isd.data_ptr(), cd.data_ptr(), fd.data_ptr(),
                                                          ctypes.c_double(eps), W0.data_ptr()))
    return W0, torch.from_numpy(np.ascontiguousarray(H0)).to(dev)

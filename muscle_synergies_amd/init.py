"""Host-side initialisation of W and H (not part of the hot loop).

``find_synergies`` never passes ``init`` unless the user does, so the reference's default path is
sklearn's ``_initialize_nmf`` with ``init=None`` -> NNDSVDa (``sklearn/decomposition/_nmf.py:221-373``)
on top of a randomized SVD (``sklearn/utils/extmath.py:287-372, 531-605, 895-953``).  This module
implements those published algorithms with NumPy/SciPy so that the product does not depend on
sklearn's private API; drawing from the same ``RandomState`` in the same order makes the result match
sklearn's for a given ``random_state`` (checked in ``tests/test_init.py`` against fixtures captured
from sklearn 1.7.2).
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


def _norm(x):
    """Dot-product based Euclidean norm (``_nmf.py:42-50``)."""
    x = np.ravel(x)
    return np.sqrt(np.dot(x, x))


def _svd_flip_u(u, v):
    """Sign convention on the columns of u (``extmath.py:895-953``, u_based_decision=True)."""
    idx = np.argmax(np.abs(u.T), axis=1)
    signs = np.sign(u.T[np.arange(u.shape[1]), idx])
    u *= signs[np.newaxis, :]
    v *= signs[:, np.newaxis]
    return u, v


def _svd_flip_v(u, v):
    idx = np.argmax(np.abs(v), axis=1)
    signs = np.sign(v[np.arange(v.shape[0]), idx])
    u *= signs[np.newaxis, :]
    v *= signs[:, np.newaxis]
    return u, v


def randomized_svd(M, n_components, random_state, n_oversamples=10):
    """Halko et al. randomized SVD as configured by ``_initialize_nmf`` (all defaults)."""
    rng = check_random_state(random_state)
    n_random = n_components + n_oversamples
    n_samples, n_features = M.shape
    n_iter = 7 if n_components < 0.1 * min(M.shape) else 4
    transpose = n_samples < n_features
    A = M.T if transpose else M
    Q = rng.normal(size=(A.shape[1], n_random)).astype(A.dtype, copy=False)
    for _ in range(n_iter):  # LU-normalised power iterations (n_iter > 2)
        Q, _ = linalg.lu(A @ Q, permute_l=True, check_finite=False)
        Q, _ = linalg.lu(A.T @ Q, permute_l=True, check_finite=False)
    Q, _ = linalg.qr(A @ Q, mode="economic", check_finite=False)
    B = Q.T @ A
    Uhat, s, Vt = linalg.svd(B, full_matrices=False, lapack_driver="gesdd")
    U = Q @ Uhat
    if not transpose:
        U, Vt = _svd_flip_u(U, Vt)
        return U[:, :n_components], s[:n_components], Vt[:n_components, :]
    U, Vt = _svd_flip_v(U, Vt)
    return Vt[:n_components, :].T, s[:n_components], U[:, :n_components].T


def exact_svd(M, n_components):
    """Leading singular triplets from LAPACK (gesdd) with sklearn's sign convention (``svd_flip``)."""
    U, s, Vt = linalg.svd(M, full_matrices=False, lapack_driver="gesdd")
    if M.shape[0] >= M.shape[1]:
        U, Vt = _svd_flip_u(U, Vt)
    else:
        U, Vt = _svd_flip_v(U, Vt)
    return U[:, :n_components], s[:n_components], Vt[:n_components, :]


def initialize_nmf(X, n_components, init=None, eps=1e-6, random_state=None, svd_solver="randomized"):
    """W0 (T x k), H0 (k x m) in ``X.dtype``; same options and error messages as sklearn.

    ``svd_solver='randomized'`` is sklearn's behaviour; ``'exact'`` takes the triplets from a full LAPACK SVD
    (what the on-device :func:`nndsvd_init_batched` reproduces through the Gram matrix)."""
    X = np.asarray(X)
    if (X < 0).any():
        raise ValueError("Negative values in data passed to NMF initialization.")
    n_samples, n_features = X.shape
    if init is not None and init != "random" and n_components > min(n_samples, n_features):
        raise ValueError(
            "init = '{}' can only be used when n_components <= min(n_samples, n_features)".format(init)
        )
    if init is None:
        init = "nndsvda" if n_components <= min(n_samples, n_features) else "random"

    if init == "random":
        avg = np.sqrt(X.mean() / n_components)
        rng = check_random_state(random_state)
        H = avg * rng.standard_normal(size=(n_components, n_features)).astype(X.dtype, copy=False)
        W = avg * rng.standard_normal(size=(n_samples, n_components)).astype(X.dtype, copy=False)
        np.abs(H, out=H)
        np.abs(W, out=W)
        return W, H

    if init not in ("nndsvd", "nndsvda", "nndsvdar"):
        raise ValueError(
            "Invalid init parameter: got %r instead of one of %r"
            % (init, (None, "random", "nndsvd", "nndsvda", "nndsvdar"))
        )

    if svd_solver == "exact":
        U, S, V = exact_svd(X, n_components)
    else:
        U, S, V = randomized_svd(X, n_components, random_state)
    W = np.zeros_like(U)
    H = np.zeros_like(V)
    W[:, 0] = np.sqrt(S[0]) * np.abs(U[:, 0])
    H[0, :] = np.sqrt(S[0]) * np.abs(V[0, :])
    for j in range(1, n_components):  # Boutsidis & Gallopoulos split of the +/- parts
        x, y = U[:, j], V[j, :]
        x_p, y_p = np.maximum(x, 0), np.maximum(y, 0)
        x_n, y_n = np.abs(np.minimum(x, 0)), np.abs(np.minimum(y, 0))
        x_p_nrm, y_p_nrm = _norm(x_p), _norm(y_p)
        x_n_nrm, y_n_nrm = _norm(x_n), _norm(y_n)
        m_p, m_n = x_p_nrm * y_p_nrm, x_n_nrm * y_n_nrm
        if m_p > m_n:
            u, v, sigma = x_p / x_p_nrm, y_p / y_p_nrm, m_p
        else:
            u, v, sigma = x_n / x_n_nrm, y_n / y_n_nrm, m_n
        lbd = np.sqrt(S[j] * sigma)
        W[:, j] = lbd * u
        H[j, :] = lbd * v
    W[W < eps] = 0
    H[H < eps] = 0
    if init == "nndsvda":
        avg = X.mean()
        W[W == 0] = avg
        H[H == 0] = avg
    elif init == "nndsvdar":
        rng = check_random_state(random_state)
        avg = X.mean()
        W[W == 0] = abs(avg * rng.standard_normal(size=len(W[W == 0])) / 100)
        H[H == 0] = abs(avg * rng.standard_normal(size=len(H[H == 0])) / 100)
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

"""NumPy restatement of the NMF multiplicative-update ("mu") solver.

TEST INFRASTRUCTURE ONLY -- this file is the *checker*, never the thing that
is shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``muscle_synergies_amd`` must never do so.

What it restates
----------------
The reference (elvis-sik/muscle_synergies) performs synergy extraction in
``src/muscle_synergies/analysis.py:862-863``::

    model = NMF(n_components=n_components, **sklearn_kwargs)
    transformed_signal = model.fit_transform(matrix)

i.e. all arithmetic lives in the third-party dependency **scikit-learn**, which
is not vendored under ``/root/reference`` (pinned there as
``scikit-learn>=0.21, <=0.24`` in ``requirements.txt:3``; the image ships
scikit-learn **1.7.2**, whose ``solver='mu'`` / ``beta_loss='frobenius'`` update
equations are the same Lee-Seung rules).  The functions below restate the
published algorithm, each citing the ``sklearn/decomposition/_nmf.py`` (1.7.2)
lines it follows.  Notation is sklearn's: ``X (T x m) ~= W (T x k) @ H (k x m)``
with one muscle per column of ``X`` (``analysis.py:734-746``).

Pinning
-------
The reference's own test-suite never touches ``analysis.py``,
so no golden vector exists upstream.  The oracle is pinned instead against
outputs of the reference itself: ``tests/golden/make_golden.py`` imports
``muscle_synergies.find_synergies`` / sklearn *in the build container* and
stores inputs + outputs under ``tests/golden/``; ``tests/test_oracle.py``
replays them through this file (bit-exact on CPU).
"""

from __future__ import annotations

import numpy as np

#: ``EPSILON = np.finfo(np.float32).eps`` for fp32 *and* fp64 (_nmf.py:39).
EPSILON = np.finfo(np.float32).eps


def squared_norm(x: np.ndarray):
    """``sklearn/utils/extmath.py:19-44``: ``np.dot`` of the raveled array."""
    x = np.ravel(x, order="K")
    return np.dot(x, x)


def beta_divergence_frobenius(X, W, H, square_root: bool = False):
    """Frobenius branch of ``_beta_divergence`` (_nmf.py:120-134).

    ``res = ||X - W H||_F^2 / 2``; with ``square_root`` returns ``sqrt(2 res)``.
    """
    X = np.atleast_2d(X)
    W = np.atleast_2d(W)
    H = np.atleast_2d(H)
    res = squared_norm(X - np.dot(W, H)) / 2.0
    if square_root:
        return np.sqrt(res * 2)
    return res


def multiplicative_update_w(X, W, H, l1_reg_W=0.0, l2_reg_W=0.0):
    """beta=2 branch of ``_multiplicative_update_w`` (_nmf.py:540-554, 615-631).

    ``W *= (X H^T) / (W (H H^T))`` with exact-zero denominators set to EPSILON.
    Updates ``W`` in place and returns it.
    """
    numerator = X @ H.T  # safe_sparse_dot(X, H.T), _nmf.py:543
    HHt = np.dot(H, H.T)  # _nmf.py:553
    denominator = np.dot(W, HHt)  # _nmf.py:554
    if l1_reg_W > 0:
        denominator += l1_reg_W
    if l2_reg_W > 0:
        denominator = denominator + l2_reg_W * W
    denominator[denominator == 0] = EPSILON  # _nmf.py:620
    numerator /= denominator
    W *= numerator
    return W


def multiplicative_update_h(X, W, H, l1_reg_H=0.0, l2_reg_H=0.0):
    """beta=2 branch of ``_multiplicative_update_h`` (_nmf.py:638-640, 701-728).

    ``H *= (W^T X) / multi_dot([W^T, W, H])``; ``multi_dot`` picks the cheaper
    association -- ``(W^T W) H`` whenever ``k (T + m) < 2 T m``.
    """
    numerator = W.T @ X  # _nmf.py:639
    denominator = np.linalg.multi_dot([W.T, W, H])  # _nmf.py:640
    if l1_reg_H > 0:
        denominator += l1_reg_H
    if l2_reg_H > 0:
        denominator = denominator + l2_reg_H * H
    denominator[denominator == 0] = EPSILON  # _nmf.py:706
    numerator /= denominator
    H *= numerator
    return H


def fit_multiplicative_update(
    X,
    W,
    H,
    max_iter: int = 200,
    tol: float = 1e-4,
    l1_reg_W=0.0,
    l1_reg_H=0.0,
    l2_reg_W=0.0,
    l2_reg_H=0.0,
    update_H: bool = True,
    check_every: int = 10,
    err_trace: list | None = None,
):
    """``_fit_multiplicative_update`` for beta_loss=2 (_nmf.py:731-893).

    W is updated first, then H with the *new* W (_nmf.py:834, 854).  Only when
    ``tol > 0`` and every ``check_every``-th iteration (10 in sklearn,
    _nmf.py:872) the Frobenius error is evaluated and the loop stops when
    ``(previous_error - error) / error_at_init < tol`` (_nmf.py:883).

    ``W`` and ``H`` are modified in place.  Returns ``(W, H, n_iter)``.
    """
    error_at_init = beta_divergence_frobenius(X, W, H, square_root=True)  # :827
    previous_error = error_at_init
    if err_trace is not None:
        err_trace.append(float(error_at_init))

    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        W = multiplicative_update_w(X, W, H, l1_reg_W, l2_reg_W)
        if update_H:
            H = multiplicative_update_h(X, W, H, l1_reg_H, l2_reg_H)
        if tol > 0 and n_iter % check_every == 0:
            error = beta_divergence_frobenius(X, W, H, square_root=True)
            if err_trace is not None:
                err_trace.append(float(error))
            if (previous_error - error) / error_at_init < tol:
                break
            previous_error = error
    return W, H, n_iter


def compute_regularization(n_samples, n_features, alpha_W=0.0, alpha_H="same", l1_ratio=0.0):
    """``_BaseNMF._compute_regularization`` (_nmf.py:1254-1265)."""
    alpha_H = alpha_W if alpha_H == "same" else alpha_H
    l1_reg_W = n_features * alpha_W * l1_ratio
    l1_reg_H = n_samples * alpha_H * l1_ratio
    l2_reg_W = n_features * alpha_W * (1.0 - l1_ratio)
    l2_reg_H = n_samples * alpha_H * (1.0 - l1_ratio)
    return l1_reg_W, l1_reg_H, l2_reg_W, l2_reg_H


def nmf_mu_fit(X, W0, H0, max_iter=200, tol=1e-4, alpha_W=0.0, alpha_H="same", l1_ratio=0.0):
    """``NMF(solver='mu', init='custom').fit_transform(X, W=W0, H=H0)``.

    Follows ``fit_transform`` / ``_fit_transform`` (_nmf.py:1594-1734): run the
    loop, then recompute ``reconstruction_err_`` (_nmf.py:1628-1630).
    Returns a dict ``{W, H, n_iter, reconstruction_err}``; inputs are copied.
    """
    X = np.asarray(X)
    W = np.array(W0, dtype=X.dtype, order="C", copy=True)
    H = np.array(H0, dtype=X.dtype, order="C", copy=True)
    regs = compute_regularization(X.shape[0], X.shape[1], alpha_W, alpha_H, l1_ratio)
    W, H, n_iter = fit_multiplicative_update(
        X, W, H, max_iter, tol, regs[0], regs[1], regs[2], regs[3], update_H=True
    )
    err = beta_divergence_frobenius(X, W, H, square_root=True)
    return {"W": W, "H": H, "n_iter": n_iter, "reconstruction_err": err}


def nmf_mu_transform(X, H, max_iter=200, tol=1e-4, alpha_W=0.0, alpha_H="same", l1_ratio=0.0):
    """``NMF(solver='mu').transform(X)`` with fixed components ``H``.

    ``_check_w_h`` with ``update_H=False`` (_nmf.py:1219-1243) starts W at
    ``sqrt(X.mean() / k)`` everywhere; only W is updated (_nmf.py:1736-1763).
    """
    X = np.asarray(X)
    H = np.array(H, dtype=X.dtype, order="C", copy=True)
    k = H.shape[0]
    avg = np.sqrt(X.mean() / k)
    W = np.full((X.shape[0], k), avg, dtype=X.dtype)
    regs = compute_regularization(X.shape[0], X.shape[1], alpha_W, alpha_H, l1_ratio)
    W, H, n_iter = fit_multiplicative_update(
        X, W, H, max_iter, tol, regs[0], regs[1], regs[2], regs[3], update_H=False
    )
    return {"W": W, "H": H, "n_iter": n_iter}


def vaf(X, W, H):
    """Uncentered VAF of ``analysis.py:597-667``.

    Returns ``(vaf_all, vaf_per_column)`` with
    ``vaf = 1 - sum((X - W H)^2) / sum(X^2)`` over all entries / per column.
    """
    X = np.asarray(X)
    err = X - W @ H
    vaf_all = 1 - np.sum(err**2, axis=(0, 1)) / np.sum(X**2, axis=(0, 1))
    vaf_col = 1 - np.sum(err**2, axis=0) / np.sum(X**2, axis=0)
    return vaf_all, vaf_col


# ---------------------------------------------------------------------------
# Kullback-Leibler loss (beta_loss = 1) -- SURVEY.md section 8 row f-4.  Dense X only.
# ---------------------------------------------------------------------------
EPS64 = np.finfo(np.float64).eps


def kl_divergence(X, W, H, square_root: bool = False):
    """beta = 1 branch of ``_beta_divergence`` (_nmf.py:140-161, 185-189): generalised KL divergence
    ``sum(X log(X / WH)) - sum(X) + sum(WH)`` with zeros of X skipped and WH clamped at EPSILON."""
    WH_data = np.dot(W, H).ravel()
    X_data = X.ravel()
    indices = X_data > EPSILON
    WH_data = WH_data[indices]
    X_data = X_data[indices]
    WH_data[WH_data < EPSILON] = EPSILON
    sum_WH = np.dot(np.sum(W, axis=0), np.sum(H, axis=1))
    div = X_data / WH_data
    res = np.dot(X_data, np.log(div))
    res += sum_WH - X_data.sum()
    if square_root:
        res = max(res, 0)
        return np.sqrt(2 * res)
    return res


def kl_update_w(X, W, H, l1_reg_W=0.0, l2_reg_W=0.0):
    """beta = 1 branch of ``_multiplicative_update_w`` (_nmf.py:556-591, 615-631):
    ``W *= ((X / WH) H^T) / colsum(H)``."""
    WH_safe_X = np.dot(W, H)
    WH_safe_X[WH_safe_X < EPSILON] = EPSILON
    np.divide(X, WH_safe_X, out=WH_safe_X)
    numerator = WH_safe_X @ H.T
    H_sum = np.sum(H, axis=1)
    denominator = H_sum[np.newaxis, :]
    if l1_reg_W > 0:
        denominator = denominator + l1_reg_W
    if l2_reg_W > 0:
        denominator = denominator + l2_reg_W * W
    denominator = np.array(np.broadcast_to(denominator, numerator.shape))
    denominator[denominator == 0] = EPSILON
    numerator /= denominator
    W *= numerator
    return W


def kl_update_h(X, W, H, l1_reg_H=0.0, l2_reg_H=0.0):
    """beta = 1 branch of ``_multiplicative_update_h`` (_nmf.py:642-684, 701-728):
    ``H *= (W^T (X / WH)) / colsum(W)``."""
    WH_safe_X = np.dot(W, H)
    WH_safe_X[WH_safe_X < EPSILON] = EPSILON
    np.divide(X, WH_safe_X, out=WH_safe_X)
    numerator = W.T @ WH_safe_X
    W_sum = np.sum(W, axis=0)
    W_sum[W_sum == 0] = 1.0
    denominator = W_sum[:, np.newaxis]
    if l1_reg_H > 0:
        denominator = denominator + l1_reg_H
    if l2_reg_H > 0:
        denominator = denominator + l2_reg_H * H
    denominator = np.array(np.broadcast_to(denominator, numerator.shape))
    denominator[denominator == 0] = EPSILON
    numerator /= denominator
    H *= numerator
    return H


def fit_multiplicative_update_kl(X, W, H, max_iter=200, tol=1e-4, l1_reg_W=0.0, l1_reg_H=0.0, l2_reg_W=0.0,
                                 l2_reg_H=0.0, update_H=True, check_every=10, err_trace=None):
    """``_fit_multiplicative_update`` with beta_loss = 1 (_nmf.py:731-893; gamma = 1).  After the H update,
    entries of H below float64 eps are set to 0 (_nmf.py:866-868; the analogous W rule only applies to
    beta_loss < 1)."""
    error_at_init = kl_divergence(X, W, H, square_root=True)
    previous_error = error_at_init
    if err_trace is not None:
        err_trace.append(float(error_at_init))
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        W = kl_update_w(X, W, H, l1_reg_W, l2_reg_W)
        if update_H:
            H = kl_update_h(X, W, H, l1_reg_H, l2_reg_H)
            H[H < EPS64] = 0.0
        if tol > 0 and n_iter % check_every == 0:
            error = kl_divergence(X, W, H, square_root=True)
            if err_trace is not None:
                err_trace.append(float(error))
            if (previous_error - error) / error_at_init < tol:
                break
            previous_error = error
    return W, H, n_iter


def nmf_mu_fit_kl(X, W0, H0, max_iter=200, tol=1e-4, alpha_W=0.0, alpha_H="same", l1_ratio=0.0):
    """``NMF(solver='mu', beta_loss='kullback-leibler', init='custom').fit_transform(X, W=W0, H=H0)``."""
    X = np.asarray(X)
    W = np.array(W0, dtype=X.dtype, order="C", copy=True)
    H = np.array(H0, dtype=X.dtype, order="C", copy=True)
    regs = compute_regularization(X.shape[0], X.shape[1], alpha_W, alpha_H, l1_ratio)
    W, H, n_iter = fit_multiplicative_update_kl(X, W, H, max_iter, tol, regs[0], regs[1], regs[2], regs[3])
    err = kl_divergence(X, W, H, square_root=True)
    return {"W": W, "H": H, "n_iter": n_iter, "reconstruction_err": err}


# ---------------------------------------------------------------------------
# T-sharded restatement: the same iteration written as per-shard passes plus
# one sum over shards.  Used by the world_size-2 gloo test to check the
# multi-GPU orchestration without a GPU.
# ---------------------------------------------------------------------------


def shard_pass(X_s, W_s, H, l1_reg_W=0.0, l2_reg_W=0.0):
    """One shard's part of an iteration: update its rows of W, return the
    shard-local sums ``W_s^T X_s`` (k x m) and ``W_s^T W_s`` (k x k)."""
    multiplicative_update_w(X_s, W_s, H, l1_reg_W, l2_reg_W)
    return W_s.T @ X_s, W_s.T @ W_s


def h_update_from_sums(WtX, WtW, H, l1_reg_H=0.0, l2_reg_H=0.0):
    """H update from the summed-over-shards ``W^T X`` and ``W^T W``."""
    denominator = WtW @ H
    if l1_reg_H > 0:
        denominator += l1_reg_H
    if l2_reg_H > 0:
        denominator = denominator + l2_reg_H * H
    denominator[denominator == 0] = EPSILON
    H *= WtX / denominator
    return H


def kl_shard_pass(X_s, W_s, H, l1_reg_W=0.0, l2_reg_W=0.0):
    """Kullback-Leibler flavour of :func:`shard_pass`: update the shard's rows of W (row-local: the denominator is
    ``rowsum(H)``), return ``W_s^T (X_s / W_s H)`` (k x m) and ``colsum(W_s)`` in column 0 of a k x k block."""
    kl_update_w(X_s, W_s, H, l1_reg_W, l2_reg_W)
    WH = np.dot(W_s, H)
    WH[WH < EPSILON] = EPSILON
    k = H.shape[0]
    second = np.zeros((k, k), dtype=W_s.dtype)
    second[:, 0] = W_s.sum(axis=0)
    return W_s.T @ (X_s / WH), second


def kl_h_update_from_sums(WtQ, second, H, l1_reg_H=0.0, l2_reg_H=0.0):
    """H update of the Kullback-Leibler loss from the summed-over-shards ``W^T (X / WH)`` and ``colsum(W)``
    (_nmf.py:663-684, 866-868)."""
    W_sum = second[:, 0].copy()
    W_sum[W_sum == 0] = 1.0
    denominator = W_sum[:, np.newaxis]
    if l1_reg_H > 0:
        denominator = denominator + l1_reg_H
    if l2_reg_H > 0:
        denominator = denominator + l2_reg_H * H
    denominator = np.array(np.broadcast_to(denominator, WtQ.shape))
    denominator[denominator == 0] = EPSILON
    H *= WtQ / denominator
    H[H < EPS64] = 0.0
    return H


def kl_divergence_columns(X, W, H):
    """The generalised Kullback-Leibler divergence split by column (element by element ``x log(x / wh) - x + wh`` with zeros of
    X skipped and WH clamped): sums to :func:`kl_divergence`."""
    WH = np.dot(W, H)
    WHc = np.where(WH < EPSILON, EPSILON, WH)
    pos = X > EPSILON
    term = np.where(pos, X * np.log(np.where(pos, X, 1.0) / WHc) - X + WH, WH)
    return term.sum(axis=0)

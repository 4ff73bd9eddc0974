"""``HipNMF`` -- estimator with the surface of ``sklearn.decomposition.NMF`` that the reference uses,
running the multiplicative-update solver on an MI355X through ``libhip_nmf.so``.

The reference constructs ``NMF(n_components=k, **sklearn_kwargs)`` and calls ``fit_transform``
(``src/muscle_synergies/analysis.py:862-863``), then reads ``components_`` (``:875, 879``) and hands the
model object to the user (``:882``), who may look at ``n_iter_``, ``reconstruction_err_``,
``n_components_`` or call ``transform`` / ``inverse_transform`` (``analysis.py:763-767``).  This class
duck-types exactly that, for ``solver='mu'`` with ``beta_loss`` ``'frobenius'`` (the reference's default)
or ``'kullback-leibler'`` on dense input; parameter
names, defaults, validation messages and warnings follow sklearn 1.7.2
(``sklearn/decomposition/_nmf.py:1140-1265, 1538-1763``).
"""

from __future__ import annotations

import warnings

import numpy as np

from . import _lib, engine
from .init import initialize_nmf


def _convergence_warning_class():
    try:
        from sklearn.exceptions import ConvergenceWarning

        return ConvergenceWarning
    except Exception:  # sklearn not importable: keep the semantics with a local class

        class ConvergenceWarning(UserWarning):
            pass

        return ConvergenceWarning


def _check_init(A, shape, whom):
    """``_check_init`` of sklearn (``_nmf.py:68-83``)."""
    A = np.asarray(A)
    if A.ndim != 2:
        raise ValueError(f"Expected 2D array, got {A.ndim}D array instead passed to {whom}.")
    if A.shape[0] != shape[0]:
        raise ValueError(
            f"Array with wrong first dimension passed to {whom}. Expected {shape[0]}, but got {A.shape[0]}."
        )
    if A.shape[1] != shape[1]:
        raise ValueError(
            f"Array with wrong second dimension passed to {whom}. Expected {shape[1]}, but got {A.shape[1]}."
        )
    if (A < 0).any():
        raise ValueError(f"Negative values in data passed to {whom}.")
    if np.max(A) == 0:
        raise ValueError(f"Array passed to {whom} is full of zeros.")
    return A


class HipNMF:
    """NMF by multiplicative updates (Frobenius or Kullback-Leibler loss) on an AMD MI355X.

    Parameters mirror ``sklearn.decomposition.NMF``; only ``solver='mu'`` with
    ``beta_loss in ('frobenius', 2, 'kullback-leibler', 1)`` is implemented here -- ``find_synergies``
    routes everything else to sklearn, as the reference does.
    """

    def __init__(self, n_components=None, *, init=None, solver="mu", beta_loss="frobenius", tol=1e-4,
                 max_iter=200, random_state=None, alpha_W=0.0, alpha_H="same", l1_ratio=0.0, verbose=0,
                 shuffle=False, device=None):
        self.n_components = n_components
        self.init = init
        self.solver = solver
        self.beta_loss = beta_loss
        self.tol = tol
        self.max_iter = max_iter
        self.random_state = random_state
        self.alpha_W = alpha_W
        self.alpha_H = alpha_H
        self.l1_ratio = l1_ratio
        self.verbose = verbose
        self.shuffle = shuffle
        self.device = device

    # -- sklearn estimator protocol (just enough for inspection / cloning by hand) ----------------
    def get_params(self, deep=True):
        return {k: getattr(self, k) for k in (
            "n_components", "init", "solver", "beta_loss", "tol", "max_iter", "random_state", "alpha_W",
            "alpha_H", "l1_ratio", "verbose", "shuffle", "device")}

    def set_params(self, **params):
        for k, v in params.items():
            if k not in self.get_params():
                raise ValueError(f"Invalid parameter {k!r} for estimator {type(self).__name__}")
            setattr(self, k, v)
        return self

    def __repr__(self):
        return f"HipNMF(n_components={self.n_components!r}, init={self.init!r}, tol={self.tol!r}, max_iter={self.max_iter!r})"

    # -- validation --------------------------------------------------------------------------------
    MAX_FEATURES = 512   # widest shape compiled into libhip_nmf.so (nmf_big.hpp: the general-shape kernels beyond the 128 x 32 of
    MAX_COMPONENTS = 64  # nmf_wide.hpp, both losses); HIPNMF_ERR_UNSUPPORTED beyond

    @staticmethod
    def supports(solver="cd", beta_loss="frobenius", n_features=None, n_components=None, **_ignored) -> bool:
        """True when these NMF kwargs (and, when given, this shape) select the path this engine implements."""
        if not (solver == "mu" and beta_loss in ("frobenius", 2, 2.0, "kullback-leibler", 1, 1.0)):
            return False
        if n_features is not None and n_features > HipNMF.MAX_FEATURES:
            return False
        if n_components is not None and n_components > HipNMF.MAX_COMPONENTS:
            return False
        return True

    def _check_params(self):
        if self.solver != "mu":
            raise ValueError(f"HipNMF implements solver='mu' only (got {self.solver!r})")
        if self.beta_loss not in ("frobenius", 2, 2.0, "kullback-leibler", 1, 1.0):
            raise NotImplementedError(
                f"HipNMF implements beta_loss 'frobenius' and 'kullback-leibler' (got {self.beta_loss!r}); "
                "use sklearn for other losses"
            )
        if not (isinstance(self.max_iter, (int, np.integer)) and self.max_iter >= 1):
            raise ValueError(f"The 'max_iter' parameter of NMF must be an int in the range [1, inf). Got {self.max_iter!r} instead.")
        if not (self.tol >= 0):
            raise ValueError(f"The 'tol' parameter of NMF must be a float in the range [0, inf). Got {self.tol!r} instead.")
        if not (self.alpha_W >= 0):
            raise ValueError("alpha_W must be >= 0")
        if self.alpha_H != "same" and not (self.alpha_H >= 0):
            raise ValueError("alpha_H must be >= 0 or 'same'")
        if not (0 <= self.l1_ratio <= 1):
            raise ValueError("l1_ratio must be in [0, 1]")
        if self.init not in (None, "random", "nndsvd", "nndsvda", "nndsvdar", "custom"):
            raise ValueError(
                f"The 'init' parameter of NMF must be a str among {{'random', 'nndsvd', 'nndsvda', 'nndsvdar', 'custom'}} or None. Got {self.init!r} instead."
            )
        if self.init == "nndsvd":
            warnings.warn(
                "The multiplicative update ('mu') solver cannot update zeros present in the initialization, "
                "and so leads to poorer results when used jointly with init='nndsvd'. You may try "
                "init='nndsvda' or init='nndsvdar' instead.", UserWarning)

    @staticmethod
    def _validate_X(X, whom="NMF initialization"):
        """float32 stays float32, everything else becomes float64 (``validate_data(dtype=[f64, f32])``)."""
        arr = X.to_numpy() if hasattr(X, "to_numpy") else np.asarray(X)
        if arr.ndim != 2:
            raise ValueError(f"Expected 2D array, got {arr.ndim}D array instead")
        if arr.dtype != np.float32:
            arr = arr.astype(np.float64, copy=False)
        if arr.shape[0] == 0 or arr.shape[1] == 0:
            raise ValueError(f"Found array with {arr.shape[0]} sample(s) (shape={arr.shape}) while a minimum of 1 is required.")
        if not np.isfinite(arr).all():
            raise ValueError("Input X contains NaN or infinity.")
        if (arr < 0).any():
            raise ValueError(f"Negative values in data passed to {whom}.")
        return arr

    def _regularization(self, n_samples, n_features):
        """``_compute_regularization`` (``_nmf.py:1254-1265``)."""
        alpha_H = self.alpha_W if self.alpha_H == "same" else self.alpha_H
        return (n_features * self.alpha_W * self.l1_ratio, n_samples * alpha_H * self.l1_ratio,
                n_features * self.alpha_W * (1.0 - self.l1_ratio), n_samples * alpha_H * (1.0 - self.l1_ratio))

    # -- fitting -----------------------------------------------------------------------------------
    def fit_transform(self, X, y=None, W=None, H=None):
        """Learn the factorisation and return W (``_nmf.py:1594-1636``)."""
        return self._fit_prepared(*self._prepare(X, W, H))

    def _prepare(self, X, W=None, H=None):
        """Everything of ``fit_transform`` that happens on the host before the solver runs: parameter and input validation
        and the initialisation (the only consumer of ``random_state`` / NumPy's global generator).  ``find_synergies``
        prepares the ranks of a range in order and then lets the fits run concurrently."""
        self._check_params()
        whom = "NMF (input X)" if self.init == "custom" else "NMF initialization"
        columns = getattr(X, "columns", None)
        X = self._validate_X(X, whom)
        T, m = X.shape
        k = m if self.n_components in (None, "auto") else int(self.n_components)
        if self.init == "custom":
            if W is None or H is None:
                raise ValueError("init='custom' needs both W and H")
            H = _check_init(H, (k, m), "NMF (input H)")
            W = _check_init(W, (T, k), "NMF (input W)")
            if H.dtype != X.dtype or W.dtype != X.dtype:
                raise TypeError(
                    "H and W should have the same dtype as X. Got H.dtype = {} and W.dtype = {}.".format(H.dtype, W.dtype))
            W0, H0 = W, H
        else:
            if W is not None or H is not None:
                warnings.warn("When init!='custom', provided W or H are ignored. Set  init='custom' to use them as initialization.",
                              RuntimeWarning)
            W0, H0 = initialize_nmf(X, k, init=self.init, random_state=self.random_state)
        return X, columns, W0, H0

    def _fit_prepared(self, X, columns, W0, H0):
        """The solver on a prepared problem (see ``_prepare``); sets the fitted attributes and returns W."""
        T, m = X.shape
        k = W0.shape[1]
        l1w, l1h, l2w, l2h = self._regularization(T, m)
        try:
            res = engine.fit_batched(X, W0, H0, max_iter=self.max_iter, tol=self.tol, l1_reg_W=l1w, l1_reg_H=l1h,
                                     l2_reg_W=l2w, l2_reg_H=l2h, beta_loss=self.beta_loss, device=self.device,
                                     return_numpy=True)
        except _lib.HipNmfError as e:
            if e.code != _lib.HIPNMF_ERR_UNSUPPORTED:
                raise
            return self._fit_transform_sklearn(X, W0, H0, columns, str(e))
        n_iter = int(res.n_iter[0])
        if n_iter == self.max_iter and self.tol > 0:
            warnings.warn("Maximum number of iterations %d reached. Increase it to improve convergence." % self.max_iter,
                          _convergence_warning_class())
        self.reconstruction_err_ = res.reconstruction_err[0]
        self.n_components_ = k
        self.components_ = res.H[0]
        self.n_iter_ = n_iter
        self.n_features_in_ = m
        if columns is not None:
            self.feature_names_in_ = np.asarray(columns, dtype=object)
        self.vaf_ = res.vaf[0]
        self.kernel_ms_ = res.kernel_ms
        return res.W[0]

    def _fit_transform_sklearn(self, X, W0, H0, columns, why):
        """A configuration the library refuses (``HIPNMF_ERR_UNSUPPORTED``) runs on scikit-learn's own mu solver from
        the same starting point, as every other unsupported configuration of the reference's call does -- loudly."""
        from sklearn.decomposition import NMF

        warnings.warn(f"{why}; running scikit-learn on the CPU instead", RuntimeWarning, stacklevel=3)
        model = NMF(n_components=W0.shape[1], init="custom", solver="mu", beta_loss=self.beta_loss, tol=self.tol,
                    max_iter=self.max_iter, alpha_W=self.alpha_W, alpha_H=self.alpha_H, l1_ratio=self.l1_ratio)
        W = model.fit_transform(X, W=np.array(W0, dtype=X.dtype, order="C"), H=np.array(H0, dtype=X.dtype, order="C"))
        self.reconstruction_err_ = model.reconstruction_err_
        self.n_components_ = model.n_components_
        self.components_ = model.components_
        self.n_iter_ = model.n_iter_
        self.n_features_in_ = X.shape[1]
        if columns is not None:
            self.feature_names_in_ = np.asarray(columns, dtype=object)
        sse = ((X - W @ model.components_) ** 2).sum(axis=0)
        xsq = (X ** 2).sum(axis=0)
        self.vaf_ = np.concatenate([[1.0 - sse.sum() / xsq.sum()], 1.0 - sse / xsq])
        self.kernel_ms_ = None
        return W

    def fit(self, X, y=None, **params):
        self.fit_transform(X, **params)
        return self

    def transform(self, X):
        """W for new data with the fitted components fixed (``_nmf.py:1736-1763``; W starts at
        ``sqrt(X.mean() / k)``, ``:1238-1240``)."""
        if not hasattr(self, "components_"):
            raise RuntimeError("This HipNMF instance is not fitted yet. Call 'fit' with appropriate arguments before using this estimator.")
        self._check_params()
        X = self._validate_X(X, "NMF (input X)")
        T, m = X.shape
        if m != self.n_features_in_:
            raise ValueError(f"X has {m} features, but HipNMF is expecting {self.n_features_in_} features as input.")
        k = self.n_components_
        H = np.ascontiguousarray(self.components_, dtype=X.dtype)
        avg = np.sqrt(X.mean() / k)
        W0 = np.full((T, k), avg, dtype=X.dtype)
        l1w, l1h, l2w, l2h = self._regularization(T, m)
        res = engine.fit_batched(X, W0, H, max_iter=self.max_iter, tol=self.tol, update_H=False, l1_reg_W=l1w,
                                 l1_reg_H=l1h, l2_reg_W=l2w, l2_reg_H=l2h, beta_loss=self.beta_loss,
                                 device=self.device, return_numpy=True)
        if int(res.n_iter[0]) == self.max_iter and self.tol > 0:
            warnings.warn("Maximum number of iterations %d reached. Increase it to improve convergence." % self.max_iter,
                          _convergence_warning_class())
        return res.W[0]

    def inverse_transform(self, X=None, Xt=None):
        """``W @ components_`` (host side; a k-wide product is not a hot path)."""
        W = X if X is not None else Xt
        if not hasattr(self, "components_"):
            raise RuntimeError("This HipNMF instance is not fitted yet.")
        return np.asarray(W) @ self.components_

"""ctypes wrapper of the C restatement (``oracle/nmf_mu_oracle.c``) -- checker only, see that file."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libnmf_mu_oracle.so")


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "nmf_mu_oracle.c")):
        subprocess.run(["make", "-s", "-C", HERE], check=True)
    lib = ctypes.CDLL(LIB)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.nmf_oracle_fit.restype = ctypes.c_int
    lib.nmf_oracle_fit.argtypes = [dp, dp, dp, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, ctypes.c_double, dp]
    lib.nmf_oracle_frobenius_error.restype = ctypes.c_double
    lib.nmf_oracle_frobenius_error.argtypes = [dp, dp, dp, ctypes.c_long, ctypes.c_int, ctypes.c_int, dp]
    return lib


def nmf_mu_fit_c(X, W0, H0, max_iter=200, tol=1e-4, update_H=True, l1_reg_W=0.0, l1_reg_H=0.0, l2_reg_W=0.0,
                 l2_reg_H=0.0, check_every=10):
    """Double-precision fit through the C oracle; returns ``{W, H, n_iter, reconstruction_err}``."""
    lib = load()
    X = np.ascontiguousarray(X, dtype=np.float64)
    W = np.array(W0, dtype=np.float64, order="C", copy=True)
    H = np.array(H0, dtype=np.float64, order="C", copy=True)
    T, m = X.shape
    k = H.shape[0]
    err = ctypes.c_double()
    ptr = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))  # noqa: E731
    n = lib.nmf_oracle_fit(ptr(X), ptr(W), ptr(H), T, m, k, int(max_iter), float(tol), int(check_every),
                           int(bool(update_H)), l1_reg_W, l1_reg_H, l2_reg_W, l2_reg_H, ctypes.byref(err))
    if n < 0:
        raise ValueError("bad argument to nmf_oracle_fit")
    return {"W": W, "H": H, "n_iter": n, "reconstruction_err": err.value}

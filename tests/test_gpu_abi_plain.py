"""The C ABI driven without PyTorch: plain ctypes + the HIP runtime for device buffers (the stub of
INTEGRATION.md section 3, executed).  Runs in a subprocess so that no torch-loaded HIP runtime is in the picture."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = textwrap.dedent(
    r'''
    import ctypes, os, sys
    import numpy as np
    sys.path.insert(0, os.environ["HIPNMF_REPO"])
    from oracle import nmf_mu_oracle as orc
    from muscle_synergies_amd.synth import emg_matrix, random_init

    assert "torch" not in sys.modules
    hip = ctypes.CDLL("libamdhip64.so")
    lib = ctypes.CDLL(os.path.join(os.environ["HIPNMF_REPO"], "muscle_synergies_amd", "lib", "libhip_nmf.so"))
    assert "torch" not in sys.modules

    class Problem(ctypes.Structure):                       # struct hipnmf_problem (include/hip_nmf.h)
        _fields_ = [("struct_size", ctypes.c_int32), ("batch", ctypes.c_int32), ("n_samples", ctypes.c_int64),
                    ("n_features", ctypes.c_int32), ("n_components", ctypes.c_int32), ("x_layout", ctypes.c_int32),
                    ("update_h", ctypes.c_int32), ("w_layout", ctypes.c_int32), ("loss", ctypes.c_int32),
                    ("ldx", ctypes.c_int64), ("x_batch_stride", ctypes.c_int64), ("max_iter", ctypes.c_int32),
                    ("check_every", ctypes.c_int32), ("tol", ctypes.c_double), ("l1_reg_W", ctypes.c_double),
                    ("l1_reg_H", ctypes.c_double), ("l2_reg_W", ctypes.c_double), ("l2_reg_H", ctypes.c_double)]

    vp = ctypes.c_void_p
    hip.hipMalloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [vp]
    lib.hipnmf_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    lib.hipnmf_destroy.argtypes = [vp]
    lib.hipnmf_fit_batched_f64.argtypes = [vp, ctypes.POINTER(Problem)] + [vp] * 7
    lib.hipnmf_last_error.restype = ctypes.c_char_p

    def dev(arr):
        p = vp()
        assert hip.hipMalloc(ctypes.byref(p), arr.nbytes) == 0
        assert hip.hipMemcpy(p, arr.ctypes.data_as(vp), arr.nbytes, 1) == 0      # hipMemcpyHostToDevice
        return p

    def host(p, like):
        out = np.empty_like(like)
        assert hip.hipMemcpy(out.ctypes.data_as(vp), p, out.nbytes, 2) == 0      # hipMemcpyDeviceToHost
        return out

    T, m, k = 3000, 8, 3
    X = emg_matrix(12, T=T, m=m, k_true=3, dtype=np.float64)     # F-contiguous = channel-major, as DataFrame.to_numpy()
    W0, H0 = random_init(X, k, seed=4)
    h = vp()
    assert lib.hipnmf_create(0, ctypes.byref(h)) == 0, lib.hipnmf_last_error()
    dX, dW, dH = dev(X), dev(W0), dev(H0)
    err, nit, sse, xsq = np.zeros(1), np.zeros(1, np.int32), np.zeros(m), np.zeros(m)
    dErr, dNit, dSse, dXsq = dev(err), dev(nit), dev(sse), dev(xsq)
    p = Problem(ctypes.sizeof(Problem), 1, T, m, k, 1, 1, 0, 0, T, T * m, 300, 10, 1e-4, 0, 0, 0, 0)
    rc = lib.hipnmf_fit_batched_f64(h, ctypes.byref(p), dX, dW, dH, dErr, dNit, dSse, dXsq)
    assert rc == 0, lib.hipnmf_last_error()
    W, H = host(dW, W0), host(dH, H0)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=300, tol=1e-4)
    assert int(host(dNit, nit)[0]) == ref["n_iter"]
    np.testing.assert_allclose(W @ H, ref["W"] @ ref["H"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(host(dErr, err)[0], ref["reconstruction_err"], rtol=1e-9)
    vaf_col = 1 - host(dSse, sse) / host(dXsq, xsq)
    np.testing.assert_allclose(vaf_col, 1 - ((X - W @ H) ** 2).sum(0) / (X ** 2).sum(0), atol=1e-9)
    p.max_iter = 0                                           # errors come back as codes + message, not exceptions
    assert lib.hipnmf_fit_batched_f64(h, ctypes.byref(p), dX, dW, dH, dErr, dNit, dSse, dXsq) == -1
    assert b"max_iter" in lib.hipnmf_last_error()
    for q in (dX, dW, dH, dErr, dNit, dSse, dXsq):
        hip.hipFree(q)
    assert lib.hipnmf_destroy(h) == 0
    assert "torch" not in sys.modules
    print("PLAIN-ABI-OK")
    '''
)


def test_c_abi_without_torch():
    env = dict(os.environ, HIPNMF_REPO=ROOT)
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "PLAIN-ABI-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]

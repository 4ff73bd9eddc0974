"""Threading clause of include/hip_nmf.h ("distinct handles may be driven concurrently from different host threads"),
exercised without PyTorch: plain ctypes + the HIP runtime for device buffers, N host threads with one handle each, every
solver path of the library, results compared BITWISE with the same calls run one after the other on one handle.

    python3 tools/abi_threads_stress.py --threads 3 --rounds 2 [--only wide_sliced] [--list]

The reference's seam is re-entrant by construction (a fresh sklearn estimator per call,
src/muscle_synergies/analysis.py:862-863; the rank loop :907-912 has no shared state): so must the C ABI be.
Prints one line per case with the kernel that ran, then "ABI-THREADS-OK paths=<comma separated>".
"""
import argparse
import ctypes
import os
import random
import sys
import threading
import time

import numpy as np

ROOT = os.environ.get("HIPNMF_REPO") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from muscle_synergies_amd import _lib as L  # noqa: E402  (ctypes declarations only; nothing is loaded by the import)
from muscle_synergies_amd.preprocess import EnvelopeParams, SosfiltParams  # noqa: E402
from muscle_synergies_amd.synth import emg_matrix, random_init, raw_emg  # noqa: E402

assert "torch" not in sys.modules
vp = ctypes.c_void_p
hip = ctypes.CDLL("libamdhip64.so")
lib = ctypes.CDLL(os.environ.get("HIPNMF_LIBRARY", L.LIB_PATH))
L._declare(lib)
for sfx in ("f32", "f64"):
    getattr(lib, "hipnmf_random_init_" + sfx).argtypes = [vp, vp, ctypes.c_uint64, ctypes.c_int32, vp, vp, vp]
    for name in ("hipnmf_rank_sweep_", "hipnmf_rank_sweep_stop_"):
        getattr(lib, name + sfx).argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_uint64,
                                             ctypes.c_int32, vp, vp, vp, vp, vp, vp, vp]
hip.hipMalloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int]
hip.hipFree.argtypes = [vp]
hip.hipSetDevice.argtypes = [ctypes.c_int]


class Fail(RuntimeError):
    pass


def ok(rc, what):
    if rc != 0:
        msg = lib.hipnmf_last_error()
        raise Fail("%s -> %d: %s" % (what, rc, msg.decode() if msg else ""))


class Dev:
    """Device buffers of one call, freed together."""

    def __init__(self):
        self.ptrs = []

    def put(self, arr):
        arr = np.ascontiguousarray(arr) if not (arr.flags.c_contiguous or arr.flags.f_contiguous) else arr
        p = vp()
        if hip.hipMalloc(ctypes.byref(p), max(arr.nbytes, 16)) != 0:
            raise Fail("hipMalloc")
        if arr.nbytes and hip.hipMemcpy(p, arr.ctypes.data_as(vp), arr.nbytes, 1) != 0:
            raise Fail("hipMemcpy H2D")
        self.ptrs.append(p)
        return p

    def get(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        if out.nbytes and hip.hipMemcpy(out.ctypes.data_as(vp), p, out.nbytes, 2) != 0:
            raise Fail("hipMemcpy D2H")
        return out

    def free(self):
        for p in self.ptrs:
            hip.hipFree(p)
        self.ptrs = []


def problem(B, T, m, k, *, x_layout, ldx, xbs, w_layout=L.W_ROW_MAJOR, max_iter=60, tol=0.0, loss=L.LOSS_FROBENIUS,
            update_h=1, check_every=10):
    p = L.Problem()
    p.struct_size = ctypes.sizeof(L.Problem)
    p.batch, p.n_samples, p.n_features, p.n_components = B, T, m, k
    p.x_layout, p.update_h, p.w_layout, p.loss = x_layout, update_h, w_layout, loss
    p.ldx, p.x_batch_stride, p.max_iter, p.check_every, p.tol = ldx, xbs, max_iter, check_every, tol
    return p


_inputs = {}
_inputs_lock = threading.Lock()


def fit_inputs(dtype, B, T, m, k, row_major):
    """Host inputs of one batch (cached: every thread uploads its own device copy of the same bytes)."""
    key = (np.dtype(dtype).name, B, T, m, k, row_major)
    with _inputs_lock:
        if key not in _inputs:
            Xs, Ws, Hs = [], [], []
            for b in range(B):
                X = emg_matrix(100 + b, T=T, m=m, k_true=min(k, 4), dtype=dtype)  # F-order = channel-major
                W0, H0 = random_init(X, k, seed=b)
                Xs.append(np.ascontiguousarray(X) if row_major else np.ascontiguousarray(X.T))
                Ws.append(W0)
                Hs.append(H0)
            _inputs[key] = (np.stack(Xs), np.stack(Ws), np.stack(Hs))
        return _inputs[key]


def fit_case(dtype, B, T, m, k, *, variant=0, threads=0, row_major=False, max_iter=60, tol=0.0, loss=0, expect=None, max_slices=0):
    """hipnmf_fit_batched_*: returns (kernel name, [W, H, err, n_iter, sse, xsq])."""
    sfx = "f32" if dtype == np.float32 else "f64"

    def run(h):
        X, W0, H0 = fit_inputs(dtype, B, T, m, k, row_major)
        d = Dev()
        try:
            dX, dW, dH = d.put(X), d.put(W0), d.put(H0)
            dE, dN = d.put(np.zeros(B, dtype)), d.put(np.zeros(B, np.int32))
            dS, dQ = d.put(np.zeros((B, m), dtype)), d.put(np.zeros((B, m), dtype))
            p = problem(B, T, m, k, x_layout=L.X_ROW_MAJOR if row_major else L.X_CHANNEL_MAJOR, ldx=m if row_major else T,
                        xbs=T * m, max_iter=max_iter, tol=tol, loss=loss)
            ok(lib.hipnmf_set_tuning(h, threads, max_slices, variant), "set_tuning")
            ok(getattr(lib, "hipnmf_fit_batched_" + sfx)(h, ctypes.byref(p), dX, dW, dH, dE, dN, dS, dQ), "fit_batched")
            name = lib.hipnmf_last_kernel(h).decode()
            ok(lib.hipnmf_set_tuning(h, 0, 0, 0), "set_tuning")
            out = [d.get(dW, W0.shape, dtype), d.get(dH, H0.shape, dtype), d.get(dE, (B,), dtype), d.get(dN, (B,), np.int32),
                   d.get(dS, (B, m), dtype), d.get(dQ, (B, m), dtype)]
            if expect and expect not in name:
                raise Fail("expected a kernel matching %r, ran %r" % (expect, name))
            return name, out
        finally:
            d.free()

    return run


def ragged_case(dtype, Ts, m, k, max_iter=40):
    sfx = "f32" if dtype == np.float32 else "f64"

    def run(h):
        d = Dev()
        try:
            B = len(Ts)
            desc = np.zeros((B, 4), np.int64)
            xs, ws, hs = [], [], []
            xoff = woff = 0
            for b, T in enumerate(Ts):
                X, W0, H0 = fit_inputs(dtype, 1, T, m, k, False)
                ld = (T + 3) // 4 * 4
                xb = np.zeros((m, ld), dtype)
                xb[:, :T] = X[0]
                wb = np.zeros((k, ld), dtype)
                wb[:, :T] = W0[0].T
                desc[b] = (T, xoff, ld, woff)
                xoff += xb.size
                woff += wb.size
                xs.append(xb.ravel())
                ws.append(wb.ravel())
                hs.append(H0[0])
            Xp, Wp, Hp = np.concatenate(xs), np.concatenate(ws), np.stack(hs)
            dX, dW, dH = d.put(Xp), d.put(Wp), d.put(Hp)
            dE, dN = d.put(np.zeros(B, dtype)), d.put(np.zeros(B, np.int32))
            p = problem(B, max(Ts), m, k, x_layout=L.X_CHANNEL_MAJOR, ldx=max(Ts), xbs=1, w_layout=L.W_COMPONENT_MAJOR,
                        max_iter=max_iter)
            ok(getattr(lib, "hipnmf_fit_ragged_" + sfx)(h, ctypes.byref(p), desc.ctypes.data_as(vp), dX, dW, dH, dE, dN, None, None),
               "fit_ragged")
            name = lib.hipnmf_last_kernel(h).decode()
            return name + "[ragged]", [d.get(dW, Wp.shape, dtype), d.get(dH, Hp.shape, dtype), d.get(dE, (B,), dtype)]
        finally:
            d.free()

    return run


def sweep_case(dtype, B, T, m, k_min, k_max, stop, max_iter=40):
    sfx = "f32" if dtype == np.float32 else "f64"

    def run(h):
        d = Dev()
        try:
            X, _, _ = fit_inputs(dtype, B, T, m, k_min, False)
            nk = k_max - k_min + 1
            dX = d.put(X)
            dW = d.put(np.zeros((B, T, k_max), dtype))
            nH = sum(B * k * m for k in range(k_min, k_max + 1))
            dH, dV, dSel = d.put(np.zeros(nH, dtype)), d.put(np.zeros((B, nk), dtype)), d.put(np.zeros(B, np.int32))
            dE, dN = d.put(np.zeros((B, nk), dtype)), d.put(np.zeros((B, nk), np.int32))
            p = problem(B, T, m, k_max, x_layout=L.X_CHANNEL_MAJOR, ldx=T, xbs=T * m, max_iter=max_iter)
            fn = getattr(lib, ("hipnmf_rank_sweep_stop_" if stop else "hipnmf_rank_sweep_") + sfx)
            ok(fn(h, ctypes.byref(p), k_min, k_max, 0.9, 7, 0, dX, dW, dH, dV, dSel, dE, dN), "rank_sweep")
            return ("rank_sweep_stop" if stop else "rank_sweep"), [d.get(dH, (nH,), dtype), d.get(dV, (B, nk), dtype),
                                                                   d.get(dSel, (B,), np.int32), d.get(dN, (B, nk), np.int32)]
        finally:
            d.free()

    return run


def random_init_case(dtype, B, T, m, k):
    sfx = "f32" if dtype == np.float32 else "f64"

    def run(h):
        d = Dev()
        try:
            X, W0, H0 = fit_inputs(dtype, B, T, m, k, False)
            dX, dW, dH = d.put(X), d.put(np.zeros_like(W0)), d.put(np.zeros_like(H0))
            p = problem(B, T, m, k, x_layout=L.X_CHANNEL_MAJOR, ldx=T, xbs=T * m)
            ok(getattr(lib, "hipnmf_random_init_" + sfx)(h, ctypes.byref(p), 11, 3, dX, dW, dH), "random_init")
            return "random_init", [d.get(dW, W0.shape, dtype), d.get(dH, H0.shape, dtype)]
        finally:
            d.free()

    return run


def envelope_case(dtype, B, T, m, window, n_out):
    sfx = "f32" if dtype == np.float32 else "f64"

    def run(h):
        d = Dev()
        try:
            raw = np.stack([raw_emg(b, T, m) for b in range(B)]).astype(dtype)  # [B, T, m] row-major
            dR = d.put(raw)
            To = n_out if n_out else T
            dO = d.put(np.zeros((B, m, To), dtype))
            p = EnvelopeParams(ctypes.sizeof(EnvelopeParams), B, T, m, L.X_ROW_MAJOR, m, T * m, window, 1, n_out, 1, 0, 0)
            ok(getattr(lib, "hipnmf_emg_envelope_" + sfx)(h, ctypes.byref(p), dR, dO), "emg_envelope")
            return "emg_envelope", [d.get(dO, (B, m, To), dtype)]
        finally:
            d.free()

    return run


_SOS = np.array([[2.91464945e-05, 5.82929890e-05, 2.91464945e-05, 1.0, -1.86689228, 0.87521455],
                 [1.0, 2.0, 1.0, 1.0, -1.93296719, 0.94170979]])  # an order-4 low-pass as scipy.signal.butter(.., output="sos") lays it out


def sosfilt_case(dtype, B, T, m, zero_lag, mode=0):
    sfx = "f32" if dtype == np.float32 else "f64"

    def run(h):
        d = Dev()
        try:
            raw = np.stack([raw_emg(b, T, m) for b in range(B)]).astype(dtype)
            dR, dO = d.put(raw), d.put(np.zeros((B, m, T), dtype))
            p = SosfiltParams(ctypes.sizeof(SosfiltParams), B, T, m, L.X_ROW_MAJOR, m, T * m, 2, zero_lag, -1, 1, 1, mode)
            ok(getattr(lib, "hipnmf_sosfilt_" + sfx)(h, ctypes.byref(p), _SOS.ctypes.data_as(vp), None, dR, dO), "sosfilt")
            return "sosfilt_scan" if mode else "sosfilt", [d.get(dO, (B, m, T), dtype)]
        finally:
            d.free()

    return run


def shard_case(dtype, T, m, k, *, wide=False, loss=0, iters=3):
    """hipnmf_shard_pass / _hupdate / _residual (one rank: sums go straight back in), `iters` iterations.  Narrow layouts
    (channel-major X, component-major W) or the general-shape ones (row-major X, W padded to 16 components: W_ROW_MAJOR_PAD16)."""
    sfx = "f32" if dtype == np.float32 else "f64"

    def run(h):
        d = Dev()
        try:
            X, W0, H0 = fit_inputs(dtype, 1, T, m, k, wide)
            if wide:
                vec = 16 // np.dtype(dtype).itemsize
                ldx, kp = (m + vec - 1) // vec * vec, (k + 15) // 16 * 16
                Xc = np.zeros((1, T, ldx), dtype)
                Xc[:, :, :m] = X
                Wc = np.zeros((1, T, kp), dtype)
                Wc[:, :, :k] = W0
                p = problem(1, T, m, k, x_layout=L.X_ROW_MAJOR, ldx=ldx, xbs=T * ldx, w_layout=L.W_ROW_MAJOR_PAD16, max_iter=1, loss=loss)
            else:
                Xc, Wc = X, np.ascontiguousarray(W0.transpose(0, 2, 1))
                p = problem(1, T, m, k, x_layout=L.X_CHANNEL_MAJOR, ldx=T, xbs=T * m, w_layout=L.W_COMPONENT_MAJOR, max_iter=1)
            dX, dW, dH = d.put(Xc), d.put(Wc), d.put(H0)
            dS = d.put(np.zeros((1, k * m + k * k), dtype))
            dE, dQ = d.put(np.zeros((1, m), dtype)), d.put(np.zeros((1, m), dtype))
            for _ in range(iters):
                ok(getattr(lib, "hipnmf_shard_pass_" + sfx)(h, ctypes.byref(p), dX, dW, dH, dS), "shard_pass")
                ok(getattr(lib, "hipnmf_shard_hupdate_" + sfx)(h, ctypes.byref(p), dH, dS), "shard_hupdate")
            ok(getattr(lib, "hipnmf_shard_residual_" + sfx)(h, ctypes.byref(p), dX, dW, dH, dE, dQ), "shard_residual")
            return ("shard_wide" if wide else "shard_narrow") + ("_kl" if loss else ""), [
                d.get(dW, Wc.shape, dtype), d.get(dH, H0.shape, dtype), d.get(dS, (1, k * m + k * k), dtype), d.get(dE, (1, m), dtype),
                d.get(dQ, (1, m), dtype)]
        finally:
            d.free()

    return run


def tsharded_case(dtype, T, m, k, *, wide=False, max_iter=25):
    """hipnmf_fit_tsharded_* with a HOST-SIDE all-reduce callback (one rank: it synchronises the stream, reads the buffer back,
    counts the call and leaves the sums as they are) -- the C frames call back into Python from several threads at once."""
    sfx = "f32" if dtype == np.float32 else "f64"
    hip.hipStreamSynchronize.argtypes = [vp]

    def run(h):
        d = Dev()
        calls = []

        def cb(buf, count, elem_size, stream, user):
            if hip.hipStreamSynchronize(vp(stream)) != 0:
                return 1
            host = np.empty(count, dtype)
            if hip.hipMemcpy(host.ctypes.data_as(vp), vp(buf), count * elem_size, 2) != 0:
                return 1
            calls.append((count, float(host.sum())))
            return 0

        fn_cb = L.ALLREDUCE_FN(cb)
        try:
            X, W0, H0 = fit_inputs(dtype, 1, T, m, k, wide)
            if wide:
                vec = 16 // np.dtype(dtype).itemsize
                ldx, kp = (m + vec - 1) // vec * vec, (k + 15) // 16 * 16
                Xc = np.zeros((1, T, ldx), dtype)
                Xc[:, :, :m] = X
                Wc = np.zeros((1, T, kp), dtype)
                Wc[:, :, :k] = W0
                p = problem(1, T, m, k, x_layout=L.X_ROW_MAJOR, ldx=ldx, xbs=T * ldx, w_layout=L.W_ROW_MAJOR_PAD16, max_iter=max_iter)
            else:
                Xc, Wc = X, np.ascontiguousarray(W0.transpose(0, 2, 1))
                p = problem(1, T, m, k, x_layout=L.X_CHANNEL_MAJOR, ldx=T, xbs=T * m, w_layout=L.W_COMPONENT_MAJOR, max_iter=max_iter)
            dX, dW, dH = d.put(Xc), d.put(Wc), d.put(H0)
            dE, dN = d.put(np.zeros(1, dtype)), d.put(np.zeros(1, np.int32))
            dS, dQ = d.put(np.zeros((1, m), dtype)), d.put(np.zeros((1, m), dtype))
            fn = getattr(lib, "hipnmf_fit_tsharded_" + sfx)
            fn.argtypes = [vp, vp, vp, vp, vp, L.ALLREDUCE_FN, vp, vp, vp, vp, vp]
            fn.restype = ctypes.c_int
            ok(fn(h, ctypes.addressof(p), dX, dW, dH, fn_cb, None, dE, dN, dS, dQ), "fit_tsharded")
            if len(calls) != max_iter + 1 or calls[0][0] != k * m + k * k or calls[-1][0] != 2 * m:
                raise Fail("fit_tsharded: %d all-reduce calls, sizes %s" % (len(calls), sorted({c for c, _ in calls})))
            return "fit_tsharded_wide" if wide else "fit_tsharded", [
                d.get(dW, Wc.shape, dtype), d.get(dH, H0.shape, dtype), d.get(dE, (1,), dtype), d.get(dS, (1, m), dtype),
                np.array([s for _, s in calls])]
        finally:
            d.free()

    return run


def nndsvd_case(dtype, B, T, m, k):
    """hipnmf_gram, then (host: eigenvectors of the Gram matrix) hipnmf_nndsvd_stats and hipnmf_nndsvd_write."""
    sfx = "f32" if dtype == np.float32 else "f64"

    def run(h):
        d = Dev()
        try:
            X, _, _ = fit_inputs(dtype, B, T, m, k, True)
            dX = d.put(X)
            p = problem(B, T, m, k, x_layout=L.X_ROW_MAJOR, ldx=m, xbs=T * m)
            dG, dC = d.put(np.zeros((B, m, m))), d.put(np.zeros((B, m)))
            ok(getattr(lib, "hipnmf_gram_" + sfx)(h, ctypes.byref(p), dX, dG, dC), "gram")
            G = d.get(dG, (B, m, m), np.float64)
            V, inv_s = np.zeros((B, k, m)), np.zeros((B, k))
            for b in range(B):
                w, v = np.linalg.eigh(G[b])
                V[b] = v[:, ::-1][:, :k].T
                inv_s[b] = 1.0 / np.sqrt(np.maximum(w[::-1][:k], 1e-300))
            dV, dI = d.put(V), d.put(inv_s)
            dSt = d.put(np.zeros((B, k, 4)))
            ok(getattr(lib, "hipnmf_nndsvd_stats_" + sfx)(h, ctypes.byref(p), dX, dV, dI, dSt), "nndsvd_stats")
            st = d.get(dSt, (B, k, 4), np.float64)
            coef = np.ones((B, k, 2))
            coef[:, :, 0] = 0.5
            dCo, dF = d.put(coef), d.put(np.full(B, 0.01))
            dW = d.put(np.zeros((B, T, k), dtype))
            ok(getattr(lib, "hipnmf_nndsvd_write_" + sfx)(h, ctypes.byref(p), dX, dV, dI, dCo, dF, 1e-6, dW), "nndsvd_write")
            return "gram+nndsvd_stats+nndsvd_write", [G, d.get(dC, (B, m), np.float64), st, d.get(dW, (B, T, k), dtype)]
        finally:
            d.free()

    return run


f32, f64 = np.float32, np.float64
CASES = {
    # narrow kernels (nmf_kernels.hpp, nmf_rowlane.hpp, nmf_small.hpp)
    "persistent": fit_case(f32, 24, 3000, 12, 4, variant=4, expect="fit_persistent_kernel"),
    "persistent_f64_stop": fit_case(f64, 6, 2500, 8, 3, variant=1, max_iter=200, tol=1e-4),
    "rowlane": fit_case(f32, 24, 3000, 16, 5, variant=5, row_major=True, expect="fit_rowlane_kernel"),
    "rowlane_kl": fit_case(f32, 8, 2000, 16, 6, row_major=True, loss=1),
    "small": fit_case(f32, 40, 200, 16, 5, variant=6, expect="fit_small_kernel"),
    "coop": fit_case(f32, 1, 10000, 16, 5, variant=3, max_iter=120, expect="fit_coop_kernel"),
    "coop_f64": fit_case(f64, 1, 6000, 8, 4, variant=3, max_iter=80, expect="fit_coop_kernel"),
    "sliced_graph": fit_case(f32, 1, 60000, 16, 5, variant=2, max_iter=140, expect="slice_pass"),
    "sliced_graph_stop": fit_case(f64, 2, 30000, 8, 3, variant=2, max_iter=200, tol=1e-5, expect="slice_pass"),
    # wide kernels: the same template instance with different dynamic-LDS sizes (W cache rows follow T)
    "wide_a": fit_case(f32, 6, 700, 64, 12, variant=1, expect="fit_wide_kernel"),
    "wide_b": fit_case(f32, 3, 5000, 64, 12, variant=1, expect="fit_wide_kernel"),
    "wide_f64": fit_case(f64, 3, 900, 40, 10, variant=1, expect="fit_wide_kernel"),
    "wide_kl": fit_case(f32, 3, 900, 64, 12, loss=1, expect="fit_wide_kernel", max_slices=1),  # (max_slices = 1: one workgroup per matrix)
    "wide4_kl": fit_case(f32, 3, 900, 64, 6, loss=1, expect="fit_wide4_kernel", max_slices=1),
    "wide4_kl_32": fit_case(f32, 130, 400, 24, 7, loss=1, row_major=True, expect="fit_wide4_kernel<32,2,8,1,1>"),
    "kl_long_sliced": fit_case(f32, 2, 12000, 24, 6, loss=1, row_major=True, max_iter=20, expect="big1_pass_kernel<float,16"),
    "kl_long_sliced_f64": fit_case(f64, 1, 9000, 12, 4, loss=1, max_iter=20, expect="big1_pass_kernel<double,16"),
    "wide4d_kl": fit_case(f64, 3, 700, 64, 6, loss=1, expect="fit_wide4d_kernel<64,2,8,1,2,1>", max_slices=1),
    "wide4d_kl_16": fit_case(f64, 130, 300, 12, 4, loss=1, row_major=True, expect="fit_wide4d_kernel<16,1,8,1,2,1>"),
    "wide4_a": fit_case(f32, 6, 600, 64, 8, variant=1, expect="fit_wide4_kernel"),
    "wide4_b": fit_case(f32, 3, 4000, 64, 8, variant=1, expect="fit_wide4_kernel"),
    "wide4d": fit_case(f64, 4, 1200, 64, 6, variant=1, expect="fit_wide4d_kernel"),
    "wide_sliced": fit_case(f32, 1, 20000, 64, 12, variant=2, max_iter=140, expect="[sliced]"),
    "wide4_sliced": fit_case(f32, 1, 20000, 64, 8, variant=2, max_iter=140, expect="[sliced]"),
    "wide4d_sliced_stop": fit_case(f64, 1, 20000, 64, 5, variant=2, max_iter=200, tol=1e-5, expect="[sliced]"),
    "wide_sliced_auto": fit_case(f64, 1, 20000, 64, 4, max_iter=140, expect="[sliced]"),
    # general shapes (nmf_big1.hpp one-pass kernel: a kernel-node graph of two launches per iteration, per-handle workspace;
    # nmf_big.hpp for float64 and the Kullback-Leibler loss) and the one-wave 256-channel instance
    "big": fit_case(f32, 2, 3000, 300, 20, max_iter=140, expect="big1_pass_kernel<float,32,4"),
    "big_stop": fit_case(f32, 3, 1500, 144, 48, max_iter=200, tol=1e-4, expect="big1_pass_kernel<float,48,2"),
    "big_f64": fit_case(f64, 2, 1200, 130, 33, max_iter=140, expect="big_pass_w_kernel<double"),
    "big1_f64": fit_case(f64, 2, 1200, 200, 20, max_iter=140, expect="big1_pass_kernel<double,32,2,2"),
    "big_kl": fit_case(f32, 2, 1500, 300, 20, loss=1, expect="big1_pass_kernel<float,32,4,4,true,2,1>"),
    "big_kl_two_pass": fit_case(f32, 2, 1200, 300, 40, loss=1, expect="big_pass_w_kernel<float"),
    "big_ragged": ragged_case(f32, [500, 900, 700], 300, 20),
    "wide_xl": fit_case(f32, 3, 2000, 256, 16, expect="fit_wide_kernel<float,256"),
    # time-shard building blocks and the native sharded loop with a host-side collective callback
    "shard_narrow": shard_case(f32, 40000, 16, 5),
    "shard_narrow_f64": shard_case(f64, 20000, 8, 3),
    "shard_wide": shard_case(f32, 6000, 200, 20, wide=True),
    "shard_wide_kl": shard_case(f64, 3000, 64, 8, wide=True, loss=1),
    "fit_tsharded": tsharded_case(f32, 40000, 16, 5),
    "fit_tsharded_wide": tsharded_case(f32, 5000, 64, 12, wide=True),
    # on-device NNDSVD building blocks
    "nndsvd": nndsvd_case(f32, 4, 3000, 24, 5),
    "nndsvd_f64_wide": nndsvd_case(f64, 2, 1500, 100, 8),
    # the other entry points
    "ragged": ragged_case(f32, [900, 1400, 700, 1100], 12, 4),
    "ragged_wide": ragged_case(f32, [500, 900, 700], 48, 6),
    "rank_sweep": sweep_case(f32, 12, 1500, 16, 2, 5, False),
    "rank_sweep_stop": sweep_case(f32, 12, 1500, 16, 2, 5, True),
    "random_init": random_init_case(f64, 5, 800, 40, 6),
    "envelope": envelope_case(f32, 6, 6000, 16, 101, 0),
    "envelope_tn": envelope_case(f64, 4, 5000, 8, 51, 200),
    "sosfilt": sosfilt_case(f32, 6, 6000, 16, 1),
    "sosfilt_causal": sosfilt_case(f64, 3, 4000, 8, 0),
    "sosfilt_scan": sosfilt_case(f32, 6, 6000, 16, 1, mode=1),
    "sosfilt_scan_f64": sosfilt_case(f64, 3, 12000, 4, 1, mode=1),
    "sosfilt_scan_short": sosfilt_case(f32, 40, 700, 8, 1, mode=1),      # one wave per series
    "sosfilt_scan_long": sosfilt_case(f64, 2, 50000, 4, 1, mode=1),       # scan over blocks (five launches, workspace)
    "envelope_short": envelope_case(f32, 40, 900, 8, 41, 0),              # one wave per series
    "envelope_long_f64": envelope_case(f64, 2, 20000, 4, 200, 0),         # eight waves, the CU's whole LDS
}


def same(a, b):
    return len(a) == len(b) and all(x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes() for x, y in zip(a, b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--only", default="", help="comma separated case names (default: all)")
    ap.add_argument("--same-order", action="store_true", help="every thread runs the cases in the same order (same case at the same time)")
    ap.add_argument("--churn", type=int, default=0, help="every thread destroys its handle and creates a new one after this many cases (0: one handle per thread)")
    ap.add_argument("--seconds", type=float, default=0.0, help="keep starting rounds until this much time has passed (overrides --rounds as the upper bound)")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args()
    names = [n for n in (args.only.split(",") if args.only else CASES) if n]
    if args.list:
        print("\n".join(CASES))
        return 0
    hip.hipSetDevice(args.device)

    def new_handle():
        h = vp()
        ok(lib.hipnmf_create(args.device, ctypes.byref(h)), "hipnmf_create")
        return h

    # reference: one handle, one thread, one call after the other (twice: the calls must be deterministic to begin with)
    h0 = new_handle()
    base, kernels = {}, {}
    for n in names:
        kernels[n], base[n] = CASES[n](h0)
        _, again = CASES[n](h0)
        if not same(base[n], again):
            print("NOT DETERMINISTIC when run alone:", n, kernels[n], flush=True)
            return 2
        print("%-22s %s" % (n, kernels[n]), flush=True)
    ok(lib.hipnmf_destroy(h0), "hipnmf_destroy")

    errors = []
    rounds_done = [0] * args.threads
    start = threading.Barrier(args.threads)

    def worker(tid):
        try:
            hip.hipSetDevice(args.device)
            h = new_handle()
            order = list(names)
            start.wait()
            n_cases = 0
            t_end = time.monotonic() + args.seconds if args.seconds > 0 else None
            for r in range(args.rounds if t_end is None else 10**9):
                if t_end is not None and time.monotonic() > t_end:
                    break
                rounds_done[tid] = r + 1
                if not args.same_order:
                    random.Random(1000 * tid + r).shuffle(order)
                for n in order:
                    n_cases += 1
                    if args.churn and n_cases % args.churn == 0:
                        ok(lib.hipnmf_destroy(h), "hipnmf_destroy")
                        h = new_handle()
                    try:
                        _, out = CASES[n](h)
                    except Fail as e:
                        errors.append("thread %d round %d case %s: %s" % (tid, r, n, e))
                        continue
                    if not same(out, base[n]):
                        bad = [i for i, (x, y) in enumerate(zip(out, base[n])) if x.tobytes() != y.tobytes()]
                        errors.append("thread %d round %d case %s: outputs %s differ from the sequential run" % (tid, r, n, bad))
            ok(lib.hipnmf_destroy(h), "hipnmf_destroy")
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (tid, e))

    t0 = time.perf_counter()
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(args.threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    for e in errors[:40]:
        print("FAIL", e, flush=True)
    if errors:
        print("ABI-THREADS-FAILED %d errors, threads=%d rounds=%d (%.1f s)" % (len(errors), args.threads, args.rounds, dt))
        return 1
    print("ABI-THREADS-OK threads=%d rounds=%s churn=%d cases=%d (%.1f s) paths=%s" % (args.threads, "/".join(map(str, rounds_done)), args.churn, len(names), dt,
                                                                              ",".join(sorted(set(kernels.values())))))
    return 0


if __name__ == "__main__":
    sys.exit(main())

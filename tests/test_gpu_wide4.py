"""GPU parity of fit_wide4_kernel (nmf_wide4.hpp): fp32, 33..128 channels, at most 8 components, every contraction on
v_mfma_f32_4x4x1 -- against the NumPy oracle through the host API, and against the 16x16x4 formulation it replaces
(HIPNMF_WIDE4=0) on the same inputs."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT
from oracle import nmf_mu_oracle as orc
from muscle_synergies_amd.synth import emg_matrix, random_init

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rel(X, W, H, ref):
    xn = np.linalg.norm(X.astype(np.float64))
    wh = W.astype(np.float64) @ H.astype(np.float64)
    wr = ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)
    return np.linalg.norm(wh - wr) / xn


def _case(T, m, k, seed=0):
    X = emg_matrix(seed, T=T, m=m, k_true=min(6, m), dtype=np.float32)
    W0, H0 = random_init(X, k, seed)
    return X, W0, H0


def _handle(threads=0):
    from muscle_synergies_amd import _lib

    h = _lib.Handle(0)
    h.set_tuning(threads, 0, 1)  # variant 1: one workgroup per matrix (a single long matrix would take the row-sliced path)
    return h


@pytest.mark.parametrize("m", [33, 48, 50, 64, 65, 96, 100, 128])
@pytest.mark.parametrize("k", [1, 3, 4, 5, 8])
def test_wide4_shape_sweep(m, k):
    import muscle_synergies_amd as ms

    for T, threads in ((1, 0), (15, 256), (16, 0), (17, 0), (250, 256), (1003, 0)):
        h = _handle(threads)
        X, W0, H0 = _case(T, m, k, seed=m * 100 + k + T)
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=20, tol=0.0)
        res = ms.fit_batched(np.ascontiguousarray(X), W0, H0, max_iter=20, tol=0.0, handle=h)
        kern = h.last_kernel()
        # the library's own choice of waves: three per SIMD where that instance exists (5..8 components up to 64 channels)
        nw = 4 if threads == 256 else 12 if (4 < k <= 8 and m <= 64) else 8
        assert kern.startswith("fit_wide4_kernel<") and f",{nw}," in kern, kern
        assert int(res.n_iter[0]) == 20
        assert _rel(X, res.W[0], res.H[0], ref) <= TOL, (m, k, T)
        assert abs(float(res.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL
        np.testing.assert_allclose(res.W[0], ref["W"], rtol=5e-4, atol=1e-6)
        np.testing.assert_allclose(res.H[0], ref["H"], rtol=5e-4, atol=1e-6)
        assert (res.W[0] >= 0).all() and (res.H[0] >= 0).all()
        va, vc = orc.vaf(X.astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
        assert abs(res.vaf[0, 0] - va) <= TOL
        np.testing.assert_allclose(res.vaf[0, 1:], vc, atol=5e-5)


@pytest.mark.parametrize("m,k,T", [(64, 8, 9000), (128, 6, 6001), (48, 4, 20000), (96, 7, 5000)])
@pytest.mark.parametrize("threads", [256, 512, 768])
def test_wide4_rows_beyond_the_lds_cache(m, k, T, threads):
    """Matrices longer than the W cache: the first rows live in LDS for the whole fit, the rest streams -- a batch, so that
    both kinds of subtile run in every workgroup; and the same with the cache switched off."""
    import muscle_synergies_amd as ms

    X, W0, H0 = _case(T, m, k, seed=300)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=12, tol=0.0)
    for lds in ("1", "0"):
        env = dict(os.environ, HIPNMF_LDS_W=lds)
        code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
h = _lib.Handle(0); h.set_tuning({threads}, 0, 1)
X = emg_matrix(300, T={T}, m={m}, k_true=min(6, {m}), dtype=np.float32); W0, H0 = random_init(X, {k}, 300)
Xb = np.stack([X, X[::-1].copy(), X]); Wb = np.stack([W0, W0[::-1].copy(), W0]); Hb = np.stack([H0, H0, H0])
r = ms.fit_batched(Xb, Wb, Hb, max_iter=12, tol=0.0, handle=h)
assert h.last_kernel().startswith('fit_wide4_kernel<'), h.last_kernel()
np.save(sys.argv[1], np.concatenate([r.W[0].ravel(), r.H[0].ravel(), r.W[2].ravel(), r.reconstruction_err[:1].astype(np.float32)]))
"""
        out = os.path.join(tempfile.gettempdir(), f"w4_{os.getpid()}_{m}_{k}_{threads}_{lds}.npy")
        r = subprocess.run([sys.executable, "-c", code, out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        v = np.load(out)
        os.remove(out)
        W = v[: T * k].reshape(T, k)
        H = v[T * k: T * k + k * m].reshape(k, m)
        W2 = v[T * k + k * m: 2 * T * k + k * m].reshape(T, k)
        assert _rel(X, W, H, ref) <= TOL, (lds,)
        np.testing.assert_array_equal(W, W2)  # matrices 0 and 2 are the same problem: same bits, whatever ran beside them
        assert abs(float(v[-1]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL


@pytest.mark.parametrize("m,k", [(64, 8), (40, 3), (128, 5)])
def test_wide4_stop_rule_regularisation_transform(m, k):
    import muscle_synergies_amd as ms

    h = _handle()
    X, W0, H0 = _case(700, m, k, seed=41)
    for tol in (1e-3, 3e-4):
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=300, tol=tol)
        res = ms.fit_batched(X, W0, H0, max_iter=300, tol=tol, handle=h)
        assert h.last_kernel().startswith("fit_wide4_kernel<")
        assert int(res.n_iter[0]) == ref["n_iter"], (int(res.n_iter[0]), ref["n_iter"])
        assert _rel(X, res.W[0], res.H[0], ref) <= TOL
    regs = dict(l1_reg_W=0.02, l1_reg_H=0.03, l2_reg_W=0.05, l2_reg_H=0.01)
    Wr, Hr, _ = orc.fit_multiplicative_update(X, W0.copy(), H0.copy(), max_iter=30, tol=0.0, **regs)
    res = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0, handle=h, **regs)
    assert _rel(X, res.W[0], res.H[0], {"W": Wr, "H": Hr}) <= TOL
    Wt = np.full_like(W0, np.sqrt(X.mean() / k))
    Wt_ref, _, _ = orc.fit_multiplicative_update(X, Wt.copy(), Hr.copy(), max_iter=25, tol=0.0, update_H=False)
    res_t = ms.fit_batched(X, Wt, Hr, max_iter=25, tol=0.0, update_H=False, handle=h)
    np.testing.assert_array_equal(res_t.H[0], Hr)
    np.testing.assert_allclose(res_t.W[0], Wt_ref, rtol=5e-4, atol=1e-7)


def test_wide4_ragged_batch_and_padded_rows():
    """Trials of unequal length (hipnmf_fit_ragged_*), and a row-major X whose rows are padded (ldx > n_features)."""
    import torch

    import muscle_synergies_amd as ms

    m, k = 64, 7
    h = _handle()
    Ts = [17, 300, 64, 1025, 5]
    cases = [_case(T, m, k, seed=70 + s) for s, T in enumerate(Ts)]
    res = ms.fit_ragged([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], max_iter=25, tol=0.0, handle=h)
    assert h.last_kernel().startswith("fit_wide4_kernel<"), h.last_kernel()
    for s, (X, W0, H0) in enumerate(cases):
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=25, tol=0.0)
        assert _rel(X, res.W[s].cpu().numpy(), res.H[s].cpu().numpy(), ref) <= TOL, s
    X, W0, H0 = _case(333, 40, 6, seed=5)
    buf = torch.zeros((1, 333, 48), device="cuda")
    buf[0, :, :40] = torch.from_numpy(X).cuda()
    r = ms.fit_batched(buf[:, :, :40], torch.from_numpy(W0).cuda()[None], torch.from_numpy(H0).cuda()[None], max_iter=20, tol=0.0, handle=h)
    assert h.last_kernel().startswith("fit_wide4_kernel<")
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=20, tol=0.0)
    assert _rel(X, r.W[0].cpu().numpy(), r.H[0].cpu().numpy(), ref) <= TOL


def test_wide4_agrees_with_the_16x16x4_formulation_and_is_deterministic():
    """Same batch through both formulations (separate processes: the choice is read once per process): W H equal to rounding;
    two runs of the 4x4x1 kernel bitwise equal."""
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {ROOT!r})
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_batch_torch
X, W0, H0 = emg_batch_torch(300, T=1500, m=64, k=8, k_true=6, device='cuda:0', seed=9)
Xr = X.transpose(1, 2).contiguous()
a = ms.fit_batched(Xr, W0, H0, max_iter=60, tol=0.0)
b = ms.fit_batched(Xr, W0, H0, max_iter=60, tol=0.0)
assert torch.equal(a.W, b.W) and torch.equal(a.H, b.H)
print('KERNEL', _lib.get_handle(0).last_kernel())
np.save(sys.argv[1], torch.bmm(a.W, a.H).cpu().numpy())
"""
    outs = []
    for flag in ("1", "0"):
        out = os.path.join(tempfile.gettempdir(), f"w4_vs_w16_{os.getpid()}_{flag}.npy")
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, HIPNMF_WIDE4=flag), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        assert ("KERNEL fit_wide4_kernel<" if flag == "1" else "KERNEL fit_wide_kernel<float,64,16") in r.stdout, r.stdout
        outs.append(np.load(out))
        os.remove(out)
    scale = np.abs(outs[1]).max()
    assert np.abs(outs[0] - outs[1]).max() <= 2e-5 * scale


# ------------------------------------------------------------------------------------------------ float64: fit_wide4d_kernel
def _case64(T, m, k, seed=0):
    X = emg_matrix(seed, T=T, m=m, k_true=min(6, m), dtype=np.float64)
    W0, H0 = random_init(X, k, seed)
    return X, W0, H0


@pytest.mark.parametrize("m", [33, 48, 50, 64, 65, 96, 100, 128])
@pytest.mark.parametrize("k", [1, 3, 4, 5, 8])
def test_wide4d_shape_sweep_fp64(m, k):
    """float64 (what a DataFrame carries): 33..64 channels, at most 8 components on v_mfma_f64_4x4x4 (nmf_wide4d.hpp)."""
    import muscle_synergies_amd as ms

    for T, threads in ((1, 0), (15, 256), (16, 0), (17, 0), (250, 256), (1003, 0)):
        h = _handle(threads)
        X, W0, H0 = _case64(T, m, k, seed=m * 100 + k + T)
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=40, tol=0.0)
        for layout in ("C", "F"):
            Xl = np.ascontiguousarray(X) if layout == "C" else np.asfortranarray(X)
            res = ms.fit_batched(Xl, W0, H0, max_iter=40, tol=0.0, handle=h)
            kern = h.last_kernel()
            assert kern.startswith("fit_wide4d_kernel<") and f",{4 if (threads == 256 or m > 64) else 8}," in kern, kern
            np.testing.assert_allclose(res.W[0], ref["W"], rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(res.H[0], ref["H"], rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(res.reconstruction_err[0], ref["reconstruction_err"], rtol=1e-9, atol=1e-12 * np.linalg.norm(X))
            va, vc = orc.vaf(X, ref["W"], ref["H"])
            assert abs(res.vaf[0, 0] - va) <= 1e-10
            np.testing.assert_allclose(res.vaf[0, 1:], vc, atol=1e-10)


@pytest.mark.parametrize("m,k,T", [(64, 8, 5000), (48, 4, 9000), (40, 7, 3001), (128, 8, 2500), (96, 3, 4000)])
@pytest.mark.parametrize("threads", [256, 512])
def test_wide4d_rows_beyond_the_lds_cache_stop_rule_and_batch(m, k, T, threads):
    import muscle_synergies_amd as ms

    h = _handle(threads)
    cases = [_case64(T - 37 * s, m, k, seed=500 + s) for s in range(3)]
    Tm = min(c[0].shape[0] for c in cases)
    Xb = np.stack([c[0][:Tm] for c in cases]); Wb = np.stack([c[1][:Tm] for c in cases]); Hb = np.stack([c[2] for c in cases])
    res = ms.fit_batched(Xb, Wb, Hb, max_iter=25, tol=0.0, handle=h)
    assert h.last_kernel().startswith("fit_wide4d_kernel<")
    for s in range(3):
        ref = orc.nmf_mu_fit(Xb[s], Wb[s], Hb[s], max_iter=25, tol=0.0)
        np.testing.assert_allclose(res.W[s], ref["W"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(res.H[s], ref["H"], rtol=1e-9, atol=1e-13)
    X, W0, H0 = cases[0][0][:700], cases[0][1][:700], cases[0][2]
    for tol in (1e-3, 2e-4):
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=400, tol=tol)
        r = ms.fit_batched(X, W0, H0, max_iter=400, tol=tol, handle=h)
        assert int(r.n_iter[0]) == ref["n_iter"]
        np.testing.assert_allclose(r.W[0], ref["W"], rtol=1e-9, atol=1e-13)
    regs = dict(l1_reg_W=0.02, l1_reg_H=0.03, l2_reg_W=0.05, l2_reg_H=0.01)
    Wr, Hr, _ = orc.fit_multiplicative_update(X, W0.copy(), H0.copy(), max_iter=30, tol=0.0, **regs)
    r = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0, handle=h, **regs)
    np.testing.assert_allclose(r.W[0], Wr, rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(r.H[0], Hr, rtol=1e-9, atol=1e-13)
    Wt = np.full_like(W0, np.sqrt(X.mean() / k))
    Wt_ref, _, _ = orc.fit_multiplicative_update(X, Wt.copy(), Hr.copy(), max_iter=25, tol=0.0, update_H=False)
    rt = ms.fit_batched(X, Wt, Hr, max_iter=25, tol=0.0, update_H=False, handle=h)
    np.testing.assert_array_equal(rt.H[0], Hr)
    np.testing.assert_allclose(rt.W[0], Wt_ref, rtol=1e-9, atol=1e-13)


def test_wide4d_ragged_and_find_synergies_float64_frame():
    """Trials of unequal length in float64, and the reference's own call on a 64-channel float64 DataFrame: no fallback."""
    import warnings

    import pandas as pd

    import muscle_synergies_amd as ms

    h = _handle()
    m, k = 64, 6
    Ts = [17, 300, 64, 1025, 5]
    cases = [_case64(T, m, k, seed=90 + s) for s, T in enumerate(Ts)]
    res = ms.fit_ragged([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], max_iter=25, tol=0.0, handle=h)
    assert h.last_kernel().startswith("fit_wide4d_kernel<"), h.last_kernel()
    for s, (X, W0, H0) in enumerate(cases):
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=25, tol=0.0)
        np.testing.assert_allclose(res.W[s].cpu().numpy(), ref["W"], rtol=1e-9, atol=1e-13)
    X = emg_matrix(3, T=900, m=64, k_true=5, dtype=np.float64)
    df = pd.DataFrame(X, columns=[f"ch{i}" for i in range(64)])
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        out = ms.find_synergies(df, 5, solver="mu", max_iter=60, tol=0.0, init="random", random_state=0)
    assert out.model.components_.dtype == np.float64


def test_wide4_17_to_32_channels_two_rows_per_instruction():
    """fit_wide4_kernel<32, ..> / <16, ..>: up to 32 / 16 channels the W^T X products take two / four rows per instruction (CBSZ = 3 / 2).  The library
    sends these shapes there at k = 8 for batches; HIPNMF_FORCE_WIDE=1 (read once per process) sends every shape."""
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
sys.path.insert(0, {os.path.join(ROOT, "tests")!r})
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
from oracle import nmf_mu_oracle as orc
bad = 0
for m in (3, 8, 9, 16, 17, 20, 24, 32):
    for k in (1, 4, 5, 8):
        if k > m: continue
        for T, threads in ((1, 0), (16, 256), (33, 0), (700, 512), (4000, 0)):
            h = _lib.Handle(0); h.set_tuning(threads, 0, 1)
            X = emg_matrix(m * 10 + k, T=T, m=m, k_true=min(5, m), dtype=np.float32); W0, H0 = random_init(X, k, m + k)
            r = ms.fit_batched(np.stack([X, X]), np.stack([W0, W0]), np.stack([H0, H0]), max_iter=25, tol=0.0, handle=h)
            assert h.last_kernel().startswith('fit_wide4_kernel<16,' if m <= 16 else 'fit_wide4_kernel<32,'), h.last_kernel()
            ref = orc.nmf_mu_fit(X, W0, H0, max_iter=25, tol=0.0)
            xn = np.linalg.norm(X.astype(np.float64))
            d = np.linalg.norm(r.W[1].astype(np.float64) @ r.H[1].astype(np.float64) - ref['W'].astype(np.float64) @ ref['H'].astype(np.float64)) / xn
            e = abs(float(r.reconstruction_err[1]) - float(ref['reconstruction_err'])) / xn
            if not (d <= 1e-5 and e <= 1e-5 and int(r.n_iter[0]) == 25 and np.array_equal(r.W[0], r.W[1])):
                print('MISMATCH', m, k, T, threads, d, e); bad += 1
print('problems', bad)
"""
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPNMF_FORCE_WIDE="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "problems 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_wide4_kullback_leibler_up_to_32_channels():
    """Round 5: fit_wide4_kernel<32 / 16, KQ, 4, 1, LOSS = 1> -- the Kullback-Leibler flavour on the layouts with 32 / 16 lanes per row
    (W'^T Q' takes two / four rows per instruction).  HIPNMF_FORCE_WIDE=1 sends every shape there: fixed iteration count, stop rule,
    regularisation, transform and trials of unequal length against the oracle's restatement of _nmf.py:556-591, 642-684."""
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
sys.path.insert(0, {os.path.join(ROOT, "tests")!r})
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
from oracle import nmf_mu_oracle as orc
import os
WAVES = os.environ['HIPNMF_KL_WAVES']
bad = 0
KL = dict(beta_loss='kullback-leibler')
def relwh(X, W, H, Wr, Hr):
    return np.linalg.norm(W.astype(np.float64) @ H.astype(np.float64) - Wr.astype(np.float64) @ Hr.astype(np.float64)) / np.linalg.norm(X.astype(np.float64))
for m in (3, 8, 9, 16, 17, 20, 24, 32):
    for k in (1, 4, 5, 8):
        if k > m: continue
        for T in (1, 16, 33, 700, 2100):
            X = emg_matrix(m * 10 + k, T=T, m=m, k_true=min(5, m), dtype=np.float32); W0, H0 = random_init(X, k, m + k)
            for layout in ('F', 'C'):
                Xl = np.asfortranarray(X) if layout == 'F' else np.ascontiguousarray(X)
                r = ms.fit_batched(np.stack([Xl, Xl]), np.stack([W0, W0]), np.stack([H0, H0]), max_iter=25, tol=0.0, **KL)
                name = _lib.get_handle(0).last_kernel()
                assert name.startswith('fit_wide4_kernel<16,' if m <= 16 else 'fit_wide4_kernel<32,') and name.endswith(',%s,1,1>' % WAVES), name
                Wr, Hr, _ = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 25, 0.0)
                d = relwh(X, r.W[1], r.H[1], Wr, Hr)
                ref_err = np.sqrt(2 * max(orc.kl_divergence(X.astype(np.float64), Wr.astype(np.float64), Hr.astype(np.float64)), 0.0))
                e = abs(float(r.reconstruction_err[1]) - ref_err) / max(ref_err, 1.0)
                if T == 1:  # one row is fitted exactly: sqrt(2 KL) of float32 rounding noise -- compare the divergences on the scale of sum(X)
                    e = abs(float(r.reconstruction_err[1]) ** 2 - ref_err ** 2) / 2 / float(X.sum()) * 100
                if not (d <= 3e-5 and e <= 1e-4 and int(r.n_iter[0]) == 25 and np.array_equal(r.W[0], r.W[1]) and (r.W[0] >= 0).all() and (r.H[0] >= 0).all()):
                    print('MISMATCH', m, k, T, layout, d, e); bad += 1
# stop rule per matrix, regularisation, transform, ragged
for m, k in ((32, 8), (20, 5), (12, 6)):
    Xu = [emg_matrix(90 + s + m, T=600, m=m, k_true=min(6, m), dtype=np.float32) for s in range(4)]
    iu = [random_init(x, k, s) for s, x in enumerate(Xu)]
    res = ms.fit_batched(np.stack(Xu), np.stack([w for w, _ in iu]), np.stack([h for _, h in iu]), max_iter=200, tol=2e-3, **KL)
    assert _lib.get_handle(0).last_kernel().endswith(',%s,1,1>' % WAVES)
    for b in range(4):
        Wr, Hr, n_it = orc.fit_multiplicative_update_kl(Xu[b], iu[b][0].copy(), iu[b][1].copy(), 200, 2e-3)
        if int(res.n_iter[b]) != n_it or relwh(Xu[b], res.W[b], res.H[b], Wr, Hr) > 5e-5:
            print('STOP RULE', m, k, b, int(res.n_iter[b]), n_it); bad += 1
    regs = dict(l1_reg_W=0.02, l1_reg_H=0.03, l2_reg_W=0.05, l2_reg_H=0.01)
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(Xu[0], iu[0][0].copy(), iu[0][1].copy(), 30, 0.0, *regs.values())
    r = ms.fit_batched(Xu[0], iu[0][0], iu[0][1], max_iter=30, tol=0.0, **KL, **regs)
    if relwh(Xu[0], r.W[0], r.H[0], Wr, Hr) > 3e-5:
        print('REG', m, k); bad += 1
    Wt = np.full_like(iu[0][0], np.sqrt(Xu[0].mean() / k))
    Wt_ref, _, _ = orc.fit_multiplicative_update_kl(Xu[0], Wt.copy(), Hr.copy(), 20, 0.0, update_H=False)
    rt = ms.fit_batched(Xu[0], Wt, Hr, max_iter=20, tol=0.0, update_H=False, **KL)
    if not (np.array_equal(rt.H[0], Hr) and np.allclose(rt.W[0], Wt_ref, rtol=2e-3, atol=1e-6)):
        print('TRANSFORM', m, k); bad += 1
    Xs = [emg_matrix(80 + s, T=500 + 37 * s, m=m, k_true=min(6, m), dtype=np.float32) for s in range(5)]
    ir = [random_init(x, k, s) for s, x in enumerate(Xs)]
    rr = ms.fit_ragged(Xs, [w for w, _ in ir], [h for _, h in ir], max_iter=25, tol=0.0, **KL)
    assert _lib.get_handle(0).last_kernel().endswith(',%s,1,1>' % WAVES), _lib.get_handle(0).last_kernel()
    for b in range(5):
        Wr, Hr, _ = orc.fit_multiplicative_update_kl(Xs[b], ir[b][0].copy(), ir[b][1].copy(), 25, 0.0)
        if relwh(Xs[b], rr.W[b].cpu().numpy(), rr.H[b].cpu().numpy(), Wr, Hr) > 3e-5:
            print('RAGGED', m, k, b); bad += 1
print('problems', bad)
"""
    for waves in ("4", "8"):  # the 256- and the 512-thread instances (the library picks by batch size: hipnmf_wide.hip)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPNMF_FORCE_WIDE="1", HIPNMF_KL_WAVES=waves, HIPNMF_KL_SLICED="0"), capture_output=True,
                           text=True, timeout=1500)
        assert r.returncode == 0 and "problems 0" in r.stdout, waves + r.stdout[-2000:] + r.stderr[-2000:]


def test_wide4d_kullback_leibler():
    """Round 5: fit_wide4d_kernel<MP, KQ, 4, 1, WPE, LOSS = 1> -- float64 Kullback-Leibler on v_mfma_f64_4x4x4, 1..128 channels with
    at most 8 components (HIPNMF_FORCE_WIDE=1: the shapes of the lane mappings too): fixed iteration count in both layouts, stop
    rule per matrix, regularisation, transform and trials of unequal length against the oracle (_nmf.py:556-591, 642-684)."""
    code = f"""
import os, sys, numpy as np
sys.path.insert(0, {ROOT!r})
sys.path.insert(0, {os.path.join(ROOT, "tests")!r})
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
from oracle import nmf_mu_oracle as orc
bad = 0
KL = dict(beta_loss='kullback-leibler')
def relwh(X, W, H, Wr, Hr):
    return np.linalg.norm(W @ H - Wr @ Hr) / np.linalg.norm(X)
for m in (3, 16, 17, 32, 33, 48, 64, 65, 96, 100, 128):
    for k in (1, 4, 5, 8):
        if k > m: continue
        for T in (1, 16, 33, 700, 1603):
            X = emg_matrix(m * 10 + k, T=T, m=m, k_true=min(5, m), dtype=np.float64); W0, H0 = random_init(X, k, m + k)
            for layout in ('F', 'C'):
                Xl = np.asfortranarray(X) if layout == 'F' else np.ascontiguousarray(X)
                r = ms.fit_batched(np.stack([Xl, Xl]), np.stack([W0, W0]), np.stack([H0, H0]), max_iter=25, tol=0.0, **KL)
                name = _lib.get_handle(0).last_kernel()
                MP = 16 if m <= 16 else 32 if m <= 32 else 48 if m <= 48 else 64 if m <= 64 else 96 if m <= 96 else 128
                assert name.startswith('fit_wide4d_kernel<%d,' % MP) and name.endswith(',1>') and name.split(',')[2] == (os.environ['HIPNMF_KL_WAVES'] if MP <= 64 else '4'), name
                Wr, Hr, _ = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 25, 0.0)
                d = relwh(X, r.W[1], r.H[1], Wr, Hr)
                ref_err = np.sqrt(2 * max(orc.kl_divergence(X, Wr, Hr), 0.0))
                e = abs(float(r.reconstruction_err[1]) ** 2 - ref_err ** 2) / 2 / float(X.sum())
                if not (d <= 1e-11 and e <= 1e-12 and int(r.n_iter[0]) == 25 and np.array_equal(r.W[0], r.W[1]) and (r.W[0] >= 0).all() and (r.H[0] >= 0).all()):
                    print('MISMATCH', m, k, T, layout, d, e); bad += 1
for m, k in ((64, 8), (100, 5), (32, 8), (12, 6)):
    Xu = [emg_matrix(90 + s + m, T=600, m=m, k_true=min(6, m), dtype=np.float64) for s in range(4)]
    iu = [random_init(x, k, s) for s, x in enumerate(Xu)]
    res = ms.fit_batched(np.stack(Xu), np.stack([w for w, _ in iu]), np.stack([h for _, h in iu]), max_iter=200, tol=2e-3, **KL)
    assert _lib.get_handle(0).last_kernel().startswith('fit_wide4d_kernel<')
    for b in range(4):
        Wr, Hr, n_it = orc.fit_multiplicative_update_kl(Xu[b], iu[b][0].copy(), iu[b][1].copy(), 200, 2e-3)
        if int(res.n_iter[b]) != n_it or relwh(Xu[b], res.W[b], res.H[b], Wr, Hr) > 1e-10:
            print('STOP RULE', m, k, b, int(res.n_iter[b]), n_it); bad += 1
    regs = dict(l1_reg_W=0.02, l1_reg_H=0.03, l2_reg_W=0.05, l2_reg_H=0.01)
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(Xu[0], iu[0][0].copy(), iu[0][1].copy(), 30, 0.0, *regs.values())
    r = ms.fit_batched(Xu[0], iu[0][0], iu[0][1], max_iter=30, tol=0.0, **KL, **regs)
    if relwh(Xu[0], r.W[0], r.H[0], Wr, Hr) > 1e-11:
        print('REG', m, k); bad += 1
    Wt = np.full_like(iu[0][0], np.sqrt(Xu[0].mean() / k))
    Wt_ref, _, _ = orc.fit_multiplicative_update_kl(Xu[0], Wt.copy(), Hr.copy(), 20, 0.0, update_H=False)
    rt = ms.fit_batched(Xu[0], Wt, Hr, max_iter=20, tol=0.0, update_H=False, **KL)
    if not (np.array_equal(rt.H[0], Hr) and np.allclose(rt.W[0], Wt_ref, rtol=1e-9, atol=1e-13)):
        print('TRANSFORM', m, k); bad += 1
    Xs = [emg_matrix(80 + s, T=500 + 37 * s, m=m, k_true=min(6, m), dtype=np.float64) for s in range(5)]
    ir = [random_init(x, k, s) for s, x in enumerate(Xs)]
    rr = ms.fit_ragged(Xs, [w for w, _ in ir], [h for _, h in ir], max_iter=25, tol=0.0, **KL)
    assert _lib.get_handle(0).last_kernel().startswith('fit_wide4d_kernel<'), _lib.get_handle(0).last_kernel()
    for b in range(5):
        Wr, Hr, _ = orc.fit_multiplicative_update_kl(Xs[b], ir[b][0].copy(), ir[b][1].copy(), 25, 0.0)
        if relwh(Xs[b], rr.W[b].cpu().numpy(), rr.H[b].cpu().numpy(), Wr, Hr) > 1e-11:
            print('RAGGED', m, k, b); bad += 1
print('problems', bad)
"""
    for waves in ("4", "8"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPNMF_FORCE_WIDE="1", HIPNMF_KL_WAVES=waves, HIPNMF_KL_SLICED="0"), capture_output=True,
                           text=True, timeout=1500)
        assert r.returncode == 0 and "problems 0" in r.stdout, waves + r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype,m,k,T,B,kernel", [
    (np.float64, 16, 5, 600, 640, "fit_wide4d_kernel<16,2"),    # short float64 matrices, up to 16 channels
    (np.float64, 8, 4, 900, 640, "fit_wide4d_kernel<16,1"),
    (np.float64, 12, 8, 3000, 640, "fit_wide4d_kernel<16,2"),   # 7, 8 components: any length
    (np.float64, 16, 5, 3000, 640, "fit_persistent_kernel<double"),  # long: the lane mapping stays
    (np.float64, 24, 3, 4000, 640, "fit_wide4d_kernel<32,1"),   # 17..32 channels in float64: always
    (np.float32, 24, 3, 300, 640, "fit_wide4_kernel<32,1"),     # 17..32 channels in float32: short, or 7 / 8 components
    (np.float32, 32, 6, 2500, 640, "fit_wide4_kernel<32,2"),
    (np.float32, 20, 7, 6000, 640, "fit_wide4_kernel<32,2"),
    (np.float32, 32, 4, 6000, 640, "fit_persistent_kernel<float"),
    (np.float32, 16, 5, 400, 300, "fit_wide4_kernel<16,2"),     # up to 16 channels in float32: 257..600 samples (four rows per W^T X instruction)
    (np.float32, 7, 3, 500, 300, "fit_wide4_kernel<16,1"),
    (np.float32, 16, 5, 400, 640, "fit_small_kernel<float,16,5,8>"),  # ... unless the batch gives most SIMDs a wave: one wave per matrix
    (np.float64, 8, 4, 400, 800, "fit_small_kernel<double,8,4,8>"),
    (np.float64, 8, 4, 400, 640, "fit_wide4d_kernel<16,1"),
    (np.float32, 16, 5, 900, 640, "fit_persistent_kernel<float"),
    (np.float64, 16, 5, 200, 640, "fit_small_kernel<double,16,5>"),  # float64 with 9..16 channels: one wave per matrix up to 256 samples ...
    (np.float64, 12, 4, 100, 640, "fit_wide4d_kernel<16,1"),   # ... except the very short ones (half of its 256 rows would be padding)
    (np.float64, 8, 7, 128, 640, "fit_wide4d_kernel<16,2"),
    (np.float64, 8, 4, 200, 640, "fit_small_kernel<double,8,4>"),
])
def test_batches_of_narrow_shapes_routed_to_the_4x4_kernels(dtype, m, k, T, B, kernel):
    """hipnmf_api.hip::wide_preferred: batches (at least half as many matrices as CUs) of shapes the lane mappings also hold
    run on fit_wide4_kernel / fit_wide4d_kernel where that measured faster.  Same answers either way (oracle)."""
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    # (B = 640: more than two matrices per CU -- with fewer, long matrices may take the row-sliced path instead)
    X0 = emg_matrix(m + k, T=T, m=m, k_true=min(5, m), dtype=dtype)
    W00, H00 = random_init(X0, k, m)
    Xb = np.stack([X0 * (1.0 + 0.01 * b) for b in range(B)]).astype(dtype)
    Wb = np.stack([W00] * B)
    Hb = np.stack([H00 * (1.0 + 0.001 * b) for b in range(B)]).astype(dtype)
    h = _lib.Handle(0)
    res = ms.fit_batched(torch.from_numpy(Xb).cuda(), torch.from_numpy(Wb).cuda(), torch.from_numpy(Hb).cuda(), max_iter=30, tol=0.0, handle=h)
    assert h.last_kernel().startswith(kernel), h.last_kernel()
    for b in (0, 77, B - 1):
        ref = orc.nmf_mu_fit(Xb[b], Wb[b], Hb[b], max_iter=30, tol=0.0)
        W, H = res.W[b].cpu().numpy(), res.H[b].cpu().numpy()
        if dtype == np.float64:
            np.testing.assert_allclose(W, ref["W"], rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(H, ref["H"], rtol=1e-9, atol=1e-13)
        else:
            assert _rel(Xb[b], W, H, ref) <= TOL


def test_every_compiled_4x4_instance_matches_oracle():
    """Every (channel padding, component quads, waves) instance of fit_wide4_kernel / fit_wide4d_kernel, three iterations on a
    matrix with a ragged last subtile, against the oracle (see tests/test_gpu_small_long.py for why every instance)."""
    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
sys.path.insert(0, {os.path.join(ROOT, "tests")!r})
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
from oracle import nmf_mu_oracle as orc
bad, seen = 0, set()
for dtype, chans, waves in ((np.float32, (16, 32, 48, 64, 96, 128), (256, 512, 768)), (np.float64, (16, 32, 48, 64, 96, 128), (256, 512))):
    for MP in chans:
        for k in (3, 7):
            for threads in waves:
                m = MP - 1 if k == 3 else MP
                h = _lib.Handle(0); h.set_tuning(threads, 0, 1)
                X = emg_matrix(MP + k, T=203, m=m, k_true=min(5, m), dtype=dtype); W0, H0 = random_init(X, k, MP + k)
                r = ms.fit_batched(np.stack([X, X]), np.stack([W0, W0]), np.stack([H0, H0]), max_iter=3, tol=0.0, handle=h)
                name = h.last_kernel()
                assert name.startswith('fit_wide4_kernel<%d,' % MP if dtype == np.float32 else 'fit_wide4d_kernel<%d,' % MP), name
                seen.add(name)
                ref = orc.nmf_mu_fit(X, W0, H0, max_iter=3, tol=0.0)
                lim = 3e-6 if dtype == np.float32 else 1e-12
                d = max(np.abs(r.W[1] - ref['W']).max() / np.abs(ref['W']).max(), np.abs(r.H[1] - ref['H']).max() / np.abs(ref['H']).max())
                e = abs(float(r.reconstruction_err[1]) - float(ref['reconstruction_err'])) / np.linalg.norm(X)
                if not (d <= lim and e <= max(lim, 1e-6 if dtype == np.float32 else 1e-12)):
                    print('MISMATCH', name, m, k, d, e); bad += 1
print('instances', len(seen), 'problems', bad)
"""
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPNMF_FORCE_WIDE="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "problems 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # fp32: 4 paddings x 2 quads x 3 wave counts + 2 x 2 x 2 (96 / 128 channels: 4, 8 waves); fp64: 4 x 2 x 2 + 2 x 2 (256 threads only)
    assert "instances %d " % (4 * 2 * 3 + 2 * 2 * 2 + 4 * 2 * 2 + 2 * 2) in r.stdout, r.stdout[-300:]


# ---- Kullback-Leibler on the 4x4x1 kernel (round 4: fit_wide4_kernel<MP, KQ, 4, 1, LOSS = 1>, 33..128 channels, k <= 8, fp32) ----
@pytest.mark.parametrize("m,k", [(33, 8), (48, 4), (64, 8), (64, 3), (65, 5), (96, 8), (100, 1), (128, 8), (128, 6)])
@pytest.mark.parametrize("T", [16, 250, 1003])
def test_wide4_kullback_leibler_shape_sweep(m, k, T):
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    h1 = _lib.Handle(0)
    h1.set_tuning(0, 1, 0)  # one workgroup per matrix (a lone 1 000-row matrix would take the row-sliced one-pass kernel: test_gpu_big.py)
    X = emg_matrix(m * 7 + k, T=T, m=m, k_true=min(6, m), dtype=np.float32)
    W0, H0 = random_init(X, k, 3)
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 25, 0.0)
    for layout in ("F", "C"):
        Xl = np.asfortranarray(X) if layout == "F" else np.ascontiguousarray(X)
        res = ms.fit_batched(Xl, W0, H0, max_iter=25, tol=0.0, beta_loss="kullback-leibler", handle=h1)
        name = h1.last_kernel()
        assert name.startswith("fit_wide4_kernel<") and name.endswith(",8,1,1>"), name  # (a batch of at most one matrix per CU: 8 waves)
        xn = np.linalg.norm(X.astype(np.float64))
        d = np.linalg.norm(res.W[0].astype(np.float64) @ res.H[0] - Wr.astype(np.float64) @ Hr) / xn
        assert d <= 3e-5, (layout, d)
        ref_err = np.sqrt(2 * max(orc.kl_divergence(X.astype(np.float64), Wr.astype(np.float64), Hr.astype(np.float64)), 0.0))
        assert abs(float(res.reconstruction_err[0]) - ref_err) <= 1e-4 * max(ref_err, 1.0)
        assert (res.W[0] >= 0).all() and (res.H[0] >= 0).all()


def test_wide4_kullback_leibler_batch_stop_rule_regularisation_transform_ragged():
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    m, k = 64, 8
    Xs, Ws, Hs = [], [], []
    for s in range(5):
        X = emg_matrix(80 + s, T=500 + 37 * s if s else 500, m=m, k_true=6, dtype=np.float32)
        W0, H0 = random_init(X, k, s)
        Xs.append(X), Ws.append(W0), Hs.append(H0)
    # stop rule per matrix of a uniform batch
    Xu = [emg_matrix(90 + s, T=600, m=m, k_true=6, dtype=np.float32) for s in range(4)]
    iu = [random_init(x, k, s) for s, x in enumerate(Xu)]
    res = ms.fit_batched(np.stack(Xu), np.stack([w for w, _ in iu]), np.stack([h for _, h in iu]), max_iter=200, tol=2e-3,
                         beta_loss="kullback-leibler")
    assert _lib.get_handle(0).last_kernel().endswith(",8,1,1>")
    for b in range(4):
        Wr, Hr, n_it = orc.fit_multiplicative_update_kl(Xu[b], iu[b][0].copy(), iu[b][1].copy(), 200, 2e-3)
        assert int(res.n_iter[b]) == n_it
        xn = np.linalg.norm(Xu[b].astype(np.float64))
        assert np.linalg.norm(res.W[b].astype(np.float64) @ res.H[b] - Wr.astype(np.float64) @ Hr) / xn <= 5e-5
    # regularisation and transform
    regs = dict(l1_reg_W=0.02, l1_reg_H=0.03, l2_reg_W=0.05, l2_reg_H=0.01)
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(Xu[0], iu[0][0].copy(), iu[0][1].copy(), 30, 0.0, *regs.values())
    r = ms.fit_batched(Xu[0], iu[0][0], iu[0][1], max_iter=30, tol=0.0, beta_loss="kullback-leibler", **regs)
    xn = np.linalg.norm(Xu[0].astype(np.float64))
    assert np.linalg.norm(r.W[0].astype(np.float64) @ r.H[0] - Wr.astype(np.float64) @ Hr) / xn <= 3e-5
    Wt = np.full_like(iu[0][0], np.sqrt(Xu[0].mean() / k))
    Wt_ref, _, _ = orc.fit_multiplicative_update_kl(Xu[0], Wt.copy(), Hr.copy(), 20, 0.0, update_H=False)
    rt = ms.fit_batched(Xu[0], Wt, Hr, max_iter=20, tol=0.0, beta_loss="kullback-leibler", update_H=False)
    np.testing.assert_array_equal(rt.H[0], Hr)
    np.testing.assert_allclose(rt.W[0], Wt_ref, rtol=2e-3, atol=1e-6)
    # trials of unequal length
    rr = ms.fit_ragged(Xs, Ws, Hs, max_iter=25, tol=0.0, beta_loss="kullback-leibler")
    assert _lib.get_handle(0).last_kernel().endswith(",8,1,1>")
    for b in range(5):
        Wr, Hr, _ = orc.fit_multiplicative_update_kl(Xs[b], Ws[b].copy(), Hs[b].copy(), 25, 0.0)
        xn = np.linalg.norm(Xs[b].astype(np.float64))
        W, H = rr.W[b].cpu().numpy(), rr.H[b].cpu().numpy()
        assert np.linalg.norm(W.astype(np.float64) @ H - Wr.astype(np.float64) @ Hr) / xn <= 3e-5


@pytest.mark.gpu
@pytest.mark.parametrize("m,k,T,B,loss,kernel", [
    (24, 6, 700, 5, "frobenius", "fit_wide4d_kernel<32,2"),        # one workgroup per matrix either way: the 4x4x4 kernel's (round 5)
    (12, 4, 600, 3, "frobenius", "fit_wide4d_kernel<16,1"),
    (32, 8, 3000, 2, "frobenius", "fit_wide4d_kernel<32,2,4"),  # 7 / 8 components on 17..32 channels: never the lane mappings up to 50 000 rows ([sliced])
    (20, 7, 9000, 1, "frobenius", "fit_wide4d_kernel<32,2,4"),
    (24, 6, 9000, 1, "frobenius", "fit_wide4d_kernel<32,2,4"),     # 17..32 channels: every k, cooperative form of the lane mappings included
    (16, 8, 9000, 2, "frobenius", "fit_wide4d_kernel<16,2,4"),     # up to 16 channels: with 7 / 8 components
    (16, 5, 3000, 3, "frobenius", "fit_coop_kernel<double"),        # up to 16 channels and 6 components the cooperative form stays
    (8, 4, 900, 3, "frobenius", "fit_"),                             # up to 8 channels: the lane mappings (whichever form)
    (24, 6, 700, 5, "kullback-leibler", "fit_wide4d_kernel<32,2"),  # Kullback-Leibler beyond 8 channels: at every batch size
    (12, 3, 600, 1, "kullback-leibler", "fit_wide4d_kernel<16,1"),
    (12, 3, 2500, 1, "kullback-leibler", "big1_pass_kernel<double,16"),  # ... and row-sliced on the one-pass kernel once long enough
    (8, 4, 900, 3, "kullback-leibler", "fit_persistent_kernel<double"),
])
def test_float64_small_batches_routing_and_parity(m, k, T, B, loss, kernel):
    """Round 5: float64 calls of the reference's own size (a handful of matrices) leave the (G = 4) lane mappings where the
    4x4x4 / 16x16x4 kernels measured faster (hipnmf_api.hip, fit_batched_impl; tools/probes/f64_small_batch_ab.sh): the route taken
    and parity with the oracle through it."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    Xs = [emg_matrix(31 * m + b, T=T, m=m, k_true=min(5, m), dtype=np.float64) for b in range(B)]
    inits = [random_init(x, k, b) for b, x in enumerate(Xs)]
    res = ms.fit_batched(np.stack(Xs), np.stack([w for w, _ in inits]), np.stack([h for _, h in inits]), max_iter=40, tol=0.0, beta_loss=loss)
    name = _lib.get_handle(0).last_kernel()
    assert name.startswith(kernel), name
    if m <= 8 and loss == "frobenius":
        assert not name.startswith("fit_wide"), name
    for b in range(B):
        if loss == "frobenius":
            ref = orc.nmf_mu_fit(Xs[b], inits[b][0], inits[b][1], max_iter=40, tol=0.0)
            Wr, Hr = ref["W"], ref["H"]
        else:
            Wr, Hr, _ = orc.fit_multiplicative_update_kl(Xs[b], inits[b][0].copy(), inits[b][1].copy(), 40, 0.0)
        d = np.linalg.norm(res.W[b] @ res.H[b] - Wr @ Hr) / np.linalg.norm(Xs[b])
        assert d <= 1e-10, (b, d)
        np.testing.assert_allclose(res.H[b], Hr, rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("m,k,T,B,kernel", [
    (32, 8, 600, 5, "fit_wide4_kernel<32,2"),         # one workgroup per matrix either way: the 4x4x1 kernel's
    (20, 3, 300, 7, "fit_wide4_kernel<32,1"),
    (24, 6, 3000, 40, "fit_wide_kernel<float,32,16"),  # k >= 6 on long matrices: the row-sliced 16x16x4 kernel
    (32, 8, 9000, 1, "fit_wide_kernel<float,32,16"),   # k = 8: the cooperative lane-mapping form too
    (32, 4, 3000, 40, "fit_persistent_kernel<float,4,8,4"),  # k <= 5 beyond 1 000 rows stays
    (24, 6, 9000, 1, "fit_coop_kernel<float,4,8,6"),
])
def test_float32_17_to_32_channels_small_batches_routing_and_parity(m, k, T, B, kernel):
    """Round 5: the same finding for float32 on 17..32 channels (tools/probes/f32_small_batch_ab.sh): route and parity."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    Xs = [emg_matrix(17 * m + b, T=T, m=m, k_true=min(5, m), dtype=np.float32) for b in range(B)]
    inits = [random_init(x, k, b) for b, x in enumerate(Xs)]
    res = ms.fit_batched(np.stack(Xs), np.stack([w for w, _ in inits]), np.stack([h for _, h in inits]), max_iter=40, tol=0.0)
    name = _lib.get_handle(0).last_kernel()
    assert name.startswith(kernel), name
    for b in range(min(B, 4)):
        ref = orc.nmf_mu_fit(Xs[b], inits[b][0], inits[b][1], max_iter=40, tol=0.0)
        assert _rel(Xs[b], res.W[b], res.H[b], ref) <= TOL, b

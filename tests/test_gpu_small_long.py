"""GPU parity of the fit_small_kernel instances with 6 / 8 / 10 / 12 / 16 tiles of 64 rows in registers (one WAVE per matrix, nothing
but registers and a little LDS inside an iteration; nmf_small.hpp, inst_small_long.hpp): matrices of 257..1024 samples, the
sizes between the reference's time-normalised cycles and long recordings.  Variant 6 pins the kernel; batches that give every
SIMD a wave get it by themselves."""
import numpy as np
import pytest

from oracle import nmf_mu_oracle as orc
from muscle_synergies_amd.synth import emg_matrix, random_init

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rel(X, W, H, ref):
    xn = np.linalg.norm(X.astype(np.float64))
    return np.linalg.norm(W.astype(np.float64) @ H.astype(np.float64) - ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)) / xn


CASES = [  # dtype, m, k, T, tiles: the smallest compiled tile count that holds T rows
    (np.float32, 16, 5, 257, 6), (np.float32, 16, 8, 384, 6), (np.float32, 9, 3, 300, 6), (np.float32, 8, 8, 400, 8), (np.float32, 16, 8, 512, 8),
    (np.float32, 3, 2, 511, 8), (np.float32, 16, 6, 513, 10), (np.float32, 16, 6, 640, 10), (np.float32, 5, 1, 600, 10), (np.float32, 16, 7, 520, None),
    (np.float32, 12, 5, 768, 12), (np.float32, 8, 6, 700, 12),
    (np.float32, 16, 3, 769, 16), (np.float32, 16, 1, 1024, 16), (np.float32, 8, 5, 1000, 16), (np.float32, 7, 4, 900, 16),
    (np.float64, 8, 6, 300, 6), (np.float64, 8, 5, 384, 6), (np.float64, 8, 4, 512, 8), (np.float64, 4, 2, 400, 8), (np.float64, 8, 3, 768, 12),
    (np.float64, 6, 1, 600, 12),
]


@pytest.mark.parametrize("dtype,m,k,T,tiles", CASES)
def test_small_kernel_long_instances_match_oracle(dtype, m, k, T, tiles):
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    h = _lib.Handle(0)
    h.set_tuning(0, 0, 6)
    if tiles is None:  # 7, 8 components beyond 512 samples: no instance (the registers run out) -- refused under variant 6
        X = emg_matrix(1, T=T, m=m, k_true=3, dtype=dtype)
        W0, H0 = random_init(X, k, 1)
        with pytest.raises(_lib.HipNmfError, match="fit_small_kernel"):
            ms.fit_batched(X, W0, H0, max_iter=2, tol=0.0, handle=h)
        return
    lim = TOL if dtype == np.float32 else 1e-10
    Xs = [emg_matrix(50 + i, T=T, m=m, k_true=min(4, m), dtype=dtype) for i in range(3)]
    inits = [random_init(x, k, 60 + i) for i, x in enumerate(Xs)]
    W0, H0 = np.stack([w for w, _ in inits]), np.stack([hh for _, hh in inits])
    for order in ("C", "F"):
        Xb = np.stack(Xs) if order == "C" else np.ascontiguousarray(np.stack(Xs).transpose(0, 2, 1)).transpose(0, 2, 1)
        res = ms.fit_batched(Xb, W0, H0, max_iter=40, tol=0.0, handle=h)
        name = h.last_kernel()
        want = "fit_small_kernel<%s,%d,%d" % ("float" if dtype == np.float32 else "double", 8 if m <= 8 else 16, k)
        assert name == want + (">" if tiles == 4 else ",%d>" % tiles), name
        for i in range(3):
            ref = orc.nmf_mu_fit(Xs[i], W0[i], H0[i], max_iter=40, tol=0.0)
            assert _rel(Xs[i], res.W[i], res.H[i], ref) <= lim
            assert abs(float(res.reconstruction_err[i]) - float(ref["reconstruction_err"])) / np.linalg.norm(Xs[i]) <= max(lim, 1e-12)
            va, vc = orc.vaf(Xs[i].astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
            assert abs(res.vaf[i, 0] - va) <= max(lim, 1e-10)
    # stop rule, regularisation, transform, trials of unequal length
    ref = orc.nmf_mu_fit(Xs[0], W0[0], H0[0], max_iter=400, tol=1e-3)
    r = ms.fit_batched(Xs[0], W0[0], H0[0], max_iter=400, tol=1e-3, handle=h)
    if dtype == np.float64:
        assert int(r.n_iter[0]) == ref["n_iter"]
    else:
        assert abs(int(r.n_iter[0]) - ref["n_iter"]) <= 10
    regs = dict(l1_reg_W=0.02, l1_reg_H=0.03, l2_reg_W=0.05, l2_reg_H=0.01)
    Wr, Hr, _ = orc.fit_multiplicative_update(Xs[1], W0[1].copy(), H0[1].copy(), max_iter=30, tol=0.0, **regs)
    r = ms.fit_batched(Xs[1], W0[1], H0[1], max_iter=30, tol=0.0, handle=h, **regs)
    assert _rel(Xs[1], r.W[0], r.H[0], {"W": Wr, "H": Hr}) <= lim
    Wt_ref, _, _ = orc.fit_multiplicative_update(Xs[2], W0[2].copy(), Hr.copy(), max_iter=20, tol=0.0, update_H=False)
    rt = ms.fit_batched(Xs[2], W0[2], Hr, max_iter=20, tol=0.0, update_H=False, handle=h)
    np.testing.assert_array_equal(rt.H[0], Hr)
    np.testing.assert_allclose(rt.W[0], Wt_ref, rtol=5e-4 if dtype == np.float32 else 1e-9, atol=1e-7)
    Tr = [T, max(1, T // 2), max(1, T - 3)]
    rr = ms.fit_ragged([Xs[i][:Tr[i]] for i in range(3)], [W0[i][:Tr[i]] for i in range(3)], [H0[i] for i in range(3)], max_iter=25, tol=0.0, handle=h)
    assert h.last_kernel().startswith("fit_small_kernel<")
    for i in range(3):
        ref = orc.nmf_mu_fit(np.ascontiguousarray(Xs[i][:Tr[i]]), W0[i][:Tr[i]], H0[i], max_iter=25, tol=0.0)
        assert _rel(Xs[i][:Tr[i]], np.asarray(rr.W[i].cpu()), np.asarray(rr.H[i].cpu()), ref) <= lim


def test_small_kernel_long_is_the_librarys_choice_for_big_batches_only():
    """At least two matrices per CU (fp64: three): fit_small_kernel with 8 tiles for 16 x 400; a smaller batch of
    the same matrices keeps a workgroup per matrix (or the 4x4 kernel); shapes outside the compiled set are refused under variant 6."""
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    X = emg_matrix(9, T=400, m=16, k_true=4, dtype=np.float32)
    W0, H0 = random_init(X, 5, 9)
    h = _lib.Handle(0)
    for B, small in ((1100, True), (600, True), (300, False)):
        Xb = torch.from_numpy(np.stack([X] * 4)).cuda().repeat(B // 4, 1, 1)
        Wb = torch.from_numpy(np.stack([W0] * 4)).cuda().repeat(B // 4, 1, 1)
        Hb = torch.from_numpy(np.stack([H0] * 4)).cuda().repeat(B // 4, 1, 1)
        r = ms.fit_batched(Xb, Wb, Hb, max_iter=30, tol=0.0, handle=h)
        assert (h.last_kernel() == "fit_small_kernel<float,16,5,8>") == small, h.last_kernel()
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=30, tol=0.0)
        assert _rel(X, r.W[B - 1].cpu().numpy(), r.H[B - 1].cpu().numpy(), ref) <= TOL
        assert torch.equal(r.W[0], r.W[B - 1])
    h.set_tuning(0, 0, 6)
    big = emg_matrix(1, T=900, m=16, k_true=3, dtype=np.float32)  # 16 channels, k = 5: 16 tiles are not compiled
    wb, hb = random_init(big, 5, 1)
    with pytest.raises(_lib.HipNmfError, match="fit_small_kernel"):
        ms.fit_batched(big, wb, hb, max_iter=2, tol=0.0, handle=h)


def test_every_compiled_one_wave_instance_matches_oracle():
    """Every (dtype, channel padding, k, tiles) instance of fit_small_kernel, three iterations against the oracle: the
    instances are spill-heavy builds of one template, and one of them (float64, 16 channels, k = 5, groups of four tiles)
    once came out of the compiler wrong while its neighbours were exact."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    h = _lib.Handle(0)
    h.set_tuning(0, 0, 6)
    table = {  # (dtype, CH): {tiles: max k}   (inst_small.hip, inst_small_long.hpp)
        (np.float32, 8): {4: 8, 6: 8, 8: 8, 10: 6, 12: 6, 16: 5},
        (np.float32, 16): {4: 8, 6: 8, 8: 8, 10: 6, 12: 6, 16: 3},
        (np.float64, 8): {4: 6, 6: 6, 8: 6, 12: 3},
        (np.float64, 16): {4: 6},
    }
    bad, n = [], 0
    for (dtype, CH), tiles in table.items():
        for nt, kmax in tiles.items():
            for k in range(1, kmax + 1):
                m = CH if (k + nt) % 2 else CH - 3
                T = 64 * nt - (k % 5)
                X = emg_matrix(nt * 100 + k, T=T, m=m, k_true=min(3, m), dtype=dtype)
                W0, H0 = random_init(X, k, nt + k)
                r = ms.fit_batched(X, W0, H0, max_iter=3, tol=0.0, handle=h)
                want = "fit_small_kernel<%s,%d,%d" % ("float" if dtype == np.float32 else "double", CH, k) + (">" if nt == 4 else ",%d>" % nt)
                assert h.last_kernel() == want, (h.last_kernel(), want)
                ref = orc.nmf_mu_fit(X, W0, H0, max_iter=3, tol=0.0)
                lim = 2e-6 if dtype == np.float32 else 1e-12
                d = max(np.abs(r.W[0] - ref["W"]).max() / max(np.abs(ref["W"]).max(), 1e-30), np.abs(r.H[0] - ref["H"]).max() / np.abs(ref["H"]).max())
                n += 1
                if not d <= lim:
                    bad.append((want, T, m, float(d)))
    assert not bad, bad
    assert n == 8 + 8 + 8 + 6 + 6 + 5 + 8 + 8 + 8 + 6 + 6 + 3 + 6 + 6 + 6 + 3 + 6

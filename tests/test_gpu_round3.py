"""Round-3 GPU tests: hardening of the cooperative kernel's fence-free exchange (every flavour, generation numbers
across the 32-bit wrap), plus the items added this round (see the individual docstrings)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import nmf_mu_oracle as orc
from muscle_synergies_amd.synth import emg_matrix, random_init

pytestmark = pytest.mark.gpu
TOL = 1e-5
SOAK = os.path.join(ROOT, "tools", "coop_soak.py")


def _soak(*args, timeout=600):
    out = subprocess.run([sys.executable, SOAK, *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("coop_soak:")][-1]
    assert "mismatches=0" in line, line
    return line


@pytest.mark.parametrize("dtype,flags,kernel", [
    ("float32", [], "fit_coop_kernel<float,1,16,5,xcd>"),          # one matrix: same-XCD exchange, plain granule stores
    ("float32", ["--device-scope"], "fit_coop_kernel<float,1,16,5>"),
    ("float64", [], "fit_coop_kernel<double"),                      # two granules per value
    ("float32", ["--matrices", "8"], "fit_coop_kernel<float,1,16,5>"),
])
def test_cooperative_exchange_every_flavour_under_load(dtype, flags, kernel):
    """300 back-to-back cooperative fits per flavour while a second stream saturates the memory system: bitwise equal
    results, and the row-sliced path (no in-kernel exchange) agrees to rounding.  (The long soak -- 10 000 fits per
    flavour -- is tools/coop_soak.py through tools/coop_soak.sh; its summary is committed under profiles/.)"""
    line = _soak("--fits", "300", "--dtype", dtype, *flags)
    assert f"kernel={kernel}" in line, line


@pytest.mark.parametrize("dtype,flags", [("float32", []), ("float32", ["--device-scope"]), ("float64", [])])
def test_cooperative_generation_numbers_across_the_wrap(dtype, flags):
    """The exchange tags every granule with a generation number; the hook starts the sequence 40 below 2^32 so that the
    80-iteration fits cross the wrap (the sequence skips 0, the cleared state of the buffers): same bits as always."""
    base = _soak("--fits", "3", "--iters", "80", "--dtype", dtype, *flags)
    wrapped = _soak("--fits", "40", "--iters", "80", "--dtype", dtype, "--gen-base", str(2**32 - 40), *flags)
    assert "gen_base=4294967256" in wrapped
    # both runs compare against the same sliced-path result to rounding; their own bitwise reference is internal


def test_cooperative_fit_matches_oracle_after_protocol_change():
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    h = _lib.Handle(0)
    h.set_tuning(0, 0, 3)
    for dtype, tol in ((np.float32, TOL), (np.float64, 1e-10)):
        X = emg_matrix(31, T=6000, dtype=dtype)
        W0, H0 = random_init(X, 4, 31)
        res = ms.fit_batched(X, W0, H0, max_iter=70, tol=0.0, handle=h)
        assert h.last_kernel().startswith("fit_coop_kernel")
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=70, tol=0.0)
        wh = res.W[0].astype(np.float64) @ res.H[0].astype(np.float64)
        wr = ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)
        assert np.linalg.norm(wh - wr) / np.linalg.norm(X) <= tol


# ------------------------------------------------------------------------------------------------ rank sweep with a real stop
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_rank_sweep_stop_at_threshold_matches_compute_all(dtype):
    """hipnmf_rank_sweep_stop_* (BASELINE config #4's "k = 2..8 with VAF >= 0.90 stop"): trials that reached the
    threshold are not fitted at higher ranks, everything that IS fitted is bit-identical to the compute-all sweep,
    `selected` is identical, and the workload really discriminates (several different ranks get selected)."""
    import torch

    from muscle_synergies_amd.engine import rank_sweep_native
    from muscle_synergies_amd.synth import emg_rank_trials_torch

    X, k_true = emg_rank_trials_torch(60, T=1500, device="cuda:0", seed=3)
    Xv = X.transpose(1, 2).contiguous()
    if dtype == "float64":
        Xv = Xv.double()
    ra = rank_sweep_native(Xv, 2, 8, vaf_threshold=0.90, max_iter=120, tol=0.0, seed=5)
    rs = rank_sweep_native(Xv, 2, 8, vaf_threshold=0.90, max_iter=120, tol=0.0, seed=5, stop_at_threshold=True)
    assert torch.equal(ra.selected, rs.selected)
    assert len(set(ra.selected.tolist())) >= 3, ra.selected.tolist()  # the histogram is not degenerate
    assert ra.ranks == rs.ranks == [2, 3, 4, 5, 6, 7, 8]
    n_skipped = 0
    for i, k in enumerate(ra.ranks):
        ran = rs.n_iter[k] > 0
        # a trial runs at rank k exactly when no smaller rank reached the threshold
        expect = (ra.selected < 0) | (ra.selected >= k)
        assert torch.equal(ran, expect), k
        assert torch.equal(rs.vaf_all[ran, i], ra.vaf_all[ran, i])
        assert torch.equal(rs.components[k][ran], ra.components[k][ran])
        assert torch.equal(rs.reconstruction_err[k][ran], ra.reconstruction_err[k][ran])
        assert torch.isnan(rs.vaf_all[~ran, i]).all() and (rs.components[k][~ran] == 0).all()
        n_skipped += int((~ran).sum())
    assert n_skipped > 60  # most of the work past the stop is really skipped


def test_rank_sweep_native_up_to_rank_8_and_wide_matches_oracle():
    """The native sweep through every rank k = 2..8 (round 2's test stopped at 6) and on a wide batch (48 channels,
    k up to 10: nmf_wide.hpp), each (trial, rank) against the oracle from the same starting point."""
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd.engine import random_init_device, rank_sweep_native

    for m, kmax, T in ((16, 8, 700), (48, 10, 300)):
        Xs = np.stack([np.ascontiguousarray(emg_matrix(900 + b, T=T, m=m, k_true=5, dtype=np.float32)) for b in range(3)])
        Xd = torch.from_numpy(Xs).cuda()
        r = rank_sweep_native(Xd, 2, kmax, vaf_threshold=0.9, max_iter=40, tol=0.0, seed=11)
        for i, k in enumerate(r.ranks):
            W0, H0 = random_init_device(Xd, k, seed=11 + k)
            # the sweep returns H, err and VAF but not W: the same fit through fit_batched (same kernel, same starting point) gives
            # the W that belongs to it, and the pair is held to the parity bar itself -- |d(WH)|, |d err| relative to |X| and |d VAF|
            # all <= 1e-5 (this test used to accept 2e-5 on VAF and rtol 2e-3 on H)
            fb = ms.fit_batched(Xd, W0, H0, max_iter=40, tol=0.0)
            assert torch.equal(fb.H, r.components[k]), (m, k)
            assert torch.equal(fb.reconstruction_err, r.reconstruction_err[k]), (m, k)
            for b in range(3):
                ref = orc.nmf_mu_fit(Xs[b], W0[b].cpu().numpy(), H0[b].cpu().numpy(), max_iter=40, tol=0.0)
                va, _ = orc.vaf(Xs[b].astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
                xn = np.linalg.norm(Xs[b].astype(np.float64))
                assert abs(float(r.vaf_all[b, i]) - va) <= 1e-5, (m, k, b)
                assert abs(float(r.reconstruction_err[k][b]) - float(ref["reconstruction_err"])) / xn <= 1e-5, (m, k, b)
                wh = fb.W[b].double().cpu().numpy() @ r.components[k][b].double().cpu().numpy()
                wr = ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)
                assert np.linalg.norm(wh - wr) / xn <= 1e-5, (m, k, b)


def test_traffic_measurements_still_name_the_kernel_the_library_launches():
    """bench.py copies roofline.traffic from profiles/traffic.json, keyed on the kernel instance: if the library no
    longer launches that instance for the entry's workload, the number must not be reported for it."""
    import json

    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    for e in json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))):
        B = min(int(e["batch"]), 320)  # the instance depends on the shape and on "more matrices than CUs", not on B itself
        dt = torch.float64 if "<double" in e["kernel"] else torch.float32  # (round 6: the float64 headline has an entry too)
        X = (torch.rand((B, e["T"], e["m"]), device="cuda", dtype=dt) if e["x_layout"] == "row"
             else torch.rand((B, e["m"], e["T"]), device="cuda", dtype=dt).transpose(1, 2))
        W0 = torch.rand((B, e["T"], e["k"]), device="cuda", dtype=dt)
        H0 = torch.rand((B, e["k"], e["m"]), device="cuda", dtype=dt)
        ms.fit_batched(X, W0, H0, max_iter=2, tol=0.0)
        assert _lib.get_handle(0).last_kernel() == e["kernel"], (e["kernel"], _lib.get_handle(0).last_kernel())


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8])
def test_kullback_leibler_on_the_matrix_pipe(k):
    """fit_rowlane_kernel's Kullback-Leibler flavour (both W H reconstructions and Q H^T on v_mfma_f32_4x4x1; the library's
    choice from k = 6 on, forced here with variant 5 for every k) vs the oracle: ragged tail, padded channels, stop rule."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    h = _lib.Handle(0)
    h.set_tuning(0, 0, 5)
    for m, T in ((16, 1001), (11, 333), (9, 12000)):
        X = emg_matrix(500 + k, T=T, m=m, k_true=min(5, m), dtype=np.float32)
        W0, H0 = random_init(X, k, seed=k)
        res = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0, beta_loss="kullback-leibler", handle=h)
        assert h.last_kernel().startswith("fit_rowlane_kernel<") and h.last_kernel().endswith("[kl]"), h.last_kernel()
        Wr, Hr, _ = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 30, 0.0)
        wh = res.W[0].astype(np.float64) @ res.H[0].astype(np.float64)
        assert np.linalg.norm(wh - Wr.astype(np.float64) @ Hr.astype(np.float64)) / np.linalg.norm(X) <= 3e-5, (m, T)
        err = orc.kl_divergence(X, Wr, Hr, square_root=True)
        assert abs(float(res.reconstruction_err[0]) - err) <= 5e-3 * err  # (the float32 oracle sums three large cancelling terms)
    X = emg_matrix(77, T=2000, m=16, dtype=np.float32)
    W0, H0 = random_init(X, k, seed=3)
    res = ms.fit_batched(X, W0, H0, max_iter=200, tol=1e-3, beta_loss="kullback-leibler", l1_reg_W=0.01, l2_reg_H=0.02, handle=h)
    _, _, n_it = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 200, 1e-3, 0.01, 0.0, 0.0, 0.02)
    assert abs(int(res.n_iter[0]) - n_it) <= 10
    h.set_tuning(0, 1, 0)  # (max_slices = 1: one workgroup per matrix -- three 2 000-row matrices would take the row-sliced one-pass kernel)
    if k >= 6:  # the library's own choice among the one-workgroup kernels
        ms.fit_batched(np.stack([X] * 3), np.stack([W0] * 3), np.stack([H0] * 3), max_iter=3, tol=0.0, beta_loss="kullback-leibler", handle=h)
        assert h.last_kernel().endswith("[kl]") and "rowlane" in h.last_kernel()


def test_find_synergies_rank_range_concurrent_threads_equal_the_sequential_loop(monkeypatch):
    """The ranks of find_synergies(df, n, max, solver='mu') run from concurrent host threads (one handle and stream each):
    same VAF table, components and iteration counts as the sequential loop (HIPNMF_RANK_THREADS=0), bit for bit."""
    import time

    import pandas as pd

    import muscle_synergies_amd as ms

    X = emg_matrix(21, T=200, m=8, k_true=3, dtype=np.float64)
    df = pd.DataFrame(X, columns=[f"m{i}" for i in range(8)])
    kw = dict(solver="mu", init="random", random_state=4, max_iter=4000, tol=1e-7)
    out = {}
    for mode in ("0", "1", "1"):
        monkeypatch.setenv("HIPNMF_RANK_THREADS", mode)
        t0 = time.perf_counter()
        out[mode] = (ms.find_synergies(df, 2, 6, **kw), time.perf_counter() - t0)
    seq, par = out["0"][0], out["1"][0]
    pd.testing.assert_frame_equal(seq.vaf_values, par.vaf_values, check_exact=True)
    assert list(seq.components) == list(par.components) == [2, 3, 4, 5, 6]
    for k in seq.components:
        pd.testing.assert_frame_equal(seq.components[k], par.components[k], check_exact=True)
        assert seq.model[k].n_iter_ == par.model[k].n_iter_
    # the default initialisation draws from NumPy's global generator: seeded by the user, the concurrent call consumes it in rank
    # order like the loop (the initialisations are computed on the calling thread), and a shared RandomState instance likewise
    for make_rs in (lambda: None, lambda: np.random.RandomState(11)):
        res = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("HIPNMF_RANK_THREADS", mode)
            np.random.seed(123)
            res[mode] = ms.find_synergies(df, 2, 5, solver="mu", max_iter=300, tol=0.0, init="nndsvdar", random_state=make_rs())
        for k in res["0"].components:
            pd.testing.assert_frame_equal(res["0"].components[k], res["1"].components[k], check_exact=True)

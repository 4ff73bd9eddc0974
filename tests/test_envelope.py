"""EMG envelope preprocessing: oracle vs the reference's recorded outputs (CPU), and the
HIP kernels vs oracle / fixtures (GPU)."""
import numpy as np
import pandas as pd
import pytest

from conftest import load_npz
from oracle import emg_envelope_oracle as eo
from muscle_synergies_amd.synth import raw_emg

CASES = ("small", "odd", "tutorial")


@pytest.fixture(scope="module")
def g6():
    return load_npz("g6_envelope.npz")


def _raw(g6, name):
    T, m, win, reduce_to, seed = (int(v) for v in g6[f"{name}_params"])
    raw = g6[f"{name}_raw"] if f"{name}_raw" in g6.files else raw_emg(seed, T, m)
    np.testing.assert_allclose(raw.sum(), g6[f"{name}_raw_sum"], rtol=1e-9)
    return raw, win, reduce_to


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_outputs(g6, name):
    raw, win, reduce_to = _raw(g6, name)
    zc = eo.zero_center(raw)
    r = eo.rms(zc, win)
    tn = eo.time_normalize(r, reduce_to)
    if f"{name}_rms" in g6.files:
        np.testing.assert_allclose(zc, g6[f"{name}_zero_center"], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(r, g6[f"{name}_rms"], rtol=1e-11)
    else:
        np.testing.assert_allclose(r[[0, 1, 499, 500, 501, 9999, 19998, 19999]], g6[f"{name}_rms_rows"], rtol=1e-10)
    np.testing.assert_allclose(tn, g6[f"{name}_time_normalize"], rtol=1e-10)
    np.testing.assert_allclose(eo.normalize(tn), g6[f"{name}_normalize"], rtol=1e-10)
    np.testing.assert_allclose(eo.envelope(raw, win, reduce_to), g6[f"{name}_normalize"], rtol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("layout", ["C", "F"])
def test_gpu_chain_matches_reference_outputs(g6, name, layout):
    from muscle_synergies_amd.preprocess import emg_envelope_batched

    raw, win, reduce_to = _raw(g6, name)
    x = np.ascontiguousarray(raw) if layout == "C" else np.asfortranarray(raw)
    out = emg_envelope_batched(x, win, reduce_to=reduce_to)[0].cpu().numpy()
    np.testing.assert_allclose(out, g6[f"{name}_normalize"], rtol=1e-9, atol=1e-12)
    tn = emg_envelope_batched(x, win, reduce_to=reduce_to, normalize=False)[0].cpu().numpy()
    np.testing.assert_allclose(tn, g6[f"{name}_time_normalize"], rtol=1e-9, atol=1e-12)
    full = emg_envelope_batched(x, win, normalize=False)[0].cpu().numpy()
    if f"{name}_rms" in g6.files:
        np.testing.assert_allclose(full, g6[f"{name}_rms"], rtol=1e-9, atol=1e-12)
    else:
        np.testing.assert_allclose(full[[0, 1, 499, 500, 501, 9999, 19998, 19999]], g6[f"{name}_rms_rows"], rtol=1e-9)


@pytest.mark.gpu
def test_gpu_dataframe_mirrors_and_fp32(g6):
    from muscle_synergies_amd import preprocess as pp

    raw, win, reduce_to = _raw(g6, "small")
    df = pd.DataFrame(raw, columns=list("abcd"))
    zc = pp.zero_center(df)
    assert list(zc.columns) == list(df.columns) and zc.index.equals(df.index)
    np.testing.assert_allclose(zc.to_numpy(), g6["small_zero_center"], rtol=1e-12, atol=1e-15)
    r = pp.rms(zc, win)
    np.testing.assert_allclose(r.to_numpy(), g6["small_rms"], rtol=1e-10)
    tn = pp.time_normalize(r, reduce_to)
    np.testing.assert_allclose(tn.index.to_numpy(), np.linspace(0, 1, reduce_to))
    np.testing.assert_allclose(tn.to_numpy(), g6["small_time_normalize"], rtol=1e-10)
    np.testing.assert_allclose(pp.normalize(tn).to_numpy(), g6["small_normalize"], rtol=1e-10)
    r2 = pp.rms(df, 0.0125, sampling_frequency=2000)  # 25 samples
    np.testing.assert_allclose(r2.to_numpy(), eo.rms(raw, 25), rtol=1e-10)
    with pytest.raises(NotImplementedError):
        pp.time_normalize(r, 10, kind="lagrange")
    # fp32 I/O (fp64 accumulation inside)
    out32 = pp.emg_envelope_batched(raw.astype(np.float32), win, reduce_to=reduce_to)[0].cpu().numpy()
    assert out32.dtype == np.float32
    np.testing.assert_allclose(out32, g6["small_normalize"], rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
def test_gpu_batch_feeds_the_solver_without_a_copy():
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd.preprocess import emg_envelope_batched

    B, T, m = 6, 4000, 8
    raw = np.stack([raw_emg(100 + b, T, m) for b in range(B)])
    env = emg_envelope_batched(raw, 200, reduce_to=400)  # [B, 400, m] view on channel-major storage
    assert tuple(env.shape) == (B, 400, m) and env.stride()[1] == 1
    for b in range(B):
        np.testing.assert_allclose(env[b].cpu().numpy(), eo.envelope(raw[b], 200, 400), rtol=1e-9, atol=1e-12)
    W0, H0 = ms.random_init_batched(env, 3, seed=0)
    r = ms.fit_batched(env, W0, H0, max_iter=50, tol=0.0)
    assert bool(torch.isfinite(r.reconstruction_err).all()) and float(r.vaf[:, 0].min()) > 0.5


# Round 4: series of 1 280 .. 20 480 samples with a window go to emg_chunk_kernel (envelope_chunk.hpp); the kernels named in the
# comments below are what HIPNMF_ENV_CHUNK=0 selects (test_gpu_envelope_kernels_without_the_chunk_kernel runs the list that way).
# Shapes chosen to reach every envelope kernel (hipnmf_envelope.hip picks by shape): emg_wg_kernel (full-length
# output of a series that fits in the registers of a workgroup), emg_wave_kernel (everything else with a window: ragged tiles, long windows,
# up-sampling, the re-basing of the running prefix past 65 536 samples, unaligned channel rows) and
# emg_fused_kernel (no window / window too long for the ring).
_KERNEL_CASES = [
    # T, m, W, reduce_to, normalize, zero_center, dtype
    (8000, 3, 201, None, True, True, np.float64),     # workgroup kernel, fp64, odd window
    (8001, 2, 200, None, True, False, np.float64),    # workgroup kernel, ragged last tile, unaligned rows (F layout)
    (20000, 2, 200, None, True, True, np.float32),    # workgroup kernel, fp32 (the benchmark shape)
    (20480, 1, 256, None, False, True, np.float32),   # workgroup kernel at its register limit, not normalised
    (9216, 2, 50, None, True, True, np.float32),      # workgroup kernel, shortest float series it takes
    (9215, 2, 50, None, True, True, np.float32),      # ... one sample shorter: wave kernel
    (20000, 2, 200, None, True, True, np.float64),    # workgroup kernel, float64: 16 waves x 7 tiles
    (20480, 1, 255, None, True, False, np.float64),   # ... at its register limit
    (8193, 2, 64, None, True, True, np.float64),      # ... shortest series of the 16-wave instance
    (4096, 2, 64, None, True, True, np.float64),      # float64 with 8 waves x 6 tiles: shortest series
    (4096, 2, 1000, None, False, True, np.float64),   # window too long for the 8-wave instance's six tiles: wave kernel
    (20000, 2, 200, 200, True, True, np.float32),     # wave kernel, time-normalised (the tutorial's chain)
    (8000, 3, 201, 150, True, True, np.float64),      # wave kernel, time-normalised, fp64
    (8192, 2, 100, 2000, False, True, np.float64),    # wave kernel, dense time normalisation
    (21000, 2, 200, None, False, True, np.float64),   # wave kernel, full length (series too long for the registers)
    (70001, 2, 150, None, True, True, np.float64),    # wave kernel, prefix re-based past 65 536 samples
    (70000, 2, 150, 1000, True, True, np.float64),    # wave kernel, time-normalised, re-based
    (70000, 1, 150, None, True, True, np.float32),    # wave kernel fp32 (too long for the workgroup kernel)
    (1000, 3, 37, 2500, True, True, np.float64),      # up-sampling
    (1023, 2, 511, 3070, True, True, np.float32),     # ring exactly tile + window + 1 with both neighbours' windows live
    (8200, 2, 257, 9000, True, True, np.float64),     # time-normalised across a re-basing of the prefix (every 4096 samples)
    (513, 2, 500, None, True, True, np.float64),      # window almost as long as the series
    (300, 2, 1, 50, False, False, np.float64),        # window of one sample
    (5000, 2, 3000, None, True, True, np.float64),    # largest ring
    (5000, 2, 4000, None, True, True, np.float64),    # beyond the ring: emg_fused_kernel
    (2, 2, 2, None, True, False, np.float64),         # two samples
    (1, 2, 1, None, False, False, np.float64),        # one sample: emg_fused_kernel
    (4096, 16, 200, 200, True, True, np.float32),     # tutorial-like chain in fp32
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", _KERNEL_CASES, ids=lambda c: f"T{c[0]}-m{c[1]}-W{c[2]}-r{c[3]}-n{int(c[4])}-z{int(c[5])}-{np.dtype(c[6]).name}")
@pytest.mark.parametrize("layout", ["C", "F"])
def test_gpu_envelope_kernels_match_oracle(case, layout):
    from muscle_synergies_amd.preprocess import emg_envelope_batched

    T, m, W, reduce_to, norm, zc, dtype = case
    B = 2
    raw = np.stack([raw_emg(300 + 7 * b + T % 97, T, m) for b in range(B)]).astype(dtype)
    raw[1] += 0.37  # a DC offset that zero_center has to remove
    x = np.ascontiguousarray(raw) if layout == "C" else np.ascontiguousarray(raw.transpose(0, 2, 1)).transpose(0, 2, 1)
    out = emg_envelope_batched(x, W, reduce_to=reduce_to, normalize=norm, zero_center=zc).cpu().numpy()
    assert out.dtype == dtype
    for b in range(B):
        ref = eo.envelope(raw[b].astype(np.float64), W, reduce_to, do_zero_center=zc, do_normalize=norm)
        if dtype == np.float32:
            np.testing.assert_allclose(out[b], ref, rtol=3e-5, atol=3e-6 * np.abs(ref).max())
        else:
            np.testing.assert_allclose(out[b], ref, rtol=1e-9, atol=1e-12)


_CHUNK_CASES = [
    # T, m, W, reduce_to, normalize, zero_center, dtype, expected kernel
    (20000, 2, 200, None, True, True, np.float32, "emg_chunk_kernel<float,41,512>"),   # the benchmark shape
    (20000, 2, 200, 200, True, True, np.float32, "emg_chunk_kernel<float,41,512>"),    # ... time-normalised
    (20000, 2, 279, None, True, False, np.float32, "emg_chunk_kernel<float,41,512>"),  # widest window with two workgroups per CU
    (20480, 1, 256, None, False, True, np.float32, "emg_chunk_kernel<float,41,512>"),  # 20 480 + 127 positions of the 20 736
    (20609, 2, 255, None, True, True, np.float32, "emg_chunk_kernel<float,41,512>"),   # the last position of the last thread
    (20610, 2, 255, None, True, True, np.float32, "emg_wave_kernel"),              # one more: no instance
    (20000, 2, 1400, 300, True, True, np.float32, "emg_wave_kernel"),              # time-normalised, LDS for one workgroup only
    (20000, 2, 1400, None, True, True, np.float32, "emg_chunk_kernel<float,41,512>"),  # full length takes it all the same
    (1280, 3, 37, None, True, True, np.float32, "emg_chunk_kernel<float,25,64>"),      # up to 64 x 33 positions: one wave per series
    (2112 - 18, 3, 37, None, True, True, np.float32, "emg_chunk_kernel<float,33,64>"),  # the longest one-wave series
    (2112 - 17, 3, 37, None, True, True, np.float32, "emg_chunk_kernel<float,9,256>"),  # one more: four waves, smallest instance
    (64, 2, 5, None, True, True, np.float32, "emg_chunk_kernel<float,5,64>"),          # shortest series the kernel takes
    (318, 2, 5, 40, True, True, np.float32, "emg_chunk_kernel<float,5,64>"),           # 320 positions
    (319, 2, 5, 40, True, True, np.float32, "emg_chunk_kernel<float,9,64>"),
    (63, 2, 5, None, True, True, np.float32, "emg_wave_kernel"),
    (300, 2, 1, 50, False, False, np.float64, "emg_chunk_kernel<double,5,64>"),        # window of one sample, time-normalised
    (577, 3, 577, 1200, True, True, np.float64, "emg_chunk_kernel<double,17,64>"),     # window = series, up-sampling, one wave
    (2304 - 18, 2, 37, 100, False, True, np.float32, "emg_chunk_kernel<float,9,256>"),   # all 256 threads own a chunk
    (2304 - 17, 2, 37, 100, False, True, np.float32, "emg_chunk_kernel<float,13,256>"),
    (5000, 2, 3000, None, True, True, np.float64, "emg_chunk_kernel<double,33,256>"),  # window longer than half the series
    (5001, 3, 5001, None, True, True, np.float64, "emg_chunk_kernel<double,33,256>"),  # window = series
    (4097, 2, 1, 4000, True, False, np.float64, "emg_chunk_kernel<double,17,256>"),    # window of one sample, dense time normalisation
    (4097, 2, 2, None, True, True, np.float32, "emg_chunk_kernel<float,17,256>"),      # even window of two
    (10000, 2, 200, None, True, True, np.float64, "emg_chunk_kernel<double,41,256>"),  # float64 at the top of its range (one workgroup per CU)
    (10000, 2, 200, 500, True, True, np.float64, "emg_wave_kernel"),               # ... time-normalised: left to the wave kernel
    (9000, 2, 200, 500, True, True, np.float64, "emg_chunk_kernel<double,41,256>"),
    (10497, 2, 1, None, True, True, np.float64, "emg_chunk_kernel<double,41,512>"),  # one position more than 256 x 41: eight waves
    (20000, 2, 200, None, True, True, np.float64, "emg_chunk_kernel<double,41,512>"),  # the benchmark shape in float64: the CU's whole LDS
    (20000, 2, 200, 300, True, True, np.float64, "emg_wave_kernel"),               # ... time-normalised: one workgroup per CU does not pay
    (20480, 1, 255, None, True, False, np.float64, "emg_wg_kernel"),               # 167 KB: does not fit
    (3000, 2, 100, 9001, True, True, np.float32, "emg_chunk_kernel<float,13,256>"),    # up-sampling
    (18560, 2, 257, None, True, True, np.float32, "emg_chunk_kernel<float,73,256>"),   # 18 688 positions: the longest four-wave series
    (18561, 2, 257, 700, True, True, np.float32, "emg_chunk_kernel<float,41,512>"),    # one more: eight waves
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", _CHUNK_CASES, ids=lambda c: f"T{c[0]}-m{c[1]}-W{c[2]}-r{c[3]}-n{int(c[4])}-z{int(c[5])}-{np.dtype(c[6]).name}")
@pytest.mark.parametrize("layout", ["C", "F"])
def test_gpu_envelope_chunk_kernel(case, layout):
    """emg_chunk_kernel (round 4): every instance boundary, both outputs, and the shapes either side that must not take it."""
    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.preprocess import emg_envelope_batched

    T, m, W, reduce_to, norm, zc, dtype, kernel = case
    B = 3
    raw = np.stack([raw_emg(900 + 11 * b + T % 89, T, m) for b in range(B)]).astype(dtype)
    raw[1] += 0.37
    raw[2, :, 0] *= 1e-3  # a quiet channel next to loud ones
    x = np.ascontiguousarray(raw) if layout == "C" else np.ascontiguousarray(raw.transpose(0, 2, 1)).transpose(0, 2, 1)
    out = emg_envelope_batched(x, W, reduce_to=reduce_to, normalize=norm, zero_center=zc).cpu().numpy()
    assert _lib.get_handle(0).last_kernel() in (kernel, kernel + "[vec]")  # ([vec]: 16-byte pieces, chosen by alignment and length)
    for b in range(B):
        ref = eo.envelope(raw[b].astype(np.float64), W, reduce_to, do_zero_center=zc, do_normalize=norm)
        for c in range(m):  # per channel: the quiet one is held to its own scale
            top = float(np.abs(ref[:, c]).max())
            if dtype == np.float32:
                np.testing.assert_allclose(out[b][:, c], ref[:, c], rtol=3e-5, atol=3e-6 * top)
            else:
                np.testing.assert_allclose(out[b][:, c], ref[:, c], rtol=1e-9, atol=1e-6 * top if W <= 8 else 1e-12 * max(top, 1.0))


@pytest.mark.gpu
def test_gpu_envelope_kernels_without_the_chunk_kernel():
    """HIPNMF_ENV_CHUNK=0: the round-2 kernels (emg_wg_kernel / emg_wave_kernel) still serve every shape of _KERNEL_CASES."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    code = """
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_envelope import _KERNEL_CASES, raw_emg, eo
from muscle_synergies_amd import _lib
from muscle_synergies_amd.preprocess import emg_envelope_batched
bad = 0; seen = set()
for T, m, W, reduce_to, norm, zc, dtype in _KERNEL_CASES:
    raw = np.stack([raw_emg(300 + 7 * b + T %% 97, T, m) for b in range(2)]).astype(dtype); raw[1] += 0.37
    out = emg_envelope_batched(np.ascontiguousarray(raw), W, reduce_to=reduce_to, normalize=norm, zero_center=zc).cpu().numpy()
    seen.add(_lib.get_handle(0).last_kernel())
    for b in range(2):
        ref = eo.envelope(raw[b].astype(np.float64), W, reduce_to, do_zero_center=zc, do_normalize=norm)
        ok = np.allclose(out[b], ref, rtol=3e-5, atol=3e-6 * np.abs(ref).max()) if dtype == np.float32 else np.allclose(out[b], ref, rtol=1e-9, atol=1e-12)
        if not ok: print('MISMATCH', T, m, W, reduce_to, norm, zc, dtype); bad += 1
print('kernels', sorted(seen)); print('problems', bad)
""" % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPNMF_ENV_CHUNK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "problems 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert "emg_wg_kernel" in r.stdout and "emg_wave_kernel" in r.stdout and "emg_fused_kernel" in r.stdout and "emg_chunk" not in r.stdout


@pytest.mark.gpu
def test_gpu_envelope_fuzz():
    """tests/fuzz_envelope_gpu.py with a fixed seed: 150 random shapes / windows / options against the oracle."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_envelope_gpu.py"), "--cases", "150", "--seed", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 problems" in r.stdout



# ------------------------------------------------------------------------------------------------ time_normalize kinds (round 3)
INDEX_KINDS = ["linear", "slinear", "nearest", "nearest-up", "previous", "next", "zero"]


@pytest.fixture(scope="module")
def g10():
    from conftest import load_npz

    return load_npz("g10_time_normalize_kinds.npz")


def test_oracle_and_host_operator_of_the_spline_kinds_match_the_reference(g10):
    """G10's 'quadratic' / 'cubic' outputs of the reference's time_normalize: (a) the oracle's own restatement of
    make_interp_spline (Cox - de Boor basis on scipy's not-a-knot knots + a dense collocation solve); (b) the banded operator the
    product builds on the host (preprocess._spline_operator) applied with NumPy -- what hipnmf_resample_weights_* computes."""
    from muscle_synergies_amd.preprocess import _spline_operator

    raw = g10["raw"]
    T = raw.shape[0]
    for reduce_to in (40, 97, 230, 2, 193):
        for kind, k in (("quadratic", 2), ("cubic", 3)):
            ref = g10[f"{kind}_{reduce_to}"]
            np.testing.assert_allclose(eo.time_normalize(raw, reduce_to, kind), ref, rtol=1e-12, atol=1e-14 * np.abs(raw).max())
            first, w = _spline_operator(T, reduce_to, k)
            assert first.dtype == np.int32 and w.shape[0] == reduce_to and (first >= 0).all() and (first + w.shape[1] <= T).all()
            got = np.stack([(w[r][:, None] * raw[first[r]: first[r] + w.shape[1]]).sum(axis=0) for r in range(reduce_to)])
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-14 * np.abs(raw).max())
    with pytest.raises(ValueError):
        _spline_operator(3, 10, 3)  # scipy needs more than k samples too


def test_oracle_time_normalize_kinds_match_the_reference(g10):
    """G10: outputs of the reference's time_normalize (scipy interp1d) for every kind it forwards."""
    raw = g10["raw"]
    for reduce_to in (40, 97, 230, 2, 193):
        for kind in INDEX_KINDS:
            got = eo.time_normalize(raw, reduce_to, kind)
            if kind in ("linear", "slinear"):
                np.testing.assert_allclose(got, g10[f"{kind}_{reduce_to}"], rtol=1e-12, atol=1e-15)
            else:
                np.testing.assert_array_equal(got, g10[f"{kind}_{reduce_to}"], err_msg=f"{kind} {reduce_to}")


@pytest.mark.gpu
def test_gpu_time_normalize_kinds(g10):
    """Every interp1d kind the reference forwards, all on the device: index kinds and (s)linear in the envelope kernels, the
    spline kinds through the banded operator; the frames carry the reference's index / columns."""
    from muscle_synergies_amd import _lib
    from muscle_synergies_amd import preprocess as pp

    raw = g10["raw"]
    df = pd.DataFrame(raw, columns=list("abc"))
    for reduce_to in (40, 97, 230, 2, 193):
        for kind in INDEX_KINDS:
            out = pp.time_normalize(df, reduce_to, kind=kind)
            assert list(out.columns) == list("abc")
            np.testing.assert_allclose(out.index.to_numpy(), np.linspace(0, 1, reduce_to))
            if kind in ("linear", "slinear"):
                np.testing.assert_allclose(out.to_numpy(), g10[f"{kind}_{reduce_to}"], rtol=1e-12, atol=1e-15)
            else:
                np.testing.assert_array_equal(out.to_numpy(), g10[f"{kind}_{reduce_to}"], err_msg=f"{kind} {reduce_to}")
        for kind in ("quadratic", "cubic"):  # round 6: on the device too (banded spline operator, hipnmf_resample_weights_*)
            out = pp.time_normalize(df, reduce_to, kind=kind)
            assert _lib.get_handle(0).last_kernel() == "resample_weights_kernel<double>"
            assert list(out.columns) == list("abc")
            np.testing.assert_allclose(out.index.to_numpy(), np.linspace(0, 1, reduce_to))
            np.testing.assert_allclose(out.to_numpy(), g10[f"{kind}_{reduce_to}"], rtol=1e-12, atol=1e-14 * np.abs(raw).max())
    with pytest.raises(NotImplementedError):
        pp.time_normalize(df, 10, kind="lagrange")
    assert pp.time_normalize(df, 10, kind=0).equals(pp.time_normalize(df, 10, kind="zero"))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gpu_index_kinds_fused_with_rms_over_many_lengths(dtype):
    """The kinds inside the fused chain (zero_center -> rms -> time_normalize(kind) -> normalize) for series lengths
    and output counts that reach every envelope kernel, against the oracle."""
    from muscle_synergies_amd.preprocess import emg_envelope_batched

    rng = np.random.default_rng(3)
    for T, n_out, win in ((2, 5, 0), (3, 2, 0), (100, 33, 5), (513, 1000, 25), (2049, 200, 200), (9300, 77, 0), (700, 701, 64),
                          (5000, 4999, 4000)):
        raw = raw_emg(int(rng.integers(1000)), T, 3).astype(dtype)
        for kind in ("nearest", "nearest-up", "previous", "next", "zero", "linear"):
            got = emg_envelope_batched(raw, win, reduce_to=n_out, kind=kind)[0].cpu().numpy()
            ref = eo.envelope(raw.astype(np.float64), win, n_out, kind=kind)
            tol = dict(rtol=1e-9, atol=1e-12) if dtype == np.float64 else dict(rtol=3e-5, atol=1e-6)
            np.testing.assert_allclose(got, ref, err_msg=f"T={T} n_out={n_out} win={win} {kind}", **tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,layout", [(np.float64, "row"), (np.float32, "row"), (np.float64, "channel")])
def test_gpu_spline_time_normalize_batched_vs_scipy(dtype, layout):
    """A batch at the reference's real size (rows of a recording -> 200 points of a gait cycle) against scipy's interp1d, both
    spline kinds, both memory orders of the input."""
    from scipy.interpolate import interp1d

    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.preprocess import time_normalize_batched

    rng = np.random.default_rng(4)
    B, T, m, n_out = 5, 6001, 7, 200
    X = np.abs(rng.standard_normal((B, T, m))).cumsum(axis=1).astype(dtype)
    Xin = X if layout == "row" else np.ascontiguousarray(X.transpose(0, 2, 1)).transpose(0, 2, 1)
    for kind in ("cubic", "quadratic", 3, 2):
        out = time_normalize_batched(Xin, n_out, kind=kind).cpu().numpy()
        assert out.shape == (B, n_out, m) and out.dtype == dtype
        assert _lib.get_handle(0).last_kernel().startswith("resample_weights_kernel<")
        ref = interp1d(np.linspace(0, 1, T), X.astype(np.float64), axis=1, kind=kind)(np.linspace(0, 1, n_out))
        tol = 1e-12 if dtype == np.float64 else 3e-7
        assert np.abs(out - ref).max() <= tol * np.abs(ref).max()
    lin = time_normalize_batched(Xin, n_out, kind="linear").cpu().numpy()  # the other kinds through the same entry point
    ref = interp1d(np.linspace(0, 1, T), X.astype(np.float64), axis=1)(np.linspace(0, 1, n_out))
    assert np.abs(lin - ref).max() <= (1e-12 if dtype == np.float64 else 3e-7) * np.abs(ref).max()

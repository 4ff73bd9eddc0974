"""IIR filter stage (``digital_filter`` / ``linear_envelope`` = scipy sosfilt /
sosfiltfilt): oracle vs the reference's recorded outputs (CPU) and the HIP kernel vs both (GPU)."""
import json

import numpy as np
import pandas as pd
import pytest

from conftest import load_npz
from oracle import sosfilt_oracle as so
from muscle_synergies_amd.synth import raw_emg

CASES = ("lp4", "hp2_fwd", "bp3", "cheby1_lp5", "cheby2_bs2_fwd", "long_lp2", "short")
# the reference's digital_filter kwargs behind every fixture (tests/golden/make_golden.py::G8_CASES)
KW = {
    "lp4": dict(critical_freqs=6, order=4, filter_type="butter", band_type="lowpass", zero_lag=True),
    "hp2_fwd": dict(critical_freqs=20, order=2, filter_type="butter", band_type="highpass", zero_lag=False),
    "bp3": dict(critical_freqs=[20, 450], order=3, filter_type="butter", band_type="bandpass", zero_lag=True),
    "cheby1_lp5": dict(critical_freqs=10, order=5, filter_type="cheby1", band_type="lowpass", zero_lag=True, cheby_param=1.0),
    "cheby2_bs2_fwd": dict(critical_freqs=[45, 55], order=2, filter_type="cheby2", band_type="bandstop", zero_lag=False,
                           cheby_param=30.0),
    "long_lp2": dict(critical_freqs=4, order=2, filter_type="butter", band_type="lowpass", zero_lag=True),
    "short": dict(critical_freqs=5, order=1, filter_type="butter", band_type="lowpass", zero_lag=True),
}


@pytest.fixture(scope="module")
def g8():
    return load_npz("g8_filters.npz")


def _case(g8, name):
    T, m, fs, zero_lag, seed = (int(v) for v in g8[f"{name}_params"])
    raw = raw_emg(seed, T, m, fs=float(fs))
    if f"{name}_raw" in g8.files:
        assert np.array_equal(raw, g8[f"{name}_raw"])  # the generator is bit-reproducible across hosts
    return raw, fs, bool(zero_lag), g8[f"{name}_sos"], g8[f"{name}_zi"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_outputs(g8, name):
    raw, fs, zero_lag, sos, zi = _case(g8, name)
    assert np.allclose(so.sosfilt_zi(sos), zi, rtol=1e-13, atol=1e-300)
    y = so.digital_filter(raw, sos, zero_lag)
    if f"{name}_filtered" in g8.files:
        assert np.array_equal(y, g8[f"{name}_filtered"])  # same operations in the same order: bit-exact
    else:
        assert np.array_equal(y[g8[f"{name}_rows"]], g8[f"{name}_filtered_rows"])
        np.testing.assert_allclose(y.sum(axis=0), g8[f"{name}_filtered_colsum"], rtol=1e-12)
    if KW[name]["band_type"] == "lowpass":
        le = so.linear_envelope(raw, sos, zero_lag)
        if f"{name}_linear_envelope" in g8.files:
            np.testing.assert_allclose(le, g8[f"{name}_linear_envelope"], rtol=1e-12, atol=1e-15)
            np.testing.assert_allclose(so.linear_envelope(raw, sos, zero_lag, zero_center=False),
                                       g8[f"{name}_linear_envelope_nocenter"], rtol=0, atol=0)
        else:
            np.testing.assert_allclose(le[g8[f"{name}_rows"]], g8[f"{name}_linear_envelope_rows"], rtol=1e-12, atol=1e-15)


def test_oracle_against_live_scipy_and_host_design(g8):
    signal = pytest.importorskip("scipy.signal")
    from muscle_synergies_amd.preprocess import design_sos

    for name in CASES:
        raw, fs, zero_lag, sos, zi = _case(g8, name)
        kw = KW[name]
        mine = design_sos(kw["filter_type"], kw["order"], fs, kw["critical_freqs"], kw["band_type"], kw.get("cheby_param"))
        np.testing.assert_allclose(mine, sos, rtol=1e-12, atol=1e-300)  # design may differ in the last bits across hosts
        ref = signal.sosfiltfilt(sos, raw, axis=0) if zero_lag else signal.sosfilt(sos, raw, axis=0)
        assert np.array_equal(so.digital_filter(raw, sos, zero_lag), ref)
        assert so.default_padlen(sos) == 3 * (2 * len(sos) + 1 - min((sos[:, 2] == 0).sum(), (sos[:, 5] == 0).sum()))
    with pytest.raises(ValueError, match="must be greater than padlen"):
        so.sosfiltfilt(g8["lp4_sos"], np.ones((15, 2)))
    with pytest.raises(ValueError, match="filter type not understood"):
        design_sos("bessel", 2, 100, 5)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_filter_is_bit_exact_in_fp64(g8, name):
    from muscle_synergies_amd.preprocess import sosfilt_batched

    raw, fs, zero_lag, sos, zi = _case(g8, name)
    ref = so.digital_filter(raw, sos, zero_lag)
    for arr in (np.ascontiguousarray(raw), np.asfortranarray(raw)):  # row-major and channel-major inputs
        got = sosfilt_batched(arr, sos, zero_lag=zero_lag)[0].cpu().numpy()
        assert got.shape == raw.shape and got.dtype == np.float64
        assert np.array_equal(got, ref), np.abs(got - ref).max()
    if f"{name}_filtered" in g8.files:
        assert np.array_equal(got, g8[f"{name}_filtered"])
    else:
        assert np.array_equal(got[g8[f"{name}_rows"]], g8[f"{name}_filtered_rows"])
    if KW[name]["band_type"] == "lowpass":
        le = sosfilt_batched(raw, sos, zero_lag=zero_lag, zero_center=True, rectify=True)[0].cpu().numpy()
        np.testing.assert_allclose(le, so.linear_envelope(raw, sos, zero_lag), rtol=1e-10, atol=1e-13)  # mean: other order
        le_nc = sosfilt_batched(raw, sos, zero_lag=zero_lag, rectify=True)[0].cpu().numpy()
        assert np.array_equal(le_nc, so.linear_envelope(raw, sos, zero_lag, zero_center=False))


@pytest.mark.gpu
def test_gpu_filter_batches_fp32_and_errors(g8):
    import torch

    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.preprocess import sosfilt_batched

    sos = g8["lp4_sos"]
    # 70 recordings x 3 channels = 210 series: several waves, the last one partly filled; T not a tile multiple
    raw = np.stack([raw_emg(300 + b, 1333, 3) for b in range(70)])
    got = sosfilt_batched(raw, sos, zero_lag=True, rectify=True).cpu().numpy()
    for b in (0, 21, 69):
        assert np.array_equal(got[b], so.linear_envelope(raw[b], sos, True, zero_center=False))
    # more sections than the usual one or two (order-8 band-pass = 8 sections), forward only
    from scipy import signal

    sos8 = signal.butter(8, [20, 400], btype="bandpass", output="sos", fs=2000)
    assert sos8.shape == (8, 6)
    got8 = sosfilt_batched(raw[:2], sos8, zero_lag=False).cpu().numpy()
    assert np.array_equal(got8[1], so.sosfilt(sos8, raw[1])[0])
    got8 = sosfilt_batched(raw[:2], sos8, zero_lag=True).cpu().numpy()
    assert np.array_equal(got8[0], so.sosfiltfilt(sos8, raw[0]))
    # float32 samples: filtered in fp64 from the float values, rounded to float at the end
    raw32 = raw_emg(100, 600, 4).astype(np.float32)
    got32 = sosfilt_batched(raw32, sos, zero_lag=True)[0].cpu().numpy()
    assert got32.dtype == np.float32
    np.testing.assert_array_equal(got32, g8["lp4_filtered_from_f32"].astype(np.float32))
    # a device tensor that is a transposed view (channel-major) is used in place
    xt = torch.from_numpy(np.ascontiguousarray(raw[:4].transpose(0, 2, 1))).cuda().transpose(1, 2)
    assert np.array_equal(sosfilt_batched(xt, sos).cpu().numpy(), np.stack([so.sosfiltfilt(sos, raw[b]) for b in range(4)]))
    # explicit padlen, including none at all
    assert np.array_equal(sosfilt_batched(raw[0], sos, padlen=0)[0].cpu().numpy(), so.sosfiltfilt(sos, raw[0], padlen=0))
    assert np.array_equal(sosfilt_batched(raw[0], sos, padlen=100)[0].cpu().numpy(), so.sosfiltfilt(sos, raw[0], padlen=100))
    # scipy's errors
    with pytest.raises(ValueError, match="must be greater than padlen, which is 15"):
        sosfilt_batched(np.ones((15, 2)), sos)
    with pytest.raises(ValueError, match="sos\\[:, 3\\] should be all ones"):
        sosfilt_batched(raw[0], sos * 2.0)
    with pytest.raises(_lib.HipNmfError, match="outside the compiled kernel set"):
        sosfilt_batched(raw[0], np.tile(sos, (5, 1)))


@pytest.mark.gpu
def test_gpu_dataframe_functions_match_the_reference_outputs(g8):
    import muscle_synergies_amd.preprocess as pp

    for name in ("lp4", "cheby1_lp5", "bp3", "hp2_fwd"):
        raw, fs, zero_lag, sos, zi = _case(g8, name)
        df = pd.DataFrame(raw, columns=[f"m{j}" for j in range(raw.shape[1])], index=np.arange(len(raw)) / fs)
        out = pp.digital_filter(df, sampling_frequency=fs, **KW[name])
        assert list(out.columns) == list(df.columns) and out.index.equals(df.index) and out is not df
        np.testing.assert_allclose(out.to_numpy(), g8[f"{name}_filtered"], rtol=1e-12, atol=1e-300)
        if KW[name]["band_type"] == "lowpass":
            kw = {k: v for k, v in KW[name].items() if k != "band_type"}
            le = pp.linear_envelope(df, sampling_frequency=fs, **kw)
            np.testing.assert_allclose(le.to_numpy(), g8[f"{name}_linear_envelope"], rtol=1e-10, atol=1e-13)
            le_nc = pp.linear_envelope(df, sampling_frequency=fs, zero_center_=False, **kw)
            np.testing.assert_allclose(le_nc.to_numpy(), g8[f"{name}_linear_envelope_nocenter"], rtol=1e-12, atol=1e-300)
    df2 = df.copy()
    same = pp.digital_filter(df2, sampling_frequency=fs, inplace=True, **KW["hp2_fwd"])
    assert same is df2 and np.allclose(df2.to_numpy(), g8["hp2_fwd_filtered"], rtol=1e-12)
    with pytest.raises(ValueError, match="filter type not understood"):
        pp.digital_filter(df, 5, 100, 2, filter_type="bessel")
    # the whole filter-based chain on a batch, feeding the solver without a copy
    import muscle_synergies_amd as ms
    from oracle import emg_envelope_oracle as eo

    raw = np.stack([raw_emg(500 + b, 4000, 6) for b in range(5)])
    env = pp.linear_envelope_batched(raw, 6, 2000, 4, reduce_to=300)
    assert tuple(env.shape) == (5, 300, 6) and env.stride(1) == 1
    sos = pp.design_sos("butter", 4, 2000, 6)
    for b in (0, 4):
        ref = eo.normalize(eo.time_normalize(so.linear_envelope(raw[b], sos), 300))
        np.testing.assert_allclose(env[b].cpu().numpy(), ref, rtol=1e-9, atol=1e-12)
    X = env.clamp_min(0)  # a low-pass filter can undershoot slightly below zero
    W0, H0 = ms.random_init_batched(X, 3, seed=0)
    res = ms.fit_batched(X, W0, H0, max_iter=50, tol=0.0)
    assert float(res.vaf[:, 0].min()) > 0.8


@pytest.mark.gpu
def test_gpu_filter_fuzz():
    """tests/fuzz_sosfilt_gpu.py with a fixed seed: random designs (1-8 sections), series counts, lengths, paddings and
    options against the oracle; fp64 bit-identical."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_sosfilt_gpu.py"), "--cases", "120", "--seed", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 problems" in r.stdout


# ---- the time-parallel mode (csrc/sosfilt_scan.hpp, hipnmf_sosfilt_params.mode = HIPNMF_SOSFILT_SCAN) --------------------------
SCAN_TOL = 1e-10  # relative to the output's largest magnitude; measured 1e-12 .. 2e-11 for the reference's own designs


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_scan_mode_matches_the_reference_outputs(g8, name):
    """The fixtures captured from the reference's digital_filter / linear_envelope (G8), through the chunked filter: same result
    to rounding times the filter's conditioning (not bit for bit: fused multiply-adds, another order of the sums)."""
    from muscle_synergies_amd.preprocess import sosfilt_batched

    raw, fs, zero_lag, sos, zi = _case(g8, name)
    ref = so.digital_filter(raw, sos, zero_lag)
    scale = np.abs(ref).max()
    for arr in (np.ascontiguousarray(raw), np.asfortranarray(raw)):
        got = sosfilt_batched(arr, sos, zero_lag=zero_lag, mode="scan")[0].cpu().numpy()
        assert got.shape == raw.shape and got.dtype == np.float64
        assert np.abs(got - ref).max() <= SCAN_TOL * scale, np.abs(got - ref).max() / scale
    if KW[name]["band_type"] == "lowpass":
        le = sosfilt_batched(raw, sos, zero_lag=zero_lag, zero_center=True, rectify=True, mode="scan")[0].cpu().numpy()
        ref_le = so.linear_envelope(raw, sos, zero_lag)
        assert np.abs(le - ref_le).max() <= SCAN_TOL * np.abs(ref_le).max()


@pytest.mark.gpu
def test_gpu_scan_mode_sizes_dtypes_and_fallbacks(g8):
    """Every compiled chunk length (16 / 40 / 80 samples per thread), fp32 and fp64, a batch larger than the chip, the tail of the
    last chunk, and the shapes the mode hands to the sequential kernel (longer than 20 480 extended samples, or n_samples not a
    multiple of the 16-byte vector) -- which then answer bit for bit."""
    import torch

    from muscle_synergies_amd.preprocess import sosfilt_batched

    sos = g8["lp4_sos"]
    for T, B, m in ((3000, 5, 3), (4096, 2, 2), (9000, 3, 2), (10200, 2, 1), (20000, 2, 2), (20400, 1, 2), (64, 7, 5), (2000, 600, 2)):
        raw = np.stack([raw_emg(500 + b % 7, T, m) for b in range(B)])
        for zero_lag in (True, False):
            got = sosfilt_batched(raw, sos, zero_lag=zero_lag, zero_center=True, rectify=True, mode="scan").cpu().numpy()
            for b in (0, B - 1):
                ref = so.linear_envelope(raw[b], sos, zero_lag)
                assert np.abs(got[b] - ref).max() <= SCAN_TOL * np.abs(ref).max(), (T, B, m, zero_lag)
        raw32 = raw.astype(np.float32)
        got32 = sosfilt_batched(raw32, sos, zero_lag=True, mode="scan").cpu().numpy()
        ex32 = sosfilt_batched(raw32, sos, zero_lag=True, mode="exact").cpu().numpy()
        assert got32.dtype == np.float32
        np.testing.assert_allclose(got32, ex32, rtol=0, atol=2e-7 * np.abs(ex32).max())  # both round an fp64 result to float
    # longer than one workgroup's 20 224 extended samples: the block scan (round 4), and the exact mode stays bit-identical
    for T in (20481, 30000, 1333, 1001):
        raw = raw_emg(9, T, 2)
        a = sosfilt_batched(raw, sos, mode="scan")[0].cpu().numpy()
        assert np.abs(a - so.sosfiltfilt(sos, raw)).max() <= SCAN_TOL * np.abs(a).max()
    raw = raw_emg(9, 20481, 2)
    assert np.array_equal(sosfilt_batched(raw, sos, mode="exact")[0].cpu().numpy(), so.sosfiltfilt(sos, raw))
    with pytest.raises(KeyError):
        sosfilt_batched(raw, sos, mode="fast")
    # determinism
    x = torch.from_numpy(np.stack([raw_emg(40 + b, 20000, 4) for b in range(8)]).astype(np.float32)).cuda()
    y1, y2 = sosfilt_batched(x, sos, rectify=True, mode="scan"), sosfilt_batched(x, sos, rectify=True, mode="scan")
    assert torch.equal(y1, y2)


@pytest.mark.gpu
def test_gpu_scan_mode_kernel_choice_and_shapes_of_the_second_version(g8):
    """sosfilt_chunk_kernel (round 4, the whole extended series in LDS): lengths that are no multiple of the 16-byte vector,
    unaligned rows, a padding longer than the workgroup, both ends of every instance; what it does not hold goes to
    sosfilt_scan_kernel (float64 beyond 256 x 41 extended samples) or to the sequential kernels; HIPNMF_SOS_CHUNK=0 restores
    the first version."""
    import os
    import subprocess
    import sys

    from conftest import ROOT
    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.preprocess import sosfilt_batched

    sos = g8["lp4_sos"]  # two sections: padlen 15
    h = _lib.get_handle(0)
    for T, dtype, padlen, want in ((1001, np.float64, None, "sosfilt_chunk_kernel<double,2,17,64>"),
                                   (2624 - 30, np.float32, None, "sosfilt_chunk_kernel<float,2,41,64>"),
                                   (2624 - 29, np.float32, None, "sosfilt_chunk_kernel<float,2,17>"),
                                   (100, np.float64, None, "sosfilt_chunk_kernel<double,2,5,64>"),
                                   (290, np.float32, None, "sosfilt_chunk_kernel<float,2,5,64>"),
                                   (291, np.float32, None, "sosfilt_chunk_kernel<float,2,9,64>"),
                                   (4352 - 30, np.float64, None, "sosfilt_chunk_kernel<double,2,17>"),
                                   (4352 - 29, np.float64, None, "sosfilt_chunk_kernel<double,2,25>"),
                                   (10496 - 30, np.float32, None, "sosfilt_chunk_kernel<float,2,41>"),
                                   (10496 - 29, np.float32, None, "sosfilt_chunk_kernel<float,2,49>"),
                                   (10496 - 28, np.float64, None, "sosfilt_scan_kernel<double,2,80>"),
                                   (10496 - 29, np.float64, None, "sosfilt_block_kernel<double,2,41>"),  # odd length: the first version does not take it, two blocks do
                                   (20224 - 30, np.float32, None, "sosfilt_chunk_kernel<float,2,79>"),
                                   (20224 - 28, np.float32, None, "sosfilt_scan_kernel<float,2,80>"),
                                   (20000, np.float64, None, "sosfilt_chunk_kernel<double,2,41,512>"),  # eight waves, the CU's whole LDS
                                   (16000, np.float64, None, "sosfilt_scan_kernel<double,2,80>"),         # below 80 % of its positions: the first version
                                   (20224 - 27, np.float32, None, "sosfilt_block_kernel<float,2,79>"),
                                   (3001, np.float32, 700, "sosfilt_chunk_kernel<float,2,25>"),
                                   (16640 - 30, np.float32, None, "sosfilt_chunk_kernel<float,2,65>"),
                                   (8448 - 29, np.float64, None, "sosfilt_chunk_kernel<double,2,41>"),
                                   (333, np.float64, 300, "sosfilt_chunk_kernel<double,2,17,64>")):
        raw = raw_emg(70 + T % 13, T, 3).astype(dtype)
        x = np.ascontiguousarray(raw.T)[:, :T].T if T % 2 else raw  # channel-major rows of odd length: unaligned
        got = sosfilt_batched(x, sos, zero_lag=True, zero_center=True, rectify=True, padlen=padlen, mode="scan")[0].cpu().numpy()
        name = h.last_kernel()
        assert name == want or (want == "sosfiltx" and name.startswith("sosfilt2_kernel<")), (T, dtype, name)
        import scipy.signal as ss

        v = np.abs(raw - raw.mean(axis=0, dtype=np.float64).astype(dtype))  # centred and rectified in the samples' precision
        ref = ss.sosfiltfilt(sos, v.astype(np.float64), axis=0, **({} if padlen is None else {"padlen": padlen}))
        tol = SCAN_TOL if dtype == np.float64 else 3e-7
        assert np.abs(got - ref).max() <= tol * np.abs(ref).max(), (T, dtype, np.abs(got - ref).max() / np.abs(ref).max())
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
from muscle_synergies_amd import _lib
from muscle_synergies_amd.preprocess import sosfilt_batched
from muscle_synergies_amd.synth import raw_emg
from oracle import sosfilt_oracle as so
import json
sos = np.array(json.loads(%r))
raw = raw_emg(3, 20000, 2)
got = sosfilt_batched(raw, sos, zero_lag=True, zero_center=True, rectify=True, mode='scan')[0].cpu().numpy()
ref = so.linear_envelope(raw, sos, True)
print(_lib.get_handle(0).last_kernel(), float(np.abs(got - ref).max() / np.abs(ref).max()) <= 1e-10)
""" % (ROOT, json.dumps(np.asarray(sos).tolist()))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HIPNMF_SOS_CHUNK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "sosfilt_scan_kernel<double,2,80> True" in r.stdout, r.stdout[-1000:] + r.stderr[-1000:]


@pytest.mark.gpu
def test_gpu_scan_mode_long_series_block_scan(g8):
    """Series longer than one workgroup holds (sosfilt_block_kernel, round 4: state pass, scan over the blocks, full pass, per
    direction): whole recordings of 10^5 .. 10^6 samples.  Block boundaries, a last block of a few samples, both dtypes and
    directions, 1 to 8 sections, centring / rectification, a padding that reaches into the neighbouring block; against scipy."""
    import scipy.signal as ss

    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.preprocess import digital_filter, sosfilt_batched

    h = _lib.get_handle(0)
    designs = {"lp4": g8["lp4_sos"], "bp6": ss.butter(3, [20, 450], btype="bandpass", fs=2000.0, output="sos"),
               "hp16": ss.butter(16, 30.0, btype="highpass", fs=2000.0, output="sos"), "lp1": ss.butter(1, 5.0, fs=2000.0, output="sos")}
    for T, dtype, name, zero_lag, padlen in ((20481, np.float64, "lp4", True, None), (2 * 20224 - 30, np.float32, "lp4", True, None),
                                             (2 * 20224 - 29, np.float32, "lp4", True, None), (2 * 10496 + 3, np.float64, "bp6", False, None),
                                             (100003, np.float64, "hp16", True, None), (250001, np.float32, "lp1", True, None),
                                             (61000, np.float64, "lp4", True, 5000), (31000, np.float32, "bp6", True, 300)):
        sos = designs[name]
        raw = raw_emg(20 + T % 11, T, 3).astype(dtype)
        raw[:, 1] += 0.4
        got = sosfilt_batched(raw, sos, zero_lag=zero_lag, zero_center=True, rectify=(name == "lp4"), padlen=padlen, mode="scan")[0].cpu().numpy()
        assert h.last_kernel().startswith("sosfilt_block_kernel<%s" % ("float" if dtype == np.float32 else "double")), (T, h.last_kernel())
        v = raw - raw.mean(axis=0, dtype=np.float64).astype(dtype)
        if name == "lp4":
            v = np.abs(v)
        kw = {} if padlen is None else {"padlen": padlen}
        ref = ss.sosfiltfilt(sos, v.astype(np.float64), axis=0, **kw) if zero_lag else ss.sosfilt(sos, v.astype(np.float64), axis=0)
        err = np.abs(got - ref).max() / np.abs(ref).max()
        # (the 16th-order high-pass is the worst conditioned of the designs: its conditioning, not the algorithm, sets 1e-8)
        assert err <= (5e-7 if dtype == np.float32 else 1e-8 if name == "hp16" else SCAN_TOL), (T, dtype, name, err)
    # a batch of long recordings, and the reference-facing single-frame function in scan mode
    raw = np.stack([raw_emg(60 + b, 50000, 4) for b in range(5)])
    got = sosfilt_batched(raw, designs["lp4"], zero_lag=True, zero_center=True, rectify=True, mode="scan").cpu().numpy()
    for b in (0, 4):
        ref = so.linear_envelope(raw[b], designs["lp4"], True)
        assert np.abs(got[b] - ref).max() <= SCAN_TOL * np.abs(ref).max()
    df = pd.DataFrame(raw[2], columns=list("abcd"))
    out = digital_filter(df, 6, 2000, 4, mode="scan")
    ref = ss.sosfiltfilt(ss.butter(4, 6, fs=2000, output="sos"), raw[2], axis=0)
    assert list(out.columns) == list(df.columns) and np.abs(out.to_numpy() - ref).max() <= SCAN_TOL * np.abs(ref).max()


@pytest.mark.gpu
def test_gpu_scan_mode_fuzz():
    """tests/fuzz_sosfilt_gpu.py --mode scan: random designs (1-8 sections), lengths up to 20 400, paddings, options, layouts."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_sosfilt_gpu.py"), "--cases", "160", "--seed", "11", "--mode", "scan"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 problems" in r.stdout

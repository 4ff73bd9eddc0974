#!/usr/bin/env python3
"""Randomised cross-check of the IIR filter stage (sosfilt2_kernel<real, LPS>: tiles of 64 samples, one lane per
section, a second wave moving the data, mirrored backward tiles) against the NumPy oracle: 1 to 8 sections of
Butterworth / Chebyshev designs (odd orders: a first-order section), series counts around the per-wave counts, lengths
around the tile size, scipy's default edge padding / none / explicit, forward-only and zero-lag, the linear envelope's
zero-centring and rectification, both dtypes and memory orders.  fp64 results must be BIT-IDENTICAL to the oracle (which
is pinned bit-exactly to scipy by tests/test_filters.py) unless zero_center is on (the mean is summed in another order).
Part of the test infrastructure (it imports oracle/); run by tests/test_filters.py::test_gpu_filter_fuzz or by hand."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import signal
from muscle_synergies_amd.preprocess import sosfilt_batched
from muscle_synergies_amd.synth import raw_emg
from oracle import sosfilt_oracle as so

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=80)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--mode", default="exact", choices=["exact", "scan"],
                help="scan: the time-parallel kernel (sosfilt_scan.hpp); agreement with the oracle to --tol relative to the output's "
                     "largest magnitude instead of bit identity")
ap.add_argument("--tol", type=float, default=1e-10)
ap.add_argument("--report", action="store_true", help="print every case's relative error")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
bad = 0
for case in range(a.cases):
    fs = 2000.0
    kind = rng.choice(["butter", "cheby1", "cheby2"])
    band = rng.choice(["lowpass", "highpass", "bandpass"])
    order = int(rng.integers(1, 9 if band != "bandpass" else 5))
    wn = [float(rng.uniform(5, 200)), float(rng.uniform(300, 900))] if band == "bandpass" else float(rng.uniform(5, 900))
    if kind == "butter":
        sos = signal.butter(order, wn, btype=band, output="sos", fs=fs)
    elif kind == "cheby1":
        sos = signal.cheby1(order, 1.0, wn, btype=band, output="sos", fs=fs)
    else:
        sos = signal.cheby2(order, 30.0, wn, btype=band, output="sos", fs=fs)
    if sos.shape[0] > 8:
        continue
    dtype = np.float64 if rng.random() < 0.65 else np.float32
    T = int(rng.choice([40, 63, 64, 65, 100, 127, 128, 129, 191, 192, 193, 500, 1000, 1333, 2048, 2500, 4097] +
                       ([4000, 4096, 8191, 12000, 20000, 20400, 20481, 33000, 40449, 61000] if a.mode == "scan" else [])))
    B = int(rng.choice([1, 2, 3, 7, 11, 17, 33]))
    m = int(rng.choice([1, 2, 3, 4, 5]))
    zero_lag = bool(rng.random() < 0.7)
    zc, rect = bool(rng.random() < 0.3), bool(rng.random() < 0.4)
    edge = so.default_padlen(sos)
    r = rng.random()
    padlen = None if r < 0.6 else (0 if r < 0.75 else int(rng.integers(1, max(2, min(T - 1, 300)))))
    if zero_lag and T <= (edge if padlen is None else padlen):
        continue
    layout = rng.choice(["C", "F"])
    raw = np.stack([raw_emg(9000 + 31 * case + b, T, m) for b in range(B)]).astype(dtype)
    x = np.ascontiguousarray(raw) if layout == "C" else np.ascontiguousarray(raw.transpose(0, 2, 1)).transpose(0, 2, 1)
    desc = (f"case {case}: {np.dtype(dtype).name} {kind} {band} order={order} sections={sos.shape[0]} B={B} T={T} m={m} "
            f"zero_lag={zero_lag} padlen={padlen} zero_center={zc} rectify={rect} layout={layout}")
    try:
        got = sosfilt_batched(x, sos, zero_lag=zero_lag, zero_center=zc, rectify=rect, padlen=padlen, mode=a.mode).cpu().numpy()
    except Exception as e:  # noqa: BLE001 -- a fuzz driver reports and goes on
        print("ERROR", desc, repr(e))
        bad += 1
        continue
    # all series of the batch as columns of one (T, B m) array; preprocessing in the sample dtype as NumPy would do it
    cols = raw.transpose(1, 0, 2).reshape(T, B * m)
    if zc:
        cols = cols - cols.mean(axis=0, dtype=np.float64).astype(dtype)
    if rect:
        cols = np.abs(cols)
    cols = cols.astype(np.float64)
    ref = so.sosfiltfilt(sos, cols, padlen=padlen) if zero_lag else so.sosfilt(sos, cols)[0]
    ref = ref.reshape(T, B, m).transpose(1, 0, 2)
    scale = float(np.abs(ref).max())
    if not np.isfinite(scale):
        continue
    if a.report:
        print("%.2e" % (np.abs(got - ref).max() / max(scale, 1e-300)), desc)
    if a.mode == "scan" and dtype == np.float64:
        ok = np.abs(got - ref).max() <= a.tol * max(scale, 1e-300)
    elif dtype == np.float64 and not zc:
        ok = np.array_equal(got, ref)
    elif dtype == np.float64:
        ok = np.allclose(got, ref, rtol=1e-9, atol=1e-11 * max(scale, 1e-300))
    else:
        ok = np.allclose(got, ref.astype(np.float32), rtol=2e-5, atol=2e-6 * max(scale, 1e-30))
    if not ok or got.shape != ref.shape:
        print("MISMATCH", desc, f"max|diff|={np.abs(got - ref).max():.3e} scale={scale:.3e}")
        bad += 1
print(f"{a.cases} cases, {bad} problems")
sys.exit(1 if bad else 0)

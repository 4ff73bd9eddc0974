import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, time
from muscle_synergies_amd import _lib
from muscle_synergies_amd.preprocess import sosfilt_batched
from muscle_synergies_amd.synth import raw_emg
from oracle import sosfilt_oracle as so
import scipy.signal as ss
h = _lib.get_handle(0)
sos = ss.butter(4, 6.0, btype="lowpass", fs=2000.0, output="sos")
sos8 = ss.butter(8, [20, 450], btype="bandpass", fs=2000.0, output="sos")
bad = 0
for T in (20481, 30000, 40448, 40449, 100000, 250001):
    for dtype in (np.float64, np.float32):
        for zl in (True, False):
            for S, nm in ((sos, "lp4"), (sos8, "bp8")):
                raw = raw_emg(3 + T % 7, T, 3).astype(dtype)
                got = sosfilt_batched(raw, S, zero_lag=zl, zero_center=True, rectify=(nm == "lp4"), mode="scan")[0].cpu().numpy()
                name = h.last_kernel()
                v = raw - raw.mean(axis=0, dtype=np.float64).astype(dtype)
                if nm == "lp4": v = np.abs(v)
                ref = ss.sosfiltfilt(S, v.astype(np.float64), axis=0) if zl else ss.sosfilt(S, v.astype(np.float64), axis=0)
                err = np.abs(got - ref).max() / np.abs(ref).max()
                tol = 1e-10 if dtype == np.float64 else 5e-7
                flag = "" if err <= tol else "  <-- BAD"
                bad += err > tol
                print(T, np.dtype(dtype).name, "zl" if zl else "causal", nm, name, "%.2e" % err, flag)
print("problems", bad)
raw = raw_emg(1, 200000, 16)
for mode in ("scan", "exact"):
    for _ in range(3):
        t0 = time.perf_counter(); sosfilt_batched(raw, sos, zero_lag=True, zero_center=True, rectify=True, mode=mode); dt = time.perf_counter() - t0
    print(mode, "16 x 200000 f64: %.2f ms wall, kernel %.3f ms" % (1e3 * dt, h.last_kernel_ms()), h.last_kernel())

"""Single-frame latency of the reference-facing preprocessing functions (one DataFrame in, one out) against the reference's own
scipy / NumPy implementation, restated by the oracles, on the node's host."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pandas as pd
import scipy.signal as ss
import muscle_synergies_amd as ms
from muscle_synergies_amd.synth import raw_emg

warnings.simplefilter("ignore")
def timeit(fn, n=15):
    fn(); fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    t.sort()
    return 1e3 * t[len(t) // 2]
for T, m in ((2000, 8), (20000, 16), (200000, 16)):
    raw = raw_emg(5, T, m)
    df = pd.DataFrame(raw, columns=[f"m{j}" for j in range(m)])
    sos = ss.butter(4, 6.0, btype="lowpass", fs=2000.0, output="sos")
    rows = []
    rows.append(("linear_envelope (hip)", timeit(lambda: ms.linear_envelope(df, 6, 2000, 4))))
    rows.append(("linear_envelope (hip, mode='scan')", timeit(lambda: ms.linear_envelope(df, 6, 2000, 4, mode="scan"))))
    rows.append(("linear_envelope (scipy)", timeit(lambda: ss.sosfiltfilt(sos, np.abs(raw - raw.mean(axis=0)), axis=0))))
    rows.append(("rms 100 ms (hip)", timeit(lambda: ms.rms(df, 200))))
    rows.append(("rms 100 ms (numpy)", timeit(lambda: np.stack([np.sqrt(np.convolve(raw[:, j] ** 2, np.ones(200) / 200, "same")) for j in range(m)], axis=1))))
    for name, v in rows:
        print(f"T={T} m={m} {name}: {v:.2f} ms")

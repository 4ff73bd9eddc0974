import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from scipy import signal
from muscle_synergies_amd.preprocess import sosfilt_batched
from muscle_synergies_amd.synth import raw_emg
from oracle import sosfilt_oracle as so
np.set_printoptions(linewidth=220, precision=3)
def run(sos, B, T, m, zero_lag, padlen, tag):
    raw = np.stack([raw_emg(100 + b, T, m) for b in range(B)])
    cols = raw.transpose(1, 0, 2).reshape(T, B * m)
    ref = so.sosfiltfilt(sos, cols, padlen=padlen) if zero_lag else so.sosfilt(sos, cols)[0]
    ref = ref.reshape(T, B, m).transpose(1, 0, 2)
    got = [sosfilt_batched(raw, sos, zero_lag=zero_lag, padlen=padlen, mode="scan").cpu().numpy() for _ in range(3)]
    ex = sosfilt_batched(raw, sos, zero_lag=zero_lag, padlen=padlen, mode="exact").cpu().numpy()
    err = np.abs(got[0] - ref) / np.abs(ref).max()
    print(tag, "exact-mode max err %.1e" % (np.abs(ex - ref).max()), "scan max err %.2e" % err.max(), "deterministic", np.array_equal(got[0], got[1]) and np.array_equal(got[1], got[2]))
    for b in range(B):
        for j in range(m):
            e = err[b, :, j]
            bad = np.nonzero(e > 1e-9)[0]
            print("  series", b, j, "max %.2e" % e.max(), "bad samples", len(bad), "first/last bad", (bad[0], bad[-1]) if len(bad) else None)
sos = signal.butter(3, 100.0, btype="lowpass", output="sos", fs=2000.0)
run(sos, 2, 192, 1, True, 118, "A")
run(sos, 1, 192, 1, True, 118, "A1")
run(sos, 1, 192, 1, False, None, "A-causal")
run(sos, 1, 300, 1, False, None, "B-causal-T300")
run(sos, 1, 1000, 1, False, None, "C-causal-T1000")
sos1 = signal.butter(1, 100.0, btype="lowpass", output="sos", fs=2000.0)
run(sos1, 1, 1000, 1, False, None, "D-1section-T1000")
run(sos1, 3, 1000, 2, False, None, "D-1section-T1000-B3m2")

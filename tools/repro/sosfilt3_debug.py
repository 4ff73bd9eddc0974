import os, sys, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1:
    os.environ["HIPNMF_SOS_V3"] = sys.argv[1]
    from scipy import signal
    from muscle_synergies_amd.preprocess import sosfilt_batched
    rng = np.random.default_rng(0)
    for order, T, m in ((4, 400, 3), (2, 400, 3), (8, 400, 3), (4, 129, 2), (4, 20000, 2)):
        x = rng.standard_normal((1, T, m))
        sos = signal.butter(order, 0.2, output="sos")
        y = sosfilt_batched(x, sos, zero_lag=True)[0].cpu().numpy()
        ref = signal.sosfiltfilt(sos, x[0], axis=0)
        d = np.abs(y - ref)
        bad = np.argwhere(d > 1e-12)
        print(f"v3={sys.argv[1]} order={order} T={T}: max diff {d.max():.3e}; first bad rows {bad[:6, 0].tolist()} n_bad={len(bad)} last bad {bad[-3:, 0].tolist() if len(bad) else []}")
else:
    for v in ("1", "0"):
        print(subprocess.run([sys.executable, __file__, v], capture_output=True, text=True).stdout)

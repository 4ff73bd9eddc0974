import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
for dt in (np.float32, np.float64):
    X = emg_matrix(5, T=10000, m=16, dtype=dt); W0, H0 = random_init(X, 5, 5)
    h = _lib.get_handle(0); h.set_tuning(0, 0, 3)
    for rep in range(3):
        r = ms.fit_batched(X[None], W0[None], H0[None], max_iter=500, tol=0.0)
    print(np.dtype(dt).name, h.last_kernel(), f"{r.kernel_ms*1e3/500:.2f} us/iter")

"""Every host entry point on a narrow, a wide and a general-shape recording, both losses: runs or raises, which kernel (sanity
matrix for the combinations the parity tests do not spell out one by one)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pandas as pd
import torch
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix

warnings.simplefilter("ignore")
h = _lib.get_handle(0)
bad = 0
for m, k in ((16, 5), (64, 8), (100, 20), (200, 12), (300, 24)):
    for loss in ("frobenius", "kullback-leibler"):
        for dtype in (np.float32, np.float64):
            X = np.stack([emg_matrix(7 + b, T=900 + 64 * b, m=m, k_true=6, dtype=dtype)[:900] for b in range(3)])
            for name, fn in (
                ("fit_restarts", lambda: ms.fit_restarts(X, k, 3, max_iter=30, tol=0.0, beta_loss=loss)),
                ("rank_sweep_batched", lambda: ms.rank_sweep_batched(torch.from_numpy(X).cuda(), max(1, k - 1), k, max_iter=30, tol=0.0, beta_loss=loss)),
                ("rank_sweep nndsvda", lambda: ms.rank_sweep_batched(torch.from_numpy(X).cuda(), k, k, max_iter=30, tol=0.0, beta_loss=loss, init="nndsvda")),
                ("find_synergies_batched", lambda: ms.find_synergies_batched([pd.DataFrame(x[: 700 + 50 * i]) for i, x in enumerate(X)], k, max_iter=30, tol=0.0, beta_loss=loss, random_state=0)),
                ("HipNMF.fit_transform", lambda: ms.HipNMF(n_components=k, solver="mu", beta_loss=loss, max_iter=30, tol=0.0, init="nndsvda").fit_transform(X[0])),
                ("HipNMF.transform", lambda: (lambda mdl: (mdl.fit(X[0]), mdl.transform(X[1]))[1])(ms.HipNMF(n_components=k, solver="mu", beta_loss=loss, max_iter=20, tol=0.0, init="random", random_state=0))),
            ):
                try:
                    with warnings.catch_warnings(record=True) as wl:
                        warnings.simplefilter("always")
                        out = fn()
                    fell = [str(w.message)[:60] for w in wl if "scikit-learn" in str(w.message) or "fall" in str(w.message).lower()]
                    print(f"m={m} k={k} {loss[:4]} {np.dtype(dtype).name} {name}: ok  {h.last_kernel()} {'FALLBACK ' + fell[0] if fell else ''}")
                    bad += bool(fell)
                except Exception as e:  # noqa: BLE001
                    print(f"m={m} k={k} {loss[:4]} {np.dtype(dtype).name} {name}: ERROR {type(e).__name__}: {str(e)[:120]}")
                    bad += 1
print("problems", bad)

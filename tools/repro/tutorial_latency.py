"""Latency of the reference's own call shape (config #1: one 8-muscle frame of a few hundred samples, n_components = 4, 200 iterations)
through HipNMF against scikit-learn on the host, same initialisation."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pandas as pd
import muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_matrix, random_init

warnings.simplefilter("ignore")
for T, m, k in ((200, 8, 4), (1000, 8, 4), (5000, 16, 5)):
    X = emg_matrix(1, T=T, m=m, k_true=min(4, m), dtype=np.float64)
    df = pd.DataFrame(X, columns=[f"m{j}" for j in range(m)])
    W0, H0 = random_init(X, k, 1)
    def hip():
        mdl = ms.HipNMF(n_components=k, init="custom", solver="mu", max_iter=200, tol=1e-4)
        return mdl.fit_transform(X, W=W0.copy(), H=H0.copy())
    def skl():
        from sklearn.decomposition import NMF
        mdl = NMF(n_components=k, init="custom", solver="mu", max_iter=200, tol=1e-4)
        return mdl.fit_transform(X, W=W0.copy(), H=H0.copy())
    def fs():
        return ms.find_synergies(df, k, solver="mu", max_iter=200, random_state=0)
    for name, fn in (("HipNMF.fit_transform", hip), ("sklearn NMF.fit_transform", skl), ("find_synergies(solver='mu')", fs)):
        fn(); fn()
        t = []
        for _ in range(20):
            t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
        t.sort()
        print(f"T={T} m={m} k={k} {name}: median {1e3 * t[len(t) // 2]:.2f} ms, min {1e3 * t[0]:.2f} ms")

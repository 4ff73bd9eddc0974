#!/usr/bin/env python3
"""Latency of one small find_synergies-style fit (tutorial scale: 200 x 8, float64, tol=1e-6) on the GPU vs sklearn."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_matrix
warnings.simplefilter("ignore")
X = emg_matrix(5, T=200, m=8, k_true=3, dtype=np.float64)
df = pd.DataFrame(X, columns=[f"m{j}" for j in range(8)])
for rep in range(3):
    t0 = time.perf_counter(); r = ms.find_synergies(df, 2, 3, solver="mu", max_iter=50_000, random_state=0); dt = time.perf_counter() - t0
print(f"GPU  find_synergies(200x8 f64, k=2..3, tol=1e-6): {dt*1e3:.2f} ms, n_iter={[m.n_iter_ for m in r.model.values()]}, VAF={r.vaf_values['All signals'].tolist()}")
from sklearn.decomposition import NMF
def sk(df):
    out = {}
    for k in (2, 3):
        m = NMF(k, solver="mu", max_iter=50_000, tol=1e-6, random_state=0); m.fit_transform(df); out[k] = m
    return out
for rep in range(2):
    t0 = time.perf_counter(); s = sk(df); dt = time.perf_counter() - t0
print(f"CPU  sklearn mu same call:                       {dt*1e3:.2f} ms, n_iter={[m.n_iter_ for m in s.values()]}")

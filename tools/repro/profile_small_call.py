import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, pandas as pd
import muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_matrix
X = emg_matrix(3, T=200, m=8, dtype=np.float64)
df = pd.DataFrame(X, columns=[f"m{j}" for j in range(8)])
for _ in range(3):
    r = ms.find_synergies(df, 2, 3, solver="mu", max_iter=50_000, tol=1e-6)
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    r = ms.find_synergies(df, 2, 3, solver="mu", max_iter=50_000, tol=1e-6)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

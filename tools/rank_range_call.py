#!/usr/bin/env python3
"""Wall time of ONE find_synergies(df, n, max, solver='mu') call on a reference-sized frame: the ranks fitted in a loop
(HIPNMF_RANK_THREADS=0) vs from concurrent host threads (default).  python tools/rank_range_call.py [--T 200 --m 8 --kmin 2 --kmax 3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_matrix

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=200); ap.add_argument("--m", type=int, default=8)
ap.add_argument("--kmin", type=int, default=2); ap.add_argument("--kmax", type=int, default=3)
ap.add_argument("--dtype", default="float64"); ap.add_argument("--reps", type=int, default=7)
a = ap.parse_args()
X = emg_matrix(21, T=a.T, m=a.m, k_true=min(3, a.m), dtype=np.dtype(a.dtype))
df = pd.DataFrame(X, columns=[f"m{i}" for i in range(a.m)])
kw = dict(solver="mu", max_iter=50_000, tol=1e-6)   # the tutorial's call: default init (nndsvda), stop rule live
for mode in ("0", "1", "0", "1"):
    os.environ["HIPNMF_RANK_THREADS"] = mode
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter(); r = ms.find_synergies(df, a.kmin, a.kmax, **kw); ts.append(time.perf_counter() - t0)
    its = [r.model[k].n_iter_ for k in r.model]
    print(f"{a.T} x {a.m} {a.dtype} k={a.kmin}..{a.kmax} ranks {'concurrent' if mode == '1' else 'in a loop   '}: best {min(ts)*1e3:7.2f} ms, median {sorted(ts)[len(ts)//2]*1e3:7.2f} ms  (iterations {its})", flush=True)

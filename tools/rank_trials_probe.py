#!/usr/bin/env python3
"""Histogram of the selected rank for the config-#4 workload (muscle_synergies_amd.synth.emg_rank_trials_torch) and a
compute-all vs stop-at-threshold comparison of the native sweep (development aid)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import muscle_synergies_amd as ms
from muscle_synergies_amd.engine import rank_sweep_native
from muscle_synergies_amd.synth import emg_rank_trials_torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--T", type=int, default=10000)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--noise", type=float, nargs="*", default=[0.12])
ap.add_argument("--kmax", type=int, default=8)
a = ap.parse_args()
for noise in a.noise:
    X, kt = emg_rank_trials_torch(a.batch, T=a.T, noise=noise, device="cuda:0")
    Xv = X.transpose(1, 2).contiguous()
    out = {}
    for stop in (False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = rank_sweep_native(Xv, 2, a.kmax, vaf_threshold=0.90, max_iter=a.iters, tol=0.0, seed=1, stop_at_threshold=stop)
        torch.cuda.synchronize()
        out[stop] = (r, time.perf_counter() - t0)
    ra, rs = out[False][0], out[True][0]
    hist = torch.bincount(ra.selected.clamp(min=0), minlength=a.kmax + 1).tolist()
    same = bool(torch.equal(ra.selected, rs.selected))
    fits_all = a.batch * (a.kmax - 1)
    fits_stop = int((rs.n_iter[2] > 0).sum() if False else sum(int((rs.n_iter[k] > 0).sum()) for k in rs.ranks))
    print(f"noise={noise}: selected histogram (index = rank, 0 = none) {hist}; k_true histogram {torch.bincount(kt, minlength=a.kmax + 1).tolist()}; "
          f"stop == all: {same}; fits {fits_stop}/{fits_all}; wall all {out[False][1]*1e3:.0f} ms, stop {out[True][1]*1e3:.0f} ms; "
          f"VAF at k=2: min {float(ra.vaf_all[:,0].min()):.3f} max {float(ra.vaf_all[:,0].max()):.3f}", flush=True)

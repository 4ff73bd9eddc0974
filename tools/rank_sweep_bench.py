#!/usr/bin/env python3
"""BASELINE.json config #4 on one GPU: per-trial rank sweep k = 2..8, 500 iterations each, VAF >= 0.90 selection."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_batch_torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--iters", type=int, default=500)
a = ap.parse_args()
X, _, _ = emg_batch_torch(a.batch, device="cuda:0")
Xv = X.transpose(1, 2)
for rep in range(2):
    t0 = time.perf_counter()
    r = ms.rank_sweep_batched(Xv, 2, 8, vaf_threshold=0.90, max_iter=a.iters, tol=0.0, seed=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
units = a.batch * a.iters * 7
print(f"B={a.batch} k=2..8 x {a.iters} it: wall {dt*1e3:.1f} ms, kernels {r.kernel_ms:.1f} ms -> {units/dt/1e6:.2f} M matrix-it/s "
      f"(wall), selected-rank histogram {torch.bincount(r.selected.clamp(min=0), minlength=9).tolist()}, "
      f"mean VAF per rank {[round(float(v), 4) for v in r.vaf_all.mean(dim=0)]}")

#!/usr/bin/env python3
"""Probe: the persistent kernel with every workgroup reading the SAME X (multi-restart fit: X served by L2 / MALL)
against the regular batch (every workgroup streams its own X from HBM).  Same arithmetic, different memory traffic:
the gap is what HBM costs the headline kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_batch_torch

B, T, iters = 2048, 10000, 100
X, W0, H0 = emg_batch_torch(B, T=T, device="cuda:0")
for rep in range(2):
    r = ms.fit_batched(X.transpose(1, 2), W0, H0, max_iter=iters, tol=0.0)
print(f"own X   : B={B} kernel {r.kernel_ms:.2f} ms  {B*iters/r.kernel_ms/1e3:.3f} M matrix-it/s")
for rep in range(2):
    rr = ms.fit_restarts(X[:1].transpose(1, 2), 5, n_restarts=B, max_iter=iters, tol=0.0)
print(f"shared X: R={B} kernel {rr.kernel_ms:.2f} ms  {B*iters/rr.kernel_ms/1e3:.3f} M matrix-it/s")

#!/usr/bin/env python3
"""Host-resident fit rate against the chunk size of the transfer pipeline (engine._fit_batched_pipelined), headline batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import muscle_synergies_amd as ms
from muscle_synergies_amd import engine
from muscle_synergies_amd.synth import emg_batch_torch

B = 4096
X, W0, H0 = emg_batch_torch(B, T=10000, m=16, k=5, device="cuda:0")
Xr = X.transpose(1, 2).contiguous()
for _ in range(2):
    t0 = time.perf_counter(); r = ms.fit_batched(Xr, W0, H0, max_iter=500, tol=0.0); torch.cuda.synchronize(); dev_s = time.perf_counter() - t0
print(f"device-resident {B*500/dev_s/1e6:.2f} M matrix-it/s ({dev_s*1e3:.1f} ms)")
Xh, Wh, Hh = Xr.cpu().numpy(), W0.cpu().numpy(), H0.cpu().numpy()
for lanes in (1,):
    for chunk in (0, 128, 256, 512, 768, 1024, 2048):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); rh = ms.fit_batched(Xh, Wh, Hh, max_iter=500, tol=0.0, host_chunk=chunk); best = min(best, time.perf_counter() - t0)
        print(f"lanes={lanes} chunk={chunk:5d}: {B*500/best/1e6:.2f} M matrix-it/s ({best*1e3:.1f} ms)  {dev_s/best:.3f} of device-resident", flush=True)

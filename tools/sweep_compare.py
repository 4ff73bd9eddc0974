#!/usr/bin/env python3
"""Config #4 (1024 trials, k = 2..8, 500 iterations) through the Python-driven sweep and through hipnmf_rank_sweep_f32."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_batch_torch
X, _, _ = emg_batch_torch(1024, k=5, device="cuda:0")
Xv = X.transpose(1, 2)
for name, fn in (("python", lambda: ms.rank_sweep_batched(Xv, 2, 8, max_iter=500, tol=0.0, seed=1)),
                 ("native", lambda: ms.rank_sweep_native(Xv, 2, 8, max_iter=500, tol=0.0, seed=1))):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(name, rep, f"{dt*1e3:.1f} ms", f"{1024*500*7/dt/1e6:.2f} M matrix-it/s", r.selected[:4].tolist(), flush=True)

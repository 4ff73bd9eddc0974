#!/usr/bin/env python3
"""Where the host-resident pipeline's time goes: waits for uploads, fits, waits for downloads (headline batch, chunk 256)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import muscle_synergies_amd as ms
from muscle_synergies_amd import engine
from muscle_synergies_amd.synth import emg_batch_torch

B = 4096
X, W0, H0 = emg_batch_torch(B, T=10000, m=16, k=5, device="cuda:0")
Xh, Wh, Hh = X.transpose(1, 2).contiguous().cpu().numpy(), W0.cpu().numpy(), H0.cpu().numpy()
del X
orig_fit = engine.fit_batched
log = []
def traced(*a, **k):
    if k.get("_inputs_ready"):
        t0 = time.perf_counter(); r = orig_fit(*a, **k); log.append(("fit", t0, time.perf_counter())); return r
    return orig_fit(*a, **k)
engine.fit_batched = traced
for rep in range(3):
    log.clear()
    engine._pipeline_trace = tr = []
    r = None
    t0 = time.perf_counter(); r = traced(Xh, Wh, Hh, max_iter=500, tol=0.0, host_chunk=256); t1 = time.perf_counter()
    fits = [(b - a) * 1e3 for _, a, b in log]
    gaps = [(log[i + 1][1] - log[i][2]) * 1e3 for i in range(len(log) - 1)]
    print(f"rep {rep}: total {(t1-t0)*1e3:.1f} ms; first fit starts at {(log[0][1]-t0)*1e3:.1f} ms; fits sum {sum(fits):.1f} (min {min(fits):.2f} max {max(fits):.2f}); "
          f"gaps sum {sum(gaps):.1f} (max {max(gaps):.2f}); after last fit {(t1-log[-1][2])*1e3:.1f} ms; kernel_ms sum {r.kernel_ms:.1f}")
    for stage in ("alloc", "upload", "download", "small"):
        ev = sorted((i, (a - t0) * 1e3, (b - a) * 1e3) for s_, i, a, b in tr if s_ == stage)
        print("   ", stage, " ".join(f"{i}:{st:.0f}+{du:.1f}" for i, st, du in ev))
# page-fault cost of fresh result arrays
t0 = time.perf_counter(); w = np.empty((B, 10000, 5), np.float32); w.fill(0); print(f"np.empty + fill 0.82 GB: {(time.perf_counter()-t0)*1e3:.1f} ms")
t0 = time.perf_counter(); w.fill(1); print(f"second fill: {(time.perf_counter()-t0)*1e3:.1f} ms")

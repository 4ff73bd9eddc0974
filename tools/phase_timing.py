#!/usr/bin/env python3
"""Per-phase cycle counts of one iteration of the persistent kernel (needs the HIPNMF_TIMING build variant:
python -m muscle_synergies_amd.build --variant timing --flag=-DHIPNMF_TIMING; run with
HIPNMF_LIBRARY=muscle_synergies_amd/lib/libhip_nmf_timing.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_batch_torch

B = 1024
X, W0, H0 = emg_batch_torch(B, T=10000, device="cuda:0")
for rep in range(2):
    r = ms.fit_batched(X.transpose(1, 2), W0, H0, max_iter=100, tol=0.0)
names = ["row pass", "wave reduce", "barrier 1", "wave-0 epilogue", "barrier 2", "load H regs"]
w0 = r.sse_col[:, :6].double().mean(dim=0).tolist()
print(f"kernel {r.kernel_ms:.2f} ms for {B} x 100 iterations; shader-clock cycles per iteration, wave 0:")
for n, a in zip(names, w0):
    print(f"  {n:16s} {a:9.1f}")
print(f"  total            {sum(w0):9.1f}")
rp = r.xsq_col[:, :8].double().mean(dim=0).tolist()
wt = r.xsq_col[:, 8:16].double().mean(dim=0).tolist()
print("row pass per wave :", " ".join(f"{v:8.0f}" for v in rp))
print("barrier-1 wait    :", " ".join(f"{v:8.0f}" for v in wt))

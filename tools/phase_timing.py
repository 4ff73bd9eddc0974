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
w1 = r.xsq_col[:, :6].double().mean(dim=0).tolist()
print(f"kernel {r.kernel_ms:.2f} ms for {B} x 100 iterations; s_memtime ticks (100 MHz -> 10 ns each) per iteration")
for n, a, b in zip(names, w0, w1):
    print(f"  {n:16s} wave 0: {a:9.1f}   wave 1: {b:9.1f}")
print(f"  total            wave 0: {sum(w0):9.1f}   wave 1: {sum(w1):9.1f}")

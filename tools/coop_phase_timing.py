#!/usr/bin/env python3
"""Per-phase cycle counts of one iteration of fit_coop_kernel (BASELINE config #2: one 16 x 10 000 matrix).  Needs the timing build:
    python -m muscle_synergies_amd.build --variant timing --flag=-DHIPNMF_TIMING
    HIPNMF_LIBRARY=muscle_synergies_amd/lib/libhip_nmf_timing.so python tools/coop_phase_timing.py
(the stamps overwrite the per-column outputs of the fit: a development build, never the product library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_batch_torch

names = ["row pass (W update + sums of this slice)", "wave reduction + workgroup barrier", "publish: combine waves, tagged 8-byte stores, barrier",
         "poll + sum the S slices' records", "wave-0 H update + workgroup barrier", "H, HH^T back into registers"]
for dt in (torch.float32,):
    X, W0, H0 = emg_batch_torch(1, T=10000, device="cuda:0")
    X, W0, H0 = X.to(dt), W0.to(dt), H0.to(dt)
    for rep in range(3):
        r = ms.fit_batched(X.transpose(1, 2).contiguous(), W0, H0, max_iter=500, tol=0.0)
    clk = 0.0
    print(f"{_lib.get_handle(0).last_kernel()}: kernel {r.kernel_ms:.3f} ms for 500 iterations = {r.kernel_ms * 2:.2f} us per iteration")
    for who, arr in (("wave 0 of slice 0", r.sse_col[0]), ("last wave of slice 0", r.xsq_col[0])):
        v = arr[:6].double().tolist()
        tot = sum(v)
        print(f"  {who}: shader-clock cycles per iteration (total {tot:.0f}; {r.kernel_ms * 2e-6 / max(tot, 1) * 1e9:.3f} ns per cycle if the loop were the whole kernel)")
        for n, c in zip(names, v):
            print(f"    {n:58s} {c:8.0f}  {100 * c / tot:5.1f} %")

#!/usr/bin/env python3
"""Phase durations of sosfilt_chunk_kernel (time-parallel IIR filter) from shader-clock stamps.  Needs the timing build:
    python -m muscle_synergies_amd.build --variant sostiming --flag=-DHIPNMF_SOS_TIMING --only hipnmf_sosfilt
    HIPNMF_LIBRARY=muscle_synergies_amd/lib/libhip_nmf_sostiming.so python tools/sos_phase_timing.py [--B 1024 --T 20000 --m 16]
(the stamps overwrite the first nine output samples of every series: a development build, never the product library)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from muscle_synergies_amd import _lib
from muscle_synergies_amd.preprocess import sosfilt_batched

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1024)
ap.add_argument("--T", type=int, default=20000)
ap.add_argument("--m", type=int, default=16)
ap.add_argument("--dtype", default="float32")
ap.add_argument("--causal", action="store_true")
a = ap.parse_args()
sos = np.array([[2.91464945e-05, 5.82929890e-05, 2.91464945e-05, 1.0, -1.86689228, 0.87521455], [1.0, 2.0, 1.0, 1.0, -1.93296719, 0.94170979]])
x = torch.randn((a.B, a.m, a.T), device="cuda", dtype=getattr(torch, a.dtype)).transpose(1, 2)
for B in (a.B, max(1, 256 // a.m)):  # the batch, then one workgroup per CU (each alone on its CU)
    xb = x[:B]
    for rep in range(2):
        y = sosfilt_batched(xb, sos, zero_lag=not a.causal, zero_center=True, rectify=True, mode="scan")
    torch.cuda.synchronize()
    st = y.transpose(1, 2).reshape(B * a.m, a.T)[:, :13].double().cpu().numpy()
    names = ["load (HBM -> regs)", "mean + LDS staging + extension + chunk", "forward zero-state", "forward scan", "forward correction",
             "backward (all three)", "output -> LDS", "LDS -> HBM stores"]
    tot = st[:, :8].sum(axis=1)
    print(f"{_lib.get_handle(0).last_kernel()}  series={B * a.m}  kernel {_lib.get_handle(0).last_kernel_ms():.3f} ms; shader-clock cycles, mean (min .. max) over the workgroups")
    for i, n in enumerate(names):
        print(f"  {n:42s} {st[:, i].mean():9.0f}  ({st[:, i].min():7.0f} .. {st[:, i].max():7.0f})   {100 * st[:, i].mean() / tot.mean():5.1f} %")
    print(f"  {'workgroup lifetime':42s} {tot.mean():9.0f}  ({tot.min():7.0f} .. {tot.max():7.0f})")
    for i, n in enumerate(["  phase 2a: mean (cvt, sum, shuffles, barrier)", "  phase 2b: pre-process + 79 LDS stores + barrier", "  phase 2c: odd extension + barrier",
                           "  phase 2d: chunk reads + cvt (+ the zero-state recurrence the compiler moves up)"]):
        print(f"  {n:50s} {st[:, 9 + i].mean():9.0f}  ({st[:, 9 + i].min():7.0f} .. {st[:, 9 + i].max():7.0f})")

#!/usr/bin/env python3
"""BASELINE config #2: one synthetic 16 x 10 000 EMG matrix, k = 5, fp32, 500 mu iterations -- per solver path
(1 persistent single workgroup, 2 row-sliced launches + hipGraph, 3 cooperative multi-workgroup kernel)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_batch_torch

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, nargs="*", default=[10000])
ap.add_argument("--batch", type=int, nargs="*", default=[1])
ap.add_argument("--iters", type=int, default=500)
ap.add_argument("--variants", type=int, nargs="*", default=[1, 2, 3, 0])
a = ap.parse_args()
h = _lib.get_handle(0)
names = {0: "auto", 1: "persistent", 2: "sliced+graph", 3: "cooperative"}
for T in a.T:
    for B in a.batch:
        X, W0, H0 = emg_batch_torch(B, T=T, device="cuda:0")
        Xv = X.transpose(1, 2)
        for v in a.variants:
            h.set_tuning(0, 0, v)
            try:
                for rep in range(3):
                    r = ms.fit_batched(Xv, W0, H0, max_iter=a.iters, tol=0.0)
            except _lib.HipNmfError as e:
                print(f"T={T} B={B} {names[v]:13s}: {e}")
                continue
            print(f"T={T} B={B} {names[v]:13s}: {r.kernel_ms:8.3f} ms for {a.iters} iterations = {r.kernel_ms*1e3/a.iters:6.2f} us/iter, "
                  f"{B*a.iters/r.kernel_ms/1e3:7.3f} M matrix-it/s  err={float(r.reconstruction_err[0]):.5f}", flush=True)
h.set_tuning(0, 0, 0)

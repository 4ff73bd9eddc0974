#!/usr/bin/env python3
"""Quick throughput probe of the batched fit (development aid, not the contract bench)."""
import argparse, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_batch_torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--T", type=int, default=10000)
ap.add_argument("--m", type=int, default=16)
ap.add_argument("--k", type=int, default=5)
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--threads", type=int, nargs="*", default=[256, 512])
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--max-slices", type=int, nargs="*", default=[0])
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--loss", default="frobenius")
ap.add_argument("--tol", type=float, default=0.0, help="> 0: the stop rule is live (a residual pass every 10 iterations)")
ap.add_argument("--dtype", default="float32")
ap.add_argument("--rowmajor", action="store_true", help="X as [B, T, m] C-contiguous (row-major) instead of channel-major")
a = ap.parse_args()
X, W0, H0 = emg_batch_torch(a.batch, T=a.T, m=a.m, k=a.k, k_true=min(5, a.m), device="cuda:0")
if a.dtype == "float64":
    X, W0, H0 = X.double(), W0.double(), H0.double()
Xv = X.transpose(1, 2)  # [B, T, m] view of channel-major storage
if a.rowmajor:
    Xv = Xv.contiguous()
h = _lib.get_handle(0)
for nt, msl in [(t, s) for t in a.threads for s in a.max_slices]:
    h.set_tuning(nt, msl, a.variant)
    for rep in range(a.reps):
        t0 = time.perf_counter()
        r = ms.fit_batched(Xv, W0, H0, max_iter=a.iters, tol=a.tol, beta_loss=a.loss)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        its = a.batch * a.iters / (r.kernel_ms * 1e-3)
        gbs = its * X.element_size() * a.T * (a.m + 2 * a.k) / 1e9
        print(f"threads={nt} max_slices={msl} rep={rep} wall={dt*1e3:.1f} ms kernel={r.kernel_ms:.1f} ms  {its/1e6:.3f} M matrix-it/s  {gbs:.0f} GB/s algorithmic  err0={float(r.reconstruction_err[0]):.4f}  {h.last_kernel()}", flush=True)

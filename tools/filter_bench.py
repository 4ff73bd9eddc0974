#!/usr/bin/env python3
"""Throughput probe of the batched IIR filter stage (row f-1): samples/s and algorithmic GB/s
(read the raw samples once + write the filtered samples once)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muscle_synergies_amd import _lib
from muscle_synergies_amd.preprocess import design_sos, sosfilt_batched

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--T", type=int, default=20000)
ap.add_argument("--m", type=int, default=16)
ap.add_argument("--orders", type=int, nargs="*", default=[2, 4, 8])
ap.add_argument("--dtypes", nargs="*", default=["float32", "float64"])
ap.add_argument("--modes", nargs="*", default=["exact", "scan"])
ap.add_argument("--zero-lag", nargs="*", type=int, default=[1, 0], help="1: sosfiltfilt, 0: sosfilt")
a = ap.parse_args()
h = _lib.get_handle(0)
for dtn in a.dtypes:
    dt = getattr(torch, dtn)
    raw = torch.randn((a.batch, a.m, a.T), device="cuda:0", dtype=dt).transpose(1, 2)
    for order in a.orders:
        sos = design_sos("butter", order, 2000, 6)
        for zero_lag in [bool(z) for z in a.zero_lag]:
            for mode in a.modes:
                for rep in range(3):
                    out = sosfilt_batched(raw, sos, zero_lag=zero_lag, zero_center=True, rectify=True, mode=mode)
                ms = h.last_kernel_ms()
                alg = 2 * raw.element_size() * a.batch * a.m * a.T
                print(f"B={a.batch} T={a.T} m={a.m} {dtn} order={order} ({len(sos)} sections) zero_lag={zero_lag} mode={mode}: {ms:.3f} ms, "
                      f"{a.batch*a.m*a.T/ms/1e6:.1f} G samples/s, {alg/ms/1e6:.0f} GB/s algorithmic", flush=True)
                del out

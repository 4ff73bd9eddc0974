#!/usr/bin/env python3
"""Throughput probe of the EMG envelope preprocessing (row f-1): samples/s and algorithmic GB/s."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muscle_synergies_amd import _lib
from muscle_synergies_amd.preprocess import emg_envelope_batched

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--T", type=int, default=20000)
ap.add_argument("--m", type=int, default=16)
ap.add_argument("--window", type=int, default=200)
ap.add_argument("--dtype", default="float32")
ap.add_argument("--no-normalize", action="store_true", help="skip the division by the channel maximum (second sweep over the output)")
ap.add_argument("--no-center", action="store_true", help="skip zero_center (the mean pass over the raw samples)")
ap.add_argument("--reps", type=int, default=15)
ap.add_argument("--reduce-to", nargs="*", default=["none", "200"], help="output lengths to time: 'none' (full length) or a number of points")
a = ap.parse_args()
dt = getattr(torch, a.dtype)
raw = torch.randn((a.batch, a.m, a.T), device="cuda:0", dtype=dt).transpose(1, 2)
h = _lib.get_handle(0)
for reduce_to in [None if r == "none" else int(r) for r in a.reduce_to]:
    times = []
    for rep in range(a.reps + 2):
        out = emg_envelope_batched(raw, a.window, reduce_to=reduce_to, normalize=not a.no_normalize,
                                   zero_center=not a.no_center)
        if rep >= 2:
            times.append(h.last_kernel_ms())
    times.sort()
    ms = times[len(times) // 2]  # median; the minimum is printed too
    n_out = reduce_to or a.T
    esz = raw.element_size()
    alg = esz * a.batch * a.m * (a.T + n_out)
    print(f"B={a.batch} T={a.T} m={a.m} W={a.window} {a.dtype} reduce_to={reduce_to} normalize={not a.no_normalize} "
          f"zero_center={not a.no_center}: {ms:.3f} ms (min {times[0]:.3f}), "
          f"{a.batch*a.m*a.T/ms/1e6:.1f} G samples/s, {alg/ms/1e6:.0f} GB/s algorithmic (read raw once + write out)")

#!/usr/bin/env python3
"""Randomised cross-check of the envelope kernels (emg_wg_kernel / emg_wave_kernel / emg_fused_kernel: the library picks
by shape) against the NumPy oracle: series lengths around the tile and ring sizes, windows of every parity and length,
time normalisation up and down, both memory orders and dtypes, all four option combinations.  Part of the test
infrastructure (it imports oracle/); run by tests/test_envelope.py::test_gpu_envelope_fuzz or by hand."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from muscle_synergies_amd.preprocess import emg_envelope_batched
from muscle_synergies_amd.synth import raw_emg
from oracle import emg_envelope_oracle as eo

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=120)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
T_CHOICES = [1, 2, 3, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1000, 1023, 1025, 2047, 2049, 4095, 4096, 4097, 5000,
             8191, 8192, 8193, 12288, 16384, 20000, 20479, 20480, 20481, 24000, 65535, 65536, 65537, 70000, 131073]
bad = 0
for case in range(a.cases):
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    T = int(rng.choice(T_CHOICES))
    m = int(rng.choice([1, 2, 3, 5]))
    B = int(rng.choice([1, 2, 3]))
    wmax = min(T, 3900)
    W = int(rng.choice([0, 1, 2, 3, 7, 8, 63, 64, 65, 100, 199, 200, 201, 255, 256, 257, 500, 511, 512, 513, 1000, 2000, 3583, 3584, 3585, 3900]))
    W = min(W, wmax)
    r = rng.random()
    if r < 0.45 or T < 2:
        reduce_to = None
    elif r < 0.8:
        reduce_to = int(rng.choice([2, 3, 50, 200, 1000]))
    else:
        reduce_to = int(min(3 * T + 1, 60000))  # up-sampling
    norm, zc = bool(rng.random() < 0.6), bool(rng.random() < 0.7)
    layout = rng.choice(["C", "F"])
    raw = np.stack([raw_emg(5000 + 13 * case + b, T, m) for b in range(B)]).astype(dtype)
    raw[-1] += 0.25
    x = np.ascontiguousarray(raw) if layout == "C" else np.ascontiguousarray(raw.transpose(0, 2, 1)).transpose(0, 2, 1)
    desc = f"case {case}: {np.dtype(dtype).name} B={B} T={T} m={m} W={W} reduce_to={reduce_to} normalize={norm} zero_center={zc} layout={layout}"
    try:
        out = emg_envelope_batched(x, W, reduce_to=reduce_to, normalize=norm, zero_center=zc).cpu().numpy()
    except Exception as e:  # noqa: BLE001 -- a fuzz driver reports and goes on
        print("ERROR", desc, repr(e))
        bad += 1
        continue
    for b in range(B):
        ref = eo.envelope(raw[b].astype(np.float64), W, reduce_to, do_zero_center=zc, do_normalize=norm)
        with np.errstate(invalid="ignore"):
            scale = np.nanmax(np.abs(ref)) if ref.size else 0.0
        if not np.isfinite(scale):
            continue
        # The kernels take a window sum as a difference of fp64 prefix sums over at most 4096 + W samples: an absolute
        # error of ~eps * 4096 sigma^2 in the sum, which the root amplifies where the true mean square is far below
        # sigma^2 -- only windows of a few samples get there, hence the 1/W term
        if dtype == np.float32:
            ok = np.allclose(out[b], ref, rtol=3e-5, atol=3e-6 * scale + 1e-30, equal_nan=True)
        else:
            # W <= 8: single samples next to a zero crossing are ~1e-6 of the signal's power; sqrt(eps * 4096) ~ 1e-6
            atol = 1e-6 if 0 < W <= 8 else 1e-12 + 2e-9 / max(W, 1)
            ok = np.allclose(out[b], ref, rtol=1e-9, atol=atol * max(scale, 1e-300), equal_nan=True)
        if not ok or out[b].shape != ref.shape:
            err = np.nanmax(np.abs(out[b] - ref)) if out[b].shape == ref.shape else float("nan")
            print("MISMATCH", desc, f"b={b} max|diff|={err:.3e} scale={scale:.3e} shapes {out[b].shape} {ref.shape}")
            bad += 1
            break
print(f"{a.cases} cases, {bad} problems")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Back-to-back cooperative fits under a bandwidth hog: every result must equal the first bit for bit.

    python tools/coop_soak.py --fits 10000 --dtype float32 [--device-scope] [--gen-base 0xFFFFFF00] [--matrices 1]

The cooperative kernel (fit_coop_kernel, nmf_kernels.hpp) exchanges one record per workgroup and iteration through
global memory as {value bits, generation} granules without fences; this driver is the soak the protocol is
qualified with (float: same-XCD and device-scope flavours; double: two granules per value).  Environment of the
flavour / the generation offset is set before the library is loaded.  Exit code 0 = every fit bitwise equal to the
first and equal to the row-sliced path to rounding; prints one summary line.
"""
import argparse
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--fits", type=int, default=1000)
ap.add_argument("--dtype", default="float32")
ap.add_argument("--device-scope", action="store_true", help="HIPNMF_COOP_XCD=0: the device-scope exchange for one matrix too")
ap.add_argument("--gen-base", default=None, help="HIPNMF_COOP_GEN_BASE: start of the generation numbers (test hook)")
ap.add_argument("--matrices", type=int, default=1)
ap.add_argument("--T", type=int, default=10_000)
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--hog-every", type=int, default=4)
a = ap.parse_args()
if a.device_scope:
    os.environ["HIPNMF_COOP_XCD"] = "0"
if a.gen_base is not None:
    os.environ["HIPNMF_COOP_GEN_BASE"] = str(a.gen_base)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import muscle_synergies_amd as ms  # noqa: E402
from muscle_synergies_amd import _lib  # noqa: E402
from muscle_synergies_amd.synth import emg_matrix, random_init  # noqa: E402

dt = np.float32 if a.dtype == "float32" else np.float64
Xs, Ws, Hs = [], [], []
for b in range(a.matrices):
    X = np.ascontiguousarray(emg_matrix(77 + b, T=a.T, dtype=dt))
    W0, H0 = random_init(X, 5, 77 + b)
    Xs.append(X), Ws.append(W0), Hs.append(H0)
Xd = torch.from_numpy(np.stack(Xs)).cuda()
Wd, Hd = torch.from_numpy(np.stack(Ws)).cuda(), torch.from_numpy(np.stack(Hs)).cuda()
h = _lib.Handle(0)
h.set_tuning(0, 0, 3)
first = ms.fit_batched(Xd, Wd, Hd, max_iter=a.iters, tol=0.0, handle=h)
kernel = h.last_kernel()
assert kernel.startswith("fit_coop_kernel"), kernel
hog_stream = torch.cuda.Stream()
big = torch.empty(512 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda")
other = torch.empty_like(big)
bad = 0
t0 = time.perf_counter()
for rep in range(a.fits):
    if a.hog_every and rep % a.hog_every == 0:
        with torch.cuda.stream(hog_stream):  # ~1 GB of traffic per copy, queued ahead of the fits
            other.copy_(big)
            big.copy_(other)
    r = ms.fit_batched(Xd, Wd, Hd, max_iter=a.iters, tol=0.0, handle=h)
    if not (torch.equal(r.W, first.W) and torch.equal(r.H, first.H) and torch.equal(r.reconstruction_err, first.reconstruction_err)):
        bad += 1
        print(f"MISMATCH at fit {rep}", flush=True)
        if bad > 5:
            break
torch.cuda.synchronize()
wall = time.perf_counter() - t0
assert h.last_kernel() == kernel, (h.last_kernel(), kernel)  # no silent change of path in between
h.set_tuning(0, 0, 2)
sliced = ms.fit_batched(Xd, Wd, Hd, max_iter=a.iters, tol=0.0, handle=h)
wh_c = (first.W.double() @ first.H.double()).cpu().numpy()
wh_s = (sliced.W.double() @ sliced.H.double()).cpu().numpy()
rel = max(float(np.linalg.norm(wh_c[b] - wh_s[b]) / np.linalg.norm(Xs[b])) for b in range(a.matrices))
tol = 1e-5 if dt == np.float32 else 1e-10
print(f"coop_soak: kernel={kernel} dtype={a.dtype} matrices={a.matrices} T={a.T} iters={a.iters} gen_base={a.gen_base} "
      f"fits={rep + 1} mismatches={bad} rel|WH - WH_sliced|={rel:.2e} wall={wall:.1f}s "
      f"library={os.path.basename(os.environ.get('HIPNMF_LIBRARY', 'libhip_nmf.so'))}", flush=True)
sys.exit(0 if bad == 0 and rel <= tol else 1)

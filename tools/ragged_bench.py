#!/usr/bin/env python3
"""Timing probe of the ragged / multi-trial entry points on reference-shaped input (development aid).

The reference's real multi-trial loop is ``find_synergies`` over the gait cycles ``project/segment.py:160-207`` cuts
(``analysis.py:907-912``): float64 frames of 8-16 muscles, a few hundred to a few thousand rows each, k = 2..8.

    python3 tools/ragged_bench.py --entry fit_ragged --dtype float64 --m 12 --k 4 --trials 40 --tmin 200 --tmax 600
    python3 tools/ragged_bench.py --entry rank_sweep --m 16 --kmin 2 --kmax 8 --trials 60 --tmin 1500 --tmax 1500
    python3 tools/ragged_bench.py --entry find_synergies_batched ...

Prints one line: entry, shape, best-of-reps wall ms, device ms, kernel(s)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init

ap = argparse.ArgumentParser()
ap.add_argument("--entry", default="fit_ragged", choices=["fit_ragged", "rank_sweep", "rank_sweep_native", "find_synergies_batched"])
ap.add_argument("--dtype", default="float64")
ap.add_argument("--m", type=int, default=12)
ap.add_argument("--k", type=int, default=4)
ap.add_argument("--kmin", type=int, default=2)
ap.add_argument("--kmax", type=int, default=6)
ap.add_argument("--trials", type=int, default=40)
ap.add_argument("--tmin", type=int, default=200)
ap.add_argument("--tmax", type=int, default=600)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--loss", default="frobenius")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--suite", action="store_true", help="the audit's whole list of cases in this one process (tools/routing_ragged_audit.sh)")
args = ap.parse_args()


def run(a):
    dt = np.dtype(a.dtype)
    rng = np.random.default_rng(1)
    Ts = [int(t) for t in rng.integers(a.tmin, a.tmax + 1, size=a.trials)]
    Xs = [np.ascontiguousarray(emg_matrix(100 + b, T=Ts[b], m=a.m, k_true=min(4, a.m), dtype=dt)) for b in range(a.trials)]
    h = _lib.get_handle(0)
    kernels = set()


    def once():
        if a.entry == "fit_ragged":
            ini = [random_init(Xs[b], a.k, b) for b in range(a.trials)]
            Xd = [torch.from_numpy(x).cuda() for x in Xs]
            Wd = [torch.from_numpy(i[0]).cuda() for i in ini]
            Hd = [torch.from_numpy(i[1]).cuda() for i in ini]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = ms.fit_ragged(Xd, Wd, Hd, max_iter=a.iters, tol=0.0, beta_loss=a.loss)
            torch.cuda.synchronize()
            dt_w = time.perf_counter() - t0
            kernels.add(h.last_kernel())
            return dt_w, r.kernel_ms
        if a.entry in ("rank_sweep", "rank_sweep_native"):
            if a.tmin != a.tmax:
                raise SystemExit("rank sweeps take trials of one length (--tmin == --tmax)")
            Xd = torch.from_numpy(np.stack(Xs)).cuda()
            torch.cuda.synchronize()
            fn = ms.rank_sweep_batched if a.entry == "rank_sweep" else ms.rank_sweep_native
            t0 = time.perf_counter()
            r = fn(Xd, a.kmin, a.kmax, vaf_threshold=0.9, max_iter=a.iters, tol=0.0, seed=3)
            torch.cuda.synchronize()
            dt_w = time.perf_counter() - t0
            kernels.add(h.last_kernel())
            return dt_w, r.kernel_ms
        import pandas

        cols = [f"m{j}" for j in range(a.m)]
        dfs = [pandas.DataFrame(x, columns=cols) for x in Xs]
        t0 = time.perf_counter()
        ms.find_synergies_batched(dfs, a.kmin, a.kmax, max_iter=a.iters, tol=0.0, init="random", random_state=0, beta_loss=a.loss)
        torch.cuda.synchronize()
        dt_w = time.perf_counter() - t0
        kernels.add(h.last_kernel())
        return dt_w, float("nan")


    best_w, best_k = float("inf"), float("inf")
    for _ in range(a.reps):
        w, k_ms = once()
        best_w, best_k = min(best_w, w), min(best_k, k_ms)
    rows = sum(Ts)
    print(f"{a.entry} {a.dtype} {a.loss} m={a.m} k={a.k if a.entry == 'fit_ragged' else f'{a.kmin}..{a.kmax}'} trials={a.trials} rows={a.tmin}..{a.tmax} "
          f"iters={a.iters}: wall {best_w * 1e3:.2f} ms, device {best_k:.2f} ms, {rows * a.iters / (best_w * 1e3) / 1e3:.1f} M row-it/s  {sorted(kernels)}", flush=True)


if not args.suite:
    run(args)
else:
    import copy

    def case(**kw):
        a = copy.copy(args)
        for k_, v_ in kw.items():
            setattr(a, k_, v_)
        run(a)

    for dt_ in ("float64", "float32"):
        for m_ in (8, 12, 16):
            for n_, lo_, hi_ in ((6, 200, 600), (40, 200, 600), (300, 200, 600), (40, 800, 1500), (40, 2000, 5000)):
                case(entry="fit_ragged", dtype=dt_, m=m_, k=4, trials=n_, tmin=lo_, tmax=hi_)
        case(entry="fit_ragged", dtype=dt_, m=16, k=2, trials=40, tmin=200, tmax=600)
        case(entry="fit_ragged", dtype=dt_, m=16, k=8, trials=40, tmin=200, tmax=600)
        case(entry="fit_ragged", dtype=dt_, m=16, k=8, trials=40, tmin=2000, tmax=5000)
        case(entry="fit_ragged", dtype=dt_, m=16, k=8, trials=300, tmin=800, tmax=1500)
        case(entry="fit_ragged", dtype=dt_, m=12, k=4, trials=40, tmin=200, tmax=5000)
        case(entry="fit_ragged", dtype=dt_, m=12, k=4, trials=40, tmin=200, tmax=600, loss="kullback-leibler")
        case(entry="rank_sweep", dtype=dt_, m=16, kmin=2, kmax=8, trials=60, tmin=1500, tmax=1500)
        case(entry="rank_sweep_native", dtype=dt_, m=16, kmin=2, kmax=8, trials=60, tmin=1500, tmax=1500)
        case(entry="rank_sweep_native", dtype=dt_, m=8, kmin=2, kmax=6, trials=300, tmin=400, tmax=400)
        case(entry="rank_sweep_native", dtype=dt_, m=12, kmin=2, kmax=8, trials=12, tmin=3000, tmax=3000)
        case(entry="find_synergies_batched", dtype=dt_, m=12, kmin=2, kmax=6, trials=40, tmin=200, tmax=600)
        case(entry="find_synergies_batched", dtype=dt_, m=16, kmin=2, kmax=8, trials=12, tmin=2000, tmax=5000)

#!/usr/bin/env python3
"""Throughput probe of the time-sharded building blocks on one GPU (config #5, one rank's shard)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muscle_synergies_amd.tsharded import HipShardOps, fit_tsharded

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=25_000_000)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
m, k = 16, 5
X = torch.rand((1, a.T, m), generator=g, device=dev)          # [B, T, m] row-major (converted once by HipShardOps)
W0 = torch.rand((1, a.T, k), generator=g, device=dev) + 0.1
H0 = torch.rand((1, k, m), generator=g, device=dev) + 0.1
ops = HipShardOps(X, W0, H0)
del X, W0
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    res = fit_tsharded(ops, max_iter=a.iters, tol=0.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gb = a.iters * 4 * a.T * 26 / 1e9
    print(f"T={a.T} iters={a.iters} wall={dt*1e3:.1f} ms -> {dt/a.iters*1e3:.3f} ms/iter, {gb/dt:.0f} GB/s algorithmic, err={float(res.reconstruction_err[0]):.3f}", flush=True)

# pure pass / h-update timings on the stream (no Python loop overhead between events)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn, n in (("shard_pass", ops.shard_pass, 10), ("h_update", lambda: ops.h_update(ops.sums), 50), ("residual", ops.residual, 10)):
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name}: {ms:.3f} ms per call" + (f" = {4*a.T*26/1e9/(ms*1e-3):.0f} GB/s algorithmic" if name == "shard_pass" else ""), flush=True)

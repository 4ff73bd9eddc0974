#!/usr/bin/env python3
"""Sweep the slice geometry of the time-shard pass (one rank's shard of config #5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muscle_synergies_amd import _lib
from muscle_synergies_amd.tsharded import HipShardOps

T = int(sys.argv[1]) if len(sys.argv) > 1 else 25_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.rand((1, T, 16), generator=g, device=dev)
W0 = torch.rand((1, T, 5), generator=g, device=dev) + 0.1
H0 = torch.rand((1, 5, 16), generator=g, device=dev) + 0.1
ops = HipShardOps(X, W0, H0)
del X, W0
h = _lib.get_handle(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for threads in (256, 512):
    for max_slices in (256, 512, 1024, 2048, 4096, 8192):
        h.set_tuning(threads, max_slices, 0)
        ops.shard_pass(); torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops.shard_pass()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"threads={threads} max_slices={max_slices}: {ms:.3f} ms per pass = {4*T*26/1e9/(ms*1e-3):.0f} GB/s algorithmic", flush=True)
h.set_tuning(0, 0, 0)

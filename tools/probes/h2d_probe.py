#!/usr/bin/env python3
"""Host <-> device transfer rates on this box (sizing of engine._fit_batched_pipelined): pageable, pinned, staged, registered."""
import time, threading, os
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch

dev = torch.device("cuda:0")
N = 1 << 30  # 1 GiB
src = np.random.default_rng(0).random(N // 4, dtype=np.float32)
dst = torch.empty(N // 4, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))

def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best

ts = torch.from_numpy(src)
print("pageable H2D 1 thread      %.1f GB/s" % (N / t(lambda: dst.copy_(ts)) / 1e9))
pin = torch.empty(N // 4, dtype=torch.float32, pin_memory=True)
print("pinned H2D                 %.1f GB/s" % (N / t(lambda: dst.copy_(pin, non_blocking=True)) / 1e9))
print("pinned D2H                 %.1f GB/s" % (N / t(lambda: pin.copy_(dst, non_blocking=True)) / 1e9))
pn = pin.numpy()
print("memcpy -> pinned 1 thread  %.1f GB/s" % (N / t(lambda: np.copyto(pn, src)) / 1e9))
for nt in (2, 4, 8, 16):
    pool = ThreadPoolExecutor(nt)
    n = len(src) // nt
    def work(i): np.copyto(pn[i * n:(i + 1) * n], src[i * n:(i + 1) * n])
    print("memcpy -> pinned %2d threads %.1f GB/s" % (nt, N / t(lambda: list(pool.map(work, range(nt)))) / 1e9))
for nt in (2, 4):
    pool = ThreadPoolExecutor(nt)
    n = len(src) // nt
    streams = [torch.cuda.Stream(dev) for _ in range(nt)]
    def work(i):
        with torch.cuda.stream(streams[i]):
            dst[i * n:(i + 1) * n].copy_(ts[i * n:(i + 1) * n])
        streams[i].synchronize()
    print("pageable H2D %d threads     %.1f GB/s" % (nt, N / t(lambda: list(pool.map(work, range(nt)))) / 1e9))
# staged: 4 threads copy 32 MiB pieces into pinned slots and enqueue async H2D
piece = 32 << 20
slots = [torch.empty(piece // 4, dtype=torch.float32, pin_memory=True) for _ in range(8)]
def staged(nt):
    pool = ThreadPoolExecutor(nt)
    npieces = N // piece
    def work(w):
        st = torch.cuda.Stream(dev)
        ev = None
        for j in range(w, npieces, nt):
            slot = slots[w * 2 + (j // nt) % 2] if nt <= 4 else slots[w % 8]
            if ev is not None and (j // nt) % 2 == 0: st.synchronize()
            np.copyto(slot.numpy(), src[j * piece // 4:(j + 1) * piece // 4])
            with torch.cuda.stream(st):
                dst[j * piece // 4:(j + 1) * piece // 4].copy_(slot, non_blocking=True)
            ev = True
            st.synchronize()
        st.synchronize()
    return lambda: list(pool.map(work, range(nt)))
for nt in (2, 4):
    print("staged via pinned, %d threads %.1f GB/s" % (nt, N / t(staged(nt)) / 1e9))
rt = torch.cuda.cudart()
t0 = time.perf_counter(); rc = rt.cudaHostRegister(src.ctypes.data, N, 0); dt = time.perf_counter() - t0
print("hipHostRegister 1 GiB: rc", rc, "%.3f s" % dt)
if int(rc) == 0:
    print("registered H2D             %.1f GB/s" % (N / t(lambda: dst.copy_(ts, non_blocking=True)) / 1e9))
    t0 = time.perf_counter(); rt.cudaHostUnregister(src.ctypes.data); print("unregister %.3f s" % (time.perf_counter() - t0))

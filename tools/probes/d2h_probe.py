#!/usr/bin/env python3
"""D2H into fresh host memory while a chip-filling kernel runs: pageable vs hipHostRegister'ed destination."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_batch_torch

dev = torch.device("cuda:0")
rt = torch.cuda.cudart()
X, W0, H0 = emg_batch_torch(512, T=10000, m=16, k=5, device="cuda:0")
Xr = X.transpose(1, 2).contiguous()
src = torch.rand((256, 10000, 5), device=dev)  # 51 MB
ms.fit_batched(Xr, W0, H0, max_iter=500, tol=0.0)
def busy():
    ms.fit_batched(Xr, W0, H0, max_iter=500, tol=0.0)  # ~25 ms of a persistent kernel on every CU
for label, reg in (("pageable", False), ("registered", True)):
    for during in (False, True):
        dst = np.empty((256, 10000, 5), np.float32)
        t0 = time.perf_counter()
        if reg:
            rc = rt.cudaHostRegister(dst.ctypes.data, dst.nbytes, 0)
        t_reg = time.perf_counter() - t0
        th = threading.Thread(target=busy) if during else None
        if th: th.start(); time.sleep(0.003)
        st = torch.cuda.Stream(dev)
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            torch.from_numpy(dst).copy_(src, non_blocking=True)
        st.synchronize()
        dt = time.perf_counter() - t0
        if th: th.join()
        ok = np.array_equal(dst, src.cpu().numpy())
        print(f"{label:10s} kernel running={during}: register {t_reg*1e3:6.1f} ms, copy {dt*1e3:6.1f} ms = {dst.nbytes/dt/1e9:5.1f} GB/s, pinned={torch.from_numpy(dst).is_pinned()} ok={ok}")
        if reg: rt.cudaHostUnregister(dst.ctypes.data)
big = np.empty((4096, 10000, 5), np.float32)
t0 = time.perf_counter(); rc = rt.cudaHostRegister(big.ctypes.data, big.nbytes, 0); print(f"register fresh 0.82 GB: {(time.perf_counter()-t0)*1e3:.1f} ms rc={rc}")
t0 = time.perf_counter(); rt.cudaHostUnregister(big.ctypes.data); print(f"unregister: {(time.perf_counter()-t0)*1e3:.1f} ms")

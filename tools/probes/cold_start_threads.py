#!/usr/bin/env python3
"""Cold-start probe (round 6): N host threads make their FIRST call into the library at the same moment -- the first launch of a
kernel function loads its code object, and two such first launches racing inside the HIP runtime crashed 3 of 30 fresh processes
(the rank range of find_synergies).  Plain ctypes + hipMalloc, no torch: a fresh child per repetition.

    python3 tools/probes/cold_start_threads.py --reps 40                       # with the library's first-use lock (default)
    HIPNMF_FIRST_LAUNCH_LOCK=0 python3 tools/probes/cold_start_threads.py --reps 40   # the A/B: the lock disabled
"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, ctypes, threading
sys.path.insert(0, %r)
import numpy as np
from muscle_synergies_amd import _lib as L
from muscle_synergies_amd.engine import make_problem
from muscle_synergies_amd.synth import emg_matrix, random_init
hip = ctypes.CDLL("libamdhip64.so"); lib = ctypes.CDLL(L.LIB_PATH); L._declare(lib)
vp = ctypes.c_void_p
hip.hipMalloc.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]; hip.hipMemcpy.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_int]
NT = %d
X = emg_matrix(2, T=2000, m=8, dtype=np.float64)
start = threading.Barrier(NT)
errs = []
def dev(a):
    p = vp(); assert hip.hipMalloc(ctypes.byref(p), a.nbytes) == 0
    assert hip.hipMemcpy(p, a.ctypes.data_as(vp), a.nbytes, 1) == 0
    return p
def work(i):
    try:
        hip.hipSetDevice(0)
        k = 2 + i %% 5  # different ranks: different kernel instances, several code objects
        W0, H0 = random_init(X, k, k)
        dX, dW, dH = dev(np.ascontiguousarray(X)), dev(W0), dev(H0)
        h = vp(); assert lib.hipnmf_create(0, ctypes.byref(h)) == 0
        p = make_problem(1, 2000, 8, k, x_layout=L.X_ROW_MAJOR, ldx=8, x_batch_stride=16000, max_iter=50, tol=0.0)
        start.wait()
        rc = lib.hipnmf_fit_batched_f64(h, ctypes.byref(p), dX, dW, dH, None, None, None, None)
        if rc: errs.append((i, rc, lib.hipnmf_last_error()))
        lib.hipnmf_destroy(h)
    except BaseException as e:
        errs.append((i, repr(e)))
ts = [threading.Thread(target=work, args=(i,)) for i in range(NT)]
[t.start() for t in ts]; [t.join() for t in ts]
print("MARK", errs)
"""
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=40); ap.add_argument("--threads", type=int, default=8)
a = ap.parse_args()
bad = 0
for r in range(a.reps):
    p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, a.threads)], capture_output=True, text=True, timeout=300)
    ok = p.returncode == 0 and "MARK []" in p.stdout
    bad += not ok
    if not ok:
        print(f"rep {r}: rc={p.returncode} {p.stdout.strip()[-200:]} {p.stderr.strip()[-300:]}", flush=True)
print(f"COLD-START lock={os.environ.get('HIPNMF_FIRST_LAUNCH_LOCK', '1')} threads={a.threads} reps={a.reps} failures={bad}")

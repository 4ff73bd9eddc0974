#!/usr/bin/env python3
"""Interpreter-exit cases of the engine, each in a FRESH child process, repeated.

    python3 tools/exit_cases.py --reps 20 [--only pool_idle,daemon_inflight] [--log gpurun_out/r06/exit_cases.log]

What is checked: the child exits with code 0 (no abort / segfault at teardown, whatever state the engine is in when the
interpreter ends), and printed its marker line before.  A failing repetition's stderr tail (faulthandler is on) goes to the log.
Never a re-exec of a process that has touched the GPU: every repetition is a child started from this GPU-free parent.

Cases
  plain_fit          one fit, then fall off the end of the script
  pool_idle          find_synergies over a rank range: the 8 worker threads of analysis._rank_pool idle, each with a cached handle
  threads_gone       8 threads made a fit each and ended; their cached handles are still in _lib._handles at exit
  user_handle_global a Handle and device tensors kept in module globals (destroyed by module teardown, after atexit)
  handle_churn       200 x create / fit / destroy from 4 threads, then exit
  pipeline_exit      host-resident pipelined fit (worker threads, hipHostRegister), then exit at once
  scatter_exit       devices=[0, 0] scatter (one thread + handle per slice), then exit at once
  daemon_inflight    a DAEMON thread is inside fits when the main thread ends: the exit hook must leave its handle alone
  sys_exit_in_thread the main thread calls sys.exit(0) while a non-daemon worker still fits (joined by the interpreter)
  shutdown_twice     _lib.shutdown() called by the host, then again by atexit; a late Handle() raises cleanly
  plain_ctypes_leak  no torch, no Python package: ctypes + hipMalloc, a handle and its workspace leaked at exit
  c_host_late_destroy  a compiled C host (tools/abi_hosts/late_destroy.c): hipnmf_destroy from an exit handler registered BEFORE
                     the library was loaded (it runs after the library's own exit hook: the destroy must not touch the runtime)
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRE = f"""
import sys, os, threading, time
sys.path.insert(0, {ROOT!r})
import numpy as np
"""
PKG = PRE + """
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
def data(seed, T=3000, m=16, k=5, dt=np.float32):
    X = emg_matrix(seed, T=T, m=m, dtype=dt); W0, H0 = random_init(X, k, seed); return X, W0, H0
"""

CASES = {
    "plain_fit": PKG + """
X, W0, H0 = data(1)
r = ms.fit_batched(X, W0, H0, max_iter=50, tol=0.0, device="cuda:0")
print("MARK", float(r.reconstruction_err[0]))
""",
    "pool_idle": PKG + """
import pandas
df = pandas.DataFrame(emg_matrix(2, T=2000, m=8, dtype=np.float64), columns=list("abcdefgh"))
res = ms.find_synergies(df, 2, 6, solver="mu", max_iter=100, init="random", random_state=0)
print("MARK", type(res).__name__, sum(t.name.startswith("hipnmf-rank") for t in threading.enumerate()))
""",
    "threads_gone": PKG + """
out = [None] * 8
def work(i):
    X, W0, H0 = data(10 + i, T=1500 + 100 * i)
    out[i] = float(ms.fit_batched(X, W0, H0, max_iter=40, tol=0.0, device="cuda:0").reconstruction_err[0])
ts = [threading.Thread(target=work, args=(i,)) for i in range(8)]
[t.start() for t in ts]; [t.join() for t in ts]
print("MARK", len(_lib._handles), out[0])
""",
    "user_handle_global": PKG + """
import torch
KEEP_H = _lib.Handle(0)
X, W0, H0 = data(3)
KEEP_T = [torch.as_tensor(X, device="cuda:0"), torch.as_tensor(W0, device="cuda:0")]
r = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0, device="cuda:0", handle=KEEP_H)
KEEP_R = r
print("MARK", float(r.reconstruction_err[0]))
""",
    "handle_churn": PKG + """
def work(i):
    X, W0, H0 = data(20 + i, T=800)
    for _ in range(50):
        h = _lib.Handle(0)
        ms.fit_batched(X, W0, H0, max_iter=5, tol=0.0, device="cuda:0", handle=h)
        h.close()
ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
[t.start() for t in ts]; [t.join() for t in ts]
print("MARK", len(_lib._live))
""",
    "pipeline_exit": PKG + """
B = 600
X = np.stack([emg_matrix(40 + b % 7, T=2000, m=16, dtype=np.float32) for b in range(B)])
ini = [random_init(X[b], 5, b) for b in range(B)]
W0, H0 = np.stack([i[0] for i in ini]), np.stack([i[1] for i in ini])
r = ms.fit_batched(X, W0, H0, max_iter=20, tol=0.0, device="cuda:0")
print("MARK", float(r.reconstruction_err[0]))
""",
    "scatter_exit": PKG + """
B = 64
X = np.stack([emg_matrix(50 + b % 5, T=1500, m=12, dtype=np.float64) for b in range(B)])
ini = [random_init(X[b], 4, b) for b in range(B)]
W0, H0 = np.stack([i[0] for i in ini]), np.stack([i[1] for i in ini])
r = ms.fit_batched(X, W0, H0, max_iter=20, tol=0.0, devices=[0, 0])
print("MARK", float(r.reconstruction_err[0]))
""",
    "daemon_inflight": PKG + """
X, W0, H0 = data(5, T=20000)
started = threading.Event()
def loop():
    while True:
        ms.fit_batched(X, W0, H0, max_iter=200, tol=0.0, device="cuda:0")
        started.set()
t = threading.Thread(target=loop, daemon=True); t.start()
started.wait(); time.sleep(0.01)
print("MARK daemon alive", t.is_alive())
""",
    "sys_exit_in_thread": PKG + """
X, W0, H0 = data(6, T=20000)
def work():
    for _ in range(5):
        ms.fit_batched(X, W0, H0, max_iter=100, tol=0.0, device="cuda:0")
    print("worker done", flush=True)
t = threading.Thread(target=work); t.start()
time.sleep(0.3)
print("MARK", flush=True)
sys.exit(0)
""",
    "shutdown_twice": PKG + """
X, W0, H0 = data(7)
ms.fit_batched(X, W0, H0, max_iter=10, tol=0.0, device="cuda:0")
_lib.shutdown()
try:
    _lib.Handle(0)
    print("late handle was created")
except _lib.HipNmfError as e:
    print("MARK late handle refused:", e.code)
""",
    "plain_ctypes_leak": PRE + """
import ctypes
from muscle_synergies_amd import _lib as L
assert "torch" not in sys.modules
hip = ctypes.CDLL("libamdhip64.so"); lib = ctypes.CDLL(L.LIB_PATH); L._declare(lib)
h = ctypes.c_void_p(); assert lib.hipnmf_create(0, ctypes.byref(h)) == 0
p = ctypes.c_void_p(); hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]; assert hip.hipMalloc(ctypes.byref(p), 1 << 20) == 0
out = ctypes.c_double(); assert lib.hipnmf_diag_stream_gbs(h, 1 << 20, 64, 3, ctypes.byref(out)) == 0
print("MARK", out.value > 0)
""",
    # a compiled C host (tools/abi_hosts/late_destroy.c): hipnmf_destroy from an exit handler that runs AFTER the library's own
    "c_host_late_destroy": PRE + f"""
import subprocess, tempfile
exe = os.path.join(tempfile.mkdtemp(), "late_destroy")
subprocess.run(["gcc", "-O1", "-o", exe, os.path.join({ROOT!r}, "tools", "abi_hosts", "late_destroy.c"), "-ldl"], check=True)
from muscle_synergies_amd import _lib as L
r = subprocess.run([exe, L.LIB_PATH, "early"], capture_output=True, text=True)
sys.stdout.write(r.stdout); sys.stderr.write(r.stderr)
assert r.returncode == 0 and "LATE-DESTROY-OK" in r.stdout, r.returncode
""",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--log", default="")
    ap.add_argument("--timeout", type=float, default=300.0)
    ap.add_argument("--native-bt", action="store_true",
                    help="preload tools/probes/segv_bt.c (compiled here with gcc): a native backtrace on SIGSEGV / SIGABRT, also when the "
                         "process dies inside C exit handlers where faulthandler is gone")
    a = ap.parse_args()
    names = [n for n in (a.only.split(",") if a.only else CASES) if n]
    log = open(a.log, "w") if a.log else None

    def say(*x):
        line = " ".join(str(v) for v in x)
        print(line, flush=True)
        if log:
            log.write(line + "\n")
            log.flush()

    env = dict(os.environ, PYTHONFAULTHANDLER="1", PYTHONUNBUFFERED="1")
    fh = ["-X", "faulthandler"]
    if a.native_bt:
        import tempfile

        so = os.path.join(tempfile.mkdtemp(), "segv_bt.so")
        subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", so, os.path.join(ROOT, "tools", "probes", "segv_bt.c")], check=True)
        env["LD_PRELOAD"] = so
        env.pop("PYTHONFAULTHANDLER")
        fh = []  # (faulthandler would take the signal first)
    bad = 0
    for n in names:
        t0 = time.monotonic()
        fails = []
        for rep in range(a.reps):
            try:
                r = subprocess.run([sys.executable, *fh, "-c", CASES[n]], capture_output=True, text=True,
                                   env=env, timeout=a.timeout)
                rc, out, err = r.returncode, r.stdout, r.stderr
            except subprocess.TimeoutExpired as e:
                rc, out, err = -999, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), "TIMEOUT"
            if rc != 0 or "MARK" not in out:
                keep = [ln for ln in err.splitlines() if not ln.startswith("Extension modules:")]
                fails.append((rep, rc, out[-400:], "\n".join(keep)[-9000:]))
        say("%-28s reps=%d failures=%d (%.1f s)" % (n, a.reps, len(fails), time.monotonic() - t0))
        for rep, rc, out, err in fails[:3]:
            say("  rep %d rc=%d\n  stdout: %s\n  stderr: %s" % (rep, rc, out.strip(), err.strip()))
        bad += len(fails)
    say("EXIT-CASES-%s cases=%d reps=%d failures=%d" % ("OK" if not bad else "FAILED", len(names), a.reps, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""Interpreter / process exit in whatever state the engine is in (profiles/r06_abort_hunt.md).

Round 6 found two ways a process that had computed everything correctly still died at exit:
  * SIGSEGV inside libhsa-runtime64 <- libamdhip64's exit handler after hipLaunchCooperativeKernel had been issued from two host
    threads (the rank range of find_synergies on one long float64 frame did exactly that: 8 / 8 runs) -- the library now launches
    its cooperative kernel with an ordinary launch;
  * SIGABRT ("terminate called without an active exception") when a daemon thread was inside a fit while the main thread ran the
    exit handlers (1 / 8 runs) -- the exit hook of muscle_synergies_amd._lib now drains and gates native calls first.
  * (found by the full-suite runs that followed) SIGSEGV inside hipLaunchKernel when several threads make their first launches at
    once -- HIPNMF_LAUNCH takes a process-wide lock around the first launch of every (function, device); the pool_idle case is the
    reproducer (HIPNMF_FIRST_LAUNCH_LOCK=0: 4 of 40 fresh processes).
Every case runs in a fresh child process (tools/exit_cases.py) and must exit with code 0 after printing its marker."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_every_exit_case_ends_with_code_zero():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exit_cases.py"), "--reps", "2"], capture_output=True, text=True,
                       timeout=1500)
    assert r.returncode == 0 and "EXIT-CASES-OK" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]


CHILD = """
import sys, threading
sys.path.insert(0, {root!r})
import numpy as np
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
X = emg_matrix(2, T=2000, m=16, dtype=np.float64)
kern = []
def work(k):
    W0, H0 = random_init(X, k, k)
    ms.fit_batched(X, W0, H0, max_iter=60, tol=0.0, device="cuda:0")
    kern.append(_lib.get_handle(0).last_kernel())
ts = [threading.Thread(target=work, args=(5,)) for _ in range(4)]
[t.start() for t in ts]; [t.join() for t in ts]
print("MARK", sorted(set(kern)))
"""


def test_cooperative_fits_from_several_threads_do_not_kill_the_process_at_exit():
    """The minimal form of the crash: the same one-matrix float64 fit (routed to fit_coop_kernel) from four threads, then a
    plain exit.  With HIPNMF_COOP_LAUNCH=1 (the cooperative launch API, round 5's launch) this child dies with SIGSEGV in the
    HSA runtime's shutdown; with the ordinary launch it must leave with 0."""
    for _ in range(3):
        r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=300)
        assert "fit_coop_kernel<double" in r.stdout, r.stdout + r.stderr[-2000:]  # the premise: the cooperative kernel really ran
        assert r.returncode == 0, (r.returncode, r.stderr[-2000:])

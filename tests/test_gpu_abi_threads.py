"""The threading clause of the C ABI (include/hip_nmf.h: "distinct handles may be driven concurrently from different host
threads"; SURVEY.md section 8b: "this is how the 8-GPU scatter runs"), executed: tools/abi_threads_stress.py -- plain ctypes + the
HIP runtime, no torch -- drives EVERY solver path of the library (persistent, row-per-lane, one-wave, cooperative, narrow sliced
as a replayed hipGraph, wide / wide4 / wide4d, the wide row-sliced hipGraph path, ragged, both rank sweeps, random init,
envelope, filter, KL; shapes that share a template instance with different dynamic-LDS sizes) from 2, 3 and 8 host threads with
one handle each while the threads also hipMalloc / hipMemcpy / hipFree their own buffers, and compares every output BITWISE with
the same calls run one after the other on one handle.  The reference's seam is re-entrant by construction (a fresh estimator per
call, src/muscle_synergies/analysis.py:862-863; independent ranks, :907-912)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PATHS = ("fit_persistent_kernel<", "fit_rowlane_kernel<", "fit_small_kernel<", "fit_coop_kernel<", "slice_pass_kernel<",
         "fit_wide_kernel<", "fit_wide4_kernel<", "fit_wide4d_kernel<", "[sliced]", "[ragged]", "[kl]", "rank_sweep,",
         "rank_sweep_stop", "random_init", "emg_envelope", "sosfilt,", "sosfilt_scan",
         # round 4 / 5 entry points: the general-shape kernels (one-pass fp32, two-pass fp64 / KL), the 256-channel one-wave
         # instance, the shard building blocks (narrow and general-shape layouts, KL), the native sharded loop with a host-side
         # collective callback, the NNDSVD building blocks
         "big1_pass_kernel<", "big_pass_w_kernel<double", "big_pass_w_kernel<float", "fit_wide_kernel<float,256", "shard_narrow,",
         "shard_wide,", "shard_wide_kl", "fit_tsharded,", "fit_tsharded_wide", "gram+nndsvd_stats+nndsvd_write")


def _stress(*args, timeout=900):
    env = dict(os.environ, HIPNMF_REPO=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "abi_threads_stress.py"), *args], capture_output=True,
                       text=True, env=env, timeout=timeout)
    assert r.returncode == 0 and "ABI-THREADS-OK" in r.stdout, r.stdout[-6000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("threads", [2, 3, 8])
def test_every_path_from_concurrent_threads_is_bitwise_the_sequential_result(threads):
    out = _stress("--threads", str(threads), "--rounds", "2")
    line = [ln for ln in out.splitlines() if ln.startswith("ABI-THREADS-OK")][-1]
    for path in PATHS:
        assert path in line + ",", (path, line)


def test_graph_replayed_and_cooperative_paths_in_lock_step():
    """The chip-filling paths -- both row-sliced families replayed as hipGraphs, the cooperative kernel -- with every thread in
    the same case at the same time (round 3's failure: 'operation failed due to a previous error during capture')."""
    _stress("--threads", "3", "--rounds", "4", "--same-order", "--only",
            "wide_sliced,wide4_sliced,wide4d_sliced_stop,wide_sliced_auto,sliced_graph,sliced_graph_stop,coop,coop_f64,big,big_stop,"
            "big_f64,big_kl,big_kl_two_pass")


def test_rank_range_on_long_and_wide_frames_runs_concurrently_with_torch_in_the_process():
    """find_synergies(df, 2, 6) through the Python host (torch's null-stream copies on the calling thread while the pool
    threads fit): the frames round 3 had to gate out -- 16 x 10 000 (cooperative), 8 x 6 000 fp64, 64 x 20 000 fp64 (wide
    row-sliced) -- identical to the sequential loop."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "repro", "rank_threads_long_matrix.py")], capture_output=True,
                       text=True, env=dict(os.environ, GRAFT_REPO_ROOT=ROOT), timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if "identical=" in ln]
    assert len(lines) == 3 and all("identical=True" in ln for ln in lines), r.stdout


def test_a_host_threads_legacy_stream_copy_cannot_break_a_fit_anymore():
    """The root cause, pinned: while thread A runs graph-replayed fits, thread B does nothing but synchronous hipMemcpy's of its
    own buffer.  With stream capture inside the library B's copies failed with hipErrorStreamCaptureImplicit and A's fits with
    hipErrorStreamCaptureInvalidated; now neither notices the other."""
    script = r'''
import ctypes, os, sys, threading
import numpy as np
sys.argv = ["x"]
sys.path.insert(0, os.path.join(os.environ["HIPNMF_REPO"], "tools"))
import abi_threads_stress as S
h = ctypes.c_void_p(); S.ok(S.lib.hipnmf_create(0, ctypes.byref(h)), "create")
name, base = S.CASES["wide4_sliced"](h)
stop, bad = threading.Event(), []
def copier():
    buf = np.zeros(1 << 18, np.float32); p = ctypes.c_void_p()
    assert S.hip.hipMalloc(ctypes.byref(p), buf.nbytes) == 0
    n = 0
    while not stop.is_set():
        rc = S.hip.hipMemcpy(buf.ctypes.data_as(ctypes.c_void_p), p, buf.nbytes, 2)
        if rc: bad.append(rc)
        n += 1
    S.hip.hipFree(p); print("copies", n)
t = threading.Thread(target=copier); t.start()
for i in range(40):
    for c in ("wide4_sliced", "sliced_graph"):
        _, out = S.CASES[c](h)
        if c == "wide4_sliced": assert S.same(out, base)
stop.set(); t.join()
assert not bad, bad[:5]
print("LEGACY-COPY-OK")
'''
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=dict(os.environ, HIPNMF_REPO=ROOT),
                       timeout=900)
    assert r.returncode == 0 and "LEGACY-COPY-OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]

#!/usr/bin/env python3
"""Second bisect of the exit crash: which (dtype, shape, path) in worker threads makes the process segfault at exit."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TEMPLATE = """
import sys, os, threading
sys.path.insert(0, {root!r})
{env}
import numpy as np
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
X = emg_matrix(2, T={T}, m={m}, dtype=np.{dt})
kern = []
def work(k):
    W0, H0 = random_init(X, k, k)
    ms.fit_batched(X, W0, H0, max_iter=100, tol=0.0, device='cuda:0')
    kern.append(_lib.get_handle(0).last_kernel())
    {after}
ks = {ks}
if {threads}:
    ts = [threading.Thread(target=work, args=(k,)) for k in ks]
    [t.start() for t in ts]; [t.join() for t in ts]
else:
    [work(k) for k in ks]
print('MARK', sorted(set(kern)))
"""
def variant(name, dt="float64", m=8, T=2000, ks=(2, 3, 4, 5, 6), threads=True, env="", after="pass"):
    return name, TEMPLATE.format(root=ROOT, dt=dt, m=m, T=T, ks=list(ks), threads=threads, env=env, after=after)
V = [
 variant("f64_8ch_k2..6_threads"),
 variant("f64_8ch_k2..6_main_thread", threads=False),
 variant("f64_8ch_k4_x5_threads", ks=(4, 4, 4, 4, 4)),
 variant("f64_8ch_k4_one_thread", ks=(4,)),
 variant("f32_8ch_k2..6_threads", dt="float32"),
 variant("f32_16ch_k5_x5_threads", dt="float32", m=16, ks=(5,) * 5),
 variant("f64_16ch_k5_x5_threads", m=16, ks=(5,) * 5),
 variant("f64_8ch_threads_COOP0", env="os.environ['HIPNMF_COOP']='0'"),
 variant("f64_8ch_threads_GRAPH0", env="os.environ['HIPNMF_GRAPH']='0'"),
 variant("f64_8ch_threads_release_in_thread", after="_lib.release_thread_handles()"),
 variant("f64_8ch_T300_threads", T=300),
 variant("f64_8ch_T20000_threads", T=20000),
]
env = dict(os.environ, PYTHONFAULTHANDLER="1", PYTHONUNBUFFERED="1")
for name, code in V:
    rcs, out = [], ""
    for _ in range(3):
        r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, env=env, timeout=300)
        rcs.append(r.returncode)
        out = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    print(f"{name:38s} rc={rcs} {out[:300]}", flush=True)

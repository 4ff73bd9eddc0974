#!/usr/bin/env python3
"""Bisect of the interpreter-exit crash of the rank-range call (round 6; tools/exit_cases.py case pool_idle: SIGSEGV 8 / 8).
Each variant is a fresh child process; prints rc per variant."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PRE = f"import sys, os, threading, time\nsys.path.insert(0, {ROOT!r})\nimport numpy as np\n"
V = {
 "pandas_only": PRE + "import pandas\nprint('MARK')\n",
 "torch_pandas": PRE + "import torch, pandas\ntorch.zeros(4, device='cuda:0').sum().item()\nprint('MARK')\n",
 "pkg_import_only": PRE + "import muscle_synergies_amd as ms\nprint('MARK')\n",
 "one_rank": PRE + """
import pandas, muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_matrix
df = pandas.DataFrame(emg_matrix(2, T=2000, m=8, dtype=np.float64), columns=list('abcdefgh'))
res = ms.find_synergies(df, 3, solver='mu', max_iter=100, init='random', random_state=0)
print('MARK')
""",
 "range_loop_no_pool": PRE + """
os.environ['HIPNMF_RANK_THREADS'] = '0'
import pandas, muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_matrix
df = pandas.DataFrame(emg_matrix(2, T=2000, m=8, dtype=np.float64), columns=list('abcdefgh'))
res = ms.find_synergies(df, 2, 6, solver='mu', max_iter=100, init='random', random_state=0)
print('MARK')
""",
 "range_pool": PRE + """
import pandas, muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_matrix
df = pandas.DataFrame(emg_matrix(2, T=2000, m=8, dtype=np.float64), columns=list('abcdefgh'))
res = ms.find_synergies(df, 2, 6, solver='mu', max_iter=100, init='random', random_state=0)
print('MARK')
""",
 "range_pool_no_shutdown_hook": PRE + """
import pandas, muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix
df = pandas.DataFrame(emg_matrix(2, T=2000, m=8, dtype=np.float64), columns=list('abcdefgh'))
res = ms.find_synergies(df, 2, 6, solver='mu', max_iter=100, init='random', random_state=0)
import atexit; atexit.unregister(_lib.shutdown)
print('MARK')
""",
 "range_pool_explicit_shutdown_then_exit": PRE + """
import pandas, muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix
df = pandas.DataFrame(emg_matrix(2, T=2000, m=8, dtype=np.float64), columns=list('abcdefgh'))
res = ms.find_synergies(df, 2, 6, solver='mu', max_iter=100, init='random', random_state=0)
_lib.shutdown(); print('after shutdown', flush=True)
import torch; torch.cuda.synchronize(); print('MARK', flush=True)
""",
 "range_pool_os_exit": PRE + """
import pandas, muscle_synergies_amd as ms
from muscle_synergies_amd.synth import emg_matrix
df = pandas.DataFrame(emg_matrix(2, T=2000, m=8, dtype=np.float64), columns=list('abcdefgh'))
res = ms.find_synergies(df, 2, 6, solver='mu', max_iter=100, init='random', random_state=0)
print('MARK', flush=True); os._exit(0)
""",
 "pool_fit_batched_no_pandas": PRE + """
import muscle_synergies_amd as ms
from concurrent.futures import ThreadPoolExecutor
from muscle_synergies_amd.synth import emg_matrix, random_init
X = emg_matrix(2, T=2000, m=8, dtype=np.float64)
def work(k):
    W0, H0 = random_init(X, k, k)
    return float(ms.fit_batched(X, W0, H0, max_iter=100, tol=0.0, device='cuda:0').reconstruction_err[0])
pool = ThreadPoolExecutor(max_workers=8)
print('MARK', list(pool.map(work, range(2, 7))))
""",
 "hipnmf_model_in_threads": PRE + """
import muscle_synergies_amd as ms, threading
from muscle_synergies_amd.synth import emg_matrix
X = emg_matrix(2, T=2000, m=8, dtype=np.float64)
out = []
def work(k):
    out.append(ms.HipNMF(k, solver='mu', max_iter=100, init='random', random_state=0).fit_transform(X).shape)
ts = [threading.Thread(target=work, args=(k,)) for k in range(2, 7)]
[t.start() for t in ts]; [t.join() for t in ts]
print('MARK', out)
""",
}
env = dict(os.environ, PYTHONFAULTHANDLER="1", PYTHONUNBUFFERED="1")
for name, code in V.items():
    rcs = []
    for _ in range(3):
        r = subprocess.run([sys.executable, "-X", "faulthandler", "-c", code], capture_output=True, text=True, env=env, timeout=300)
        rcs.append(r.returncode)
    tail = [l for l in r.stderr.splitlines() if "amdgpu.ids" not in l and "Warning" not in l and "warnings.warn" not in l][-6:]
    print(f"{name:42s} rc={rcs} mark={'MARK' in r.stdout}", *(["\n    " + "\n    ".join(tail)] if any(rcs) else []), flush=True)

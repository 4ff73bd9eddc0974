#!/bin/bash
# native backtrace of the exit-time crash after cooperative launches from a worker thread (round 6)
R=$(cd "$(dirname "$0")/../.." && pwd)
gcc -shared -fPIC -O1 -o /tmp/segv_bt.so $R/tools/probes/segv_bt.c || exit 1
cat > /tmp/crash.py <<PY
import sys, os, threading, faulthandler
sys.path.insert(0, "$R")
import numpy as np
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
faulthandler.disable()
X = emg_matrix(2, T=2000, m=16, dtype=np.float64)
def work(k):
    W0, H0 = random_init(X, k, k)
    ms.fit_batched(X, W0, H0, max_iter=100, tol=0.0, device='cuda:0')
    print(_lib.get_handle(0).last_kernel(), flush=True)
n = int(os.environ.get("NTHREADS", "5"))
ts = [threading.Thread(target=work, args=(5,)) for _ in range(n)]
[t.start() for t in ts]; [t.join() for t in ts]
print("MARK", flush=True)
PY
for n in 1 2 5; do
  echo "== NTHREADS=$n without preload"; NTHREADS=$n python3 /tmp/crash.py > /dev/null 2>&1; echo "rc=$?"
  echo "== NTHREADS=$n with backtrace preload"; NTHREADS=$n LD_PRELOAD=/tmp/segv_bt.so python3 /tmp/crash.py 2>&1 | grep -v amdgpu.ids | tail -40; echo "rc=${PIPESTATUS[0]}"
done

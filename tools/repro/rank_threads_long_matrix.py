"""The ranks of find_synergies from concurrent threads on frames that take the chip-filling paths (cooperative kernel, row-sliced
hipGraph replay), with torch in the process.  Round 3: "operation failed due to a previous error during capture" for the long /
wide frames (an open stream capture is invalidated by any legacy-stream call of another thread -- here torch's null-stream
copies; profiles/r04_threads_root_cause.md).  Round 4: the library builds its graphs node by node, no frame-size gate is left,
and this prints identical=True for every frame."""
import os, sys, time, numpy as np, pandas as pd
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix
import muscle_synergies_amd.analysis as _an
for dtype, T, m in ((np.float32, 10000, 16), (np.float64, 6000, 8), (np.float64, 20000, 64)):
    X = emg_matrix(5, T=T, m=m, k_true=4, dtype=dtype)
    df = pd.DataFrame(X, columns=[f"m{i}" for i in range(m)])
    kw = dict(solver="mu", init="random", random_state=2, max_iter=300, tol=0.0)
    res = {}
    for mode in ("0", "1", "1"):
        os.environ["HIPNMF_RANK_THREADS"] = mode
        t0 = time.perf_counter(); r = ms.find_synergies(df, 2, 6, **kw); dt = time.perf_counter() - t0
        res[mode] = (r, dt)
    same = all(np.array_equal(res["0"][0].components[k].to_numpy(), res["1"][0].components[k].to_numpy()) for k in res["0"][0].components)
    print(np.dtype(dtype).name, T, m, "loop %.1f ms, concurrent %.1f ms, identical=%s" % (res["0"][1] * 1e3, res["1"][1] * 1e3, same), flush=True)

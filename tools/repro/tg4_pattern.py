"""Where exactly the float64 16-channel k = 5 one-wave kernel goes wrong when its W update walks groups of FOUR tiles
(HIPNMF_SMALL_F64_16_5_TG=4 build, lib/libhip_nmf_tg4.so): error per (tile, component) after ONE iteration, with and without
the H update, against the oracle.  HIPNMF_LIBRARY=.../libhip_nmf_tg4.so python3 tools/repro/tg4_pattern.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
from oracle import nmf_mu_oracle as orc
np.set_printoptions(linewidth=200, precision=2)
h = _lib.Handle(0); h.set_tuning(0, 0, 6)
m, k, T = 16, 5, 256
X = emg_matrix(21, T=T, m=m, k_true=3, dtype=np.float64); W0, H0 = random_init(X, k, 1)
for upd in (False, True):
    for iters in (1, 2):
        r = ms.fit_batched(X, W0, H0, max_iter=iters, tol=0.0, update_H=upd, handle=h)
        Wr, Hr, _ = orc.fit_multiplicative_update(X, W0.copy(), H0.copy(), max_iter=iters, tol=0.0, update_H=upd)
        dW = np.abs(r.W[0] - Wr) / np.abs(Wr).max()
        print("update_H", upd, "iters", iters, h.last_kernel(), "max dW %.2e  max dH %.2e" % (dW.max(), np.abs(r.H[0] - Hr).max()))
        print(" per (tile, component) max rel dW:\n", dW.reshape(4, 64, k).max(axis=1))
        bad = np.argwhere(dW > 1e-9)
        if len(bad):
            print(" bad rows: count", len(set(bad[:, 0])), "lanes", sorted(set(bad[:, 0] % 64))[:70])
            t, c = bad[0]
            print(" first bad: row", t, "comp", c, "got", r.W[0][t, c], "want", Wr[t, c], "W0", W0[t, c], "ratio got/want", r.W[0][t, c] / Wr[t, c])

"""Probe that isolated the miscompiled one-wave instance: every float64 16-channel fit_small_kernel instance, one
iteration against the oracle under variant 6; prints BAD where W or H differ.  Run on the GPU box: python tools/repro/dbg_small_f64_16.py"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib
from muscle_synergies_amd.synth import emg_matrix, random_init
from oracle import nmf_mu_oracle as orc
h = _lib.Handle(0); h.set_tuning(0, 0, 6)
for m in (9, 12, 15, 16):
    for k in (1, 2, 3, 4, 5, 6):
        for T in (64, 200, 256):
            X = emg_matrix(m + k, T=T, m=m, k_true=min(3, m), dtype=np.float64); W0, H0 = random_init(X, k, 1)
            r = ms.fit_batched(X, W0, H0, max_iter=1, tol=0.0, handle=h)
            ref = orc.nmf_mu_fit(X, W0, H0, max_iter=1, tol=0.0)
            dW = np.abs(r.W[0] - ref["W"]).max(); dH = np.abs(r.H[0] - ref["H"]).max()
            print(m, k, T, h.last_kernel(), "dW=%.2e dH=%.2e" % (dW, dH), "BAD" if max(dW, dH) > 1e-9 else "")

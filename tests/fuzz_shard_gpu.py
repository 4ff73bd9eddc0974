#!/usr/bin/env python3
"""Randomised cross-check of the time-shard building blocks (hipnmf_shard_pass / _hupdate / _residual through
HipShardOps: slice_pass_rowlane_kernel for fp32 with 9-16 channels, slice_pass_kernel otherwise, reduce_slices,
hupdate, the residual pair) against the NumPy oracle: one long matrix cut into 1-4 shards emulated on one GPU (the
all-reduce replaced by an explicit sum of the shards' partial sums), shard lengths of every residue mod 4 and around the
slice sizes, both constructors (host arrays / tensors already in the engine's layouts), regularisation, update_H off.
Part of the test infrastructure (it imports oracle/); run by tests/test_gpu_round2.py::test_shard_fuzz or by hand."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from muscle_synergies_amd.tsharded import HipShardOps
from muscle_synergies_amd.synth import emg_matrix, random_init
from oracle import nmf_mu_oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--verbose", action="store_true")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
bad = 0
for case in range(a.cases):
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    m = int(rng.choice([1, 3, 4, 8, 9, 12, 16, 16, 16, 17, 32]))
    k = int(rng.integers(1, min(m, 8) + 1))
    T = int(rng.choice([5, 64, 257, 1000, 1024, 2047, 4096, 10001, 65536, 70003, 262144]))
    n_sh = int(rng.choice([1, 2, 3, 4]))
    iters = int(rng.choice([1, 3, 12]))
    update_H = bool(rng.random() < 0.85)
    reg = (0.0,) * 4 if rng.random() < 0.75 else tuple(float(v) for v in rng.random(4) * 0.05)
    native = bool(rng.random() < 0.5)
    cuts = sorted(set([0, T] + [int(c) for c in rng.integers(1, T, size=n_sh - 1)])) if T > n_sh else [0, T]
    if native:  # from_native wants shard lengths that are multiples of 4 except for the last one (padding rows are zero)
        cuts = sorted(set([0, T] + [c // 4 * 4 for c in cuts[1:-1] if c // 4 * 4 > 0]))
    X = emg_matrix(7000 + case, T=T, m=m, k_true=min(5, m), dtype=dtype)
    W0, H0 = random_init(X, k, seed=case)
    desc = (f"case {case}: {np.dtype(dtype).name} T={T} m={m} k={k} shards at {cuts} it={iters} upH={update_H} "
            f"reg={reg != (0.0,) * 4} native={native}")
    if a.verbose:
        print("RUN", desc, flush=True)
    kw = dict(l1_reg_W=reg[0], l1_reg_H=reg[1], l2_reg_W=reg[2], l2_reg_H=reg[3], update_H=update_H)
    try:
        ops = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if native:
                ld = (hi - lo + 3) // 4 * 4
                Xc = torch.zeros((1, m, ld), dtype=torch.from_numpy(X[:1]).dtype, device="cuda:0")
                Wc = torch.zeros((1, k, ld), dtype=Xc.dtype, device="cuda:0")
                Xc[0, :, :hi - lo] = torch.from_numpy(np.ascontiguousarray(X[lo:hi].T)).cuda()
                Wc[0, :, :hi - lo] = torch.from_numpy(np.ascontiguousarray(W0[lo:hi].T)).cuda()
                ops.append(HipShardOps.from_native(Xc, Wc, torch.from_numpy(H0[None].copy()).cuda(), T=hi - lo, **kw))
            else:
                ops.append(HipShardOps(np.ascontiguousarray(X[lo:hi]), W0[lo:hi], H0, **kw))
        for _ in range(iters):
            s = None
            for o in ops:
                p = o.shard_pass()
                s = p.clone() if s is None else s + p
            if update_H:  # as fit_tsharded does: a transform leaves H alone
                for o in ops:
                    o.h_update(s)
        sse = xsq = None
        for o in ops:
            q_sse, q_xsq = o.residual()
            sse = q_sse.clone() if sse is None else sse + q_sse
            xsq = q_xsq.clone() if xsq is None else xsq + q_xsq
        W = torch.cat([o.result_W() for o in ops], dim=1).cpu().numpy()[0]
        H = ops[0].result_H().cpu().numpy()[0]
    except Exception as e:  # noqa: BLE001 -- a fuzz driver reports and goes on
        print("ERROR", desc, repr(e))
        bad += 1
        continue
    Wr, Hr, _ = orc.fit_multiplicative_update(X, W0.copy(), H0.copy(), iters, 0.0, reg[0], reg[1], reg[2], reg[3], update_H=update_H)
    xn = max(np.linalg.norm(X.astype(np.float64)), 1e-30)
    d = np.linalg.norm(W.astype(np.float64) @ H.astype(np.float64) - Wr.astype(np.float64) @ Hr.astype(np.float64)) / xn
    lim = 2e-5 if dtype == np.float32 else 1e-9
    R = X.astype(np.float64) - Wr.astype(np.float64) @ Hr.astype(np.float64)
    e_sse = np.abs(sse.cpu().numpy()[0, :m].astype(np.float64) - (R ** 2).sum(axis=0)).max() / xn ** 2
    e_xsq = np.abs(xsq.cpu().numpy()[0, :m].astype(np.float64) - (X.astype(np.float64) ** 2).sum(axis=0)).max() / xn ** 2
    if not d <= lim or W.shape != Wr.shape or not max(e_sse, e_xsq) <= 10 * lim:
        print("MISMATCH", desc, f"rel dWH={d:.3e} sse {e_sse:.2e} xsq {e_xsq:.2e} shapes {W.shape} {Wr.shape}")
        bad += 1
print(f"{a.cases} cases, {bad} problems")
sys.exit(1 if bad else 0)

"""Multi-rank orchestration of the time-sharded solver, world_size 2 over gloo on the CPU.

The product's per-shard compute is HIP only; here the *orchestration* (`fit_tsharded`: shard bounds, packed
all-reduce, replicated H update, stop rule, error / VAF assembly) is exercised with an oracle-backed
shard-ops object injected by the test, and compared with the unsharded oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from muscle_synergies_amd.synth import emg_matrix, random_init
from muscle_synergies_amd.tsharded import fit_tsharded, shard_bounds
from oracle import nmf_mu_oracle as orc


class OracleShardOps:
    """CPU stand-in for HipShardOps (test only): one matrix, rows [lo, hi)."""

    def __init__(self, X_s, W_s, H):
        self.X = np.ascontiguousarray(X_s)
        self.W = np.array(W_s, copy=True)
        self.H = np.array(H, copy=True)
        self.k, self.m = self.H.shape

    def shard_pass(self):
        a, b = orc.shard_pass(self.X, self.W, self.H)
        return torch.from_numpy(np.concatenate([a.ravel(), b.ravel()])[None, :].copy())

    def h_update(self, sums):
        s = sums.numpy()[0]
        km = self.k * self.m
        orc.h_update_from_sums(s[:km].reshape(self.k, self.m), s[km:].reshape(self.k, self.k), self.H)

    def residual(self):
        r = self.X - self.W @ self.H
        return torch.from_numpy((r ** 2).sum(axis=0)[None, :]), torch.from_numpy((self.X ** 2).sum(axis=0)[None, :])

    def result_W(self):
        return torch.from_numpy(self.W)[None]

    def result_H(self):
        return torch.from_numpy(self.H)[None]


class OracleKLShardOps(OracleShardOps):
    """The same stand-in for beta_loss='kullback-leibler': sums = [W^T (X / WH) | colsum(W) in column 0], the residual's first
    output is the divergence per column, residual_squared() the squared error (what HipShardOps(beta_loss=...) does)."""

    kl = True

    def shard_pass(self):
        a, b = orc.kl_shard_pass(self.X, self.W, self.H)
        return torch.from_numpy(np.concatenate([a.ravel(), b.ravel()])[None, :].copy())

    def h_update(self, sums):
        s = sums.numpy()[0]
        km = self.k * self.m
        orc.kl_h_update_from_sums(s[:km].reshape(self.k, self.m), s[km:].reshape(self.k, self.k), self.H)

    def residual(self):
        return (torch.from_numpy(orc.kl_divergence_columns(self.X, self.W, self.H)[None, :]),
                torch.from_numpy((self.X ** 2).sum(axis=0)[None, :]))

    def residual_squared(self):
        return OracleShardOps.residual(self)


class SharedHOracleShardOps(OracleShardOps):
    """Sub-shard flavour for MultiShardOps: ``H`` is a torch tensor shared by all sub-shards of a rank."""

    def __init__(self, X_s, W_s, H_shared):
        self.X = np.ascontiguousarray(X_s)
        self.W = np.array(W_s, copy=True)
        self.Ht = H_shared            # torch tensor [1, k, m] (float64), updated in place
        self.H = self.Ht.numpy()[0]   # NumPy view of the same memory
        self.k, self.m = self.H.shape

    @property
    def Wc(self):
        return torch.from_numpy(self.W)[None]

    def result_H(self):
        return self.Ht


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, T, tol, max_iter, out_dir, kl=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = emg_matrix(21, T=T, m=16, dtype=np.float64)
        W0, H0 = random_init(X, 5, 21)
        lo, hi = shard_bounds(T, world)[rank]
        ops = (OracleKLShardOps if kl else OracleShardOps)(X[lo:hi], W0[lo:hi], H0)
        res = fit_tsharded(ops, max_iter=max_iter, tol=tol)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), W=res.W_local.numpy()[0], H=res.H.numpy()[0],
                 n_iter=res.n_iter, err=res.reconstruction_err.numpy(), vaf=res.vaf.numpy(), lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tol,max_iter", [(0.0, 25), (1e-3, 200)])
def test_two_rank_sharded_equals_unsharded(tmp_path, tol, max_iter):
    T, world = 1002, 2  # not a multiple of the shard alignment on purpose
    mp.spawn(_worker, args=(world, _free_port(), T, tol, max_iter, str(tmp_path)), nprocs=world, join=True)
    X = emg_matrix(21, T=T, m=16, dtype=np.float64)
    W0, H0 = random_init(X, 5, 21)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=max_iter, tol=tol)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert [int(p["lo"]) for p in parts] == [0, 504] and int(parts[-1]["hi"]) == T
    W = np.concatenate([p["W"] for p in parts], axis=0)
    assert W.shape == ref["W"].shape
    for p in parts:
        assert int(p["n_iter"]) == ref["n_iter"]
        np.testing.assert_allclose(p["H"], ref["H"], rtol=1e-9)  # replicated and equal on every rank
        np.testing.assert_allclose(p["err"][0], ref["reconstruction_err"], rtol=1e-9)
        va, vc = orc.vaf(X, ref["W"], ref["H"])
        np.testing.assert_allclose(p["vaf"][0], np.r_[va, vc], rtol=1e-9)
    np.testing.assert_allclose(W, ref["W"], rtol=1e-9)
    if tol > 0:
        assert ref["n_iter"] % 10 == 0 and ref["n_iter"] < max_iter


@pytest.mark.parametrize("tol,max_iter", [(0.0, 25), (1e-3, 200)])
def test_two_rank_sharded_kullback_leibler_equals_unsharded(tmp_path, tol, max_iter):
    """The Kullback-Leibler flavour of the orchestration (round 4): sums = [W^T (X / WH) | colsum(W)], the stop rule on
    sqrt(2 KL) all-reduced per column, VAF from one extra squared-error pass."""
    T, world = 1002, 2
    mp.spawn(_worker, args=(world, _free_port(), T, tol, max_iter, str(tmp_path), True), nprocs=world, join=True)
    X = emg_matrix(21, T=T, m=16, dtype=np.float64)
    W0, H0 = random_init(X, 5, 21)
    ref = orc.nmf_mu_fit_kl(X, W0, H0, max_iter=max_iter, tol=tol)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    W = np.concatenate([p["W"] for p in parts], axis=0)
    for p in parts:
        assert int(p["n_iter"]) == ref["n_iter"]
        np.testing.assert_allclose(p["H"], ref["H"], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(p["err"][0], ref["reconstruction_err"], rtol=1e-9)
        va, vc = orc.vaf(X, ref["W"], ref["H"])
        np.testing.assert_allclose(p["vaf"][0], np.r_[va, vc], rtol=1e-9)
    np.testing.assert_allclose(W, ref["W"], rtol=1e-9, atol=1e-300)


def _multi_worker(rank, world, port, T, out_dir):
    from muscle_synergies_amd.tsharded import MultiShardOps

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = emg_matrix(22, T=T, m=16, dtype=np.float64)
        W0, H0 = random_init(X, 5, 22)
        lo, hi = shard_bounds(T, world)[rank]
        cuts = [lo, lo + (hi - lo) // 3, lo + 2 * (hi - lo) // 3, hi]  # three sub-shards per rank
        H = torch.from_numpy(H0.copy())[None]

        class _Sub(SharedHOracleShardOps):
            pass

        subs = [_Sub(X[a:b], W0[a:b], H) for a, b in zip(cuts[:-1], cuts[1:])]
        for sub in subs:  # MultiShardOps checks that the sub-shards share one H tensor
            sub.H_tensor = H
        ops = MultiShardOps.__new__(MultiShardOps)
        ops.shards = subs
        res = fit_tsharded(ops, max_iter=30, tol=0.0)
        W = np.concatenate([sub.W for sub in subs], axis=0)
        Hn = np.asarray(res.H)  # MultiShardOps hands back the shared H of its first sub-shard
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), W=W, H=Hn[0] if Hn.ndim == 3 else Hn,
                 err=res.reconstruction_err.numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_with_three_sub_shards_each_equal_unsharded(tmp_path):
    """bench.py --config 5 on fewer than 8 GPUs: every rank holds several sub-shards (MultiShardOps) that share
    the replicated H; sums are added per rank, then all-reduced."""
    T, world = 3003, 2
    mp.spawn(_multi_worker, args=(world, _free_port(), T, str(tmp_path)), nprocs=world, join=True)
    X = emg_matrix(22, T=T, m=16, dtype=np.float64)
    W0, H0 = random_init(X, 5, 22)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=30, tol=0.0)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    np.testing.assert_allclose(np.concatenate([p["W"] for p in parts], axis=0), ref["W"], rtol=1e-9)
    for p in parts:
        np.testing.assert_allclose(p["H"], ref["H"], rtol=1e-9)
        np.testing.assert_allclose(p["err"][0], ref["reconstruction_err"], rtol=1e-9)


def test_shard_bounds_cover_and_align():
    for T, w in ((200_000_000, 8), (1002, 2), (10, 4), (7, 8)):
        b = shard_bounds(T, w)
        assert len(b) == w and b[0][0] == 0 and b[-1][1] == T
        assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
        assert all(lo % 4 == 0 or lo == T for lo, _ in b)

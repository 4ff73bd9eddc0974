"""bench.py --config 5's orchestration without a GPU: the sub-shard plan (`tsharded.plan_subshards`), the synthetic
recording generated shard by shard from counter seeds (`synth.emg_shard_torch` on the CPU device), several sub-shards
per rank sharing the replicated H (`MultiShardOps`), one packed all-reduce per iteration over gloo -- at world sizes
1, 2 and 4, with oracle-backed shard ops standing in for the HIP kernels.  The SAME recording must be generated and the
same factors found for every N (no scaling curve is measured here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from muscle_synergies_amd.synth import emg_shard_torch
from muscle_synergies_amd.tsharded import MultiShardOps, fit_tsharded, plan_subshards, shard_bounds
from oracle import nmf_mu_oracle as orc
from test_tsharded_gloo import SharedHOracleShardOps

T5, SUB, M, K, ITERS = 8 * 96, 96, 16, 5, 12  # eight sub-shards in total, like 2e8 rows / 2.5e7 at full size


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_arrays(sid, n):
    """One sub-shard of the recording as the benchmark generates it (native layouts -> sklearn orientation)."""
    Xs, Ws, H0 = emg_shard_torch(5, sid, n, m=M, k=K, device="cpu")
    X = Xs[0, :, :n].t().double().numpy()  # [n, m]
    W = Ws[0, :, :n].t().double().numpy()  # [n, k]
    return np.ascontiguousarray(X), np.ascontiguousarray(W), H0[0].double().numpy()


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = plan_subshards(T5, world, rank, SUB)
        H = None
        subs, rows = [], []
        for t0, n, sid in plan:
            X, W, H0 = _shard_arrays(sid, n)
            if H is None:
                H = torch.from_numpy(H0.copy())[None]
            sub = SharedHOracleShardOps(X, W, H)
            sub.H_tensor = H
            subs.append(sub)
            rows.append((t0, n, sid, float(X.sum()), float(W.sum())))
        ops = MultiShardOps.__new__(MultiShardOps)
        ops.shards = subs
        res = fit_tsharded(ops, max_iter=ITERS, tol=0.0)
        np.savez(os.path.join(out_dir, f"w{world}_rank{rank}.npz"), rows=np.array(rows), W=np.concatenate([s.W for s in subs], axis=0),
                 H=np.asarray(res.H).reshape(K, M), err=res.reconstruction_err.numpy(), vaf=res.vaf.numpy())
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    out = tmp_path_factory.mktemp("config5")
    for world in (1, 2, 4):
        mp.spawn(_worker, args=(world, _free_port(), str(out)), nprocs=world, join=True)
    return {w: [np.load(out / f"w{w}_rank{r}.npz") for r in range(w)] for w in (1, 2, 4)}


def test_plan_covers_the_recording_with_the_same_shard_ids_for_every_world_size():
    for world in (1, 2, 4, 8):
        plans = [plan_subshards(T5, world, r, SUB) for r in range(world)]
        flat = [p for plan in plans for p in plan]
        assert [p[0] for p in flat] == list(range(0, T5, SUB)) and all(p[1] == SUB for p in flat)
        assert [p[2] for p in flat] == list(range(8))  # global sub-shard indices: the same recording for every N
        assert [plan[0][0] for plan in plans] == [lo for lo, _ in shard_bounds(T5, world)]
    # full size: 2e8 rows in sub-shards of 2.5e7 -> 8, 4, 2, 1 sub-shards per rank
    for world, per_rank in ((1, 8), (2, 4), (4, 2), (8, 1)):
        plan = plan_subshards(200_000_000, world, world - 1, 25_000_000)
        assert len(plan) == per_rank and plan[-1][0] + plan[-1][1] == 200_000_000 and plan[-1][2] == 7
    # rows that do not start on a sub-shard boundary get per-rank ids (a different, but still well-defined, recording)
    odd = plan_subshards(1000, 3, 1, 96)
    assert odd[0][2] >= 10_000 and sum(n for _, n, _ in odd) == shard_bounds(1000, 3)[1][1] - shard_bounds(1000, 3)[1][0]


def test_same_recording_and_same_factors_for_every_world_size(runs):
    ref_rows = np.concatenate([p["rows"] for p in runs[1]], axis=0)
    Xfull, Wfull, H0 = [], [], None
    for t0, n, sid in plan_subshards(T5, 1, 0, SUB):
        X, W, H = _shard_arrays(sid, n)
        Xfull.append(X), Wfull.append(W)
        H0 = H if H0 is None else H0
    Xfull, Wfull = np.concatenate(Xfull), np.concatenate(Wfull)
    ref = orc.nmf_mu_fit(Xfull, Wfull, H0, max_iter=ITERS, tol=0.0)  # the unsharded oracle on the whole recording
    for world, parts in runs.items():
        rows = np.concatenate([p["rows"] for p in parts], axis=0)
        np.testing.assert_array_equal(rows, ref_rows)  # same pieces, same data (checksums), whatever N
        W = np.concatenate([p["W"] for p in parts], axis=0)
        np.testing.assert_allclose(W, ref["W"], rtol=1e-9, atol=1e-14)
        for p in parts:
            np.testing.assert_allclose(p["H"], ref["H"], rtol=1e-9)  # replicated: identical on every rank
            np.testing.assert_allclose(p["err"][0], ref["reconstruction_err"], rtol=1e-9)
            va, vc = orc.vaf(Xfull, ref["W"], ref["H"])
            np.testing.assert_allclose(p["vaf"][0], np.r_[va, vc], rtol=1e-9)

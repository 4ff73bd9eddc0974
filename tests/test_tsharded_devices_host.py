"""Config #5 in ONE process (`fit_tsharded_devices`): the thread orchestration -- shard bounds, the host-staged all-reduce
between shard threads, identical replicas of H, gathering W in row order, error propagation without a hang -- on the CPU with
oracle-backed shard ops injected (the product's per-shard compute is HIP only: tests/test_gpu_tsharded_devices.py)."""
import threading
import time

import numpy as np
import pytest
import torch

from muscle_synergies_amd.synth import emg_matrix, random_init
from muscle_synergies_amd.tsharded import HostStagedAllReduce, fit_tsharded_devices, shard_bounds
from oracle import nmf_mu_oracle as orc
from test_tsharded_gloo import OracleKLShardOps, OracleShardOps


def _factory(cls=OracleShardOps):
    return lambda Xs, Ws, H, i: cls(Xs.numpy(), Ws.numpy(), H.numpy())


@pytest.mark.parametrize("n_dev", [1, 2, 3, 5])
@pytest.mark.parametrize("tol,max_iter", [(0.0, 20), (1e-3, 200)])
def test_shard_threads_equal_the_unsharded_oracle(n_dev, tol, max_iter):
    T = 1002  # not a multiple of the shard alignment on purpose
    X = emg_matrix(21, T=T, m=16, dtype=np.float64)
    W0, H0 = random_init(X, 5, 21)
    res = fit_tsharded_devices(X, W0, H0, devices=list(range(n_dev)), max_iter=max_iter, tol=tol, _ops_factory=_factory())
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=max_iter, tol=tol)
    assert isinstance(res.W_local, np.ndarray) and res.W_local.shape == (1, T, 5) and res.H.shape == (1, 5, 16)
    assert res.n_iter == ref["n_iter"]
    np.testing.assert_allclose(res.W_local[0], ref["W"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(res.H[0], ref["H"], rtol=1e-9, atol=1e-12)
    assert abs(float(res.reconstruction_err[0]) - ref["reconstruction_err"]) <= 1e-9 * np.linalg.norm(X)
    assert res.collective["participants"] == n_dev
    # one packed sum per iteration (+ the stop rule's / the final residual reductions)
    assert res.collective["all_reduce_calls"] >= res.n_iter + 1


def test_kullback_leibler_shard_threads():
    T = 700
    X = emg_matrix(5, T=T, m=8, dtype=np.float64) + 1e-3
    W0, H0 = random_init(X, 3, 5)
    res = fit_tsharded_devices(X, W0, H0, devices=[0, 0, 0], max_iter=30, tol=0.0, _ops_factory=_factory(OracleKLShardOps))
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(X, W0.copy(), H0.copy(), 30, 0.0)
    np.testing.assert_allclose(res.W_local[0] @ res.H[0], Wr @ Hr, rtol=1e-9, atol=1e-12)


def test_more_devices_than_row_blocks_uses_fewer_threads():
    X = emg_matrix(2, T=10, m=4, dtype=np.float64)
    W0, H0 = random_init(X, 2, 2)
    res = fit_tsharded_devices(X, W0, H0, devices=list(range(8)), max_iter=5, tol=0.0, _ops_factory=_factory())
    assert res.collective["participants"] == len([b for b in shard_bounds(10, 8) if b[1] > b[0]]) == 3
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=5, tol=0.0)
    np.testing.assert_allclose(res.W_local[0], ref["W"], rtol=1e-9, atol=1e-12)


def test_a_failing_shard_thread_ends_the_others_instead_of_hanging():
    X = emg_matrix(3, T=400, m=8, dtype=np.float64)
    W0, H0 = random_init(X, 3, 3)

    class Boom(OracleShardOps):
        def shard_pass(self):
            raise RuntimeError("shard 1 lost its device")

    def factory(Xs, Ws, H, i):
        return (Boom if i == 1 else OracleShardOps)(Xs.numpy(), Ws.numpy(), H.numpy())

    t0 = time.monotonic()
    with pytest.raises(RuntimeError, match="shard 1 lost its device"):
        fit_tsharded_devices(X, W0, H0, devices=[0, 1, 2], max_iter=10, tol=0.0, _ops_factory=factory)
    assert time.monotonic() - t0 < 30
    assert not [t for t in threading.enumerate() if t.name.startswith("hipnmf-tshard")]


def test_host_staged_sum_is_bitwise_identical_for_every_participant():
    n = 4
    red = HostStagedAllReduce(n, pinned=False)
    rng = np.random.default_rng(0)
    vals = [torch.from_numpy(rng.standard_normal((1, 105)).astype(np.float32) * 10.0 ** rng.integers(-6, 6)) for _ in range(n)]
    outs = [None] * n

    def run(i):
        time.sleep(0.01 * ((i * 7) % n))  # arrival order differs from slot order
        t = vals[i].clone()
        for _ in range(3):  # repeated rounds reuse the slots
            t = red.reducer(i)(vals[i].clone())
        outs[i] = t

    ts = [threading.Thread(target=run, args=(i,)) for i in range(n)]
    [t.start() for t in ts], [t.join() for t in ts]
    want = vals[0].clone()
    for v in vals[1:]:
        want += v  # slot order
    for o in outs:
        assert torch.equal(o, want)


def test_one_recording_only():
    X = np.ones((3, 10, 4))
    with pytest.raises(ValueError, match="ONE recording"):
        fit_tsharded_devices(X, np.ones((3, 10, 2)), np.ones((3, 2, 4)), devices=[0], _ops_factory=_factory())

"""Host logic of the multi-device scatter (muscle_synergies_amd/multi_gpu.py) on the CPU: partitions, thread-per-device
execution, gather order -- with stand-ins for the per-device fits (no GPU, no library call).  The reference loop this
replaces: /root/reference/src/muscle_synergies/analysis.py:907-912 over the trials of project/segment.py:160-207."""
import threading

import numpy as np
import pytest

from muscle_synergies_amd import engine, multi_gpu, preprocess


def test_partition_weighted_is_a_contiguous_cover_balanced_by_weight():
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(1, 60))
        parts = int(rng.integers(1, 9))
        w = rng.integers(1, 5000, size=n).astype(float)
        b = multi_gpu.partition_weighted(w, parts)
        assert len(b) == parts and b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(parts - 1)) and all(lo <= hi for lo, hi in b)
        # no part exceeds its fair share by more than one item's weight
        fair = w.sum() / parts
        assert all(w[lo:hi].sum() <= fair + w.max() + 1e-6 for lo, hi in b)
    assert multi_gpu.partition_weighted([5, 5, 5, 5], 2) == [(0, 2), (2, 4)]
    assert multi_gpu.partition_weighted([10, 1, 1, 1, 1, 1, 1, 1, 1, 2], 2) == [(0, 1), (1, 10)]
    assert multi_gpu.partition_weighted([3], 4)[-1] == (1, 1) and sum(hi - lo for lo, hi in multi_gpu.partition_weighted([3], 4)) == 1
    assert multi_gpu.partition_weighted([0, 0, 0], 3) == [(0, 1), (1, 2), (2, 3)]
    with pytest.raises(ValueError):
        multi_gpu.partition_weighted([1, -1], 2)
    assert multi_gpu.partition(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)] == engine.partition(10, 4)


def test_scatter_runs_one_thread_per_device_and_keeps_batch_order():
    seen = {}
    meet = threading.Barrier(4, timeout=30)  # passes only if the four slices really run at the same time

    def work(lo, hi, d):
        meet.wait()
        seen[(lo, hi)] = (d, threading.get_ident())
        meet.wait()
        return list(range(lo, hi))

    out = multi_gpu.scatter(11, [3, 1, 3, 0], work, _bind_device=False)
    assert [(lo, hi, d) for lo, hi, d, _ in out] == [(0, 3, 3), (3, 6, 1), (6, 9, 3), (9, 11, 0)]
    assert [x for _, _, _, r in out for x in r] == list(range(11))
    assert len({t for _, t in seen.values()}) == 4 and threading.get_ident() not in {t for _, t in seen.values()}
    # fewer items than devices: the empty slices are skipped, nothing is called for them
    out = multi_gpu.scatter(2, [0, 1, 2], lambda lo, hi, d: (lo, hi), _bind_device=False)
    assert [r for _, _, _, r in out] == [(0, 1), (1, 2)]
    # ragged: balanced by rows
    out = multi_gpu.scatter(4, [0, 1], lambda lo, hi, d: (lo, hi), weights=[100, 100, 100, 900], _bind_device=False)
    assert [r for _, _, _, r in out] == [(0, 3), (3, 4)]


def test_scatter_reraises_a_worker_error_after_all_threads_ended():
    done = []

    def work(lo, hi, d):
        if d == 1:
            raise RuntimeError("device 1 failed")
        done.append(d)
        return d

    with pytest.raises(RuntimeError, match="device 1 failed"):
        multi_gpu.scatter(6, [0, 1, 2], work, _bind_device=False)
    assert sorted(done) == [0, 2]


@pytest.fixture
def two_fake_gpus(monkeypatch):
    import torch

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(multi_gpu._lib, "release_thread_handles", lambda: None)


def test_resolve_devices(two_fake_gpus):
    assert multi_gpu.resolve_devices(None) is None
    assert multi_gpu.resolve_devices("all") == [0, 1]
    assert multi_gpu.resolve_devices([1, "cuda:0", 1]) == [1, 0, 1]
    assert multi_gpu.resolve_devices(1) == [1]
    with pytest.raises(ValueError):
        multi_gpu.resolve_devices([2])
    with pytest.raises(ValueError):
        multi_gpu.resolve_devices(["cpu"])
    with pytest.raises(ValueError):
        multi_gpu.resolve_devices([])


def _fake_fit(calls):
    def fit_batched(X, W0, H0, *, device=None, return_numpy=None, **kw):
        import torch

        X = np.asarray(X)
        B, T, m = X.shape
        k = np.asarray(H0).shape[1]
        calls.append((device, B, kw.get("max_iter")))
        tag = float(device.split(":")[1])
        t = torch.from_numpy
        return engine.BatchedResult(t(np.asarray(W0) + X[:, :, :1]), t(np.asarray(H0) * 2), t(np.full(B, 7, np.int32)), t(X.sum(axis=(1, 2))),
                                    t(np.zeros((B, 1 + m))), t(np.zeros((B, m))), t(np.full((B, m), tag)), 1.0 + tag)
    return fit_batched


def test_fit_batched_devices_gathers_in_batch_order(two_fake_gpus, monkeypatch):
    calls = []
    monkeypatch.setattr(engine, "fit_batched", _fake_fit(calls))
    rng = np.random.default_rng(1)
    X, W0, H0 = rng.random((5, 6, 3)), rng.random((5, 6, 2)), rng.random((5, 2, 3))
    r = engine._fit_batched_scattered(X, W0, H0, [0, 1], None, dict(max_iter=9))
    assert sorted(calls) == [("cuda:0", 3, 9), ("cuda:1", 2, 9)]
    assert isinstance(r.W, np.ndarray) and r.W.shape == (5, 6, 2)
    np.testing.assert_array_equal(r.W, W0 + X[:, :, :1])
    np.testing.assert_array_equal(r.H, H0 * 2)
    np.testing.assert_allclose(r.reconstruction_err, X.sum(axis=(1, 2)))
    np.testing.assert_array_equal(r.xsq_col[:, 0], [0, 0, 0, 1, 1])  # which device served which matrix
    assert r.kernel_ms == 2.0
    # torch in -> CPU tensors out; a single matrix is a batch of one
    import torch

    r = engine._fit_batched_scattered(torch.from_numpy(X), torch.from_numpy(W0), torch.from_numpy(H0), [1, 1, 0], None, {})
    assert isinstance(r.W, torch.Tensor) and r.W.device.type == "cpu" and tuple(r.W.shape) == (5, 6, 2)
    r = engine._fit_batched_scattered(X[0], W0[0], H0[0], [0, 1], None, {})
    assert r.W.shape == (1, 6, 2)
    with pytest.raises(ValueError):
        engine._fit_batched_scattered(X, W0[:4], H0, [0, 1], None, {})


def test_rank_sweep_and_recordings_gather(two_fake_gpus):
    import torch

    X = np.arange(7 * 4 * 3, dtype=np.float64).reshape(7, 4, 3)
    seen = []

    def one(Xs, lo, d):
        B = Xs.shape[0]
        seen.append((lo, d, B))
        ids = torch.arange(lo, lo + B)
        return engine.RankSweepResult([2, 3], torch.stack([ids, ids + 100], 1).double(), {k: ids.view(B, 1).double() * k for k in (2, 3)},
                                      {k: ids.int() for k in (2, 3)}, {k: ids.double() for k in (2, 3)},
                                      {k: ids.view(B, 1, 1).expand(B, k, 3).double() for k in (2, 3)}, ids, float(d))

    r = engine._rank_sweep_scattered(X, [0, 1], one)
    assert sorted(seen) == [(0, 0, 4), (4, 1, 3)]
    assert r.selected.tolist() == list(range(7)) and r.vaf_all[:, 1].tolist() == [100 + i for i in range(7)]
    assert r.components[3].shape == (7, 3, 3) and r.components[3][5, 0, 0] == 5 and r.kernel_ms == 1.0
    out = preprocess._scatter_recordings(X, [1, 0, 1], lambda part, d: torch.from_numpy(np.asarray(part)) + 1000 * d)
    assert tuple(out.shape) == (7, 4, 3)
    np.testing.assert_array_equal(out[:, 0, 0].numpy() // 1000, [1, 1, 1, 0, 0, 1, 1])
    np.testing.assert_array_equal(out.numpy() % 1000, X)

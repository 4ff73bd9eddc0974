"""Host-resident batches through the chunked transfer pipeline (engine._fit_batched_pipelined): NumPy in, NumPy out, upload /
fit / download of neighbouring chunks overlapped -- and bitwise the result of the single upload-fit-download call.  The reference's
input is host memory: /root/reference/src/muscle_synergies/analysis.py:739-746."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(B, T, m, k, dtype, seed=0, order="C"):
    rng = np.random.default_rng(seed)
    X = rng.random((B, T, m)).astype(dtype) + 0.01
    if order == "F":  # per-matrix channel-major storage (DataFrame.to_numpy()), seen as [B, T, m]
        X = np.ascontiguousarray(X.transpose(0, 2, 1)).transpose(0, 2, 1)
    W0 = rng.random((B, T, k)).astype(dtype) + 0.1
    H0 = rng.random((B, k, m)).astype(dtype) + 0.1
    return X, W0, H0


@pytest.mark.parametrize("dtype,B,T,m,k,chunk,order", [(np.float32, 37, 500, 16, 5, 8, "C"), (np.float32, 20, 300, 16, 5, 7, "F"),
                                                      (np.float64, 9, 400, 8, 3, 4, "C"), (np.float32, 10, 600, 64, 8, 3, "C"),
                                                      (np.float32, 6, 250, 12, 4, 1, "F")])
def test_pipelined_fit_is_bitwise_the_single_call(dtype, B, T, m, k, chunk, order):
    import muscle_synergies_amd as ms
    from oracle import nmf_mu_oracle as orc

    X, W0, H0 = _batch(B, T, m, k, dtype, seed=B + T, order=order)
    one = ms.fit_batched(X, W0, H0, max_iter=40, tol=0.0, host_chunk=0)
    pip = ms.fit_batched(X, W0, H0, max_iter=40, tol=0.0, host_chunk=chunk)
    for name in ("W", "H", "n_iter", "reconstruction_err", "vaf", "sse_col", "xsq_col"):
        a, b = getattr(pip, name), getattr(one, name)
        assert isinstance(a, np.ndarray) and a.dtype == b.dtype and a.shape == b.shape
        np.testing.assert_array_equal(a, b, err_msg=name)
    ref = orc.nmf_mu_fit(np.ascontiguousarray(X[B - 1]), W0[B - 1], H0[B - 1], max_iter=40, tol=0.0)
    d = np.linalg.norm(pip.W[B - 1].astype(np.float64) @ pip.H[B - 1] - ref["W"].astype(np.float64) @ ref["H"]) / np.linalg.norm(X[B - 1])
    assert d <= 1e-5
    # the stop rule is per matrix: chunking cannot change when a matrix stops
    one = ms.fit_batched(X, W0, H0, max_iter=200, tol=1e-3, host_chunk=0)
    pip = ms.fit_batched(X, W0, H0, max_iter=200, tol=1e-3, host_chunk=chunk)
    np.testing.assert_array_equal(pip.n_iter, one.n_iter)
    np.testing.assert_array_equal(pip.W, one.W)


def test_automatic_chunking_and_multi_device_host_batches(monkeypatch):
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import engine

    X, W0, H0 = _batch(600, 128, 16, 5, np.float32, seed=3)
    assert engine._pipeline_chunk(X, None) == 0          # 4.9 MB: below the threshold, one call
    monkeypatch.setattr(engine, "PIPELINE_MIN_BYTES", 1 << 20)
    monkeypatch.setattr(engine, "PIPELINE_CHUNK_BYTES", 2 << 20)
    c = engine._pipeline_chunk(X, None)
    assert c == 256 and engine._pipeline_chunk(X, 700) == 0 and engine._pipeline_chunk(X, 100) == 100
    auto = ms.fit_batched(X, W0, H0, max_iter=25, tol=0.0)             # three chunks of 256 / 256 / 88
    one = ms.fit_batched(X, W0, H0, max_iter=25, tol=0.0, host_chunk=0)
    np.testing.assert_array_equal(auto.W, one.W)
    np.testing.assert_array_equal(auto.H, one.H)
    two = ms.fit_batched(X, W0, H0, max_iter=25, tol=0.0, devices=[0, 0])  # each device thread pipelines its own slice
    np.testing.assert_array_equal(two.W, one.W)
    with pytest.raises(ValueError):
        ms.fit_batched(X, W0[:, :100], H0, max_iter=5, host_chunk=64)


def test_tail_chunk_takes_the_route_of_the_whole_batch():
    """Round-5 advisor finding: the kernel family depends on the batch size (fp32 16 x 5 at 500 rows: the 4x4x1 kernel for
    batches of at least half the CUs, the lane mapping below), so a tail chunk of 44 matrices used to be fitted by another
    kernel than the 128-matrix chunks before it -- valid, but not the bits of the one-call fit.  The pipeline now names the
    whole batch to the library (hipnmf_set_batch_hint)."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    X, W0, H0 = _batch(300, 500, 16, 5, np.float32, seed=11)
    h = _lib.get_handle(0)
    one = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0, host_chunk=0)
    k_one = h.last_kernel()
    small = ms.fit_batched(X[:44], W0[:44], H0[:44], max_iter=30, tol=0.0, host_chunk=0)
    k_small = h.last_kernel()
    assert k_one != k_small, (k_one, k_small)  # the premise: 44 matrices alone are routed differently
    pip = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0, host_chunk=128)  # 128 + 128 + 44
    assert h.last_kernel() == k_one
    np.testing.assert_array_equal(pip.W, one.W)
    np.testing.assert_array_equal(pip.H, one.H)
    np.testing.assert_array_equal(pip.n_iter, one.n_iter)
    # the hint is gone afterwards: the same 44 matrices alone are routed as a small batch again
    again = ms.fit_batched(X[:44], W0[:44], H0[:44], max_iter=30, tol=0.0, host_chunk=0)
    assert h.last_kernel() == k_small
    np.testing.assert_array_equal(again.W, small.W)
    # stop rule live: n_iter per matrix must not depend on the chunking either
    one = ms.fit_batched(X, W0, H0, max_iter=200, tol=1e-3, host_chunk=0)
    pip = ms.fit_batched(X, W0, H0, max_iter=200, tol=1e-3, host_chunk=128)
    np.testing.assert_array_equal(pip.n_iter, one.n_iter)
    np.testing.assert_array_equal(pip.W, one.W)


def test_devices_argument_hygiene():
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    X, W0, H0 = _batch(12, 300, 8, 3, np.float64, seed=5)
    with pytest.raises(ValueError, match="cannot be combined"):
        ms.fit_batched(X, W0, H0, devices=[0, 0], handle=_lib.get_handle(0))
    with pytest.raises(ValueError, match="cannot be combined"):
        ms.fit_batched(X, W0, H0, devices=[0, 0], overwrite_init=True)
    ref = ms.fit_batched(X, W0, H0, max_iter=20, tol=0.0, host_chunk=0)
    out = ms.fit_batched(X, W0, H0, max_iter=20, tol=0.0, devices=[0, 0], host_chunk=0)  # forwarded to the workers
    np.testing.assert_array_equal(out.W, ref.W)
    with pytest.raises(TypeError):
        ms.fit_batched(X.astype(np.float16), W0, H0, max_iter=5, host_chunk=4)  # reported by the ordinary path
    Xn = X[:, ::-1]  # negative stride: no pipeline, the ordinary path copies
    r = ms.fit_batched(Xn, W0, H0, max_iter=20, tol=0.0, host_chunk=4)
    r2 = ms.fit_batched(np.ascontiguousarray(Xn), W0, H0, max_iter=20, tol=0.0, host_chunk=0)
    np.testing.assert_array_equal(r.W, r2.W)


@pytest.mark.parametrize("dtype,order,reuse", [(np.float32, "C", True), (np.float32, "F", False), (np.float64, "C", True)])
def test_host_batch_registers_once_and_is_bitwise_the_single_call(dtype, order, reuse):
    """ms.HostBatch: the caller's arrays page-locked once, registration / device slots / (reuse_outputs) result arrays kept from
    call to call; results bitwise those of fit_batched; nothing stays registered after close()."""
    import torch

    import muscle_synergies_amd as ms

    X, W0, H0 = _batch(160, 600, 16, 5, dtype, seed=21, order=order)
    one = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0, host_chunk=0)
    hb = ms.HostBatch(X, W0, H0, host_chunk=64, reuse_outputs=reuse)
    assert hb.is_registered(X) and hb.is_registered(W0) and hb.is_registered(X[10:20]) and torch.from_numpy(W0).is_pinned()
    assert not hb.is_registered(H0)  # 51 KB: a small array shares its pages with the allocator's heap and is never page-locked
    for _ in range(3):  # the second and third call reuse everything
        r = hb.fit(max_iter=30, tol=0.0)
        np.testing.assert_array_equal(r.W, one.W)
        np.testing.assert_array_equal(r.H, one.H)
        np.testing.assert_array_equal(r.reconstruction_err, one.reconstruction_err)
    if reuse:
        assert hb.fit(max_iter=30, tol=0.0).W is r.W  # the batch's own result array
    # another rank: new starting points are registered on first sight, the slots follow the geometry
    rng = np.random.default_rng(5)
    W3, H3 = rng.random((160, 600, 3)).astype(dtype) + 0.1, rng.random((160, 3, 16)).astype(dtype) + 0.1
    r3 = hb.fit(W3, H3, max_iter=20, tol=0.0)
    np.testing.assert_array_equal(r3.W, ms.fit_batched(X, W3, H3, max_iter=20, tol=0.0, host_chunk=0).W)
    assert hb.is_registered(W3)
    with pytest.raises(ValueError):
        hb.fit(max_iter=5, devices=[0])
    hb.close()
    hb.close()
    assert not torch.from_numpy(W0).is_pinned() and not torch.from_numpy(W3).is_pinned()
    with pytest.raises(ValueError, match="closed"):
        hb.fit(max_iter=5)
    # X kept in HBM after the first call
    with ms.HostBatch(X, W0, H0, keep_on_device=True, reuse_outputs=reuse) as hk:
        for _ in range(2):
            rk = hk.fit(max_iter=30, tol=0.0)
            np.testing.assert_array_equal(rk.W, one.W)
            np.testing.assert_array_equal(rk.vaf, one.vaf)


def test_host_batches_from_several_threads_leave_no_registration_behind():
    import threading

    import torch

    import muscle_synergies_amd as ms

    errs, keep = [], []

    def work(i):
        try:
            X, W0, H0 = _batch(120, 600, 12, 4, np.float32, seed=100 + i)
            keep.append((X, W0, H0))
            one = ms.fit_batched(X, W0, H0, max_iter=15, tol=0.0, host_chunk=0)
            for _ in range(5):
                with ms.HostBatch(X, W0, H0, host_chunk=48, reuse_outputs=bool(i % 2)) as hb:
                    assert hb.is_registered(X) and hb.is_registered(W0)
                    for _ in range(2):
                        assert np.array_equal(hb.fit(max_iter=15, tol=0.0).W, one.W)
        except BaseException as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts], [t.join() for t in ts]
    assert not errs, errs
    for X, W0, H0 in keep:
        assert not torch.from_numpy(X).is_pinned() and not torch.from_numpy(W0).is_pinned()

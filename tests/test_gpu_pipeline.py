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

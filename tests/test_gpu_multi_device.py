"""``devices=`` on every batched entry point: trials scattered over GPUs, one host thread and handle per device, results in
batch order (BASELINE.json configs #3 / #4; the loop replaced: /root/reference/src/muscle_synergies/analysis.py:907-912).
Runs whatever the number of visible GPUs: ``[0, 0]`` names the same device twice (two threads, two handles), and when the box
has more than one GPU every device takes part too.  Each check is per slice: what a slice returns must be what the
one-device call returns for those trials."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_lists():
    import torch

    n = torch.cuda.device_count()
    lists = [[0, 0], [0, 0, 0]]
    if n > 1:
        lists.append(list(range(n)))
    return lists


def _batch(B, T, m, k, dtype=np.float32, seed=0):
    from muscle_synergies_amd.synth import emg_matrix, random_init

    X = np.stack([emg_matrix(seed + b, T=T, m=m, k_true=min(4, m), dtype=dtype) for b in range(B)])
    inits = [random_init(X[b], k, seed + b) for b in range(B)]
    return X, np.stack([i[0] for i in inits]), np.stack([i[1] for i in inits])


@pytest.mark.parametrize("shape", [(7, 900, 16, 5), (5, 400, 64, 8), (3, 300, 200, 20)])
def test_fit_batched_scattered_equals_the_one_device_call(shape):
    import muscle_synergies_amd as ms
    from oracle import nmf_mu_oracle as orc

    B, T, m, k = shape
    X, W0, H0 = _batch(B, T, m, k)
    one = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0)
    for devs in _device_lists():
        r = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0, devices=devs)
        assert isinstance(r.W, np.ndarray) and r.W.shape == (B, T, k) and r.H.shape == (B, k, m)
        np.testing.assert_array_equal(r.n_iter, one.n_iter)
        if m <= 64:  # one workgroup per matrix: a matrix's result does not depend on its batch
            np.testing.assert_array_equal(r.W, one.W)
            np.testing.assert_array_equal(r.H, one.H)
        else:  # row-sliced general shapes: the slice count follows the batch size, sums regroup
            np.testing.assert_allclose(r.W, one.W, rtol=2e-4, atol=1e-6)
    ref = orc.nmf_mu_fit(X[B - 1], W0[B - 1], H0[B - 1], max_iter=30, tol=0.0)
    d = np.linalg.norm(r.W[B - 1].astype(np.float64) @ r.H[B - 1] - ref["W"].astype(np.float64) @ ref["H"]) / np.linalg.norm(X[B - 1])
    assert d <= 1e-5
    rt = ms.fit_batched_multi_gpu(X, W0, H0, max_iter=30, tol=0.0)  # every visible GPU
    np.testing.assert_allclose(rt.W, one.W, rtol=2e-4, atol=1e-6)


def test_ragged_restarts_and_sweeps_scattered():
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd.engine import fit_ragged, fit_restarts, rank_sweep_batched, rank_sweep_native
    from muscle_synergies_amd.multi_gpu import partition, partition_weighted

    Ts = [333, 1000, 77, 2049, 640, 128]
    parts = [_batch(1, T, 12, 4, seed=10 + i) for i, T in enumerate(Ts)]
    Xs, Ws, Hs = [p[0][0] for p in parts], [p[1][0] for p in parts], [p[2][0] for p in parts]
    one = fit_ragged(Xs, Ws, Hs, max_iter=40, tol=0.0)
    for devs in _device_lists():
        r = fit_ragged(Xs, Ws, Hs, max_iter=40, tol=0.0, devices=devs)
        assert len(r.W) == len(Ts) and all(tuple(w.shape) == (T, 4) for w, T in zip(r.W, Ts))
        for b in range(len(Ts)):
            np.testing.assert_array_equal(r.W[b].numpy(), one.W[b].cpu().numpy())
        np.testing.assert_array_equal(r.H.numpy(), one.H.cpu().numpy())
        np.testing.assert_array_equal(r.reconstruction_err.numpy(), one.reconstruction_err.cpu().numpy())
        assert sum(hi - lo for lo, hi in partition_weighted(Ts, len(devs))) == len(Ts)

    X, _, _ = _batch(9, 700, 8, 3, seed=40)
    Xd = torch.from_numpy(X).cuda()
    whole = rank_sweep_native(Xd, 2, 5, max_iter=60, tol=0.0, seed=3)
    for devs in _device_lists():
        r = rank_sweep_native(Xd, 2, 5, max_iter=60, tol=0.0, seed=3, devices=devs)  # counter-based draws: split-invariant
        assert r.selected.device.type == "cpu"
        np.testing.assert_array_equal(r.selected.numpy(), whole.selected.cpu().numpy())
        np.testing.assert_array_equal(r.vaf_all.numpy(), whole.vaf_all.cpu().numpy())
        for k in r.ranks:
            np.testing.assert_array_equal(r.components[k].numpy(), whole.components[k].cpu().numpy())
        rb = rank_sweep_batched(Xd, 2, 5, max_iter=60, tol=0.0, seed=3, devices=devs)
        for lo, hi in partition(9, len(devs)):  # the slice starting at trial lo draws with seed + lo
            s = rank_sweep_batched(Xd[lo:hi], 2, 5, max_iter=60, tol=0.0, seed=3 + lo)
            np.testing.assert_array_equal(rb.vaf_all[lo:hi].numpy(), s.vaf_all.cpu().numpy())
            np.testing.assert_array_equal(rb.selected[lo:hi].numpy(), s.selected.cpu().numpy())
        rr = fit_restarts(Xd, 3, 4, seed=5, max_iter=50, tol=0.0, devices=devs)
        assert tuple(rr.restart_err.shape) == (9, 4) and tuple(rr.best.W.shape) == (9, 700, 3)
        for lo, hi in partition(9, len(devs)):
            s = fit_restarts(Xd[lo:hi], 3, 4, seed=5 + lo, max_iter=50, tol=0.0)
            np.testing.assert_array_equal(rr.restart_err[lo:hi].numpy(), s.restart_err.cpu().numpy())
            np.testing.assert_array_equal(rr.best.H[lo:hi].numpy(), s.best.H.cpu().numpy())


def test_preprocessing_scattered_is_bitwise_the_one_device_result():
    from muscle_synergies_amd.preprocess import emg_envelope_batched, linear_envelope_batched, sosfilt_batched
    from muscle_synergies_amd.synth import raw_emg

    raw = np.stack([raw_emg(70 + b, 5000, 6) for b in range(7)])
    sos = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "g8_filters.npz"))["lp4_sos"]
    one_env = emg_envelope_batched(raw, 100, reduce_to=200).cpu().numpy()
    one_flt = sosfilt_batched(raw, sos, zero_center=True, rectify=True, mode="scan").cpu().numpy()
    one_lin = linear_envelope_batched(raw, 6, 2000, 4, reduce_to=200).cpu().numpy()
    for devs in _device_lists():
        env = emg_envelope_batched(raw, 100, reduce_to=200, devices=devs)
        assert env.device.type == "cpu" and tuple(env.shape) == (7, 200, 6)
        np.testing.assert_array_equal(env.numpy(), one_env)
        np.testing.assert_array_equal(sosfilt_batched(raw, sos, zero_center=True, rectify=True, mode="scan", devices=devs).numpy(), one_flt)
        np.testing.assert_array_equal(linear_envelope_batched(raw, 6, 2000, 4, reduce_to=200, devices=devs).numpy(), one_lin)


def test_find_synergies_batched_scattered_matches_trial_by_trial():
    import warnings

    import pandas as pd

    import muscle_synergies_amd as ms

    cols = [f"m{j}" for j in range(8)]
    lens = [200, 200, 350, 120, 200, 500, 200]
    dfs = [pd.DataFrame(_batch(1, T, 8, 3, dtype=np.float64, seed=90 + i)[0][0], columns=cols) for i, T in enumerate(lens)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = ms.find_synergies_batched(dfs, 2, 4, max_iter=80, tol=0.0, random_state=0)
        for devs in _device_lists():
            got = ms.find_synergies_batched(dfs, 2, 4, max_iter=80, tol=0.0, random_state=0, devices=devs)
            assert len(got) == len(dfs)
            for a, b in zip(got, one):
                pd.testing.assert_frame_equal(a.vaf_values, b.vaf_values)
                for k in (2, 3, 4):
                    pd.testing.assert_frame_equal(a.components[k], b.components[k])
                    assert a.model[k].n_iter_ == b.model[k].n_iter_
        same = [pd.DataFrame(_batch(1, 200, 8, 3, dtype=np.float64, seed=120 + i)[0][0], columns=cols) for i in range(5)]
        one = ms.find_synergies_batched(same, 3, max_iter=50, tol=0.0)  # equal lengths: the batched on-device NNDSVD
        got = ms.find_synergies_batched(same, 3, max_iter=50, tol=0.0, devices=[0, 0])
        for a, b in zip(got, one):
            pd.testing.assert_frame_equal(a.components, b.components)
        with pytest.raises(ValueError, match="invalid number of components"):
            ms.find_synergies_batched(dfs, 0, devices=[0, 0])

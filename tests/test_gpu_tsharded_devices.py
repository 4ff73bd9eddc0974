"""BASELINE config #5 on the one GPU of the test box:
(a) RCCL really executes: an ``nccl`` process group of world size 1 on cuda:0 -- ``fit_tsharded_hip`` (async mode, torch's
    stream, ``dist.all_reduce``), ``hipnmf_fit_tsharded_*`` with torch.distributed's all-reduce wrapped into the callback, and
    the same entry point with a callback that calls ``ncclAllReduce`` through RCCL's C API on the library's stream
    (``muscle_synergies_amd.rccl``) -- all against the oracle;
(b) ``fit_tsharded_devices``: the single-process form, rows sharded over ``devices=[0, 0]`` / ``[0, 0, 0]`` (one thread + handle
    per slice, host-staged sum), against the oracle and against the one-device solver."""
import os
import socket

import numpy as np
import pytest

from muscle_synergies_amd.synth import emg_matrix, random_init
from oracle import nmf_mu_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rel_wh(X, W, H, ref):
    X64 = X.astype(np.float64)
    return float(np.linalg.norm(W.astype(np.float64) @ H.astype(np.float64) - ref["W"].astype(np.float64) @ ref["H"].astype(np.float64))
                 / np.linalg.norm(X64))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rccl_world1_worker(rank, port, T, out_dir):
    import torch
    import torch.distributed as dist

    from muscle_synergies_amd.rccl import RcclComm
    from muscle_synergies_amd.tsharded import HipShardOps, fit_tsharded_hip

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    out = {"backend": dist.get_backend()}
    try:
        n_ar = [0]
        orig = dist.all_reduce

        def counting(t, *a, **kw):  # evidence that the collective was issued on device tensors
            assert t.is_cuda
            n_ar[0] += 1
            return orig(t, *a, **kw)

        dist.all_reduce = counting
        for name, dt in (("f32", np.float32), ("f64", np.float64)):
            X = emg_matrix(31, T=T, m=16, dtype=dt)
            W0, H0 = random_init(X, 5, 31)
            Xc = np.ascontiguousarray(X)
            # (1) the Python-driven loop: shard kernels asynchronous on torch's stream, dist.all_reduce = ncclAllReduce between them
            n_ar[0] = 0
            r1 = fit_tsharded_hip(Xc, W0, H0, max_iter=40, tol=0.0, device=torch.device("cuda", 0))
            out[f"py_{name}_W"], out[f"py_{name}_H"] = r1.W_local.cpu().numpy()[0], r1.H.cpu().numpy()[0]
            out[f"py_{name}_err"], out[f"py_{name}_calls"] = r1.reconstruction_err.cpu().numpy(), n_ar[0]
            # (2) the native loop, torch.distributed's all-reduce wrapped into the C callback
            n_ar[0] = 0
            r2 = HipShardOps(Xc, W0, H0).fit_native(max_iter=40, tol=0.0)
            out[f"nat_{name}_W"], out[f"nat_{name}_H"], out[f"nat_{name}_calls"] = r2.W_local.cpu().numpy()[0], r2.H.cpu().numpy()[0], n_ar[0]
            # (3) the native loop with ncclAllReduce called straight through RCCL's C API on the LIBRARY's own stream
            comm = RcclComm(rank=0, world_size=1)
            ops = HipShardOps(Xc, W0, H0)
            ops.handle.set_stream(None)  # the handle's own stream: the callback receives it
            ops.handle.set_async(False)  # ... and the library waits for that stream before it returns
            r3 = ops.fit_native(max_iter=40, tol=0.0, collective_fn=comm.callback())
            torch.cuda.synchronize()
            out[f"rccl_{name}_W"], out[f"rccl_{name}_H"], out[f"rccl_{name}_calls"] = r3.W_local.cpu().numpy()[0], r3.H.cpu().numpy()[0], comm.calls
            out[f"rccl_{name}_elements"] = comm.elements
            # stop rule through RCCL (float64: the iteration count is reproducible to the check)
            if dt == np.float64:
                r4 = HipShardOps(Xc, W0, H0).fit_native(max_iter=400, tol=1e-3, collective_fn=comm.callback())
                out["rccl_f64_stop_n_iter"] = r4.n_iter
                out["rccl_f64_stop_W"], out["rccl_f64_stop_H"] = r4.W_local.cpu().numpy()[0], r4.H.cpu().numpy()[0]
            comm.close()
        np.savez(os.path.join(out_dir, "rank0.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_rccl_executes_on_the_one_gpu_world_size_one(tmp_path):
    import torch.multiprocessing as mp

    T = 40_002
    mp.spawn(_rccl_world1_worker, args=(_free_port(), T, str(tmp_path)), nprocs=1, join=True)
    got = np.load(tmp_path / "rank0.npz")
    assert str(got["backend"]) == "nccl"
    for name, dt, lim in (("f32", np.float32, TOL), ("f64", np.float64, 1e-9)):
        X = emg_matrix(31, T=T, m=16, dtype=dt)
        W0, H0 = random_init(X, 5, 31)
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=40, tol=0.0)
        for path in ("py", "nat", "rccl"):
            assert _rel_wh(X, got[f"{path}_{name}_W"], got[f"{path}_{name}_H"], ref) <= lim, (path, name)
            assert int(got[f"{path}_{name}_calls"]) == 41, (path, name, got[f"{path}_{name}_calls"])  # one per iteration + the final residual
        assert abs(float(got[f"py_{name}_err"][0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= lim
        assert int(got[f"rccl_{name}_elements"]) == 40 * (5 * 16 + 25) + 2 * 16
        # a world of one sums nothing: the three routes run the same kernels in the same order
        np.testing.assert_array_equal(got[f"py_{name}_W"], got[f"nat_{name}_W"])
        np.testing.assert_array_equal(got[f"rccl_{name}_W"], got[f"nat_{name}_W"])
    X = emg_matrix(31, T=T, m=16, dtype=np.float64)
    W0, H0 = random_init(X, 5, 31)
    ref2 = orc.nmf_mu_fit(X, W0, H0, max_iter=400, tol=1e-3)
    assert int(got["rccl_f64_stop_n_iter"]) == ref2["n_iter"] < 400
    assert _rel_wh(X, got["rccl_f64_stop_W"], got["rccl_f64_stop_H"], ref2) <= 1e-9


@pytest.mark.parametrize("dtype,m,k", [(np.float32, 16, 5), (np.float64, 16, 5), (np.float64, 8, 3), (np.float32, 64, 8)])
@pytest.mark.parametrize("native", [False, True])
def test_single_process_time_sharding_over_device_slots(dtype, m, k, native):
    """devices=[0, 0] / [0, 0, 0]: per slice a thread, a handle, a replica of H; sums through pinned host memory."""
    from muscle_synergies_amd.tsharded import fit_tsharded_devices

    T = 30_001
    X = emg_matrix(41, T=T, m=m, k_true=min(k, 5), dtype=dtype)
    W0, H0 = random_init(X, k, 41)
    lim = TOL if dtype == np.float32 else 1e-9
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=30, tol=0.0)
    for devices in ([0], [0, 0], [0, 0, 0]):
        res = fit_tsharded_devices(np.ascontiguousarray(X), W0, H0, devices=devices, max_iter=30, tol=0.0, native=native)
        assert isinstance(res.W_local, np.ndarray) and res.W_local.shape == (1, T, k) and res.n_iter == 30
        assert _rel_wh(X, res.W_local[0], res.H[0], ref) <= lim, (devices, native)
        assert abs(float(res.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= lim
        assert res.collective["participants"] == len(devices) and res.collective["all_reduce_calls"] == 31
    if dtype == np.float64 and m == 16:  # sklearn's stop rule on the global residual, decided identically by every shard thread
        ref2 = orc.nmf_mu_fit(X, W0, H0, max_iter=400, tol=1e-3)
        res2 = fit_tsharded_devices(np.ascontiguousarray(X), W0, H0, devices=[0, 0], max_iter=400, tol=1e-3, native=native)
        assert res2.n_iter == ref2["n_iter"] < 400
        assert _rel_wh(X, res2.W_local[0], res2.H[0], ref2) <= lim


def test_one_slot_equals_the_one_device_solver_bitwise_and_subshards_share_h():
    import torch

    from muscle_synergies_amd.tsharded import fit_tsharded_devices, fit_tsharded_hip

    T = 20_003
    X = emg_matrix(43, T=T, m=16, dtype=np.float32)
    W0, H0 = random_init(X, 5, 43)
    Xc = np.ascontiguousarray(X)
    one = fit_tsharded_hip(Xc, W0, H0, max_iter=25, tol=0.0, device=torch.device("cuda", 0))
    dev = fit_tsharded_devices(Xc, W0, H0, devices=[0], max_iter=25, tol=0.0)
    np.testing.assert_array_equal(dev.W_local[0], one.W_local.cpu().numpy()[0])
    np.testing.assert_array_equal(dev.H[0], one.H.cpu().numpy()[0])
    # a device slot whose rows exceed `subshard` keeps them as sub-shards with one shared H (MultiShardOps)
    sub = fit_tsharded_devices(Xc, W0, H0, devices=[0, 0], max_iter=25, tol=0.0, subshard=4_000)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=25, tol=0.0)
    assert sub.W_local.shape == (1, T, 5) and _rel_wh(X, sub.W_local[0], sub.H[0], ref) <= TOL
    # Kullback-Leibler through the same orchestration
    Xk = emg_matrix(44, T=6_000, m=12, dtype=np.float64) + 1e-3
    Wk, Hk = random_init(Xk, 4, 44)
    kl = fit_tsharded_devices(np.ascontiguousarray(Xk), Wk, Hk, devices=[0, 0], max_iter=20, tol=0.0, beta_loss="kullback-leibler")
    Wr, Hr, _ = orc.fit_multiplicative_update_kl(Xk, Wk.copy(), Hk.copy(), 20, 0.0)
    assert np.linalg.norm(kl.W_local[0] @ kl.H[0] - Wr @ Hr) / np.linalg.norm(Xk) <= 1e-9

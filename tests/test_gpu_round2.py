"""Round-2 GPU tests: the headline kernels at the headline shape and iteration count, the matrix-pipe instance over
k = 1..8, one full-size time shard (config #5) with shard-count invariance, the cooperative exchange under load,
the in-process multi-GPU scatter on every visible device, and a 2-rank RCCL run when two GPUs are visible."""
import os
import socket

import numpy as np
import pytest

from oracle import nmf_mu_oracle as orc
from muscle_synergies_amd.synth import emg_matrix, random_init

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _rel_wh(X, W, H, ref):
    xn = np.linalg.norm(X.astype(np.float64))
    wh = W.astype(np.float64) @ H.astype(np.float64)
    wr = ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)
    return np.linalg.norm(wh - wr) / xn


# ------------------------------------------------------------------------------------------------ verdict 1(a)
@pytest.mark.parametrize("variant", [4, 5, 1])
def test_headline_batch_kernel_row_major_500_iterations(g2_full, variant):
    """bench.py's exact kernel, layout and iteration count: row-major [B, 10000, 16] fp32, k = 5, 500 iterations,
    one workgroup per matrix (variant 4: round 1's VALU instance, 5: the matrix-pipe instance, 1: the library's
    pick), against sklearn's recorded checksums (G2, matrix 0) and the oracle (every matrix)."""
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    B = 8
    Xs = [np.ascontiguousarray(emg_matrix(s, dtype=np.float32)) for s in range(B)]  # C order: [T, m] row-major
    inits = [random_init(Xs[s], 5, s) for s in range(B)]
    X = torch.from_numpy(np.stack(Xs)).cuda()
    assert X.is_contiguous() and X.shape == (B, 10000, 16)
    W0 = torch.from_numpy(np.stack([i[0] for i in inits])).cuda()
    H0 = torch.from_numpy(np.stack([i[1] for i in inits])).cuda()
    h = _lib.Handle(0)
    h.set_tuning(512, 0, variant)
    res = ms.fit_batched(X, W0, H0, max_iter=500, tol=0.0, handle=h)
    name = h.last_kernel()
    assert name.startswith({4: "fit_persistent_kernel<float,1,16,5,0>", 5: "fit_rowlane_kernel<5,", 1: "fit_"}[variant]), name
    W, H = res.W.cpu().numpy(), res.H.cpu().numpy()
    err, vaf = res.reconstruction_err.cpu().numpy(), res.vaf.cpu().numpy()
    assert (res.n_iter.cpu().numpy() == 500).all()
    # matrix 0 is G2's case 0: sklearn 1.7.2's own output after 500 iterations
    c = g2_full["cases"][0]
    assert c["dtype"] == "float32" and c["init"] == "random" and c["seed"] == 0
    g = c["iters"]["500"]
    xn = c["X_fro"]
    WH = W[0].astype(np.float64) @ H[0].astype(np.float64)
    assert abs(float(err[0]) - g["reconstruction_err"]) / xn <= TOL
    assert abs(np.sqrt((WH ** 2).sum()) - g["WH_fro"]) / xn <= TOL
    assert np.linalg.norm(WH.sum(axis=0) - np.array(g["WH_colsum"])) / np.linalg.norm(g["WH_colsum"]) <= TOL
    assert np.abs(WH[g2_full["rows"]] - np.array(g["WH_rows"])).max() <= 2e-4
    assert abs(float(vaf[0, 0]) - g["vaf_all"]) <= TOL
    np.testing.assert_allclose(vaf[0, 1:], g["vaf_col"], atol=TOL)
    for b in range(B):
        ref = orc.nmf_mu_fit(Xs[b], inits[b][0], inits[b][1], max_iter=500, tol=0.0)
        assert _rel_wh(Xs[b], W[b], H[b], ref) <= TOL, b
        assert abs(float(err[b]) - float(ref["reconstruction_err"])) / np.linalg.norm(Xs[b]) <= TOL, b
        va, vc = orc.vaf(Xs[b].astype(np.float64), ref["W"].astype(np.float64), ref["H"].astype(np.float64))
        assert abs(vaf[b, 0] - va) <= TOL
        np.testing.assert_allclose(vaf[b, 1:], vc, atol=TOL)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("m,T", [(16, 777), (9, 1000), (13, 64), (16, 5200), (12, 11000)])
def test_matrix_pipe_instance_over_ranks_and_shapes(k, m, T):
    """fit_rowlane_kernel<K> (forced with variant 5) vs the oracle: ragged tails, padded channels, T on both sides
    of the LDS cache capacity, both X layouts, stop rule, transform and regularisation."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    X = emg_matrix(1000 * k + m, T=T, m=m, k_true=min(5, m), dtype=np.float32)
    W0, H0 = random_init(X, k, k)
    h = _lib.Handle(0)
    h.set_tuning(0, 0, 5)
    for layout in ("F", "C"):
        Xl = np.asfortranarray(X) if layout == "F" else np.ascontiguousarray(X)
        res = ms.fit_batched(Xl, W0, H0, max_iter=30, tol=0.0, handle=h)
        assert h.last_kernel().startswith(f"fit_rowlane_kernel<{k},")
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=30, tol=0.0)
        assert _rel_wh(X, res.W[0], res.H[0], ref) <= TOL, (layout, m, k, T)
        assert abs(float(res.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL
        np.testing.assert_allclose(res.W[0], ref["W"], rtol=3e-4, atol=1e-6)
        np.testing.assert_allclose(res.H[0], ref["H"], rtol=3e-4, atol=1e-6)
    if T == 1000:
        res = ms.fit_batched(X, W0, H0, max_iter=300, tol=1e-3, handle=h)
        ref = orc.nmf_mu_fit(X, W0, H0, max_iter=300, tol=1e-3)
        assert int(res.n_iter[0]) == ref["n_iter"] and ref["n_iter"] % 10 == 0
        assert _rel_wh(X, res.W[0], res.H[0], ref) <= TOL
        res = ms.fit_batched(X, W0, H0, max_iter=40, tol=0.0, update_H=False, handle=h)
        Wr, Hr, _ = orc.fit_multiplicative_update(X, W0.copy(), H0.copy(), 40, 0.0, 0.0, 0.0, 0.0, 0.0, update_H=False)
        assert _rel_wh(X, res.W[0], res.H[0], {"W": Wr, "H": Hr}) <= TOL
        np.testing.assert_array_equal(res.H[0], H0)
        res = ms.fit_batched(X, W0, H0, max_iter=40, tol=0.0, handle=h, l1_reg_W=0.02, l1_reg_H=0.01, l2_reg_W=0.03,
                             l2_reg_H=0.04)
        Wr, Hr, _ = orc.fit_multiplicative_update(X, W0.copy(), H0.copy(), 40, 0.0, 0.02, 0.01, 0.03, 0.04)
        assert _rel_wh(X, res.W[0], res.H[0], {"W": Wr, "H": Hr}) <= TOL


def test_matrix_pipe_instance_is_bitwise_reproducible_and_batch_consistent():
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    B = 300  # more workgroups than CUs
    Xs = np.stack([np.ascontiguousarray(emg_matrix(50 + b % 7, T=3000, dtype=np.float32)) for b in range(B)])
    inits = [random_init(Xs[b], 6, b % 7) for b in range(B)]
    X = torch.from_numpy(Xs).cuda()
    W0 = torch.from_numpy(np.stack([i[0] for i in inits])).cuda()
    H0 = torch.from_numpy(np.stack([i[1] for i in inits])).cuda()
    h = _lib.Handle(0)
    h.set_tuning(0, 0, 5)
    a = ms.fit_batched(X, W0, H0, max_iter=50, tol=0.0, handle=h)
    b = ms.fit_batched(X, W0, H0, max_iter=50, tol=0.0, handle=h)
    assert torch.equal(a.W, b.W) and torch.equal(a.H, b.H) and torch.equal(a.reconstruction_err, b.reconstruction_err)
    assert torch.equal(a.W[0], a.W[7]) and torch.equal(a.H[3], a.H[290 - 290 % 7 + 3])  # same inputs, any slot


def test_ragged_batch_and_restarts_on_the_matrix_pipe_instance():
    """Trials of unequal length (hipnmf_fit_ragged_f32) at k = 6 and 16 channels take fit_rowlane_kernel by default;
    the multi-restart flavour shares one X between the restarts of a trial."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    Ts = [900, 64, 5003, 1500, 333]
    Xs = [np.ascontiguousarray(emg_matrix(900 + i, T=t, m=16, dtype=np.float32)) for i, t in enumerate(Ts)]
    inits = [random_init(Xs[i], 6, i) for i in range(len(Ts))]
    h = _lib.Handle(0)
    res = ms.fit_ragged(Xs, [i[0] for i in inits], [i[1] for i in inits], max_iter=40, tol=0.0, handle=h)
    assert h.last_kernel().startswith("fit_rowlane_kernel<6,")
    for i in range(len(Ts)):
        ref = orc.nmf_mu_fit(Xs[i], inits[i][0], inits[i][1], max_iter=40, tol=0.0)
        assert _rel_wh(Xs[i], res.W[i].cpu().numpy(), res.H[i].cpu().numpy(), ref) <= TOL, i
        assert abs(float(res.reconstruction_err[i]) - float(ref["reconstruction_err"])) / np.linalg.norm(Xs[i]) <= TOL
    rs = ms.fit_restarts(np.stack([Xs[0], Xs[0][::-1].copy()]), 7, n_restarts=3, seed=4, max_iter=30, tol=0.0)
    err = np.asarray(rs.restart_err.cpu() if hasattr(rs.restart_err, "cpu") else rs.restart_err)
    assert err.shape == (2, 3) and np.isfinite(err).all()
    best = np.asarray(rs.best.reconstruction_err.cpu() if hasattr(rs.best.reconstruction_err, "cpu") else rs.best.reconstruction_err)
    np.testing.assert_allclose(best, err.min(axis=1), rtol=1e-6)


@pytest.mark.parametrize("dtype,m,k,T", [(np.float64, 8, 3, 200), (np.float64, 8, 6, 256), (np.float64, 5, 2, 1), (np.float64, 16, 5, 200), (np.float64, 11, 6, 77),
                                        (np.float32, 8, 8, 130), (np.float32, 16, 5, 200), (np.float32, 13, 8, 255),
                                        (np.float32, 2, 1, 64)])
def test_one_wave_per_matrix_kernel(dtype, m, k, T):
    """fit_small_kernel (picked automatically for n_samples <= 256; variant 6 pins it): the reference's own matrix
    sizes (200 x 8 after time_normalize), stop rule, transform, regularisation, batch of unequal trials."""
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    B = 5
    Xs = [emg_matrix(300 + b, T=T, m=m, k_true=min(3, m), dtype=dtype) for b in range(B)]
    inits = [random_init(Xs[b], k, b) for b in range(B)]
    X = np.stack([np.ascontiguousarray(x) for x in Xs])
    W0, H0 = np.stack([i[0] for i in inits]), np.stack([i[1] for i in inits])
    h = _lib.Handle(0)
    h.set_tuning(0, 0, 6)
    lim = TOL if dtype == np.float32 else 1e-10
    for kw, okw in (({}, {}), ({"update_H": False}, {"update_H": False}),
                    ({"l1_reg_W": 0.02, "l1_reg_H": 0.01, "l2_reg_W": 0.03, "l2_reg_H": 0.04}, None)):
        res = ms.fit_batched(X, W0, H0, max_iter=60, tol=0.0, handle=h, **kw)
        assert h.last_kernel().startswith("fit_small_kernel<" + ("float" if dtype == np.float32 else "double"))
        for b in range(B):
            if okw is None:
                Wr, Hr, _ = orc.fit_multiplicative_update(Xs[b], W0[b].copy(), H0[b].copy(), 60, 0.0, 0.02, 0.01, 0.03, 0.04)
            else:
                Wr, Hr, _ = orc.fit_multiplicative_update(Xs[b], W0[b].copy(), H0[b].copy(), 60, 0.0, 0.0, 0.0, 0.0, 0.0, **okw)
            assert _rel_wh(Xs[b], res.W[b], res.H[b], {"W": Wr, "H": Hr}) <= lim, (kw, b)
    if T >= 64:
        res = ms.fit_batched(X, W0, H0, max_iter=2000, tol=1e-4, handle=h)
        for b in range(B):
            ref = orc.nmf_mu_fit(Xs[b], W0[b], H0[b], max_iter=2000, tol=1e-4)
            if dtype == np.float64:
                assert int(res.n_iter[b]) == ref["n_iter"]
            assert abs(float(res.reconstruction_err[b]) - float(ref["reconstruction_err"])) / np.linalg.norm(Xs[b]) <= max(lim, 1e-6)
    # the automatic choice takes it too, and the ragged entry point (trials of unequal length)
    auto = ms.fit_batched(X, W0, H0, max_iter=10, tol=0.0)
    assert _lib.get_handle(0).last_kernel().startswith("fit_small_kernel<")
    if T >= 8:
        Tr = [T, max(1, T // 2), max(1, T - 3)]
        rr = ms.fit_ragged([Xs[i][:Tr[i]] for i in range(3)], [W0[i][:Tr[i]] for i in range(3)], [H0[i] for i in range(3)],
                           max_iter=25, tol=0.0)
        for i in range(3):
            ref = orc.nmf_mu_fit(np.ascontiguousarray(Xs[i][:Tr[i]]), W0[i][:Tr[i]], H0[i], max_iter=25, tol=0.0)
            assert _rel_wh(Xs[i][:Tr[i]], np.asarray(rr.W[i].cpu()), np.asarray(rr.H[i].cpu()), ref) <= lim
    with pytest.raises(_lib.HipNmfError, match="fit_small_kernel"):
        big = emg_matrix(1, T=1100, m=m, k_true=min(3, m), dtype=dtype)  # (beyond the 16-tile instances too: tests/test_gpu_small_long.py)
        wb, hb = random_init(big, k, 1)
        ms.fit_batched(big, wb, hb, max_iter=2, tol=0.0, handle=h)


# ------------------------------------------------------------------------------------------------ verdict 1(b)
def test_full_size_rank_shard_is_shard_count_invariant():
    """Config #5's per-rank share on 8 GPUs: ONE shard of 2.5e7 rows x 16 channels (1.6 GB of X, generated on the
    device from counter seeds) through hipnmf_shard_*; the same rows as 2 and as 4 sub-shards summed on the device
    must give the same W (bitwise: the update is row-local) and the same sums / H to rounding level (another
    summation tree); the residual entry point is checked in fp64 on a row subset."""
    import torch

    from muscle_synergies_amd.synth import emg_shard_torch
    from muscle_synergies_amd.tsharded import HipShardOps, MultiShardOps

    T, m, k = 25_000_000, 16, 5
    X, W0, H0 = emg_shard_torch(3, 0, T, m=m, k=k, device="cuda:0")
    assert X.shape == (1, m, T) and float(X.min()) >= 0

    def run(n_sub, iters=3):
        H = H0.clone()
        bounds = [(i * T // n_sub) // 4 * 4 for i in range(n_sub)] + [T]
        shards = []
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            Xc = X[:, :, lo:hi].contiguous() if n_sub > 1 else X
            shards.append(HipShardOps.from_native(Xc, W0[:, :, lo:hi].clone(), H))
        ops = MultiShardOps(shards)
        sums = None
        for _ in range(iters):
            sums = ops.shard_pass()
            ops.h_update(sums)
        sse, xsq = ops.residual()
        torch.cuda.synchronize()
        W = torch.cat([s.Wc for s in shards], dim=2)
        return W, H.clone(), sums.clone(), sse.clone(), xsq.clone()

    W1, H1, S1, sse1, xsq1 = run(1)
    for n_sub in (2, 4):
        Wn, Hn, Sn, ssen, xsqn = run(n_sub)
        rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
        assert rel(Sn, S1) <= 2e-6, (n_sub, rel(Sn, S1))
        assert rel(Hn, H1) <= 2e-6
        assert rel(Wn, W1) <= 1e-5  # H differs at rounding level after the first update, W follows
        assert rel(ssen, sse1) <= 1e-5 and rel(xsqn, xsq1) <= 2e-6
        del Wn
    # first iteration alone: identical H on entry -> W bitwise identical whatever the shard count
    Wa = run(1, iters=1)[0]
    Wb = run(4, iters=1)[0]
    assert torch.equal(Wa, Wb)
    # residual of a row subset in fp64
    n = 200_000
    Xs, Ws = X[0, :, :n].double(), W1[0, :, :n].double()
    r = Xs - (Ws.t() @ H1[0].double()).t()
    sub = HipShardOps.from_native(X[:, :, :n].contiguous(), W1[:, :, :n].contiguous(), H1.clone())
    sse_s, xsq_s = sub.residual()
    torch.cuda.synchronize()
    np.testing.assert_allclose(sse_s[0].cpu().numpy(), (r * r).sum(dim=1).cpu().numpy(), rtol=2e-5)
    np.testing.assert_allclose(xsq_s[0].cpu().numpy(), (Xs * Xs).sum(dim=1).cpu().numpy(), rtol=2e-5)
    assert float(sse1.sum()) < float(xsq1.sum())  # three iterations already explain part of the signal


# ------------------------------------------------------------------------------------------------ hardening
def test_cooperative_exchange_under_load_is_bitwise_stable():
    """200 back-to-back cooperative fits (one 16 x 10 000 matrix, many workgroups exchanging records every iteration)
    while a second stream saturates the memory system: every result must equal the first bit for bit, and the
    row-sliced path (no in-kernel exchange) must agree to rounding level."""
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib

    X = emg_matrix(77, dtype=np.float32)
    W0, H0 = random_init(X, 5, 77)
    Xd = torch.from_numpy(np.ascontiguousarray(X)).cuda()[None]
    Wd, Hd = torch.from_numpy(W0).cuda()[None], torch.from_numpy(H0).cuda()[None]
    h = _lib.Handle(0)
    h.set_tuning(0, 0, 3)
    first = ms.fit_batched(Xd, Wd, Hd, max_iter=60, tol=0.0, handle=h)
    assert h.last_kernel().startswith("fit_coop_kernel")
    hog_stream = torch.cuda.Stream()
    big = torch.empty(512 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda")
    other = torch.empty_like(big)
    for rep in range(200):
        if rep % 4 == 0:
            with torch.cuda.stream(hog_stream):  # ~1 GB of traffic per copy, queued ahead of the fits
                other.copy_(big)
                big.copy_(other)
        r = ms.fit_batched(Xd, Wd, Hd, max_iter=60, tol=0.0, handle=h)
        assert torch.equal(r.W, first.W) and torch.equal(r.H, first.H), rep
        assert torch.equal(r.reconstruction_err, first.reconstruction_err), rep
    torch.cuda.synchronize()
    h.set_tuning(0, 0, 2)
    sliced = ms.fit_batched(Xd, Wd, Hd, max_iter=60, tol=0.0, handle=h)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=60, tol=0.0)
    for r in (first, sliced):
        assert _rel_wh(X, r.W[0].cpu().numpy(), r.H[0].cpu().numpy(), ref) <= TOL


def test_multi_gpu_scatter_on_every_visible_device():
    """fit_batched_multi_gpu over range(device_count()): per-device handles, workspaces and H2D staging."""
    import torch

    import muscle_synergies_amd as ms

    n = torch.cuda.device_count()
    B, T = 4 * n + 3, 900
    Xs = np.stack([np.ascontiguousarray(emg_matrix(600 + b, T=T, dtype=np.float32)) for b in range(B)])
    inits = [random_init(Xs[b], 5, b) for b in range(B)]
    W0, H0 = np.stack([i[0] for i in inits]), np.stack([i[1] for i in inits])
    ref = ms.fit_batched(Xs, W0, H0, max_iter=25, tol=0.0)
    out = ms.fit_batched_multi_gpu(Xs, W0, H0, devices=list(range(n)), max_iter=25, tol=0.0)
    np.testing.assert_allclose(out.W, ref.W, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(out.H, ref.H, rtol=2e-4, atol=1e-6)
    np.testing.assert_array_equal(out.n_iter, ref.n_iter)
    for b in range(B):  # every slice against the oracle, however many devices took part
        o = orc.nmf_mu_fit(Xs[b], inits[b][0], inits[b][1], max_iter=25, tol=0.0)
        assert _rel_wh(Xs[b], out.W[b], out.H[b], o) <= TOL


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _nccl_worker(rank, world, port, T, out_dir):
    import torch
    import torch.distributed as dist

    from muscle_synergies_amd.tsharded import fit_tsharded_hip, shard_bounds

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        X = emg_matrix(31, T=T, m=16, dtype=np.float32)
        W0, H0 = random_init(X, 5, 31)
        lo, hi = shard_bounds(T, world)[rank]
        res = fit_tsharded_hip(np.ascontiguousarray(X[lo:hi]), W0[lo:hi], H0, max_iter=40, tol=0.0,
                               device=torch.device("cuda", rank))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), W=res.W_local.cpu().numpy()[0], H=res.H.cpu().numpy()[0],
                 err=res.reconstruction_err.cpu().numpy(), lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


def test_two_rank_rccl_time_sharded_fit(tmp_path):
    """The config #5 path end to end on two GPUs: hipnmf_shard_* per rank + one RCCL all-reduce per iteration."""
    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (the driver's multi-GPU tier)")
    T, world = 40_002, 2
    mp.spawn(_nccl_worker, args=(world, _free_port(), T, str(tmp_path)), nprocs=world, join=True)
    X = emg_matrix(31, T=T, m=16, dtype=np.float32)
    W0, H0 = random_init(X, 5, 31)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=40, tol=0.0)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    W = np.concatenate([p["W"] for p in parts], axis=0)
    np.testing.assert_array_equal(parts[0]["H"], parts[1]["H"])  # replicated
    assert _rel_wh(X, W, parts[0]["H"], ref) <= TOL
    assert abs(float(parts[0]["err"][0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= TOL


# ------------------------------------------------------------------------------------------------ ABI hardening
def test_batch_larger_than_the_grid_y_limit():
    """B > 65535 with sklearn-ordered W (T x k): the layout conversions ride on grid.y / grid.z and are chunked; the
    solver itself puts the batch on grid.x."""
    import torch

    import muscle_synergies_amd as ms

    B, T, m, k = 70_001, 40, 4, 2
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.rand((B, T, m), generator=g, device="cuda") + 0.05
    W0 = torch.rand((B, T, k), generator=g, device="cuda") + 0.1
    H0 = torch.rand((B, k, m), generator=g, device="cuda") + 0.1
    res = ms.fit_batched(X, W0, H0, max_iter=12, tol=0.0)
    for b in (0, 1, 65_534, 65_535, 65_536, B - 1):
        ref = orc.nmf_mu_fit(X[b].cpu().numpy(), W0[b].cpu().numpy(), H0[b].cpu().numpy(), max_iter=12, tol=0.0)
        np.testing.assert_allclose(res.W[b].cpu().numpy(), ref["W"], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(res.H[b].cpu().numpy(), ref["H"], rtol=2e-4, atol=1e-6)
    Xc = X.transpose(1, 2).contiguous().transpose(1, 2)  # channel-major storage: the X converter is chunked too
    res2 = ms.fit_batched(Xc, W0, H0, max_iter=12, tol=0.0)
    assert torch.allclose(res2.W, res.W, rtol=2e-4, atol=1e-6)


def test_ragged_entry_point_ignores_ldx_and_checks_reserved_fields():
    """hip_nmf.h: ldx / x_batch_stride are ignored by hipnmf_fit_ragged_*; hipnmf_sosfilt_params.mode must be a known mode (the field was the must-be-zero reserved0 before round 4)."""
    import ctypes

    import torch

    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.engine import make_problem
    from muscle_synergies_amd.preprocess import SosfiltParams

    lib = _lib.load()
    h = _lib.Handle(0)
    Ts, m, k = [100, 52], 6, 3
    lds = [(t + 3) // 4 * 4 for t in Ts]
    Xs = [emg_matrix(70 + i, T=t, m=m, k_true=3, dtype=np.float64) for i, t in enumerate(Ts)]
    inits = [random_init(Xs[i], k, i) for i in range(2)]
    xo = [0, m * lds[0]]
    wo = [0, k * lds[0]]
    Xp = torch.zeros(m * sum(lds), dtype=torch.float64, device="cuda")
    Wp = torch.zeros(k * sum(lds), dtype=torch.float64, device="cuda")
    for i in range(2):
        Xp[xo[i]:xo[i] + m * lds[i]].view(m, lds[i])[:, :Ts[i]] = torch.from_numpy(np.ascontiguousarray(Xs[i].T)).cuda()
        Wp[wo[i]:wo[i] + k * lds[i]].view(k, lds[i])[:, :Ts[i]] = torch.from_numpy(np.ascontiguousarray(inits[i][0].T)).cuda()
    Hp = torch.from_numpy(np.stack([i[1] for i in inits])).cuda()
    desc = (ctypes.c_int64 * 8)(Ts[0], xo[0], lds[0], wo[0], Ts[1], xo[1], lds[1], wo[1])
    p = make_problem(2, max(Ts), m, k, x_layout=_lib.X_CHANNEL_MAJOR, ldx=0, x_batch_stride=0,
                     w_layout=_lib.W_COMPONENT_MAJOR, max_iter=25, tol=0.0)
    err = torch.empty(2, dtype=torch.float64, device="cuda")
    rc = lib.hipnmf_fit_ragged_f64(h.ptr, ctypes.byref(p), desc, ctypes.c_void_p(Xp.data_ptr()),
                                   ctypes.c_void_p(Wp.data_ptr()), ctypes.c_void_p(Hp.data_ptr()),
                                   ctypes.c_void_p(err.data_ptr()), None, None, None)
    assert rc == 0, lib.hipnmf_last_error()
    for i in range(2):
        ref = orc.nmf_mu_fit(Xs[i], inits[i][0], inits[i][1], max_iter=25, tol=0.0)
        W = Wp[wo[i]:wo[i] + k * lds[i]].view(k, lds[i])[:, :Ts[i]].t().cpu().numpy()
        np.testing.assert_allclose(W, ref["W"], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(float(err[i]), ref["reconstruction_err"], rtol=1e-9)
    sp = SosfiltParams(ctypes.sizeof(SosfiltParams), 1, 100, 4, 0, 4, 400, 1, 0, -1, 0, 0, 7)  # mode = 7: neither EXACT (0) nor SCAN (1)
    sos = (ctypes.c_double * 6)(1, 0, 0, 1, 0, 0)
    x = torch.zeros(400, dtype=torch.float32, device="cuda")
    rc = lib.hipnmf_sosfilt_f32(h.ptr, ctypes.byref(sp), sos, None, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr()))
    assert rc == _lib.HIPNMF_ERR_BAD_ARG and b"mode" in lib.hipnmf_last_error()


@pytest.mark.gpu
def test_shard_fuzz():
    """tests/fuzz_shard_gpu.py with a fixed seed: one matrix cut into 1-4 shards of arbitrary lengths, both constructors,
    every channel mapping of the shard kernels, against the oracle."""
    import subprocess
    import sys

    from conftest import ROOT

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_shard_gpu.py"), "--cases", "80", "--seed", "4"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 problems" in r.stdout


# ------------------------------------------------------------------------------------------------ native sharded loop
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_native_tsharded_fit_single_rank(dtype):
    """hipnmf_fit_tsharded_* (the sharded loop inside the library, no callback = one rank) against the oracle and
    against the Python-driven loop over the same building blocks: fixed iteration count and the stop rule."""
    from muscle_synergies_amd.tsharded import HipShardOps, fit_tsharded

    T = 20_003
    X = emg_matrix(77, T=T, m=16, dtype=dtype)
    W0, H0 = random_init(X, 5, 77)
    res = HipShardOps(np.ascontiguousarray(X), W0, H0).fit_native(max_iter=30, tol=0.0)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=30, tol=0.0)
    W, H = res.W_local.cpu().numpy()[0], res.H.cpu().numpy()[0]
    lim = TOL if dtype == np.float32 else 1e-9
    assert res.n_iter == 30 and W.shape == (T, 5)
    assert _rel_wh(X, W, H, ref) <= lim
    assert abs(float(res.reconstruction_err[0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= lim
    py = fit_tsharded(HipShardOps(np.ascontiguousarray(X), W0, H0), max_iter=30, tol=0.0)
    np.testing.assert_array_equal(py.W_local.cpu().numpy(), res.W_local.cpu().numpy())  # same kernels, same order
    np.testing.assert_array_equal(py.H.cpu().numpy(), res.H.cpu().numpy())
    np.testing.assert_allclose(res.vaf.cpu().numpy(), py.vaf.cpu().numpy(), rtol=1e-6, atol=1e-7)
    # stop rule live (float64: the iteration count is reproducible to the check)
    if dtype == np.float64:
        res2 = HipShardOps(np.ascontiguousarray(X), W0, H0).fit_native(max_iter=400, tol=1e-3)
        ref2 = orc.nmf_mu_fit(X, W0, H0, max_iter=400, tol=1e-3)
        assert res2.n_iter == ref2["n_iter"] and res2.n_iter < 400
        assert _rel_wh(X, res2.W_local.cpu().numpy()[0], res2.H.cpu().numpy()[0], ref2) <= lim


@pytest.mark.gpu
def test_native_tsharded_fit_callback_plumbing():
    """Two identical ranks emulated in one process: the all-reduce callback doubles the buffer, which is the sum over
    two ranks holding the same rows -- the fit must equal the oracle's on the matrix with every row twice."""
    from muscle_synergies_amd.tsharded import HipShardOps

    T = 3_001
    X = emg_matrix(78, T=T, m=12, dtype=np.float64)
    W0, H0 = random_init(X, 4, 78)
    calls = []

    def double(t):
        calls.append(tuple(t.shape))
        t.mul_(2.0)

    res = HipShardOps(np.ascontiguousarray(X), W0, H0).fit_native(max_iter=25, tol=0.0, all_reduce=double)
    ref = orc.nmf_mu_fit(np.vstack([X, X]), np.vstack([W0, W0]), H0, max_iter=25, tol=0.0)
    np.testing.assert_allclose(res.W_local.cpu().numpy()[0], ref["W"][:T], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(res.H.cpu().numpy()[0], ref["H"], rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(float(res.reconstruction_err[0]), float(ref["reconstruction_err"]), rtol=1e-9)
    assert calls.count((4 * 12 + 4 * 4,)) == 25 and calls.count((2 * 12,)) == 1  # one per iteration + the final residual

    def broken(t):
        raise RuntimeError("transport down")

    with pytest.raises(RuntimeError, match="transport down"):
        HipShardOps(np.ascontiguousarray(X), W0, H0).fit_native(max_iter=5, tol=0.0, all_reduce=broken)


def _gloo_native_worker(rank, world, port, T, out_dir):
    import torch
    import torch.distributed as dist

    from muscle_synergies_amd.tsharded import HipShardOps, shard_bounds

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)  # both ranks share the one GPU of the test box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X = emg_matrix(79, T=T, m=16, dtype=np.float64)
        W0, H0 = random_init(X, 5, 79)
        lo, hi = shard_bounds(T, world)[rank]
        res = HipShardOps(np.ascontiguousarray(X[lo:hi]), W0[lo:hi], H0).fit_native(max_iter=300, tol=1e-3)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), W=res.W_local.cpu().numpy()[0], H=res.H.cpu().numpy()[0],
                 err=res.reconstruction_err.cpu().numpy(), n_iter=res.n_iter)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_native_tsharded_fit_two_processes_over_gloo(tmp_path):
    """Two ranks (two processes on the one GPU, gloo as the transport) through hipnmf_fit_tsharded_f64 with
    torch.distributed's all-reduce as the callback: rows sharded, H replicated, sklearn's stop rule on the global
    residual -- against the unsharded oracle."""
    import torch.multiprocessing as mp

    T, world = 30_001, 2
    mp.spawn(_gloo_native_worker, args=(world, _free_port(), T, str(tmp_path)), nprocs=world, join=True)
    X = emg_matrix(79, T=T, m=16, dtype=np.float64)
    W0, H0 = random_init(X, 5, 79)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=300, tol=1e-3)
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    np.testing.assert_array_equal(parts[0]["H"], parts[1]["H"])  # replicated
    assert int(parts[0]["n_iter"]) == int(parts[1]["n_iter"]) == ref["n_iter"]
    W = np.concatenate([p["W"] for p in parts], axis=0)
    assert _rel_wh(X, W, parts[0]["H"], ref) <= 1e-9
    assert abs(float(parts[0]["err"][0]) - float(ref["reconstruction_err"])) / np.linalg.norm(X) <= 1e-9


# ------------------------------------------------------------------------------------------------ native init / rank sweep
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_device_random_init_law_and_invariances(dtype):
    """hipnmf_random_init_*: sklearn's init='random' law (sqrt(mean(X)/k) |N(0,1)|, _nmf.py:303-314) from the library's
    counter-based generator -- moments of the half-normal, independence of the X layout and of the batch split."""
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd.synth import emg_batch

    B, T, m, k = 6, 4000, 16, 5
    Xb = emg_batch(range(200, 200 + B), T=T, m=m).astype(dtype)  # [B, m, T]
    Xc = torch.from_numpy(Xb).cuda().transpose(1, 2)             # channel-major storage
    Xr = Xc.contiguous()                                         # row-major storage
    W0, H0 = ms.random_init_device(Xc, k, seed=11)
    W1, H1 = ms.random_init_device(Xr, k, seed=11)
    assert torch.equal(W0, W1) and torch.equal(H0, H1)           # layout of X does not matter
    Wa, Ha = ms.random_init_device(Xc[:2], k, seed=11)
    Wb, Hb = ms.random_init_device(Xc[2:], k, seed=11, first_matrix=2)
    assert torch.equal(torch.cat([Wa, Wb]), W0) and torch.equal(torch.cat([Ha, Hb]), H0)  # nor does the batch split
    W2, _ = ms.random_init_device(Xc, k, seed=12)
    assert not torch.equal(W0, W2)
    for b in range(B):
        avg = np.sqrt(Xb[b].astype(np.float64).mean() / k)
        w = W0[b].double().cpu().numpy() / avg
        assert (w > 0).all() and abs(w.mean() - np.sqrt(2 / np.pi)) < 0.02 and abs((w ** 2).mean() - 1.0) < 0.03
        assert abs(np.corrcoef(w[:-1, 0], w[1:, 0])[0, 1]) < 0.05 and abs(np.corrcoef(w[:, 0], w[:, 1])[0, 1]) < 0.05
        hh = H0[b].double().cpu().numpy() / avg
        assert (hh > 0).all() and 0.4 < hh.mean() < 1.3


@pytest.mark.gpu
def test_native_rank_sweep_equals_its_parts_and_the_oracle():
    """hipnmf_rank_sweep_f32 (config #4 as one library call) = hipnmf_random_init + hipnmf_fit_batched per rank, bit for
    bit; the VAF table, the threshold selection, and one trial against the oracle from the same starting point."""
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd.synth import emg_batch

    B, T, m = 10, 1500, 16
    Xb = emg_batch(range(300, 300 + B), T=T, m=m, k_true=4)
    X = torch.from_numpy(Xb).cuda().transpose(1, 2)
    sw = ms.rank_sweep_native(X, 2, 6, vaf_threshold=0.9, max_iter=120, tol=0.0, seed=5)
    assert sw.ranks == [2, 3, 4, 5, 6] and tuple(sw.vaf_all.shape) == (B, 5)
    for i, k in enumerate(sw.ranks):
        W0, H0 = ms.random_init_device(X, k, seed=5 + k)
        r = ms.fit_batched(X, W0, H0, max_iter=120, tol=0.0)
        assert torch.equal(r.H, sw.components[k])
        torch.testing.assert_close(r.vaf[:, 0], sw.vaf_all[:, i], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(r.reconstruction_err, sw.reconstruction_err[k])
        assert int(sw.n_iter[k][0]) == 120
        if k == 4:
            ref = orc.nmf_mu_fit(X[3].cpu().numpy(), W0[3].cpu().numpy(), H0[3].cpu().numpy(), max_iter=120, tol=0.0)
            assert abs(float(sw.reconstruction_err[k][3]) - float(ref["reconstruction_err"])) / np.linalg.norm(Xb[3]) <= TOL
    v = sw.vaf_all.cpu().numpy()
    sel = sw.selected.cpu().numpy()
    for b in range(B):
        ok = np.nonzero(v[b] >= 0.9)[0]
        assert sel[b] == (sw.ranks[ok[0]] if len(ok) else -1)
    assert (np.diff(v, axis=1) > -5e-3).all()  # VAF grows with the rank (up to local-minimum noise)
    with pytest.raises(ValueError, match="invalid number of components"):
        ms.rank_sweep_native(X, 3, 17)


@pytest.mark.gpu
def test_same_xcd_cooperative_mode_and_its_fallback(tmp_path):
    """One matrix on the cooperative kernel: by default its workgroups are picked on one XCD and exchange through that
    XCD's L2; HIPNMF_COOP_XCD=0 keeps the device-scope exchange; HIPNMF_COOP_XCD=2 makes one slice stay away, so the
    head count times out BEFORE anything is updated and the library silently runs the device-scope exchange.  All
    three must give bit-identical factors (same kernels, same summation order)."""
    import subprocess
    import sys

    from conftest import ROOT

    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import muscle_synergies_amd as ms\n"
        "from muscle_synergies_amd import _lib\n"
        "from muscle_synergies_amd.synth import emg_matrix, random_init\n"
        "X = emg_matrix(5, T=10000, m=16, dtype=np.float32); W0, H0 = random_init(X, 5, 5)\n"
        "h = _lib.get_handle(0); h.set_tuning(0, 0, 3)\n"
        "r = ms.fit_batched(X[None], W0[None], H0[None], max_iter=200, tol=0.0)\n"
        "r2 = ms.fit_batched(X[None], W0[None], H0[None], max_iter=200, tol=0.0)\n"
        "assert np.array_equal(np.asarray(r.W), np.asarray(r2.W))\n"
        "np.savez(sys.argv[1], W=np.asarray(r.W), H=np.asarray(r.H), ms=r2.kernel_ms, kern=h.last_kernel())\n" % ROOT)
    out = {}
    for mode in ("1", "0", "2"):
        path = str(tmp_path / f"mode{mode}.npz")
        env = dict(os.environ, HIPNMF_COOP_XCD=mode)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out[mode] = np.load(path)
        assert "fit_coop_kernel" in str(out[mode]["kern"])
    for mode in ("0", "2"):
        np.testing.assert_array_equal(out[mode]["W"], out["1"]["W"])
        np.testing.assert_array_equal(out[mode]["H"], out["1"]["H"])
    X = emg_matrix(5, T=10000, m=16, dtype=np.float32)
    W0, H0 = random_init(X, 5, 5)
    ref = orc.nmf_mu_fit(X, W0, H0, max_iter=200, tol=0.0)
    assert _rel_wh(X, out["1"]["W"][0], out["1"]["H"][0], ref) <= TOL

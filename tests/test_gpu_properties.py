"""Size-independent properties of the hot path at BASELINE.json's full sizes (the oracle would take
minutes there): scaling equivariance, batch independence, run-to-run determinism, monotone residual,
non-negativity, and self-consistency of the reported residual / VAF."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workload():
    import torch

    from muscle_synergies_amd.synth import emg_batch_torch

    X, W0, H0 = emg_batch_torch(4096, T=10000, m=16, k=5, device="cuda:0", seed=123)
    return X.transpose(1, 2), W0, H0  # logical [B, T, m] on channel-major storage


def test_full_batch_properties(workload):
    import torch

    import muscle_synergies_amd as ms

    X, W0, H0 = workload
    r = ms.fit_batched(X, W0, H0, max_iter=60, tol=0.0)
    assert bool((r.n_iter == 60).all())
    assert bool(torch.isfinite(r.W).all()) and bool(torch.isfinite(r.H).all())
    assert bool((r.W >= 0).all()) and bool((r.H >= 0).all())
    # run-to-run determinism: bitwise
    r2 = ms.fit_batched(X, W0, H0, max_iter=60, tol=0.0)
    assert torch.equal(r.W, r2.W) and torch.equal(r.H, r2.H) and torch.equal(r.reconstruction_err, r2.reconstruction_err)
    # reported residual and VAF agree with an fp64 recomputation from the returned factors (subset)
    idx = torch.arange(0, 4096, 97, device=X.device)
    Xd = X[idx].double()
    rec = r.W[idx].double() @ r.H[idx].double()
    err = torch.linalg.norm((Xd - rec).reshape(len(idx), -1), dim=1)
    xn = torch.linalg.norm(Xd.reshape(len(idx), -1), dim=1)
    assert float(((r.reconstruction_err[idx].double() - err).abs() / xn).max()) <= 1e-5
    vaf = 1 - ((Xd - rec) ** 2).sum(dim=(1, 2)) / (Xd ** 2).sum(dim=(1, 2))
    assert float((r.vaf[idx, 0].double() - vaf).abs().max()) <= 1e-5
    # multiplicative updates never increase the Frobenius objective
    r3 = ms.fit_batched(X, W0, H0, max_iter=120, tol=0.0)
    assert bool((r3.reconstruction_err <= r.reconstruction_err * (1 + 1e-5)).all())


def test_power_of_two_scaling_is_exact(workload):
    """(4 X, 2 W0, 2 H0) -> (2 W, 2 H): every update factor is unchanged, so the scaled run is bitwise
    twice the unscaled one (powers of two commute with fp32 rounding)."""
    import torch

    import muscle_synergies_amd as ms

    X, W0, H0 = (t[:512] for t in workload)
    a = ms.fit_batched(X, W0, H0, max_iter=40, tol=0.0)
    b = ms.fit_batched(4 * X, 2 * W0, 2 * H0, max_iter=40, tol=0.0)
    assert torch.equal(2 * a.W, b.W) and torch.equal(2 * a.H, b.H)
    assert torch.equal(4 * a.reconstruction_err, b.reconstruction_err)


def test_batch_members_are_independent(workload):
    import torch

    import muscle_synergies_amd as ms

    X, W0, H0 = (t[:300] for t in workload)
    perm = torch.randperm(300, device=X.device, generator=torch.Generator(device=X.device).manual_seed(1))
    a = ms.fit_batched(X, W0, H0, max_iter=30, tol=0.0)
    b = ms.fit_batched(X[perm], W0[perm], H0[perm], max_iter=30, tol=0.0)
    assert torch.equal(a.W[perm], b.W) and torch.equal(a.H[perm], b.H)
    c = ms.fit_batched(X[:1].expand(5, -1, -1), W0[:1].expand(5, -1, -1), H0[:1].expand(5, -1, -1), max_iter=30, tol=0.0)
    for i in range(1, 5):
        assert torch.equal(c.W[0], c.W[i]) and torch.equal(c.H[0], c.H[i])
    # a 5-matrix call may take the row-sliced path (different, but fixed, summation order)
    assert torch.allclose(c.W[0], a.W[0], rtol=1e-3, atol=1e-6)


def test_stop_rule_in_a_batch_is_per_matrix(workload):
    import muscle_synergies_amd as ms

    X, W0, H0 = (t[:64] for t in workload)
    r = ms.fit_batched(X, W0, H0, max_iter=2000, tol=1e-3)
    n = r.n_iter.cpu().numpy()
    assert (n % 10 == 0).all() and (n < 2000).all() and len(set(n.tolist())) > 1
    one = ms.fit_batched(X[5:6], W0[5:6], H0[5:6], max_iter=2000, tol=1e-3)
    assert int(one.n_iter[0]) == int(n[5])


def test_bench_launch_is_deterministic_and_matches_oracle_row_major_4096():
    """bench.py's exact headline launch -- 4096 matrices 16 x 10 000, k = 5, fp32, ROW-major X streamed in place -- twice:
    bitwise equal; a handful of its matrices against the oracle (the full batch would take the CPU minutes)."""
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.synth import emg_batch_torch
    from oracle import nmf_mu_oracle as orc

    X, W0, H0 = emg_batch_torch(4096, T=10_000, m=16, k=5, device="cuda:0", seed=0)
    Xr = X.transpose(1, 2).contiguous()
    a = ms.fit_batched(Xr, W0, H0, max_iter=20, tol=0.0)
    assert _lib.get_handle(0).last_kernel() == "fit_persistent_kernel<float,1,16,5,0>"
    b = ms.fit_batched(Xr, W0, H0, max_iter=20, tol=0.0)
    assert torch.equal(a.W, b.W) and torch.equal(a.H, b.H) and torch.equal(a.reconstruction_err, b.reconstruction_err)
    for i in (0, 1, 2047, 4095):
        x = Xr[i].cpu().numpy()
        ref = orc.nmf_mu_fit(x, W0[i].cpu().numpy(), H0[i].cpu().numpy(), max_iter=20, tol=0.0)
        wh = a.W[i].double().cpu().numpy() @ a.H[i].double().cpu().numpy()
        wr = ref["W"].astype(np.float64) @ ref["H"].astype(np.float64)
        assert np.linalg.norm(wh - wr) / np.linalg.norm(x) <= 1e-5, i

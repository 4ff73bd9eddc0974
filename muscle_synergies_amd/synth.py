"""Synthetic EMG-envelope matrices for benchmarks and parity tests.

The reference ships one 6 x 8 toy recording (``sample_data/abridged_data.csv``)
and its tutorial factorises a 200 x 8 matrix; BASELINE.json's throughput
configs are a scaled-up synthetic workload on the same call boundary
(``find_synergies`` -> NMF).  This module is the single definition of that
workload so that the host baseline, the oracle and the HIP
engine all see identical bytes.

Every operation below is either a counter-based NumPy bit generator or an
element-wise IEEE operation / sequential cumulative sum -- no BLAS, no
SIMD-width-dependent reductions -- so ``emg_matrix(seed)`` is bit-identical on
the build container and on the GPU box.
"""

from __future__ import annotations

import numpy as np

SEED_BASE = 20260000


def emg_matrix(seed: int, T: int = 10_000, m: int = 16, k_true: int = 5, dtype=np.float32,
               noise: float = 0.05, window: int = 201) -> np.ndarray:
    """One non-negative EMG-envelope-like matrix, sklearn orientation (T x m).

    ``k_true`` non-negative synergies ``S`` (k_true x m), activations ``A``
    (T x k_true) = moving average (``window`` samples, about 0.1 s at 2 kHz) of
    rectified white noise, ``X = A S + noise * |N(0,1)|``, each column scaled
    to a maximum of 1 (as the tutorial does before ``find_synergies``).

    The returned array is **F-contiguous** (channel-major ``m x T`` in memory),
    which is what ``DataFrame.to_numpy()`` hands to sklearn in the reference
    and what BASELINE.json calls a "16 x 10 000 EMG matrix".
    """
    rng = np.random.default_rng(SEED_BASE + int(seed))
    S = rng.random((k_true, m)) ** 2
    a = np.abs(rng.standard_normal((T + window - 1, k_true)))
    c = np.cumsum(a, axis=0)
    c = np.vstack([np.zeros((1, k_true)), c])
    A = (c[window:] - c[:-window]) / window  # (T, k_true)
    X = noise * np.abs(rng.standard_normal((T, m)))
    for j in range(k_true):  # explicit rank-1 accumulation: no BLAS involved
        X += A[:, j, None] * S[None, j, :]
    X /= X.max(axis=0, keepdims=True)
    return np.asfortranarray(X.astype(dtype))


def random_init(X: np.ndarray, k: int, seed: int):
    """``init='random'`` of sklearn (``_nmf.py:303-314``) with ``RandomState(seed)``.

    H is drawn before W; both are ``sqrt(X.mean() / k) * |N(0, 1)|`` cast to
    ``X.dtype``.  Returns C-contiguous ``(W0 (T x k), H0 (k x m))``.
    """
    T, m = X.shape
    avg = np.sqrt(X.mean() / k)
    rng = np.random.RandomState(seed)
    H = avg * rng.standard_normal(size=(k, m)).astype(X.dtype, copy=False)
    W = avg * rng.standard_normal(size=(T, k)).astype(X.dtype, copy=False)
    np.abs(H, out=H)
    np.abs(W, out=W)
    return np.ascontiguousarray(W), np.ascontiguousarray(H)


def raw_emg(seed: int, T: int, m: int, fs: float = 2000.0) -> np.ndarray:
    """Synthetic *raw* (signed) EMG, ``(T, m)`` float64: amplitude-modulated white noise with a small DC
    offset per channel -- input of the envelope preprocessing (``preprocess.py``)."""
    rng = np.random.default_rng(777 + int(seed))
    t = np.arange(T) / fs
    amp = 0.2 + np.abs(np.sin(2 * np.pi * (0.7 + 0.1 * np.arange(m))[None, :] * t[:, None]))
    return amp * rng.standard_normal((T, m)) + 0.01 * rng.standard_normal((1, m))


def emg_batch(seeds, T: int = 10_000, m: int = 16, k_true: int = 5, dtype=np.float32) -> np.ndarray:
    """Stack of ``emg_matrix`` results as a ``[B, m, T]`` C-contiguous array
    (each slice channel-major, the engine's native ``x_layout=1``)."""
    out = np.empty((len(seeds), m, T), dtype=dtype)
    for i, s in enumerate(seeds):
        out[i] = emg_matrix(s, T, m, k_true, dtype).T
    return out


def emg_batch_torch(B: int, T: int = 10_000, m: int = 16, k_true: int = 5, *, k: int = 5,
                    device="cuda", seed: int = 0, chunk: int = 256):
    """Device-side generator of the throughput workload (same recipe as
    ``emg_matrix`` but drawn from torch's generator, so it is not bit-identical
    to the NumPy one -- parity runs use ``emg_batch``).

    Returns ``(X [B, m, T], W0 [B, T, k], H0 [B, k, m])`` float32 tensors on
    ``device``; W0/H0 follow sklearn's ``init='random'`` scaling.
    """
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(SEED_BASE + seed)
    X = torch.empty((B, m, T), dtype=torch.float32, device=device)
    W0 = torch.empty((B, T, k), dtype=torch.float32, device=device)
    H0 = torch.empty((B, k, m), dtype=torch.float32, device=device)
    window = 201
    for lo in range(0, B, chunk):
        n = min(chunk, B - lo)
        S = torch.rand((n, k_true, m), generator=g, device=device) ** 2
        a = torch.randn((n, k_true, T + window - 1), generator=g, device=device).abs_()
        c = torch.cumsum(a.double(), dim=2)
        c = torch.cat([torch.zeros((n, k_true, 1), dtype=c.dtype, device=device), c], dim=2)
        A = ((c[:, :, window:] - c[:, :, :-window]) / window).float()  # n, k_true, T
        x = 0.05 * torch.randn((n, m, T), generator=g, device=device).abs_()
        x += torch.einsum("nkm,nkt->nmt", S, A)
        x /= x.amax(dim=2, keepdim=True)
        X[lo:lo + n] = x
        avg = torch.sqrt(x.mean(dim=(1, 2)) / k).view(n, 1, 1)
        H0[lo:lo + n] = avg * torch.randn((n, k, m), generator=g, device=device).abs_()
        W0[lo:lo + n] = avg * torch.randn((n, T, k), generator=g, device=device).abs_()
    return X, W0, H0


def emg_rank_trials_torch(B: int, T: int = 10_000, m: int = 16, *, k_lo: int = 2, k_hi: int = 6, noise: float = 0.05,
                          device="cuda", seed: int = 0, chunk: int = 256):
    """Trials for the rank-sweep workload (BASELINE.json config #4) whose answer is not the same for everybody: trial
    ``b`` is built from ``k_true[b]`` synergies, ``k_true`` cycling through ``k_lo..k_hi``.  Unlike the smoothed-noise
    activations of :func:`emg_batch_torch` (whose large common mean lets two components reach an uncentered VAF of
    0.9 whatever the true rank), the synergies here fire in bursts at different phases of a gait-like cycle (five
    cycles per trial, ``relu(sin)^3`` envelopes with 30 % trial-to-trial amplitude jitter) and each dominates its own
    group of channels, so a factorisation with fewer components than synergies misses whole bursts: the smallest rank
    with VAF >= 0.90 follows ``k_true`` and a sweep that stops at the threshold skips a trial-dependent number of
    ranks.  Rectified noise of relative size ``noise``; channels max-normalised as everywhere.

    Returns ``(X [B, m, T] float32 on ``device``, k_true [B] int64)``."""
    import math

    import torch

    g = torch.Generator(device=device)
    g.manual_seed(SEED_BASE + 4049 + seed)
    X = torch.empty((B, m, T), dtype=torch.float32, device=device)
    k_true = torch.arange(B, device=device) % (k_hi - k_lo + 1) + k_lo
    t = torch.arange(T, device=device, dtype=torch.float32).view(1, 1, T)
    comp = torch.arange(k_hi, device=device, dtype=torch.float32).view(1, k_hi, 1)
    for lo in range(0, B, chunk):
        n = min(chunk, B - lo)
        kt = k_true[lo:lo + n]
        S = torch.rand((n, k_hi, m), generator=g, device=device) ** 2
        own = (torch.arange(m, device=device).view(1, 1, m) % k_hi) == torch.arange(k_hi, device=device).view(1, k_hi, 1)
        S = 0.15 * S + own.float()
        live = (torch.arange(k_hi, device=device).view(1, k_hi) < kt.view(n, 1)).float().view(n, k_hi, 1)
        jitter = 1.0 + 0.3 * (torch.rand((n, k_hi, 1), generator=g, device=device) - 0.5)
        phase = torch.rand((n, 1, 1), generator=g, device=device)
        A = torch.sin(2 * math.pi * (5.0 * t / T - comp / k_hi + phase)).clamp_(min=0) ** 3 * jitter * live
        x = noise * torch.randn((n, m, T), generator=g, device=device).abs_()
        x += torch.einsum("nkm,nkt->nmt", S, A)
        x /= x.amax(dim=2, keepdim=True)
        X[lo:lo + n] = x
    return X, k_true


def emg_shard_torch(seed: int, shard: int, T_shard: int, m: int = 16, k_true: int = 5, *, k: int = 5,
                    device="cuda", piece: int = 4_000_000, scale=None):
    """One time shard of a very long synthetic recording, generated on the device from counter-based seeds
    (BASELINE.json config #5: a single 16 x 2e8 matrix cannot round-trip through the host).

    The recording is defined shard by shard: shard ``s`` draws from ``torch.Generator(SEED_BASE + 7919 seed + s)``,
    so any rank can build exactly its own rows; the synergies ``S`` (``k_true x m``) depend on ``seed`` only and are
    common to all shards.  Same recipe as :func:`emg_batch_torch` (rectified noise smoothed over 201 samples times
    non-negative synergies plus rectified noise); channels are scaled by ``scale`` (``[m]`` tensor; default: one
    fixed constant, since a per-channel maximum would need a pass over all shards).

    Returns ``(X [1, m, ld], W0 [1, k, ld], H0 [1, k, m])`` in the engine's native layouts (channel-major /
    component-major, ``ld = T_shard`` rounded up to a multiple of 4, padding rows zero).
    """
    import torch

    gs = torch.Generator(device=device)
    gs.manual_seed(SEED_BASE + 7919 * seed)
    S = torch.rand((k_true, m), generator=gs, device=device) ** 2
    H0 = torch.rand((1, k, m), generator=gs, device=device) * 0.5 + 0.1
    g = torch.Generator(device=device)
    g.manual_seed(SEED_BASE + 7919 * seed + 1 + shard)
    ld = (T_shard + 3) // 4 * 4
    X = torch.zeros((1, m, ld), dtype=torch.float32, device=device)
    W0 = torch.zeros((1, k, ld), dtype=torch.float32, device=device)
    window = 201
    norm = 1.0 / 3.0 if scale is None else None
    for lo in range(0, T_shard, piece):
        n = min(piece, T_shard - lo)
        a = torch.randn((1, k_true, n + window - 1), generator=g, device=device).abs_()
        A = torch.nn.functional.avg_pool1d(a, window, stride=1)[0]  # k_true x n, moving average of 201 samples
        x = 0.05 * torch.randn((m, n), generator=g, device=device).abs_()
        x.addmm_(S.t(), A)
        if scale is None:
            x *= norm
        else:
            x /= scale.view(m, 1)
        X[0, :, lo:lo + n] = x
        W0[0, :, lo:lo + n] = 0.3 * torch.randn((k, n), generator=g, device=device).abs_()
        del a, A, x
    return X, W0, H0

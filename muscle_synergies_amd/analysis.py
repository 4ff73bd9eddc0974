"""Synergy extraction with the call surface of the reference's ``muscle_synergies.analysis``.

Mirrors, for the NMF hot path only, ``find_synergies`` (``src/muscle_synergies/analysis.py:713-914``),
``vaf`` (``:597-667``) and ``SynergyRunResult`` (``:670-710``): same signatures, same validation
messages, same return structure.  The one difference is what happens at the seam
``NMF(...).fit_transform(X)`` (``:862-863``): when the keyword arguments select sklearn's
multiplicative-update solver with the Frobenius loss (``solver='mu'``), the factorisation runs on an
MI355X through :class:`~muscle_synergies_amd.hip_nmf.HipNMF`; any other solver / loss keeps going to
``sklearn.decomposition.NMF`` exactly as in the reference (whose default is ``solver='cd'``).

Plotting, filtering and the Vicon loader of the reference are out of scope and not reproduced here.
"""

from __future__ import annotations

import warnings
from collections import OrderedDict
from dataclasses import dataclass
from typing import Any, Mapping, Optional, Union

import numpy as np
import pandas

from . import _lib
from .hip_nmf import HipNMF


@dataclass
class SynergyRunResult:
    """Result of one or several factorisations (``analysis.py:670-710``).

    ``vaf_values``: one row per number of components (index = that number when a range was requested);
    ``components``: DataFrame (k x muscles) or ``{k: DataFrame}``; ``model``: the fitted estimator
    (:class:`HipNMF` or ``sklearn.decomposition.NMF``) or ``{k: estimator}``.
    """

    vaf_values: pandas.DataFrame
    components: Union[pandas.DataFrame, Mapping[int, pandas.DataFrame]]
    model: Union[Any, Mapping[int, Any]]


def vaf(original_df: pandas.DataFrame, transformed_signal=None, components=None,
        reconstructed_signal=None) -> pandas.DataFrame:
    """Uncentered variance accounted for, ``1 - ||x - x_r||^2 / ||x||^2`` (``analysis.py:597-667``).

    Returns a one-row DataFrame whose first column ``"All signals"`` is the VAF over every entry and
    whose remaining columns (labelled like ``original_df``) are the per-muscle VAFs.
    """
    if reconstructed_signal is None:
        reconstructed_signal = np.asarray(transformed_signal) @ np.asarray(components)
    x = original_df.to_numpy()
    sq_err = (x - np.asarray(reconstructed_signal)) ** 2
    sq_x = x ** 2
    overall = 1 - np.sum(sq_err, axis=(0, 1)) / np.sum(sq_x, axis=(0, 1))
    per_column = 1 - np.sum(sq_err, axis=0) / np.sum(sq_x, axis=0)
    labels = ["All signals"] + original_df.columns.tolist()
    values = [overall] + list(per_column.reshape(-1))
    return pandas.DataFrame({label: [value] for label, value in zip(labels, values)})


def _make_model(n_components: int, n_features: Optional[int] = None, **nmf_kwargs):
    """The seam: HIP engine for the mu path at the shapes it is compiled for, sklearn for everything else.

    ``solver='mu'`` runs on the GPU for up to ``HipNMF.MAX_FEATURES`` (512) muscles and ``HipNMF.MAX_COMPONENTS`` (64) synergies
    (lane mappings up to 32 x 8, matrix-pipe kernels up to 128 x 32, the general-shape kernels beyond); a call outside that goes
    to scikit-learn like any other unsupported configuration (the reference works for every shape, ``analysis.py:862-863``), with
    a warning saying so.
    """
    if HipNMF.supports(n_features=n_features, n_components=n_components, **nmf_kwargs):
        return HipNMF(n_components=n_components, **nmf_kwargs)
    from sklearn.decomposition import NMF  # the reference's own behaviour (default solver 'cd')

    if HipNMF.supports(**nmf_kwargs):
        warnings.warn(
            f"solver='mu' with {n_features} features / {n_components} components is outside the HIP engine's "
            f"compiled shapes (<= {HipNMF.MAX_FEATURES} features, <= {HipNMF.MAX_COMPONENTS} components): "
            "running scikit-learn on the CPU instead", RuntimeWarning, stacklevel=3)
    nmf_kwargs.pop("device", None)
    return NMF(n_components=n_components, **nmf_kwargs)


def _check_component_range(df: pandas.DataFrame, n_components: int, max_components: Optional[int]):
    """``validate_num_components`` (``analysis.py:829-846``)."""
    if df.empty:
        raise ValueError("empty EMG DataFrame")
    n_muscles = len(df.columns)
    message = "invalid number of components"
    if not 1 <= n_components <= n_muscles:
        raise ValueError(message)
    if max_components is not None and not n_components <= max_components <= n_muscles:
        raise ValueError(message)


_RANK_POOL = None
_RANK_POOL_PID = None


def _rank_pool():
    """The worker threads of the rank ranges: created on first use in THIS process and kept, so that each keeps its engine
    handle (stream, events, workspace: ``_lib.get_handle`` caches per thread) from call to call.  A forked child starts its
    own (the parent's threads do not exist there; ``submit`` on the inherited object would wait for ever)."""
    global _RANK_POOL, _RANK_POOL_PID
    import os

    if _RANK_POOL is None or _RANK_POOL_PID != os.getpid():
        from concurrent.futures import ThreadPoolExecutor

        _RANK_POOL = ThreadPoolExecutor(max_workers=8, thread_name_prefix="hipnmf-rank")
        _RANK_POOL_PID = os.getpid()
        if _shutdown_rank_pool not in _lib._shutdown_hooks:
            _lib._shutdown_hooks.append(_shutdown_rank_pool)  # joined before the handles its workers cached are destroyed
    return _RANK_POOL


def _shutdown_rank_pool() -> None:
    global _RANK_POOL
    import os

    pool, _RANK_POOL = _RANK_POOL, None
    if pool is not None and _RANK_POOL_PID == os.getpid():
        pool.shutdown(wait=True, cancel_futures=True)


def _ranks_concurrently(n_ranks: int, nmf_kwargs, shape=None) -> bool:
    """Whether the solver calls of a rank range run from concurrent host threads, one engine handle each: whenever they run on
    the GPU (scikit-learn's solvers keep the reference's sequential loop).  Any frame size: the C ABI's handles are independent
    (include/hip_nmf.h; tests/test_gpu_abi_threads.py drives every solver path from 2, 3 and 8 threads).  Round 3 limited this to
    frames of at most 2 048 x 32 because concurrent calls on long or wide frames failed with "operation failed due to a previous
    error during capture"; the cause was stream capture itself (any legacy-stream call of another thread invalidates an open
    capture on this HIP runtime, profiles/r04_threads_root_cause.md) and the library no longer captures.  The initialisations
    are still computed in rank order on the calling thread, so a seeded global generator or a shared ``RandomState`` is consumed
    exactly as by the loop.  ``HIPNMF_RANK_THREADS=0`` keeps the loop."""
    import os

    if n_ranks < 2 or os.environ.get("HIPNMF_RANK_THREADS", "1") == "0":
        return False
    return nmf_kwargs.get("solver") == "mu" and HipNMF.supports(**nmf_kwargs)


def _single_run(df: pandas.DataFrame, n_components: int, **nmf_kwargs) -> SynergyRunResult:
    """One factorisation + its VAF row (``analysis.py:866-882``)."""
    model = _make_model(n_components, n_features=len(df.columns), **nmf_kwargs)
    transformed = model.fit_transform(df)
    vaf_row = vaf(df, transformed_signal=transformed, components=model.components_)
    comps = pandas.DataFrame(model.components_, columns=df.columns)
    return SynergyRunResult(vaf_row, comps, model)


def find_synergies(processed_emg_df: pandas.DataFrame, n_components: int, max_components: Optional[int] = None, *,
                   max_iter: int = 100_000, tol: float = 1e-6, **sklearn_kwargs) -> SynergyRunResult:
    """Find spatial synergy components in a processed (non-negative) EMG DataFrame.

    Same contract as the reference (``analysis.py:713-914``): ``processed_emg_df`` is
    ``(num_measurements, num_muscles)``; with ``max_components=None`` one factorisation with exactly
    ``n_components`` synergies is returned, otherwise one per rank in
    ``n_components..max_components`` with dict-valued ``components`` / ``model`` and a VAF table indexed
    by rank.  ``max_iter``, ``tol`` and ``**sklearn_kwargs`` are forwarded to the estimator.

    Pass ``solver='mu'`` to run on the GPU (``device='cuda:1'`` etc. selects one); without it the call
    behaves exactly like the reference and uses sklearn's coordinate-descent solver on the CPU.

    Raises:
        ValueError: ``"empty EMG DataFrame"``, ``"invalid number of components"`` (need
            ``num_muscles >= max_components >= n_components >= 1``), or sklearn's own message for
            negative input.
    """
    _check_component_range(processed_emg_df, n_components, max_components)
    if max_components is None:
        return _single_run(processed_emg_df, n_components, max_iter=max_iter, tol=tol, **sklearn_kwargs)

    ranks = list(range(n_components, max_components + 1))
    runs = OrderedDict()
    if _ranks_concurrently(len(ranks), sklearn_kwargs, processed_emg_df.shape):
        # The fits of a rank range are independent.  Host preparation (validation, initialisation -- the only consumer of
        # random_state / NumPy's global generator) runs here, rank by rank as in the loop below; then every rank's solver call
        # goes to a worker thread, each with its own handle and stream (_lib.get_handle is per thread, the C call releases the
        # GIL), so the ranks' kernels overlap instead of queueing: the call takes the time of its longest fit, not the sum.
        models = [_make_model(rank, n_features=len(processed_emg_df.columns), max_iter=max_iter, tol=tol, **sklearn_kwargs) for rank in ranks]
        if all(isinstance(mdl, HipNMF) for mdl in models):
            prepared = [mdl._prepare(processed_emg_df) for mdl in models]

            def fit(mdl, prep):
                transformed = mdl._fit_prepared(*prep)
                vaf_row = vaf(processed_emg_df, transformed_signal=transformed, components=mdl.components_)
                return SynergyRunResult(vaf_row, pandas.DataFrame(mdl.components_, columns=processed_emg_df.columns), mdl)

            futures = [_rank_pool().submit(fit, mdl, prep) for mdl, prep in zip(models, prepared)]
            for rank, fut in zip(ranks, futures):
                runs[rank] = fut.result()
        else:
            for rank in ranks:
                runs[rank] = _single_run(processed_emg_df, rank, max_iter=max_iter, tol=tol, **sklearn_kwargs)
    else:
        for rank in ranks:
            runs[rank] = _single_run(processed_emg_df, rank, max_iter=max_iter, tol=tol, **sklearn_kwargs)

    table = pandas.concat([run.vaf_values for run in runs.values()])
    table.set_index(np.array(tuple(runs.keys())), inplace=True)
    return SynergyRunResult(
        table,
        {rank: run.components for rank, run in runs.items()},
        {rank: run.model for rank, run in runs.items()},
    )


def find_synergies_batched(processed_emg_dfs, n_components: int, max_components: Optional[int] = None, *,
                           max_iter: int = 100_000, tol: float = 1e-6, init=None, random_state=None,
                           alpha_W: float = 0.0, alpha_H="same", l1_ratio: float = 0.0, beta_loss="frobenius",
                           device=None, devices=None, _batched_init: Optional[bool] = None):
    """``find_synergies(df, ..., solver='mu')`` for a list of trials in one GPU launch per rank.

    ``devices=`` scatters the trials over several GPUs (BASELINE.json config #4; the loop this replaces is
    ``analysis.py:907-912`` over the trials of ``project/segment.py:160-207``): contiguous runs of trials balanced by
    their rows, one host thread and handle per device, no collective, the per-trial results in the order of the input.
    Every trial's fit is the same arithmetic as in the one-device call; it is the same BITS whenever the device's share of
    the batch takes the same kernel family as the whole batch would (the choice depends on the batch size: DISPATCH.md) --
    a share is routed on its own size on purpose, a GPU that receives two long trials must slice them over all its CUs.

    ``processed_emg_dfs`` is a sequence of DataFrames with the same muscles (columns) and any numbers of rows
    (e.g. the gait cycles cut by ``project/segment.py``).  Returns one :class:`SynergyRunResult` per trial,
    each identical in structure -- and, trial by trial, in values -- to what ``find_synergies`` returns for
    that DataFrame alone; the factorisations of a rank run together through ``fit_ragged``.  When all trials
    have the same length and the init is of the NNDSVD family, the starting factors come from the batched
    on-device NNDSVD (exact SVD; sklearn's randomized SVD differs from it by its own approximation error).
    """
    from .engine import fit_ragged
    from .init import initialize_nmf

    dfs = list(processed_emg_dfs)
    if not dfs:
        raise ValueError("empty EMG DataFrame")
    if devices is not None:
        from .multi_gpu import resolve_devices, scatter

        for df in dfs:  # every trial is validated before any device starts (same errors as the one-device call)
            _check_component_range(df, n_components, max_components)
        same_len = len({len(df) for df in dfs}) == 1  # decides the init path: fixed for the whole call, not per slice
        kw = dict(max_iter=max_iter, tol=tol, init=init, random_state=random_state, alpha_W=alpha_W, alpha_H=alpha_H,
                  l1_ratio=l1_ratio, beta_loss=beta_loss)
        parts = scatter(len(dfs), resolve_devices(devices),
                        lambda lo, hi, d: find_synergies_batched(dfs[lo:hi], n_components, max_components, device=f"cuda:{d}",
                                                                 _batched_init=same_len, **kw),
                        weights=[len(df) for df in dfs])
        return [r for _, _, _, part in parts for r in part]
    columns = dfs[0].columns
    for df in dfs:
        _check_component_range(df, n_components, max_components)
        if len(df.columns) != len(columns) or not (df.columns == columns).all():
            raise ValueError("all trials must have the same muscle columns")
    ranks = [n_components] if max_components is None else list(range(n_components, max_components + 1))
    arrays = [HipNMF._validate_X(df) for df in dfs]
    dtype = np.float32 if all(a.dtype == np.float32 for a in arrays) else np.float64
    arrays = [np.asarray(a, dtype=dtype) for a in arrays]
    per_trial = [OrderedDict() for _ in dfs]
    for rank in ranks:
        template = HipNMF(rank, init=init, tol=tol, max_iter=max_iter, random_state=random_state, alpha_W=alpha_W,
                          alpha_H=alpha_H, l1_ratio=l1_ratio, beta_loss=beta_loss, device=device)
        template._check_params()
        equal_len = len({a.shape[0] for a in arrays}) == 1 if _batched_init is None else bool(_batched_init)
        if equal_len and init in (None, "nndsvd", "nndsvda") and rank <= min(arrays[0].shape):
            # one batched on-device NNDSVD (exact Gram-matrix SVD) instead of one randomized SVD per trial
            from .init import nndsvd_init_batched

            W0d, H0d = nndsvd_init_batched(np.stack(arrays), rank, init=init or "nndsvda", device=device)
            inits = [(W0d[b], H0d[b]) for b in range(len(arrays))]
        else:
            inits = [initialize_nmf(a, rank, init=init, random_state=random_state) for a in arrays]
        # regularisation scales with each trial's own shape (_nmf.py:1254-1265): only equal-length trials share it
        regs = {template._regularization(a.shape[0], a.shape[1]) for a in arrays}
        if len(regs) > 1:
            raise NotImplementedError("alpha_W / alpha_H with trials of different lengths: fit them per trial")
        l1w, l1h, l2w, l2h = regs.pop()
        res = fit_ragged(arrays, [w for w, _ in inits], [h for _, h in inits], max_iter=max_iter, tol=tol,
                         l1_reg_W=l1w, l1_reg_H=l1h, l2_reg_W=l2w, l2_reg_H=l2h, beta_loss=beta_loss, device=device)
        H = res.H.cpu().numpy()
        n_iter = res.n_iter.cpu().numpy()
        err = res.reconstruction_err.cpu().numpy()
        for b, df in enumerate(dfs):
            model = HipNMF(rank, init=init, tol=tol, max_iter=max_iter, random_state=random_state, alpha_W=alpha_W,
                           alpha_H=alpha_H, l1_ratio=l1_ratio, beta_loss=beta_loss, device=device)
            model.components_, model.n_components_, model.n_features_in_ = H[b], rank, H.shape[2]
            model.n_iter_, model.reconstruction_err_ = int(n_iter[b]), err[b]
            W = res.W[b].cpu().numpy()
            vaf_row = vaf(df, transformed_signal=W, components=H[b])
            per_trial[b][rank] = SynergyRunResult(vaf_row, pandas.DataFrame(H[b], columns=df.columns), model)
    if max_components is None:
        return [runs[n_components] for runs in per_trial]
    out = []
    for runs in per_trial:
        table = pandas.concat([run.vaf_values for run in runs.values()])
        table.set_index(np.array(tuple(runs.keys())), inplace=True)
        out.append(SynergyRunResult(table, {r: run.components for r, run in runs.items()},
                                    {r: run.model for r, run in runs.items()}))
    return out

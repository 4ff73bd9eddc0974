"""Scatter of independent trials over the GPUs of one process (BASELINE.json configs #3 / #4).

The reference fits one trial after another on the host (``src/muscle_synergies/analysis.py:907-912`` over the
trials cut by ``project/segment.py:160-207``); every factorisation -- and every envelope -- is independent, so a
batch shards by trial with **no collective**: contiguous slices of the batch index, one host thread and one
library handle per device (``include/hip_nmf.h``: distinct handles may be driven concurrently; ctypes releases
the GIL during the call), results gathered on the host in batch order.  Every batched entry point of the package
takes ``devices=`` and goes through :func:`scatter`; ``devices=None`` is the ordinary single-device call.

The same device may be named more than once (``devices=[0, 0]``): two threads, two handles, one GPU -- that is
how the scatter/gather logic is exercised on a one-GPU box.
"""
from __future__ import annotations

import threading
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib


def resolve_devices(devices) -> Optional[List[int]]:
    """``None`` -> ``None`` (single-device call); ``"all"`` -> every visible GPU; otherwise a sequence of device
    indices / ``"cuda:i"`` strings / ``torch.device`` objects (repeats allowed).  Fails loudly without a GPU."""
    if devices is None:
        return None
    import torch

    n = torch.cuda.device_count()
    if n < 1:
        raise _lib.HipNmfError(_lib.HIPNMF_ERR_NO_DEVICE, "no ROCm GPU visible; the HIP NMF engine has no CPU fallback")
    if isinstance(devices, str) and devices == "all":
        return list(range(n))
    if isinstance(devices, (int, str)) or not hasattr(devices, "__iter__"):
        devices = [devices]
    out = []
    for d in devices:
        if isinstance(d, (int, np.integer)):
            idx = int(d)
        else:
            dev = torch.device(d)
            if dev.type != "cuda":
                raise ValueError(f"devices must be GPUs (got {dev})")
            idx = torch.cuda.current_device() if dev.index is None else dev.index
        if not 0 <= idx < n:
            raise ValueError(f"device index {idx} out of range: {n} GPU(s) visible")
        out.append(idx)
    if not out:
        raise ValueError("devices must name at least one GPU")
    return out


def partition_weighted(weights: Sequence[float], n_parts: int) -> List[Tuple[int, int]]:
    """Contiguous ``[lo, hi)`` ranges over ``len(weights)`` items, balanced by cumulative weight (ragged batches: rows
    per trial): an item belongs to the part its MIDPOINT on the cumulative-weight axis falls into, so no part exceeds
    its fair share by more than one item.  Every item lands in exactly one part; parts may be empty."""
    w = np.asarray(weights, dtype=np.float64)
    if w.ndim != 1 or (w < 0).any():
        raise ValueError("weights must be a 1-d sequence of non-negative numbers")
    n, n_parts = len(w), int(n_parts)
    if n_parts < 1:
        raise ValueError("n_parts must be >= 1")
    if float(w.sum()) <= 0:  # all-zero weights: by count
        w = np.ones(n)
    share = float(w.sum()) / n_parts
    mid = np.cumsum(w) - 0.5 * w
    part = np.minimum((mid / share).astype(np.int64), n_parts - 1) if n else np.zeros(0, dtype=np.int64)
    out, lo = [], 0
    for p in range(n_parts):
        hi = int(np.searchsorted(part, p, side="right"))  # `part` is non-decreasing
        out.append((lo, hi))
        lo = hi
    return out


def partition(n_items: int, n_parts: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced ``[lo, hi)`` ranges (first ``n_items % n_parts`` parts get one extra)."""
    base, extra = divmod(int(n_items), int(n_parts))
    out, lo = [], 0
    for i in range(n_parts):
        hi = lo + base + (1 if i < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def scatter(n_items: int, devices: Sequence[int], work: Callable[[int, int, int], object], *, weights=None,
            _bind_device: bool = True) -> List[Tuple[int, int, int, object]]:
    """Run ``work(lo, hi, device_index)`` for contiguous slices of ``range(n_items)``, one host thread per entry of
    ``devices``; returns ``[(lo, hi, device_index, result), ...]`` for the non-empty slices, in batch order.  The first
    exception of any worker is re-raised after all threads have ended.  ``weights``: per-item cost (ragged batches).
    ``_bind_device=False`` (tests): do not touch torch / the library, ``work`` is a stand-in."""
    devices = list(devices)
    bounds = partition(n_items, len(devices)) if weights is None else partition_weighted(weights, len(devices))
    results: list = [None] * len(devices)
    errors: list = [None] * len(devices)

    def run(i):
        lo, hi = bounds[i]
        if hi <= lo:
            return
        try:
            if _bind_device:
                import torch

                torch.cuda.set_device(devices[i])  # the thread's current device (allocations of helpers that take none)
            results[i] = work(lo, hi, devices[i])
        except BaseException as e:  # noqa: BLE001 -- re-raised by the caller's thread
            errors[i] = e
        finally:
            if _bind_device:
                _lib.release_thread_handles()  # the worker's handles (and their workspaces) end with it

    threads = [threading.Thread(target=run, args=(i,), name=f"hipnmf-scatter-{i}") for i in range(len(devices))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    return [(bounds[i][0], bounds[i][1], devices[i], results[i]) for i in range(len(devices)) if bounds[i][1] > bounds[i][0]]


def to_host(x, as_numpy: bool):
    """A worker's output tensor on the host: NumPy array or CPU torch tensor."""
    if isinstance(x, np.ndarray):
        if as_numpy:
            return x
        import torch

        return torch.from_numpy(x)
    t = x.detach().cpu()
    return t.numpy() if as_numpy else t


def cat_host(parts: list):
    """Concatenate host arrays / CPU tensors along the batch axis."""
    if isinstance(parts[0], np.ndarray):
        return np.concatenate(parts, axis=0)
    import torch

    return torch.cat(parts, dim=0)

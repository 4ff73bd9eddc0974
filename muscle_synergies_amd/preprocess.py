"""GPU EMG-envelope preprocessing: the step that *produces* the matrix handed to ``find_synergies``.

Mirrors, for the tutorial pipeline of the reference (``docs/source/tutorials/Finding muscle synergies.ipynb``),
the DataFrame functions ``zero_center`` (``src/muscle_synergies/analysis.py:230-249``), ``rms`` (``:435-507``),
``time_normalize`` (``:551-594``, linear interpolation only), ``normalize`` (``:510-525``),
``digital_filter`` (``:314-432``) and ``linear_envelope`` (``:252-311``) -- same signatures and return types
-- and adds batched entry points that run a whole chain for many recordings in one call and leave the
result on the device in the NMF engine's native layout.

All sample arithmetic happens in ``libhip_nmf.so`` (``hipnmf_emg_envelope_*``, ``hipnmf_sosfilt_*``); there is
no CPU fallback.  Filter *design* (a handful of coefficients) stays on the host with ``scipy.signal``, exactly
where the reference does it (``analysis.py:381-403``).  ``subsample`` and plotting are not part of this module.
"""

from __future__ import annotations

import os

import ctypes
from typing import Optional, Union

import numpy as np
import pandas

from . import _lib
from .engine import _as_device_tensor, _torch, _x_layout, resolve_device


class EnvelopeParams(ctypes.Structure):
    """Mirror of ``struct hipnmf_envelope_params`` (include/hip_nmf.h)."""

    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("batch", ctypes.c_int32),
        ("n_samples", ctypes.c_int64),
        ("n_channels", ctypes.c_int32),
        ("x_layout", ctypes.c_int32),
        ("ldx", ctypes.c_int64),
        ("x_batch_stride", ctypes.c_int64),
        ("window", ctypes.c_int32),
        ("zero_center", ctypes.c_int32),
        ("n_out", ctypes.c_int32),
        ("normalize", ctypes.c_int32),
        ("resample_kind", ctypes.c_int32),
        ("reserved0", ctypes.c_int32),
    ]


#: ``interp1d`` kinds evaluated on the device (``HIPNMF_RESAMPLE_*`` of include/hip_nmf.h).  ``'zero'`` / ``0`` (the
#: order-0 spline) takes the last knot <= the abscissa like ``'previous'``; ``'slinear'`` / ``1`` (the order-1 spline)
#: is the linear interpolant.  ``'quadratic'`` / ``'cubic'`` (orders 2, 3) couple all samples of a channel through a
#: banded solve and are evaluated by scipy itself on the host.
RESAMPLE_KINDS = {"linear": 0, "slinear": 0, 1: 0, "nearest": 1, "nearest-up": 2, "previous": 3, "zero": 3, 0: 3, "next": 4}
SPLINE_KINDS = ("quadratic", "cubic", 2, 3)


class SosfiltParams(ctypes.Structure):
    """Mirror of ``struct hipnmf_sosfilt_params`` (include/hip_nmf.h)."""

    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("batch", ctypes.c_int32),
        ("n_samples", ctypes.c_int64),
        ("n_channels", ctypes.c_int32),
        ("x_layout", ctypes.c_int32),
        ("ldx", ctypes.c_int64),
        ("x_batch_stride", ctypes.c_int64),
        ("n_sections", ctypes.c_int32),
        ("zero_lag", ctypes.c_int32),
        ("padlen", ctypes.c_int32),
        ("zero_center", ctypes.c_int32),
        ("rectify", ctypes.c_int32),
        ("mode", ctypes.c_int32),
    ]


def window_in_samples(window_size: Union[int, float], sampling_frequency: Optional[int]) -> int:
    """``round(window_size * sampling_frequency)`` when a rate is given (``analysis.py:493-499``)."""
    if sampling_frequency is not None:
        return round(window_size * sampling_frequency)
    return int(window_size)


def emg_envelope_batched(raw, window_size: Union[int, float] = 0, *, sampling_frequency: Optional[int] = None,
                         zero_center: bool = True, reduce_to: Optional[int] = None, normalize: bool = True,
                         device=None, kind="linear", devices=None):
    """``zero_center -> rms -> time_normalize -> normalize`` for a batch of recordings on one GPU (``devices=``: the
    recordings scattered over several, the envelopes gathered on the host as a CPU tensor ``[B, T_out, m]``).

    Args:
        raw: ``[B, T, m]`` (or ``[T, m]``) float32/float64, NumPy or torch, any dense layout.
        window_size: RMS window (samples, or seconds when ``sampling_frequency`` is given); 0 skips the RMS.
        reduce_to: number of rows after time normalisation (``None`` keeps ``T``).
        kind: ``interp1d`` kind of the time normalisation, one of :data:`RESAMPLE_KINDS`.
    Returns:
        tensor ``[B, T_out, m]`` on the device (a transposed view of channel-major storage, which
        ``fit_batched`` streams without any copy).
    """
    torch = _torch()
    if devices is not None:
        return _scatter_recordings(raw, devices, lambda part, d: emg_envelope_batched(
            part, window_size, sampling_frequency=sampling_frequency, zero_center=zero_center, reduce_to=reduce_to,
            normalize=normalize, device=f"cuda:{d}", kind=kind))
    dev = resolve_device(device)
    Xt = _as_device_tensor(raw, dev)
    if Xt.dim() == 2:
        Xt = Xt.unsqueeze(0)
    if Xt.dim() != 3:
        raise ValueError(f"raw must be [B, T, m] or [T, m], got shape {tuple(Xt.shape)}")
    if Xt.dtype not in (torch.float32, torch.float64):
        Xt = Xt.to(torch.float64)
    B, T, m = Xt.shape
    if B == 0 or T == 0 or m == 0:
        raise ValueError("empty input")
    layout, ldx, xbs, Xt = _x_layout(Xt)
    W = window_in_samples(window_size, sampling_frequency)
    if W < 0:
        raise ValueError("window_size must be >= 0")
    n_out = int(reduce_to) if reduce_to else 0
    if kind not in RESAMPLE_KINDS or isinstance(kind, bool):
        raise NotImplementedError(f"kind={kind!r}: the device evaluates {sorted(map(str, RESAMPLE_KINDS))}")
    p = EnvelopeParams(ctypes.sizeof(EnvelopeParams), B, T, m, layout, ldx, xbs, W, int(bool(zero_center)), n_out,
                       int(bool(normalize)), RESAMPLE_KINDS[kind], 0)
    out = torch.empty((B, m, n_out if n_out else T), dtype=Xt.dtype, device=dev)
    h = _lib.get_handle(dev.index)
    lib = _lib.load()
    fn = lib.hipnmf_emg_envelope_f32 if Xt.dtype == torch.float32 else lib.hipnmf_emg_envelope_f64
    torch.cuda.synchronize(dev)
    _lib.check(fn(h.ptr, ctypes.byref(p), ctypes.c_void_p(Xt.data_ptr()), ctypes.c_void_p(out.data_ptr())))
    return out.transpose(1, 2)


def _scatter_recordings(raw, devices, one):
    """Recordings ``[B, T, m]`` in contiguous slices over ``devices`` (no collective: every recording is independent);
    ``one(slice, device_index)`` returns the slice's result on its device; gathered as one CPU tensor in batch order."""
    from .multi_gpu import cat_host, resolve_devices, scatter, to_host

    torch = _torch()
    Xa = raw if isinstance(raw, torch.Tensor) else np.asarray(raw)
    if Xa.ndim == 2:
        Xa = Xa[None]
    if Xa.ndim != 3 or Xa.shape[0] == 0:
        raise ValueError(f"need [B, T, m] or [T, m] with B >= 1, got shape {tuple(Xa.shape)}")
    parts = scatter(Xa.shape[0], resolve_devices(devices), lambda lo, hi, d: one(Xa[lo:hi], d))
    return cat_host([to_host(r, False) for _, _, _, r in parts])


# ------------------------------------------------------------------------------------------------
# Spline kinds of time_normalize ('quadratic' / 'cubic' = interp1d -> make_interp_spline): a banded operator built once per
# (T, n_out, degree) on the host from scipy's own design matrices, applied to every channel of every recording by
# hipnmf_resample_weights_* (include/hip_nmf.h says why that is the same arithmetic up to rounding).
_SPLINE_TOL = 1e-18  # weights below this fraction of a row's largest are dropped (cubic: the influence decays 0.268 per sample)
_spline_cache: dict = {}
_spline_dev_cache: dict = {}


def _spline_operator(T: int, n_out: int, k: int):
    """``(first [n_out] int32, weights [n_out, taps] float64)`` with ``interp1d(linspace(0, 1, T), y, kind=k)(linspace(0, 1, n_out))
    == sum_i weights[r, i] * y[first[r] + i]``: the rows of ``E A^-1`` (A: collocation matrix of the degree-k B-splines on scipy's
    not-a-knot knots at the samples, ``scipy/interpolate/_bsplines.py:1008-1027, 1363-1580``; E: the same splines at the new axis),
    each cut to the window that holds everything above ``_SPLINE_TOL`` of its largest weight."""
    key = (int(T), int(n_out), int(k))
    hit = _spline_cache.get(key)
    if hit is not None:
        return hit
    from scipy.interpolate import BSpline
    from scipy.sparse.linalg import splu

    if T <= k:
        raise ValueError(f"a spline of degree {k} needs more than {k} samples (got {T})")  # scipy: "Got n knots, need at least ..."
    xs, xn = np.linspace(0, 1, T), np.linspace(0, 1, n_out)
    if k % 2 == 1:
        k2, t = (k + 1) // 2, xs.copy()
    else:
        k2, t = k // 2, (xs[1:] + xs[:-1]) / 2
    t = np.r_[(xs[0],) * (k + 1), t[k2:-k2], (xs[-1],) * (k + 1)]
    A = BSpline.design_matrix(xs, t, k).tocsc()
    lu = splu(A.T.tocsc())  # R = E A^-1  <=>  A^T R^T = E^T
    first = np.empty(n_out, np.int32)
    rows = []
    for lo in range(0, n_out, 256):  # blocks of output rows: the dense right-hand side stays small whatever n_out is
        E = BSpline.design_matrix(xn[lo:lo + 256], t, k)
        R = lu.solve(np.ascontiguousarray(E.T.toarray())).T  # [block, T]
        for r in range(R.shape[0]):
            sig = np.flatnonzero(np.abs(R[r]) > _SPLINE_TOL * np.abs(R[r]).max())
            first[lo + r] = sig[0]
            rows.append(R[r, sig[0]: sig[-1] + 1])
    taps = min(T, max(len(w) for w in rows))
    weights = np.zeros((n_out, taps))
    for r, w in enumerate(rows):
        f = min(int(first[r]), T - taps)  # the window slides left at the end of the series so that it stays inside
        weights[r, int(first[r]) - f: int(first[r]) - f + len(w)] = w
        first[r] = f
    if len(_spline_cache) > 16:
        _spline_cache.clear()
    _spline_cache[key] = (first, weights)
    return first, weights


def time_normalize_batched(X, reduce_to: int, kind="linear", *, device=None):
    """``time_normalize`` for a batch ``[B, T, m]`` (or ``[T, m]``), any ``interp1d`` kind of the reference on the device: the index
    kinds and ``linear`` through the envelope entry point, ``'quadratic'`` / ``'cubic'`` (2, 3) through the banded spline operator.
    Returns ``[B, reduce_to, m]`` on the device (a transposed view of channel-major storage)."""
    if kind in SPLINE_KINDS and not isinstance(kind, bool):
        return _spline_resample(X, int(reduce_to), 2 if kind in ("quadratic", 2) else 3, device)
    return emg_envelope_batched(X, 0, reduce_to=reduce_to, normalize=False, zero_center=False, kind=kind, device=device)


def _spline_resample(X, n_out: int, k: int, device=None):
    import torch

    dev = resolve_device(device)
    Xt = _as_device_tensor(X, dev)
    if Xt.dim() == 2:
        Xt = Xt.unsqueeze(0)
    if Xt.dim() != 3 or Xt.dtype not in (torch.float32, torch.float64):
        raise ValueError("time_normalize_batched takes [B, T, m] (or [T, m]) float32 / float64")
    B, T, m = Xt.shape
    layout, ldx, xbs, Xt = _x_layout(Xt)
    dkey = (dev.index, T, n_out, k)
    ops = _spline_dev_cache.get(dkey)
    if ops is None:
        first, weights = _spline_operator(T, n_out, k)
        ops = (torch.from_numpy(first).to(dev), torch.from_numpy(weights).to(dev), weights.shape[1])
        if len(_spline_dev_cache) > 16:
            _spline_dev_cache.clear()
        _spline_dev_cache[dkey] = ops
    p = EnvelopeParams(ctypes.sizeof(EnvelopeParams), B, T, m, layout, ldx, xbs, 0, 0, n_out, 0, 0, 0)
    out = torch.empty((B, m, n_out), dtype=Xt.dtype, device=dev)
    lib = _lib.load()
    fn = lib.hipnmf_resample_weights_f32 if Xt.dtype == torch.float32 else lib.hipnmf_resample_weights_f64
    torch.cuda.synchronize(dev)
    _lib.check(fn(_lib.get_handle(dev.index).ptr, ctypes.byref(p), Xt.data_ptr(), ops[0].data_ptr(), ops[1].data_ptr(), ops[2], out.data_ptr()))
    return out.transpose(1, 2)


def _frame_through_gpu(signal_df: pandas.DataFrame, **kw) -> np.ndarray:
    arr = signal_df.to_numpy()
    if arr.dtype != np.float32:
        arr = arr.astype(np.float64, copy=False)
    return emg_envelope_batched(arr, **kw)[0].cpu().numpy()


def _recreate(signal_df: pandas.DataFrame, inplace: bool, values: np.ndarray) -> pandas.DataFrame:
    if inplace:
        signal_df[:] = values
        return signal_df
    return pandas.DataFrame(values, index=signal_df.index, columns=signal_df.columns)


def zero_center(signal_df: pandas.DataFrame, inplace: bool = False) -> pandas.DataFrame:
    """Subtract the mean of each column from it (``analysis.py:230-249``)."""
    vals = _frame_through_gpu(signal_df, window_size=0, zero_center=True, normalize=False)
    return _recreate(signal_df, inplace, vals)


def rms(signal_df: pandas.DataFrame, window_size: Union[int, float], inplace: bool = False,
        sampling_frequency: Optional[int] = None) -> pandas.DataFrame:
    """Sliding-window RMS, stride 1, same shape as the input (``analysis.py:435-507``)."""
    vals = _frame_through_gpu(signal_df, window_size=window_size, sampling_frequency=sampling_frequency,
                              zero_center=False, normalize=False)
    return _recreate(signal_df, inplace, vals)


def normalize(signal_df: pandas.DataFrame, inplace: bool = False) -> pandas.DataFrame:
    """Divide each column by its max absolute value (``analysis.py:510-525``)."""
    vals = _frame_through_gpu(signal_df, window_size=0, zero_center=False, normalize=True)
    return _recreate(signal_df, inplace, vals)


def time_normalize(signal_df: pandas.DataFrame, reduce_to: int, kind="linear",
                   fill_value="extrapolate") -> pandas.DataFrame:
    """Resample to ``reduce_to`` rows on a 0..1 time axis (``analysis.py:551-594``: ``interp1d(linspace(0, 1, T), df,
    kind=kind)`` evaluated on ``linspace(0, 1, reduce_to)``).

    ``kind`` is forwarded like the reference does: ``'linear'`` (default), ``'slinear'``, ``'nearest'``,
    ``'nearest-up'``, ``'previous'``, ``'next'`` and ``'zero'`` run on the GPU (:data:`RESAMPLE_KINDS`);
    ``'quadratic'`` / ``'cubic'`` (or the spline orders 2, 3) too since round 6: the banded collocation solve behind them is
    linear in the samples and depends on the shape only, so it is built once per ``(T, reduce_to, kind)`` on the host from
    scipy's own design matrices (:func:`_spline_operator`) and applied on the device (``hipnmf_resample_weights_*``).  ``fill_value``
    is accepted for signature compatibility: both axes span exactly [0, 1], so nothing is ever extrapolated.
    """
    if kind in SPLINE_KINDS and not isinstance(kind, bool):
        arr = signal_df.to_numpy()
        if arr.dtype != np.float32:
            arr = arr.astype(np.float64, copy=False)
        vals = _spline_resample(arr, int(reduce_to), 2 if kind in ("quadratic", 2) else 3)[0].cpu().numpy()
        return pandas.DataFrame(vals, index=np.linspace(0, 1, reduce_to), columns=signal_df.columns)
    if kind not in RESAMPLE_KINDS or isinstance(kind, bool):
        raise NotImplementedError(f"interp1d kind {kind!r} is not understood")
    if signal_df.shape[0] < 2:
        raise ValueError("time_normalize needs at least two rows")
    vals = _frame_through_gpu(signal_df, window_size=0, zero_center=False, normalize=False, reduce_to=reduce_to, kind=kind)
    return pandas.DataFrame(vals, index=np.linspace(0, 1, reduce_to), columns=signal_df.columns)


# ------------------------------------------------------------------------------------------------
# IIR filters: digital_filter / linear_envelope
def design_sos(filter_type: str, order: int, sampling_frequency, critical_freqs, band_type: str = "lowpass",
               cheby_param: Optional[float] = None) -> np.ndarray:
    """Section coefficients as the reference designs them (``filter_coeffs``, ``analysis.py:381-403``)."""
    from scipy import signal

    if filter_type not in {"butter", "cheby1", "cheby2"}:
        raise ValueError("filter type not understood.")
    if filter_type == "butter":
        return signal.butter(order, critical_freqs, btype=band_type, output="sos", fs=sampling_frequency)
    coeff_func = signal.cheby1 if filter_type == "cheby1" else signal.cheby2
    return coeff_func(order, cheby_param, critical_freqs, btype=band_type, output="sos", fs=sampling_frequency)


SOSFILT_MODES = {"exact": 0, "scan": 1}  # HIPNMF_SOSFILT_EXACT / HIPNMF_SOSFILT_SCAN (include/hip_nmf.h)


def sosfilt_batched(x, sos, *, zero_lag: bool = True, zero_center: bool = False, rectify: bool = False,
                    padlen: Optional[int] = None, device=None, mode: str = "exact", devices=None):
    """``scipy.signal.sosfiltfilt(sos, x, axis=time)`` (``zero_lag``) or ``sosfilt`` for a batch of recordings.

    Args:
        x: ``[B, T, m]`` (or ``[T, m]``) float32/float64, NumPy or torch, any dense layout.
        sos: ``(n_sections, 6)`` second-order sections (``scipy.signal`` layout), at most 8 sections.
        zero_center, rectify: the two steps ``linear_envelope`` applies before its low-pass filter.
        mode: ``"exact"`` -- scipy's sequential recurrence, bit-identical in float64 (what the reference-facing
            ``digital_filter`` / ``linear_envelope`` use); ``"scan"`` -- the time-parallel kernel (every series cut into 256
            chunks filtered at once; agrees with scipy to rounding, about 1e-12 relative for the reference's 6 Hz low-pass,
            and runs about four times faster on a batch).
    Returns:
        tensor ``[B, T, m]`` on the device (transposed view of channel-major storage), dtype of ``x``; the
        arithmetic is fp64 either way.  ``devices=``: recordings scattered over several GPUs, a CPU tensor comes back.
    """
    torch = _torch()
    if devices is not None:
        return _scatter_recordings(x, devices, lambda part, d: sosfilt_batched(
            part, sos, zero_lag=zero_lag, zero_center=zero_center, rectify=rectify, padlen=padlen, device=f"cuda:{d}", mode=mode))
    dev = resolve_device(device)
    sos = np.ascontiguousarray(np.asarray(sos, dtype=np.float64))
    if sos.ndim != 2 or sos.shape[1] != 6:
        raise ValueError("sos array must be shape (n_sections, 6)")
    if not (sos[:, 3] == 1).all():
        raise ValueError("sos[:, 3] should be all ones")
    Xt = _as_device_tensor(x, dev)
    if Xt.dim() == 2:
        Xt = Xt.unsqueeze(0)
    if Xt.dim() != 3:
        raise ValueError(f"x must be [B, T, m] or [T, m], got shape {tuple(Xt.shape)}")
    if Xt.dtype not in (torch.float32, torch.float64):
        Xt = Xt.to(torch.float64)
    B, T, m = Xt.shape
    if B == 0 or T == 0 or m == 0:
        raise ValueError("empty input")
    zi = None
    if zero_lag:
        from scipy import signal

        ntaps = 2 * sos.shape[0] + 1 - min(int((sos[:, 2] == 0).sum()), int((sos[:, 5] == 0).sum()))
        edge = 3 * ntaps if padlen is None else int(padlen)
        if T <= edge:  # scipy's own message (_validate_pad)
            raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % edge)
        zi = np.ascontiguousarray(signal.sosfilt_zi(sos), dtype=np.float64)
    layout, ldx, xbs, Xt = _x_layout(Xt)
    p = SosfiltParams(ctypes.sizeof(SosfiltParams), B, T, m, layout, ldx, xbs, sos.shape[0], int(bool(zero_lag)),
                      -1 if padlen is None else int(padlen), int(bool(zero_center)), int(bool(rectify)), SOSFILT_MODES[mode])
    out = torch.empty((B, m, T), dtype=Xt.dtype, device=dev)
    h = _lib.get_handle(dev.index)
    lib = _lib.load()
    fn = lib.hipnmf_sosfilt_f32 if Xt.dtype == torch.float32 else lib.hipnmf_sosfilt_f64
    torch.cuda.synchronize(dev)
    _lib.check(fn(h.ptr, ctypes.byref(p), sos.ctypes.data_as(ctypes.c_void_p),
                  zi.ctypes.data_as(ctypes.c_void_p) if zi is not None else None,
                  ctypes.c_void_p(Xt.data_ptr()), ctypes.c_void_p(out.data_ptr())))
    return out.transpose(1, 2)


# Filter mode of the reference-facing single-frame functions (digital_filter / linear_envelope): "exact" reproduces scipy's
# sequential recurrence bit for bit; "scan" is the time-parallel kernel (equal to scipy to ~1e-12 relative, far inside the 1e-5
# the path is held to) -- one 20 000-sample frame: 2.5 ms -> 0.15 ms.  Set per call (``mode=``), per process
# (:func:`set_filter_mode`) or with HIPNMF_FILTER_MODE.
_FILTER_MODE = os.environ.get("HIPNMF_FILTER_MODE", "exact")


def set_filter_mode(mode: str) -> None:
    """Default filter mode of :func:`digital_filter` / :func:`linear_envelope`: ``"exact"`` (scipy's bits) or ``"scan"``."""
    global _FILTER_MODE
    if mode not in SOSFILT_MODES:
        raise KeyError(mode)
    _FILTER_MODE = mode


def _filter_frame(signal_df: pandas.DataFrame, sos, zero_lag: bool, inplace: bool, mode: Optional[str] = None, **pre) -> pandas.DataFrame:
    # scipy filters in float64 whatever the input dtype and the reference returns that (analysis.py:414-416)
    arr = signal_df.to_numpy().astype(np.float64, copy=False)
    vals = sosfilt_batched(arr, sos, zero_lag=zero_lag, mode=mode or _FILTER_MODE, **pre)[0].cpu().numpy()
    return _recreate(signal_df, inplace, vals)


def digital_filter(signal_df: pandas.DataFrame, critical_freqs, sampling_frequency: int, order: int,
                   filter_type: str = "butter", band_type: str = "lowpass", zero_lag: bool = True,
                   cheby_param: Optional[float] = None, inplace: bool = False, *, mode: Optional[str] = None) -> pandas.DataFrame:
    """Butterworth / Chebyshev I / II filter of any band type, forward-backward when ``zero_lag``
    (``analysis.py:314-432``); every column is filtered on the GPU.  ``mode`` (not in the reference): ``"exact"`` /
    ``"scan"``, default :func:`set_filter_mode`'s."""
    sos = design_sos(filter_type, order, sampling_frequency, critical_freqs, band_type, cheby_param)
    return _filter_frame(signal_df, sos, zero_lag, inplace, mode)


def linear_envelope(signal_df: pandas.DataFrame, critical_freqs, sampling_frequency: int, order: int,
                    filter_type: str = "butter", zero_lag: bool = True, cheby_param: Optional[float] = None,
                    zero_center_: bool = True, inplace: bool = False, *, mode: Optional[str] = None) -> pandas.DataFrame:
    """Linear envelope of raw EMG: (optional) zero-centring, rectification, low-pass filter
    (``analysis.py:252-311``) -- one fused GPU pass.  ``mode`` (not in the reference): ``"exact"`` / ``"scan"``."""
    sos = design_sos(filter_type, order, sampling_frequency, critical_freqs, "lowpass", cheby_param)
    return _filter_frame(signal_df, sos, zero_lag, inplace, mode, zero_center=zero_center_, rectify=True)


def linear_envelope_batched(raw, critical_freqs, sampling_frequency, order: int = 4, *, filter_type: str = "butter",
                            zero_lag: bool = True, cheby_param: Optional[float] = None, zero_center: bool = True,
                            reduce_to: Optional[int] = None, normalize: bool = True, device=None, mode: str = "scan", devices=None):
    """``linear_envelope -> time_normalize -> normalize`` for a batch of recordings ``[B, T, m]`` on one GPU
    (the filter-based alternative to :func:`emg_envelope_batched`).  Returns ``[B, T_out, m]`` on the device,
    channel-major underneath, ready for ``fit_batched``.  ``mode``: see :func:`sosfilt_batched`; the batched producer of X defaults
    to the time-parallel filter (the reference-facing single-frame functions keep scipy's bits)."""
    if devices is not None:  # the whole chain per slice on its device, one gather at the end
        return _scatter_recordings(raw, devices, lambda part, d: linear_envelope_batched(
            part, critical_freqs, sampling_frequency, order, filter_type=filter_type, zero_lag=zero_lag, cheby_param=cheby_param,
            zero_center=zero_center, reduce_to=reduce_to, normalize=normalize, device=f"cuda:{d}", mode=mode))
    sos = design_sos(filter_type, order, sampling_frequency, critical_freqs, "lowpass", cheby_param)
    env = sosfilt_batched(raw, sos, zero_lag=zero_lag, zero_center=zero_center, rectify=True, device=device, mode=mode)
    if reduce_to or normalize:
        env = emg_envelope_batched(env, 0, zero_center=False, reduce_to=reduce_to, normalize=normalize, device=device)
    return env

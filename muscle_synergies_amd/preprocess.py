"""GPU EMG-envelope preprocessing: the step that *produces* the matrix handed to ``find_synergies``.

Mirrors, for the tutorial pipeline of the reference (``docs/source/tutorials/Finding muscle synergies.ipynb``),
the DataFrame functions ``zero_center`` (``src/muscle_synergies/analysis.py:230-249``), ``rms`` (``:435-507``),
``time_normalize`` (``:551-594``, linear interpolation only) and ``normalize`` (``:510-525``) -- same
signatures and return types -- and adds a batched entry point that runs the whole chain for many
recordings in one call and leaves the result on the device in the NMF engine's native layout.

All arithmetic happens in ``libhip_nmf.so`` (``hipnmf_emg_envelope_*``); there is no CPU fallback.
Filters (``digital_filter`` / ``linear_envelope``), ``subsample`` and plotting are not part of this module.
"""

from __future__ import annotations

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
    ]


def window_in_samples(window_size: Union[int, float], sampling_frequency: Optional[int]) -> int:
    """``round(window_size * sampling_frequency)`` when a rate is given (``analysis.py:493-499``)."""
    if sampling_frequency is not None:
        return round(window_size * sampling_frequency)
    return int(window_size)


def emg_envelope_batched(raw, window_size: Union[int, float] = 0, *, sampling_frequency: Optional[int] = None,
                         zero_center: bool = True, reduce_to: Optional[int] = None, normalize: bool = True,
                         device=None):
    """``zero_center -> rms -> time_normalize -> normalize`` for a batch of recordings on one GPU.

    Args:
        raw: ``[B, T, m]`` (or ``[T, m]``) float32/float64, NumPy or torch, any dense layout.
        window_size: RMS window (samples, or seconds when ``sampling_frequency`` is given); 0 skips the RMS.
        reduce_to: number of rows after linear time normalisation (``None`` keeps ``T``).
    Returns:
        tensor ``[B, T_out, m]`` on the device (a transposed view of channel-major storage, which
        ``fit_batched`` streams without any copy).
    """
    torch = _torch()
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
    p = EnvelopeParams(ctypes.sizeof(EnvelopeParams), B, T, m, layout, ldx, xbs, W, int(bool(zero_center)), n_out,
                       int(bool(normalize)))
    out = torch.empty((B, m, n_out if n_out else T), dtype=Xt.dtype, device=dev)
    h = _lib.get_handle(dev.index)
    lib = _lib.load()
    fn = lib.hipnmf_emg_envelope_f32 if Xt.dtype == torch.float32 else lib.hipnmf_emg_envelope_f64
    torch.cuda.synchronize(dev)
    _lib.check(fn(h.ptr, ctypes.byref(p), ctypes.c_void_p(Xt.data_ptr()), ctypes.c_void_p(out.data_ptr())))
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
    """Resample to ``reduce_to`` rows on a 0..1 time axis (``analysis.py:551-594``); linear only."""
    if kind != "linear":
        raise NotImplementedError("the GPU time_normalize implements kind='linear' only")
    vals = _frame_through_gpu(signal_df, window_size=0, zero_center=False, normalize=False, reduce_to=reduce_to)
    return pandas.DataFrame(vals, index=np.linspace(0, 1, reduce_to), columns=signal_df.columns)

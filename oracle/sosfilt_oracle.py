"""NumPy restatement of the reference's IIR filtering stage (checker only).

TEST INFRASTRUCTURE ONLY -- never imported by ``muscle_synergies_amd``.

The reference's ``digital_filter`` (``src/muscle_synergies/analysis.py:314-432``) designs second-order
sections with ``scipy.signal.butter / cheby1 / cheby2(..., output="sos")`` (``:381-403``) and applies them
with ``scipy.signal.sosfiltfilt`` (``zero_lag=True``) or ``scipy.signal.sosfilt`` (``:405-417``) along axis 0;
``linear_envelope`` (``:252-311``) is ``zero_center -> abs -> digital_filter(band_type="lowpass")``.
The arithmetic therefore lives in SciPy, a third-party dependency that is not under ``/root/reference`` and
that the reference does not pin (it arrives through scikit-learn; 1.15.3 in this image).  This file restates
SciPy's published algorithm on plain ``(T, m)`` float64 arrays:

* ``sosfilt``      -- cascade of direct-form-II-transposed biquads, sample by sample
                      (``scipy/signal/_sosfilt.pyx::_sosfilt_float``),
* ``odd_ext``      -- ``scipy/signal/_arraytools.py::odd_ext``,
* ``sosfiltfilt``  -- ``scipy/signal/_signaltools.py::sosfiltfilt`` (odd padding of ``3 * ntaps`` samples,
                      steady-state initial conditions scaled by the first sample, forward then backward),
* ``sosfilt_zi``   -- ``_signaltools.py::sosfilt_zi`` / ``lfilter_zi`` for second-order sections.

Pinned by ``tests/golden/g8_filters.npz``: outputs of the reference's own ``digital_filter`` /
``linear_envelope`` captured by ``tests/golden/make_golden.py`` (bit-exact for the filter stage).
"""
import numpy as np


def sosfilt(sos, x, zi=None):
    """``scipy.signal.sosfilt(sos, x, axis=0, zi=zi)`` for ``x (T, m)``; ``zi (n_sections, 2, m)`` or None
    (zeros).  Returns ``(y, zf)``.  Every product and sum is rounded separately (no fused multiply-add)."""
    sos = np.asarray(sos, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    n_sections = sos.shape[0]
    T, m = x.shape
    z = np.zeros((n_sections, 2, m)) if zi is None else np.array(zi, dtype=np.float64, copy=True)
    y = np.empty_like(x)
    for n in range(T):
        x_cur = x[n].copy()
        for s in range(n_sections):
            b0, b1, b2, _, a1, a2 = sos[s]
            x_new = b0 * x_cur + z[s, 0]
            z[s, 0] = b1 * x_cur - a1 * x_new + z[s, 1]
            z[s, 1] = b2 * x_cur - a2 * x_new
            x_cur = x_new
        y[n] = x_cur
    return y, z


def lfilter_zi2(b, a):
    """``lfilter_zi`` for one normalised second-order section: solve ``(I - A^T) zi = B``."""
    IminusA = np.array([[1.0 + a[1], -1.0], [a[2], 1.0]])
    B = np.array([b[1] - a[1] * b[0], b[2] - a[2] * b[0]])
    return np.linalg.solve(IminusA, B)


def sosfilt_zi(sos):
    """``_signaltools.py::sosfilt_zi``: per-section steady-state step-response state, scaled by the DC gain of
    the sections before it."""
    sos = np.asarray(sos, dtype=np.float64)
    zi = np.empty((sos.shape[0], 2))
    scale = 1.0
    for s in range(sos.shape[0]):
        b, a = sos[s, :3], sos[s, 3:]
        zi[s] = scale * lfilter_zi2(b, a)
        scale *= b.sum() / a.sum()
    return zi


def default_padlen(sos):
    """``3 * ntaps`` with ``ntaps = 2 n_sections + 1 - min(#(b2 == 0), #(a2 == 0))`` (``sosfiltfilt``)."""
    sos = np.asarray(sos)
    ntaps = 2 * sos.shape[0] + 1
    ntaps -= min(int((sos[:, 2] == 0).sum()), int((sos[:, 5] == 0).sum()))
    return 3 * ntaps


def odd_ext(x, n):
    """``_arraytools.py::odd_ext`` along axis 0."""
    if n < 1:
        return x
    return np.concatenate((2 * x[:1] - x[n:0:-1], x, 2 * x[-1:] - x[-2:-(n + 2):-1]), axis=0)


def sosfiltfilt(sos, x, padlen=None):
    """``scipy.signal.sosfiltfilt(sos, x, axis=0)`` (``padtype='odd'``)."""
    x = np.asarray(x, dtype=np.float64)
    edge = default_padlen(sos) if padlen is None else int(padlen)
    if x.shape[0] <= edge:
        raise ValueError("The length of the input vector x must be greater than padlen, which is %d." % edge)
    ext = odd_ext(x, edge)
    zi = sosfilt_zi(sos)[:, :, None]
    y, _ = sosfilt(sos, ext, zi * ext[0])
    y, _ = sosfilt(sos, y[::-1], zi * y[-1])
    y = y[::-1]
    return y[edge:-edge] if edge > 0 else y


def digital_filter(x, sos, zero_lag=True):
    """``apply_filter`` of ``digital_filter`` (analysis.py:405-417) for given section coefficients."""
    return sosfiltfilt(sos, x) if zero_lag else sosfilt(sos, x)[0]


def linear_envelope(x, sos, zero_lag=True, zero_center=True):
    """analysis.py:296-311: optional zero-centring, full-wave rectification, low-pass filter."""
    x = np.asarray(x, dtype=np.float64)
    if zero_center:
        x = x - x.mean(axis=0)
    return digital_filter(np.abs(x), sos, zero_lag)

"""NumPy restatement of the reference's EMG-envelope preprocessing chain (checker only).

TEST INFRASTRUCTURE ONLY -- never imported by ``muscle_synergies_amd``.  Each function follows the
reference's DataFrame function in ``src/muscle_synergies/analysis.py`` on plain ``(T, m)`` arrays; pinned
by ``tests/golden/g6_envelope.npz`` (outputs of the real reference functions, captured by
``tests/golden/make_golden.py``).
"""
import numpy as np


def zero_center(x):
    """analysis.py:230-249: subtract each column's mean."""
    return x - x.mean(axis=0)


def rms(x, window):
    """analysis.py:474-491: per column ``sqrt(np.convolve(col**2, ones(W)/W, "same"))``."""
    w = (1 / float(window)) * np.ones(window)
    return np.stack([np.sqrt(np.convolve(x[:, j] ** 2, w, "same")) for j in range(x.shape[1])], axis=1)


def time_normalize(x, reduce_to, kind="linear"):
    """analysis.py:581-594: ``interp1d(linspace(0,1,T), x, kind=kind)`` evaluated on ``linspace(0,1,reduce_to)``.

    Restated per kind from scipy 1.15 ``interpolate/_interpolate.py``: ``linear`` (``_call_linear``; ``slinear`` is the
    same interpolant), ``nearest`` / ``nearest-up`` (``_call_nearest``: ``searchsorted`` of the abscissa among the
    midpoints ``x[i]/2 + x[i+1]/2``, side ``left`` -- a tie goes down -- resp. ``right``), ``previous`` / ``next``
    (``_call_previousnext``: knots shifted by one ulp so that an abscissa ON a knot takes that knot); ``zero`` (the
    order-0 spline) evaluates to ``previous``."""
    T = x.shape[0]
    xs = np.linspace(0, 1, T)
    xn = np.linspace(0, 1, reduce_to)
    if kind in ("linear", "slinear", 1):
        return np.stack([np.interp(xn, xs, x[:, j]) for j in range(x.shape[1])], axis=1)
    if kind in ("nearest", "nearest-up"):
        half = xs / 2.0
        idx = np.searchsorted(half[1:] + half[:-1], xn, side="left" if kind == "nearest" else "right")
    elif kind in ("previous", "zero", 0):
        idx = np.searchsorted(np.nextafter(xs, -np.inf), xn, side="left") - 1
    elif kind == "next":
        idx = np.searchsorted(np.nextafter(xs, np.inf), xn, side="right")
    else:
        raise ValueError(f"kind {kind!r} is not an index rule")
    return x[np.clip(idx, 0, T - 1).astype(np.intp)]


def normalize(x):
    """analysis.py:524-525: divide by the column's max absolute value."""
    return x / np.abs(x).max(axis=0)


def envelope(x, window, reduce_to=None, do_zero_center=True, do_normalize=True, kind="linear"):
    """The tutorial chain: zero_center -> rms -> time_normalize -> normalize."""
    y = zero_center(x) if do_zero_center else x
    if window:
        y = rms(y, window)
    if reduce_to:
        y = time_normalize(y, reduce_to, kind)
    if do_normalize:
        y = normalize(y)
    return y

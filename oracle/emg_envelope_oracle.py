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
    elif kind in ("quadratic", "cubic", 2, 3):
        return spline_interpolate(x, reduce_to, 2 if kind in ("quadratic", 2) else 3)
    else:
        raise ValueError(f"kind {kind!r} is not an index rule")
    return x[np.clip(idx, 0, T - 1).astype(np.intp)]


def not_a_knot(xs, k):
    """scipy 1.15 ``interpolate/_bsplines.py:1008-1027`` (``_not_a_knot``): the knot vector ``make_interp_spline`` uses for
    ``interp1d(kind='quadratic' / 'cubic')`` (``_interpolate.py:279, 397``): odd k -- the data sites without the k2 = (k + 1) / 2
    nearest to each end; even k -- the midpoints without the k / 2 nearest to each end; the end sites k + 1 times."""
    if k % 2 == 1:
        k2, t = (k + 1) // 2, np.array(xs, dtype=np.float64)
    else:
        k2, t = k // 2, (xs[1:] + xs[:-1]) / 2
    t = t[k2:-k2]
    return np.r_[(xs[0],) * (k + 1), t, (xs[-1],) * (k + 1)]


def bspline_basis(t, k, pts):
    """All B-splines of degree k on knots t at the points ``pts`` (Cox - de Boor recursion, de Boor IX(14); right-continuous,
    the last point of the domain belongs to the last interval): ``[len(pts), len(t) - k - 1]`` (dense; test sizes only)."""
    t, pts = np.asarray(t, np.float64), np.asarray(pts, np.float64)
    n = len(t) - k - 1
    last = np.searchsorted(t, t[-1], side="left") - 1  # the last non-empty interval
    B = np.zeros((len(pts), len(t) - 1))
    for i in range(len(t) - 1):
        if t[i] < t[i + 1]:
            B[:, i] = ((pts >= t[i]) & (pts < t[i + 1])) | ((i == last) & (pts == t[-1]))
    for d in range(1, k + 1):
        Bn = np.zeros((len(pts), len(t) - 1 - d))
        for i in range(len(t) - 1 - d):
            a = t[i + d] - t[i]
            b = t[i + d + 1] - t[i + 1]
            if a > 0:
                Bn[:, i] += (pts - t[i]) / a * B[:, i]
            if b > 0:
                Bn[:, i] += (t[i + d + 1] - pts) / b * B[:, i + 1]
        B = Bn
    return B[:, :n]


def spline_interpolate(x, reduce_to, k):
    """``make_interp_spline(xs, x, k)`` (collocation: the spline of degree k on the not-a-knot knots through every sample,
    ``_bsplines.py:1363-1580``) evaluated on the new axis."""
    T = x.shape[0]
    xs, xn = np.linspace(0, 1, T), np.linspace(0, 1, reduce_to)
    t = not_a_knot(xs, k)
    coef = np.linalg.solve(bspline_basis(t, k, xs), np.asarray(x, np.float64))
    return bspline_basis(t, k, xn) @ coef


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

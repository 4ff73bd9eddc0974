"""Segment-to-batch glue: turn the segments a user cuts out of one recording
into the ragged batch the engine factorises in one launch.

In the reference's project code, ``Segmenter.get_times_of(trecho, cycle, phase)`` (``project/segment.py:
160-207``) returns a ``slice`` of ``(frame, subframe)`` instants that is meant to index a
``DeviceData`` (``src/muscle_synergies/vicon_data/user_data.py:727-731``: ``dev_data[slice]`` ->
``dev_data.df.iloc[dev_data.to_index(slice)]``).  Neither file is modified or imported here: this module only
relies on that indexing protocol, so it takes

* a ``DeviceData``-like object (anything with ``.df`` and ``.to_index(slice)``) plus ``(frame, subframe)``
  slices, or
* a plain processed-EMG ``DataFrame`` plus ordinary row slices / ``(start, stop)`` pairs,

cuts the segments out (optionally after a processing function has been applied to the whole recording) and
hands them to :func:`muscle_synergies_amd.find_synergies_batched`.
"""

from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Union

import pandas

from .analysis import find_synergies_batched

SegmentRef = Union[slice, Sequence[int]]


def _row_slice(emg, segment: SegmentRef) -> slice:
    """Row range of one segment: ``DeviceData.to_index`` for ``(frame, subframe)`` slices, otherwise the slice /
    ``(start, stop)`` pair itself."""
    if isinstance(segment, slice):
        if hasattr(emg, "to_index") and (isinstance(segment.start, tuple) or isinstance(segment.stop, tuple)):
            return emg.to_index(segment)
        return segment
    start, stop = segment
    return slice(int(start), int(stop))


def segment_frames(emg, segments: Iterable[SegmentRef], processed: Optional[pandas.DataFrame] = None
                   ) -> List[pandas.DataFrame]:
    """The segments of one recording as a list of DataFrames (views of equal columns, unequal lengths).

    Args:
        emg: a ``DeviceData``-like object (``.df``, ``.to_index``) or a DataFrame.
        segments: ``Segmenter.get_times_of(...)`` results (slices of ``(frame, subframe)`` pairs), row slices, or
            ``(start_row, stop_row)`` pairs.
        processed: rows are taken from this frame instead of ``emg.df`` (e.g. the envelope of the whole
            recording, same number of rows), while the index conversion still goes through ``emg``.
    """
    df = processed if processed is not None else (emg if isinstance(emg, pandas.DataFrame) else emg.df)
    base_rows = len(emg) if isinstance(emg, pandas.DataFrame) else len(emg.df)
    if len(df) != base_rows:
        raise ValueError(f"processed frame has {len(df)} rows, the recording {base_rows}")
    out = []
    for seg in segments:
        rows = _row_slice(emg, seg)
        part = df.iloc[rows]
        if len(part) == 0:
            raise ValueError(f"segment {seg!r} selects no rows")
        out.append(part)
    if not out:
        raise ValueError("no segments given")
    return out


def find_synergies_segments(emg, segments: Iterable[SegmentRef], n_components: int,
                            max_components: Optional[int] = None, *,
                            process: Optional[Callable[[pandas.DataFrame], pandas.DataFrame]] = None,
                            per_segment: Optional[Callable[[pandas.DataFrame], pandas.DataFrame]] = None, **kw):
    """``find_synergies(..., solver='mu')`` for every segment of a recording, one GPU launch per rank.

    Args:
        process: applied once to the whole recording before cutting (e.g. ``lambda df: ms.linear_envelope(df, 6,
            2000, 4)``); must keep the number of rows.
        per_segment: applied to every segment after cutting (e.g. ``lambda df: ms.normalize(ms.time_normalize(df,
            200))`` -- segments of equal length then share one batched on-device NNDSVD).
        **kw: forwarded to :func:`find_synergies_batched` (``max_iter``, ``tol``, ``init``, ``beta_loss`` ...).
    Returns:
        one :class:`SynergyRunResult` per segment, in the order given.
    """
    base = emg if isinstance(emg, pandas.DataFrame) else emg.df
    processed = process(base) if process is not None else None
    parts = segment_frames(emg, segments, processed)
    if per_segment is not None:
        parts = [per_segment(p) for p in parts]
    return find_synergies_batched(parts, n_components, max_components, **kw)

"""Segment-to-batch glue (row f-3): the DeviceData indexing protocol (recorded from the real loader) and the
batched per-segment synergy extraction."""
import warnings

import numpy as np
import pandas as pd
import pytest

from conftest import load_json
from muscle_synergies_amd.segments import find_synergies_segments, segment_frames
from muscle_synergies_amd.synth import emg_matrix


class FakeDeviceData:
    """Stand-in with the two members of the reference's DeviceData the glue uses (user_data.py:727-760):
    ``df`` and ``to_index`` for slices of 1-based frames / 0-based subframes."""

    def __init__(self, df, num_subframes):
        self.df = df
        self.num_subframes = num_subframes

    def to_index(self, sl):
        conv = lambda fs: (fs[0] - 1) * self.num_subframes + fs[1]  # noqa: E731
        return slice(conv(sl.start), conv(sl.stop))


def test_indexing_protocol_matches_the_real_loader():
    g9 = load_json("g9_segments.json")
    df = pd.DataFrame(np.array(g9["emg"]), columns=g9["columns"])
    dev = FakeDeviceData(df, g9["num_subframes"])
    assert len(df) == g9["n_rows"] == g9["num_frames"] * g9["num_subframes"]
    segs = [slice(tuple(c["start"]), tuple(c["stop"])) for c in g9["cases"]]
    parts = segment_frames(dev, segs)
    for part, c in zip(parts, g9["cases"]):
        assert dev.to_index(slice(tuple(c["start"]), tuple(c["stop"]))) == slice(c["row_start"], c["row_stop"])
        np.testing.assert_array_equal(part.to_numpy(), np.array(c["values"]))  # = DeviceData.__getitem__
    # plain frames with row slices / pairs, and a processed frame standing in for the raw one
    plain = segment_frames(df, [slice(0, 2), (1, 4)])
    assert [len(p) for p in plain] == [2, 3]
    doubled = segment_frames(dev, segs[:1], processed=df * 2)
    np.testing.assert_array_equal(doubled[0].to_numpy(), 2 * np.array(g9["cases"][0]["values"]))
    with pytest.raises(ValueError, match="selects no rows"):
        segment_frames(df, [slice(3, 3)])
    with pytest.raises(ValueError, match="no segments"):
        segment_frames(df, [])
    with pytest.raises(ValueError, match="rows"):
        segment_frames(df, [slice(0, 2)], processed=df.iloc[:3])


@pytest.mark.gpu
def test_segments_of_a_recording_in_one_launch():
    import muscle_synergies_amd as ms

    X = np.asarray(emg_matrix(31, T=3000, m=8, k_true=3, dtype=np.float64))
    cols = [f"m{j}" for j in range(8)]
    dev = FakeDeviceData(pd.DataFrame(X, columns=cols), num_subframes=10)
    segs = [slice((1, 0), (81, 0)), slice((81, 0), (171, 5)), slice((171, 5), (300, 9))]  # 800, 905, 1294 rows
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = find_synergies_segments(dev, segs, 2, 3, max_iter=150, tol=0.0, random_state=0)
        for res, seg in zip(got, segs):
            part = dev.df.iloc[dev.to_index(seg)]
            one = ms.find_synergies(part, 2, 3, solver="mu", max_iter=150, tol=0.0, random_state=0)
            np.testing.assert_allclose(res.vaf_values.to_numpy(), one.vaf_values.to_numpy(), atol=1e-9)
            np.testing.assert_allclose(res.components[3].to_numpy(), one.components[3].to_numpy(), rtol=1e-7, atol=1e-10)
        # processing of the whole recording before cutting, per-segment time normalisation after it
        eq = find_synergies_segments(dev, segs, 3, process=lambda df: df * 0.5,
                                     per_segment=lambda df: ms.normalize(ms.time_normalize(df, 200)),
                                     max_iter=100, tol=0.0)
    assert len(eq) == 3 and all(r.vaf_values.shape == (1, 9) for r in eq)
    assert all(float(r.vaf_values.iloc[0, 0]) > 0.85 for r in eq)


@pytest.mark.gpu
def test_tutorial_flow_example_runs_end_to_end():
    import importlib.util
    import os

    from conftest import ROOT

    spec = importlib.util.spec_from_file_location("tutorial_flow", os.path.join(ROOT, "examples", "tutorial_flow.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    raw, envelope, envelope_rms, processed, result, chosen = mod.main(T=6000, quiet=True)
    assert envelope.shape == raw.shape == envelope_rms.shape and processed.shape == (1000, 8)
    assert float(processed.to_numpy().max()) == 1.0 and float(processed.to_numpy().min()) >= 0.0
    assert list(result.vaf_values.index) == [2, 3, 4, 5, 6] and 2 <= chosen <= 6
    vaf_all = result.vaf_values["All signals"].to_numpy()
    assert (np.diff(vaf_all) > -1e-3).all()  # more synergies never explain (noticeably) less

"""Host logic that needs no GPU: validation, routing at the seam, VAF, result structure, layouts."""
import warnings

import numpy as np
import pandas as pd
import pytest

import muscle_synergies_amd as ms
from muscle_synergies_amd import _lib, engine
from oracle import nmf_mu_oracle as orc


@pytest.fixture()
def V(g1):
    return pd.DataFrame(np.array(g1["V"]), columns=g1["columns"])


def test_public_names_mirror_the_reference_surface():
    for name in ("find_synergies", "vaf", "SynergyRunResult", "HipNMF", "fit_batched"):
        assert hasattr(ms, name)


def test_error_cases_match_the_reference(V, g1):
    """Messages recorded from the reference (analysis.py:829-846 and sklearn validation)."""
    e = g1["errors"]
    for name, args in {"k0": (0, None), "k9": (9, None), "3_2": (3, 2), "3_9": (3, 9)}.items():
        with pytest.raises(ValueError, match=e[name]):
            ms.find_synergies(V, args[0], args[1], solver="mu")
    with pytest.raises(ValueError, match=e["empty"]):
        ms.find_synergies(V.iloc[0:0], 2, solver="mu")
    raw = pd.DataFrame(np.array(g1["raw_emg"]), columns=g1["columns"])
    with pytest.raises(ValueError) as ei:
        ms.find_synergies(raw, 2, solver="mu")
    assert str(ei.value) == e["negative"] == "Negative values in data passed to NMF initialization."


def test_default_solver_still_routes_to_sklearn(V, g1):
    """find_synergies never sets `solver`; without solver='mu' the call must behave like the reference."""
    pytest.importorskip("sklearn")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = ms.find_synergies(V, 4, init="nndsvda", random_state=0)
    d = g1["default_solver"]
    assert type(res.model).__module__.startswith("sklearn")
    assert res.model.solver == d["solver"] == "cd"
    assert res.model.n_iter_ == d["n_iter"]
    np.testing.assert_allclose(res.model.reconstruction_err_, d["reconstruction_err"], rtol=1e-6)
    np.testing.assert_allclose(res.vaf_values.to_numpy()[0], d["vaf_values"], rtol=1e-6)


def test_mu_without_gpu_fails_loudly(V):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.HipNmfError, match="no CPU fallback"):
        ms.find_synergies(V, 2, solver="mu")
    with pytest.raises(_lib.HipNmfError):
        ms.fit_batched(np.ones((1, 8, 4), np.float32), np.ones((1, 8, 2), np.float32), np.ones((1, 2, 4), np.float32))


def test_mu_outside_the_compiled_shapes_falls_back_to_sklearn():
    """The reference works for any shape (analysis.py:862-863); solver='mu' with more than 512 channels or more than
    64 synergies is outside the compiled kernels (narrow lane mappings up to 32 x 8, matrix-pipe instances up to
    128 x 32, the general-shape kernels up to 512 x 64) and must reach scikit-learn, not an engine error."""
    pytest.importorskip("sklearn")
    rng = np.random.default_rng(5)
    wide = pd.DataFrame(rng.random((60, 513)), columns=[f"ch{j}" for j in range(513)])
    with pytest.warns(RuntimeWarning, match="outside the HIP engine's compiled shapes"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=Warning)
            warnings.simplefilter("always", category=RuntimeWarning)
            res = ms.find_synergies(wide, 3, solver="mu", init="nndsvda", max_iter=20, tol=0)
    assert type(res.model).__module__.startswith("sklearn") and res.model.solver == "mu"
    assert res.components.shape == (3, 513)
    narrow = pd.DataFrame(rng.random((90, 80)), columns=[f"ch{j}" for j in range(80)])
    with pytest.warns(RuntimeWarning):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=Warning)
            warnings.simplefilter("always", category=RuntimeWarning)
            res = ms.find_synergies(narrow, 65, solver="mu", init="random", random_state=0, max_iter=20, tol=0)
    assert type(res.model).__module__.startswith("sklearn")
    assert ms.HipNMF.supports(solver="mu", n_features=32, n_components=8)
    assert ms.HipNMF.supports(solver="mu", n_features=33, n_components=8)
    assert ms.HipNMF.supports(solver="mu", n_features=128, n_components=32)
    assert ms.HipNMF.supports(solver="mu", n_features=512, n_components=64)
    assert not ms.HipNMF.supports(solver="mu", n_features=513, n_components=8)
    assert not ms.HipNMF.supports(solver="mu", n_features=80, n_components=65)
    assert not ms.HipNMF.supports(solver="cd", n_features=8, n_components=2)


def test_vaf_matches_reference_numbers(V, g1):
    c = g1["single_k4"]
    out = ms.vaf(V, transformed_signal=np.array(c["transformed"]), components=np.array(c["components"]))
    assert list(out.columns) == c["vaf_columns"] == ["All signals"] + g1["columns"]
    assert out.shape == (1, 9)
    np.testing.assert_allclose(out.to_numpy()[0], c["vaf_values"], rtol=1e-12)
    va, vc = orc.vaf(V.to_numpy(), np.array(c["transformed"]), np.array(c["components"]))
    np.testing.assert_allclose(out.to_numpy()[0], np.r_[va, vc], rtol=1e-12)
    rec = np.array(c["transformed"]) @ np.array(c["components"])
    out2 = ms.vaf(V, reconstructed_signal=rec)
    np.testing.assert_allclose(out2.to_numpy(), out.to_numpy())


def test_range_result_structure_with_sklearn_backend(V):
    """The merge logic (analysis.py:884-894) is backend independent: exercise it on the CPU route."""
    pytest.importorskip("sklearn")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = ms.find_synergies(V, 2, 4, max_iter=50, init="nndsvda", random_state=0)
    assert list(res.components.keys()) == [2, 3, 4] == list(res.model.keys())
    assert list(res.vaf_values.index) == [2, 3, 4]
    assert list(res.vaf_values.columns) == ["All signals"] + list(V.columns)
    for k in (2, 3, 4):
        assert res.components[k].shape == (k, 8)
        assert list(res.components[k].columns) == list(V.columns)


def test_hipnmf_parameter_validation():
    X = np.abs(np.random.default_rng(0).standard_normal((20, 4)))
    with pytest.raises(ValueError, match="solver='mu' only"):
        ms.HipNMF(2, solver="cd").fit_transform(X)
    with pytest.raises(NotImplementedError):
        ms.HipNMF(2, beta_loss="itakura-saito").fit_transform(X)
    with pytest.raises(ValueError, match="max_iter"):
        ms.HipNMF(2, max_iter=0).fit_transform(X)
    with pytest.raises(ValueError, match="init"):
        ms.HipNMF(2, init="bogus").fit_transform(X)
    with pytest.raises(ValueError, match="Negative values in data passed to NMF initialization."):
        ms.HipNMF(2).fit_transform(-X)
    with pytest.raises(ValueError, match="wrong first dimension passed to NMF \\(input H\\)"):
        ms.HipNMF(2, init="custom").fit_transform(X, W=np.ones((20, 2)), H=np.ones((3, 4)))
    with pytest.raises(ValueError, match="full of zeros"):
        ms.HipNMF(2, init="custom").fit_transform(X, W=np.zeros((20, 2)), H=np.ones((2, 4)))
    with pytest.raises(TypeError, match="same dtype as X"):
        ms.HipNMF(2, init="custom").fit_transform(X, W=np.ones((20, 2), np.float32), H=np.ones((2, 4), np.float32))
    est = ms.HipNMF(3, tol=0.5)
    assert est.get_params()["n_components"] == 3 and est.set_params(tol=0.1).tol == 0.1
    assert ms.HipNMF.supports(solver="mu") and not ms.HipNMF.supports() and ms.HipNMF.supports(solver="mu", beta_loss=1) and not ms.HipNMF.supports(solver="mu", beta_loss=0.5)
    with pytest.raises(RuntimeError, match="not fitted"):
        est.transform(X)


def test_regularisation_scaling_follows_sklearn():
    est = ms.HipNMF(2, alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3)
    assert est._regularization(512, 16) == pytest.approx(orc.compute_regularization(512, 16, 0.002, 0.001, 0.3))
    est = ms.HipNMF(2, alpha_W=0.01)
    assert est._regularization(100, 8) == pytest.approx(orc.compute_regularization(100, 8, 0.01, "same", 0.0))


def test_partition_is_contiguous_and_balanced():
    for n, p in ((4096, 8), (10, 3), (2, 4), (0, 2)):
        parts = engine.partition(n, p)
        assert len(parts) == p and parts[0][0] == 0 and parts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1


def test_layout_detection_on_host_tensors():
    import torch

    B, T, m = 3, 40, 8
    c = torch.zeros((B, T, m))
    assert engine._x_layout(c)[:3] == (_lib.X_ROW_MAJOR, m, T * m)
    f = torch.zeros((B, m, T)).transpose(1, 2)  # DataFrame-style channel-major storage
    assert engine._x_layout(f)[:3] == (_lib.X_CHANNEL_MAJOR, T, T * m)
    sl = torch.zeros((B, T, 2 * m))[:, :, :m]  # row-major with a padded leading dimension
    assert engine._x_layout(sl)[:3] == (_lib.X_ROW_MAJOR, 2 * m, T * 2 * m)
    weird = torch.zeros((B, T, 2 * m))[:, :, ::2]
    lay, ld, bs, t = engine._x_layout(weird)
    assert (lay, ld, bs) == (_lib.X_ROW_MAJOR, m, T * m) and t.is_contiguous()
    df_like = torch.from_numpy(np.asfortranarray(np.zeros((T, m)))).unsqueeze(0)
    assert engine._x_layout(df_like)[:2] == (_lib.X_CHANNEL_MAJOR, T)


def test_problem_struct_roundtrip():
    p = engine.make_problem(7, 1000, 16, 5, x_layout=1, ldx=1000, x_batch_stride=16000, max_iter=33, tol=1e-3,
                            l1_reg_W=0.5, l2_reg_H=0.25, update_H=False)
    assert (p.batch, p.n_samples, p.n_features, p.n_components) == (7, 1000, 16, 5)
    assert (p.x_layout, p.update_h, p.w_layout, p.max_iter, p.check_every) == (1, 0, 0, 33, 10)
    assert (p.tol, p.l1_reg_W, p.l2_reg_H) == (1e-3, 0.5, 0.25)


def test_rank_range_concurrency_rule(monkeypatch):
    """find_synergies fits the ranks of a range from concurrent host threads only on the GPU path and only when that cannot
    change what the reference's sequential loop would compute."""
    import numpy as np

    from muscle_synergies_amd.analysis import _ranks_concurrently

    monkeypatch.delenv("HIPNMF_RANK_THREADS", raising=False)
    assert _ranks_concurrently(3, dict(solver="mu"))
    assert _ranks_concurrently(2, dict(solver="mu", random_state=7, init="random"))
    assert not _ranks_concurrently(1, dict(solver="mu"))
    assert not _ranks_concurrently(3, dict())                      # the reference's default solver: scikit-learn, sequential
    assert not _ranks_concurrently(3, dict(solver="cd"))
    assert not _ranks_concurrently(3, dict(solver="mu", beta_loss="itakura-saito"))
    assert _ranks_concurrently(3, dict(solver="mu", random_state=np.random.RandomState(0)))  # (initialisations stay in rank order on the calling thread)
    assert _ranks_concurrently(3, dict(solver="mu"), (200, 8)) and _ranks_concurrently(3, dict(solver="mu"), (2048, 32))
    # any frame size since round 4 (the C ABI's handles are independent: tests/test_gpu_abi_threads.py): the chip-filling paths too
    assert _ranks_concurrently(3, dict(solver="mu"), (10_000, 16)) and _ranks_concurrently(3, dict(solver="mu"), (500, 64))
    monkeypatch.setenv("HIPNMF_RANK_THREADS", "0")
    assert not _ranks_concurrently(3, dict(solver="mu"))


def test_filter_mode_switch_of_the_single_frame_functions():
    """digital_filter / linear_envelope keep the reference's signature and take one extra keyword, mode; the process-wide default is
    "exact" (scipy's bits) and set_filter_mode validates its argument."""
    import inspect

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import preprocess

    assert preprocess._FILTER_MODE in ("exact", "scan")
    before = preprocess._FILTER_MODE
    try:
        ms.set_filter_mode("scan")
        assert preprocess._FILTER_MODE == "scan"
        with pytest.raises(KeyError):
            ms.set_filter_mode("fast")
    finally:
        ms.set_filter_mode(before)
    for fn, ref_names in ((ms.digital_filter, ["signal_df", "critical_freqs", "sampling_frequency", "order", "filter_type", "band_type",
                                               "zero_lag", "cheby_param", "inplace"]),
                          (ms.linear_envelope, ["signal_df", "critical_freqs", "sampling_frequency", "order", "filter_type", "zero_lag",
                                                "cheby_param"])):
        params = inspect.signature(fn).parameters
        assert list(params)[: len(ref_names)] == ref_names  # the reference's positional order (analysis.py:252-432)
        assert params["mode"].kind is inspect.Parameter.KEYWORD_ONLY and params["mode"].default is None


def test_route_table_is_one_table_with_an_environment_override():
    """every fitted routing threshold of the dispatchers lives in struct hipnmf_route_table
    (csrc/hipnmf_internal.hpp); hipnmf_routes_describe() reports the values in force, HIPNMF_ROUTES overrides them by name."""
    import os
    import re
    import subprocess
    import sys

    from muscle_synergies_amd import _lib

    r = _lib.routes()  # no GPU needed: the table is host data
    assert r["f32_16ch_wide_max_rows"] == 600 and r["f64_16ch_wide_max_rows"] == 1200 and r["kl_sliced_margin"] == 0.9
    hdr = open(os.path.join(os.path.dirname(_lib.PKG), "muscle_synergies_amd", "csrc", "hipnmf_internal.hpp")).read()
    body = hdr[hdr.index("struct hipnmf_route_table {"):hdr.index("const hipnmf_route_table& hipnmf_routes();")]
    fields = re.findall(r"\b([a-z][a-z0-9_]+)\s*=\s*[-0-9.e]+", body)
    assert sorted(fields) == sorted(r), (sorted(set(fields) ^ set(r)))  # every field of the struct is reported (and overridable)
    # the dispatchers hold no fitted row-count literal of their own any more
    api = open(os.path.join(os.path.dirname(_lib.PKG), "muscle_synergies_amd", "csrc", "hipnmf_api.hip")).read()
    wp = api[api.index("bool wide_preferred("):api.index("int fit_batched_impl(")]
    # (T <= 256 / 128 stay: the reach of the one-wave kernel's four tiles, a property of that kernel and not a fitted crossover)
    assert not re.search(r"T <= \(?(?!256\b|128\b)\d{3,}", wp), "a fitted threshold is back in wide_preferred as a literal"
    code = ("import sys; sys.path.insert(0, %r); from muscle_synergies_amd import _lib; r = _lib.routes(); "
            "print(r['f32_16ch_wide_max_rows'], r['pers_s_per_row'])" % os.path.dirname(_lib.PKG))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         env=dict(os.environ, HIPNMF_ROUTES="f32_16ch_wide_max_rows=900,pers_s_per_row=3e-9,not_a_route=1"))
    assert out.returncode == 0 and out.stdout.split() == ["900.0", "3e-09"], out.stdout + out.stderr
    assert "ignoring 'not_a_route=1'" in out.stderr


def test_exit_gate_counts_calls_parks_late_comers_and_waits_for_daemons():
    """_lib._Gate (round 6): native calls in flight are counted; once the exit hook has closed the gate a thread that tries to START
    a call is parked (never returns into the runtime), the hook's own calls still pass, close() waits for the calls in flight and
    wait_parked() for the daemon threads between two calls."""
    import threading
    import time

    from muscle_synergies_amd._lib import _Fn, _Gate
    from muscle_synergies_amd import _lib

    gate = _Gate()
    saved, _lib._gate = _lib._gate, gate
    try:
        release = threading.Event()
        entered = threading.Event()
        calls = []

        def slow(x):  # stands for a ctypes function that is in flight when the exit begins
            entered.set()
            release.wait(5)
            calls.append(x)
            return x + 1

        f = _Fn(slow)
        out = []
        t1 = threading.Thread(target=lambda: out.append(f(1)), daemon=True)
        t1.start()
        assert entered.wait(5) and gate.inflight == 1
        # the exit hook: close() must wait for the call in flight ...
        closer = []
        tc = threading.Thread(target=lambda: closer.append(gate.close(5.0)))
        tc.start()
        time.sleep(0.1)
        assert gate.closing and tc.is_alive()
        # ... and a thread that wants to START a call now is parked, not let through
        t2 = threading.Thread(target=lambda: out.append(f(10)), daemon=True)
        t2.start()
        assert gate.wait_parked({t2.ident}, 5.0)
        release.set()
        tc.join(5)
        assert closer == [True] and out == [2] and calls == [1] and gate.inflight == 0
        time.sleep(0.1)
        assert t2.is_alive() and 10 not in calls  # still parked; its call never ran
        # the closing thread itself may still call (it destroys the handles)
        gate.closer = threading.get_ident()
        assert _Fn(lambda: 7)() == 7
        assert not gate.wait_parked({12345}, 0.05)  # a thread that never shows up only costs the timeout
    finally:
        _lib._gate = saved

#!/usr/bin/env python3
"""Generate the golden fixtures under ``tests/golden/``.

Runs ONLY in the build container: it imports the real reference
(``/root/reference/src/muscle_synergies``, with a stub for the missing
``seaborn``) and the image's scikit-learn 1.7.2 -- the third-party dependency
where the reference's NMF arithmetic lives -- and records inputs and outputs as
small data files.  No reference source travels; the fixtures are numbers.

    python tests/golden/make_golden.py

Fixture sets:
  G1  plumbing: abridged Vicon CSV -> EMG block -> ``find_synergies``
  G2  the mu loop at fixed iteration counts, fp32 and fp64, custom init
  G3  the tol > 0 stop rule (n_iter_, error trace)
  G4  NNDSVD / random initialisation vectors (host-side init parity)
  G5  ``transform`` (update_H=False) and regularised fits
  G6  EMG envelope preprocessing (zero_center, rms, time_normalize, normalize) -- row f-1
  G7  Kullback-Leibler loss (beta_loss='kullback-leibler') -- row f-4
  G8  IIR filters: ``digital_filter`` / ``linear_envelope`` (scipy sosfilt / sosfiltfilt) -- row f-1
  G9  ``DeviceData`` (frame, subframe) -> row indexing of the abridged recording (segment glue) -- row f-3
  G10 ``time_normalize`` with every ``interp1d`` kind the reference forwards (analysis.py:551-594) -- row f-1
"""

import json
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

os.environ.setdefault("MPLBACKEND", "Agg")
sys.modules["seaborn"] = types.ModuleType("seaborn")  # analysis.py:23 imports it, uses it only at :133
sys.path.insert(0, "/root/reference/src")

import muscle_synergies as ms  # noqa: E402  (the reference)
import sklearn  # noqa: E402
from sklearn.decomposition import NMF  # noqa: E402
from sklearn.decomposition._nmf import _beta_divergence, _initialize_nmf  # noqa: E402

from muscle_synergies_amd.synth import emg_matrix, random_init, raw_emg  # noqa: E402

warnings.simplefilter("ignore")


def tolist(a):
    return np.asarray(a).tolist()


# --------------------------------------------------------------------------- G1
def g1():
    emg = ms.load_vicon_file("/root/reference/sample_data/abridged_data.csv").emg.df
    V = emg.abs()
    out = {
        "sklearn_version": sklearn.__version__,
        "columns": list(V.columns),
        "raw_emg": tolist(emg.to_numpy()),
        "V": tolist(V.to_numpy()),
    }
    # deterministic single-rank call through the reference's own entry point
    kw = dict(solver="mu", max_iter=200, init="nndsvda", random_state=0)
    res = ms.find_synergies(V, n_components=4, **kw)
    W0, H0 = _initialize_nmf(V.to_numpy(), 4, init="nndsvda", random_state=0)
    m = NMF(4, solver="mu", init="custom", max_iter=200, tol=1e-6)
    W = m.fit_transform(V.to_numpy(), W=W0.copy(), H=H0.copy())
    assert np.array_equal(m.components_, res.model.components_)
    out["single_k4"] = {
        "kwargs": kw,
        "tol": 1e-6,
        "W0": tolist(W0),
        "H0": tolist(H0),
        "n_iter": int(res.model.n_iter_),
        "reconstruction_err": float(res.model.reconstruction_err_),
        "components": tolist(res.model.components_),
        "transformed": tolist(W),
        "vaf_columns": list(res.vaf_values.columns),
        "vaf_values": tolist(res.vaf_values.to_numpy()[0]),
    }
    # default (init=None, random_state=None) call as in SURVEY: record stats only
    res_d = ms.find_synergies(V, n_components=4, solver="mu", max_iter=200)
    out["single_k4_default_init"] = {
        "n_iter": int(res_d.model.n_iter_),
        "reconstruction_err": float(res_d.model.reconstruction_err_),
        "vaf_values": tolist(res_d.vaf_values.to_numpy()[0]),
    }
    # range call
    res_r = ms.find_synergies(V, 2, 4, **kw)
    out["range_2_4"] = {
        "kwargs": kw,
        "keys": [int(k) for k in res_r.components.keys()],
        "vaf_index": tolist(res_r.vaf_values.index),
        "vaf_values": tolist(res_r.vaf_values.to_numpy()),
        "components": {str(k): tolist(v.to_numpy()) for k, v in res_r.components.items()},
        "n_iter": {str(k): int(v.n_iter_) for k, v in res_r.model.items()},
        "reconstruction_err": {str(k): float(v.reconstruction_err_) for k, v in res_r.model.items()},
    }
    # default solver must still route to sklearn's 'cd'
    res_cd = ms.find_synergies(V, n_components=4, init="nndsvda", random_state=0)
    out["default_solver"] = {
        "solver": res_cd.model.solver,
        "n_iter": int(res_cd.model.n_iter_),
        "reconstruction_err": float(res_cd.model.reconstruction_err_),
        "vaf_values": tolist(res_cd.vaf_values.to_numpy()[0]),
    }
    # error cases
    errs = {}
    for name, args in {"k0": (0, None), "k9": (9, None), "3_2": (3, 2), "3_9": (3, 9)}.items():
        try:
            ms.find_synergies(V, args[0], args[1], solver="mu")
        except ValueError as e:
            errs[name] = str(e)
    try:
        ms.find_synergies(V.iloc[0:0], 2, solver="mu")
    except ValueError as e:
        errs["empty"] = str(e)
    try:
        ms.find_synergies(emg, 2, solver="mu")
    except ValueError as e:
        errs["negative"] = str(e)
    out["errors"] = errs
    with open(os.path.join(HERE, "g1_abridged.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("G1", out["single_k4"]["n_iter"], out["single_k4"]["reconstruction_err"], errs)


# --------------------------------------------------------------------------- G2
def run_sk(X, W0, H0, n_iter, tol=0.0, **kw):
    m = NMF(W0.shape[1], solver="mu", init="custom", tol=tol, max_iter=n_iter, **kw)
    W = m.fit_transform(X, W=W0.copy(), H=H0.copy())
    return W, m.components_.copy(), m


def g2_small():
    """T=512, m=16, k=5: full W/H stored at a few iteration counts."""
    arrays = {}
    meta = {"sklearn_version": sklearn.__version__, "T": 512, "m": 16, "k": 5, "seed": 7, "iters": [1, 2, 10, 100]}
    for dt in (np.float32, np.float64):
        tag = np.dtype(dt).name
        X = emg_matrix(7, T=512, dtype=dt)
        W0, H0 = _initialize_nmf(X, 5, init="nndsvda", random_state=0)
        arrays[f"X_{tag}"] = np.ascontiguousarray(X)
        arrays[f"W0_{tag}"] = W0
        arrays[f"H0_{tag}"] = H0
        for n in meta["iters"]:
            W, H, m = run_sk(X, W0, H0, n)
            arrays[f"W_{tag}_{n}"] = W
            arrays[f"H_{tag}_{n}"] = H
            arrays[f"err_{tag}_{n}"] = np.array(m.reconstruction_err_)
    np.savez_compressed(os.path.join(HERE, "g2_loop_T512.npz"), **arrays)
    with open(os.path.join(HERE, "g2_loop_T512.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("G2 small done")


def g2_full():
    """Config #2 shape (10 000 x 16, k=5): X/W0/H0 by recipe, outputs as checksums."""
    out = {"sklearn_version": sklearn.__version__, "T": 10000, "m": 16, "k": 5,
           "rows": [0, 1234, 4999, 9999], "cases": []}
    for dt in (np.float32, np.float64):
        for seed, init in ((0, "random"), (1, "random"), (0, "nndsvda")):
            X = emg_matrix(seed, dtype=dt)
            if init == "random":
                W0, H0 = random_init(X, 5, seed)
                Wc, Hc = _initialize_nmf(X, 5, init="random", random_state=seed)
                assert np.array_equal(W0, Wc) and np.array_equal(H0, Hc)
            else:
                W0, H0 = _initialize_nmf(X, 5, init="nndsvda", random_state=0)
            case = {"dtype": np.dtype(dt).name, "seed": seed, "init": init,
                    "X_sum": float(X.astype(np.float64).sum()),
                    "X_fro": float(np.sqrt((X.astype(np.float64) ** 2).sum())),
                    "W0_sum": float(W0.astype(np.float64).sum()), "H0_sum": float(H0.astype(np.float64).sum()),
                    "H0": tolist(H0) if init == "nndsvda" else None,
                    "W0_rows": tolist(W0[out["rows"]]) if init == "nndsvda" else None,
                    "iters": {}}
            for n in (1, 2, 10, 100, 500):
                W, H, m = run_sk(X, W0, H0, n)
                WH = (W.astype(np.float64) @ H.astype(np.float64))
                Xd = X.astype(np.float64)
                sse_col = ((Xd - WH) ** 2).sum(axis=0)
                case["iters"][str(n)] = {
                    "reconstruction_err": float(m.reconstruction_err_),
                    "H": tolist(H),
                    "W_rows": tolist(W[out["rows"]]),
                    "WH_rows": tolist(WH[out["rows"]]),
                    "WH_fro": float(np.sqrt((WH ** 2).sum())),
                    "WH_colsum": tolist(WH.sum(axis=0)),
                    "vaf_all": float(1 - sse_col.sum() / (Xd ** 2).sum()),
                    "vaf_col": tolist(1 - sse_col / (Xd ** 2).sum(axis=0)),
                }
            out["cases"].append(case)
            print("G2 full", case["dtype"], seed, init, case["iters"]["500"]["reconstruction_err"])
    with open(os.path.join(HERE, "g2_loop_T10000.json"), "w") as f:
        json.dump(out, f)


# --------------------------------------------------------------------------- G3
def g3():
    """Stop rule: tol > 0, record n_iter_ and the error trace sklearn saw."""
    out = {"cases": []}
    for dt in (np.float32, np.float64):
        for T, seed, tol, max_iter in ((512, 7, 1e-4, 2000), (512, 7, 1e-3, 2000), (2000, 3, 1e-4, 30)):
            X = emg_matrix(seed, T=T, dtype=dt)
            W0, H0 = _initialize_nmf(X, 5, init="nndsvda", random_state=0)
            W, H, m = run_sk(X, W0, H0, max_iter, tol=tol)
            # error trace: replay with sklearn's own _beta_divergence every 10 iterations
            trace = [float(_beta_divergence(X, W0, H0, 2, square_root=True))]
            for n in range(10, int(m.n_iter_) + 1, 10):
                Wn, Hn, _ = run_sk(X, W0, H0, n)
                trace.append(float(_beta_divergence(X, Wn, Hn, 2, square_root=True)))
            out["cases"].append({"dtype": np.dtype(dt).name, "T": T, "seed": seed, "tol": tol,
                                 "max_iter": max_iter, "n_iter": int(m.n_iter_),
                                 "reconstruction_err": float(m.reconstruction_err_),
                                 "err_trace": trace, "H": tolist(H)})
            print("G3", np.dtype(dt).name, T, tol, m.n_iter_)
    with open(os.path.join(HERE, "g3_stop_rule.json"), "w") as f:
        json.dump(out, f)


# --------------------------------------------------------------------------- G4
def g4():
    """Initialisation vectors: sklearn's _initialize_nmf on small matrices."""
    arrays = {}
    meta = []
    for dt in (np.float32, np.float64):
        tag = np.dtype(dt).name
        for T, m_, k, seed in ((512, 16, 5, 7), (64, 8, 3, 11), (6, 8, 4, 5)):
            X = emg_matrix(seed, T=T, m=m_, k_true=min(5, m_), dtype=dt)
            arrays[f"X_{tag}_{T}_{m_}"] = np.ascontiguousarray(X)
            for init in ("random", "nndsvd", "nndsvda", "nndsvdar"):
                W0, H0 = _initialize_nmf(X, k, init=init, random_state=3)
                arrays[f"W0_{tag}_{T}_{m_}_{k}_{init}"] = W0
                arrays[f"H0_{tag}_{T}_{m_}_{k}_{init}"] = H0
                meta.append({"dtype": tag, "T": T, "m": m_, "k": k, "init": init, "random_state": 3})
    np.savez_compressed(os.path.join(HERE, "g4_init.npz"), **arrays)
    with open(os.path.join(HERE, "g4_init.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("G4 done")


# --------------------------------------------------------------------------- G5
def g5():
    """transform (update_H=False) and regularised fits, T=512."""
    arrays = {}
    for dt in (np.float32, np.float64):
        tag = np.dtype(dt).name
        X = emg_matrix(7, T=512, dtype=dt)
        W0, H0 = _initialize_nmf(X, 5, init="nndsvda", random_state=0)
        W, H, m = run_sk(X, W0, H0, 50)
        X2 = emg_matrix(8, T=300, dtype=dt)
        m.tol = 0.0
        m.max_iter = 40
        arrays[f"H_fit_{tag}"] = H
        arrays[f"X2_{tag}"] = np.ascontiguousarray(X2)
        arrays[f"W_transform_{tag}"] = m.transform(X2)
        Wr, Hr, mr = run_sk(X, W0, H0, 60, alpha_W=0.002, alpha_H=0.001, l1_ratio=0.3)
        arrays[f"W_reg_{tag}"] = Wr
        arrays[f"H_reg_{tag}"] = Hr
        arrays[f"err_reg_{tag}"] = np.array(mr.reconstruction_err_)
    np.savez_compressed(os.path.join(HERE, "g5_transform_reg.npz"), **arrays)
    print("G5 done")


# --------------------------------------------------------------------------- G6
def g6():
    """Envelope preprocessing: outputs of the reference's own DataFrame functions."""
    import pandas as pd

    arrays = {}
    for seed, (name, (T, m, win, fs, reduce_to)) in enumerate({"small": (600, 4, 25, None, 50), "odd": (1001, 3, 0.0505, 200, 77),
                                                              "tutorial": (20000, 8, 0.5, 2000, 200)}.items()):
        raw = raw_emg(seed, T, m)
        df = pd.DataFrame(raw, columns=[f"m{j}" for j in range(m)])
        if T <= 2000:
            arrays[f"{name}_raw"] = raw
        zc = ms.zero_center(df)
        r = ms.rms(zc, window_size=win, sampling_frequency=fs)
        tn = ms.time_normalize(r, reduce_to=reduce_to)
        nm = tn / tn.max()            # tutorial cell 23
        nm2 = ms.normalize(tn)
        arrays[f"{name}_params"] = np.array([T, m, round(win * fs) if fs else win, reduce_to, seed], dtype=np.int64)
        arrays[f"{name}_raw_sum"] = np.array(raw.sum())
        if T <= 2000:
            arrays[f"{name}_zero_center"] = zc.to_numpy()
            arrays[f"{name}_rms"] = r.to_numpy()
        else:
            arrays[f"{name}_rms_rows"] = r.to_numpy()[[0, 1, 499, 500, 501, 9999, 19998, 19999]]
        arrays[f"{name}_time_normalize"] = tn.to_numpy()
        arrays[f"{name}_normalize"] = nm2.to_numpy()
        assert np.allclose(nm.to_numpy(), nm2.to_numpy())
    np.savez_compressed(os.path.join(HERE, "g6_envelope.npz"), **arrays)
    print("G6 done")


# --------------------------------------------------------------------------- G7
def g7():
    """Kullback-Leibler loss (solver='mu', beta_loss='kullback-leibler'): sklearn outputs, T=512."""
    arrays = {}
    for dt in (np.float32, np.float64):
        tag = np.dtype(dt).name
        X = emg_matrix(7, T=512, dtype=dt)
        X[::37, 3] = 0  # exact zeros exercise the X > EPSILON mask of the divergence
        W0, H0 = _initialize_nmf(X, 5, init="nndsvda", random_state=0)
        arrays[f"X_{tag}"] = np.ascontiguousarray(X)
        arrays[f"W0_{tag}"], arrays[f"H0_{tag}"] = W0, H0
        for n in (1, 2, 10, 100):
            m = NMF(5, solver="mu", beta_loss="kullback-leibler", init="custom", tol=0, max_iter=n)
            W = m.fit_transform(X, W=W0.copy(), H=H0.copy())
            arrays[f"W_{tag}_{n}"], arrays[f"H_{tag}_{n}"] = W, m.components_.copy()
            arrays[f"err_{tag}_{n}"] = np.array(m.reconstruction_err_)
        m = NMF(5, solver="mu", beta_loss="kullback-leibler", init="custom", tol=1e-4, max_iter=2000)
        W = m.fit_transform(X, W=W0.copy(), H=H0.copy())
        arrays[f"stop_n_iter_{tag}"] = np.array(m.n_iter_)
        arrays[f"stop_err_{tag}"] = np.array(m.reconstruction_err_)
        m = NMF(5, solver="mu", beta_loss="kullback-leibler", init="custom", tol=0, max_iter=40, alpha_W=0.002,
                alpha_H=0.001, l1_ratio=0.3)
        W = m.fit_transform(X, W=W0.copy(), H=H0.copy())
        arrays[f"W_reg_{tag}"], arrays[f"H_reg_{tag}"] = W, m.components_.copy()
    np.savez_compressed(os.path.join(HERE, "g7_kl.npz"), **arrays)
    print("G7 done", arrays["stop_n_iter_float64"])


# --------------------------------------------------------------------------- G8
G8_CASES = {
    # name: (T, m, fs, digital_filter kwargs)
    "lp4": (600, 4, 2000, dict(critical_freqs=6, order=4, filter_type="butter", band_type="lowpass", zero_lag=True)),
    "hp2_fwd": (500, 3, 2000, dict(critical_freqs=20, order=2, filter_type="butter", band_type="highpass", zero_lag=False)),
    "bp3": (1001, 5, 2000, dict(critical_freqs=[20, 450], order=3, filter_type="butter", band_type="bandpass", zero_lag=True)),
    "cheby1_lp5": (777, 2, 1000, dict(critical_freqs=10, order=5, filter_type="cheby1", band_type="lowpass", zero_lag=True,
                                      cheby_param=1.0)),
    "cheby2_bs2_fwd": (300, 2, 1000, dict(critical_freqs=[45, 55], order=2, filter_type="cheby2", band_type="bandstop",
                                           zero_lag=False, cheby_param=30.0)),
    "long_lp2": (20000, 8, 2000, dict(critical_freqs=4, order=2, filter_type="butter", band_type="lowpass", zero_lag=True)),
    "short": (20, 2, 100, dict(critical_freqs=5, order=1, filter_type="butter", band_type="lowpass", zero_lag=True)),
}


def g8_design(kw, fs):
    """The section coefficients exactly as the reference designs them (analysis.py:381-403) + scipy's zi."""
    from scipy import signal

    if kw["filter_type"] == "butter":
        sos = signal.butter(kw["order"], kw["critical_freqs"], btype=kw["band_type"], output="sos", fs=fs)
    else:
        f = signal.cheby1 if kw["filter_type"] == "cheby1" else signal.cheby2
        sos = f(kw["order"], kw["cheby_param"], kw["critical_freqs"], btype=kw["band_type"], output="sos", fs=fs)
    return sos, signal.sosfilt_zi(sos)


def g8():
    """Outputs of the reference's own ``digital_filter`` and ``linear_envelope`` on synthetic raw EMG."""
    import pandas as pd

    arrays = {}
    for seed, (name, (T, m, fs, kw)) in enumerate(G8_CASES.items()):
        raw = raw_emg(100 + seed, T, m, fs=float(fs))
        df = pd.DataFrame(raw, columns=[f"m{j}" for j in range(m)])
        sos, zi = g8_design(kw, fs)
        filt = ms.digital_filter(df, sampling_frequency=fs, **kw).to_numpy()
        arrays[f"{name}_params"] = np.array([T, m, fs, int(kw["zero_lag"]), 100 + seed], dtype=np.int64)
        arrays[f"{name}_sos"], arrays[f"{name}_zi"] = sos, zi
        if T <= 2000:
            arrays[f"{name}_raw"] = raw
            arrays[f"{name}_filtered"] = filt
        else:
            rows = np.array([0, 1, 2, 17, 5000, 9999, 19997, 19998, 19999])
            arrays[f"{name}_rows"] = rows
            arrays[f"{name}_filtered_rows"] = filt[rows]
            arrays[f"{name}_filtered_colsum"] = filt.sum(axis=0)
        if kw["band_type"] == "lowpass":
            le_kw = {k: v for k, v in kw.items() if k != "band_type"}
            le = ms.linear_envelope(df, sampling_frequency=fs, **le_kw).to_numpy()
            le_nc = ms.linear_envelope(df, sampling_frequency=fs, zero_center_=False, **le_kw).to_numpy()
            if T <= 2000:
                arrays[f"{name}_linear_envelope"] = le
                arrays[f"{name}_linear_envelope_nocenter"] = le_nc
            else:
                arrays[f"{name}_linear_envelope_rows"] = le[rows]
                arrays[f"{name}_linear_envelope_colsum"] = le.sum(axis=0)
    # float32 input: scipy filters in float64 and the reference returns a float64 frame
    raw32 = raw_emg(100, 600, 4).astype(np.float32)
    kw = G8_CASES["lp4"][3]
    out32 = ms.digital_filter(pd.DataFrame(raw32), sampling_frequency=2000, **kw).to_numpy()
    assert out32.dtype == np.float64
    arrays["lp4_filtered_from_f32"] = out32
    np.savez_compressed(os.path.join(HERE, "g8_filters.npz"), **arrays)
    print("G8 done")


# --------------------------------------------------------------------------- G9
def g9():
    """The indexing protocol the segment glue relies on: DeviceData.to_index / __getitem__ on the real loader."""
    emg = ms.load_vicon_file("/root/reference/sample_data/abridged_data.csv").emg
    ft = emg._frame_tracker
    out = {"n_rows": len(emg.df), "num_frames": ft.num_frames, "num_subframes": ft.num_subframes,
           "sampling_frequency": emg.sampling_frequency, "columns": list(emg.df.columns), "cases": []}
    for a, b in [((1, 0), (1, 2)), ((1, 1), (2, 1)), ((1, 0), (2, 2)), ((2, 0), (2, 2))]:
        sl = slice(a, b)
        rows = emg.to_index(sl)
        out["cases"].append({"start": list(a), "stop": list(b), "row_start": rows.start, "row_stop": rows.stop,
                             "values": tolist(emg[sl].to_numpy())})
    out["emg"] = tolist(emg.df.to_numpy())
    with open(os.path.join(HERE, "g9_segments.json"), "w") as f:
        json.dump(out, f)
    print("G9 done")


# --------------------------------------------------------------------------- G10
def g10(only=True):
    """time_normalize(kind=...) of the reference for every kind scipy's interp1d knows: shrinking, same length, stretching."""
    import pandas as pd

    arrays = {}
    raw = np.abs(raw_emg(10, 97, 3))
    arrays["raw"] = raw
    df = pd.DataFrame(raw, columns=list("abc"))
    kinds = ["linear", "slinear", "nearest", "nearest-up", "previous", "next", "zero", "quadratic", "cubic"]
    for reduce_to in (40, 97, 230, 2, 193):
        for kind in kinds:
            arrays[f"{kind}_{reduce_to}"] = ms.time_normalize(df, reduce_to=reduce_to, kind=kind).to_numpy()
    arrays["kinds"] = np.array(kinds)
    np.savez_compressed(os.path.join(HERE, "g10_time_normalize_kinds.npz"), **arrays)
    print("G10 done")


if __name__ == "__main__":
    if len(sys.argv) > 1:  # e.g. `make_golden.py g10`: regenerate the named sets only
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g1()
    g2_small()
    g2_full()
    g3()
    g4()
    g5()
    g6()
    g7()
    g8()
    g9()
    g10()

#!/usr/bin/env python3
"""DISPATCH.md: which kernel instance serves which call by default.

    gpurun -- 'python tools/dispatch_table.py'          # runs tiny calls over a grid of shapes on the GPU, writes DISPATCH.md
    python tools/dispatch_table.py --check               # CPU: DISPATCH.md against the kernels the sources define (tests/test_dispatch_doc.py)

Every row is what ``hipnmf_last_kernel`` reported for a real call with default tuning (no environment switch, no
``hipnmf_set_tuning``).  The last sections account for every ``__global__`` function of ``muscle_synergies_amd/csrc``: named by a
row, a helper launched beside one, or not reached by any default route of the grid (a candidate for deletion).
"""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "muscle_synergies_amd", "csrc")
DOC = os.path.join(ROOT, "DISPATCH.md")

# kernels that never name a call by themselves: launched beside / inside the path of a named one
HELPERS = {
    "x_to_row_major_kernel": "layout conversion of X, once per fit, when the caller's layout is not the instance's",
    "x_to_channel_major_kernel": "layout conversion of X, once per fit / per preprocessing call",
    "w_convert_kernel": "W between the caller's layout and component-major, once in and once out",
    "wide_w_convert_kernel": "W between the caller's layout and the wide kernels' padded row-major rows",
    "gather_matrices_kernel": "ragged batches: packs the distinct matrices of the descriptors",
    "hupdate_kernel": "row-sliced narrow path: H update from the slice records (after slice_pass_kernel)",
    "reduce_slices_kernel": "time-shard pass: the slices' records summed in order",
    "slice_resid_kernel": "row-sliced narrow path: residual of a slice",
    "resid_finalize_kernel": "row-sliced narrow path: error, stop rule, outputs",
    "colsum_kernel": "shard residual: column sums over the slices",
    "wide_hupdate_kernel": "[sliced] wide path: H update from the slice records",
    "wide_resid_finalize_kernel": "[sliced] wide path: error, stop rule, outputs",
    "big_hht_kernel": "general shapes, two-pass pair (the instances the one-pass kernel does not cover): H H^T per iteration",
    "big_records_kernel": "general shapes, two-pass pair: the slice records (second pass over X)",
    "big_resid_kernel": "general shapes beyond the one-pass instances: residual of a slice",
    "big_hupdate_kernel": "general shapes: H update (+ the next H H^T as per-block partial products on the one-pass path)",
    "big_hht_part_kernel": "general shapes, one-pass path: H H^T partial products before the first iteration",
    "big1_resid_kernel": "general shapes: residual of a slice on the one-pass decomposition",
    "big_resid_finalize_kernel": "general shapes: error, stop rule, outputs",
    "big_pack_sums_kernel": "shard pass on general shapes: records packed for the all-reduce",
    "big_colsum_kernel": "shard residual on general shapes: column sums over the slices",
    "slice_pass_rowlane_kernel": "time-shard pass (hipnmf_shard_pass_f32, 9..16 channels): named by bench.py --config 5, not by last_kernel",
    "random_init_kernel": "hipnmf_random_init_* / the native rank sweeps",
    "gram_kernel": "hipnmf_gram_* (on-device NNDSVD)",
    "nndsvd_stats_kernel": "hipnmf_nndsvd_stats_*",
    "nndsvd_write_kernel": "hipnmf_nndsvd_write_*",
    "env_resample_table_kernel": "time_normalize: interpolation table, once per call",
    "sos_stats_kernel": "sequential / block filter paths: per-series mean",
    "sos_scan_tables_kernel": "time-parallel filter: G and M^(2^j) tables, once per call",
    "sos_block_scan_kernel": "long series: scan over the blocks' end states",
    "diag_stream_kernel": "hipnmf_diag_stream_gbs (bench.py's measured stream ceiling)",
}


def source_kernels():
    names = set()
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hpp", ".hip")):
            continue
        text = open(os.path.join(CSRC, f)).read()
        for m in re.finditer(r"__global__", text):
            tail = text[m.end(): m.end() + 600]
            k = re.search(r"\b([a-z][a-z0-9_]*_kernel[a-z0-9_]*)\s*\(", tail)
            if k:
                names.add(k.group(1))
    return names


def family(name):
    """'fit_wide4_kernel<64,2,12,2,0>[sliced]' -> {'fit_wide4_kernel'}; 'emg_prefix_kernel+emg_output_kernel' -> both."""
    return set(re.findall(r"[a-z][a-z0-9_]*_kernel[a-z0-9_]*", name))


def parse_doc():
    rows, unreached = set(), None
    section = None
    for line in open(DOC):
        if line.startswith("## "):
            section = line[3:].strip()
        if line.startswith("|") and section and not section.startswith(("Helper", "Compiled")):
            for cell in line.strip().strip("|").split("|"):
                if "`" in cell:  # kernel names are the back-quoted cells
                    rows |= family(cell)
        if section and section.startswith("Compiled") and line.startswith("- `"):
            unreached = (unreached or set()) | {line.split("`")[1]}
    return rows, (unreached or set())


def check():
    src = source_kernels()
    named, unreached_doc = parse_doc()
    unknown = named - src
    unreached = src - named - set(HELPERS)
    errs = []
    if unknown:
        errs.append(f"DISPATCH.md names kernels the sources no longer define: {sorted(unknown)}")
    if set(HELPERS) - src:
        errs.append(f"tools/dispatch_table.py lists helpers the sources no longer define: {sorted(set(HELPERS) - src)}")
    if unreached != unreached_doc:
        errs.append(f"'Compiled but not reached' section out of date: sources say {sorted(unreached)}, DISPATCH.md says {sorted(unreached_doc)}"
                    " -- regenerate on the GPU box (tools/dispatch_table.py)")
    return errs


def generate():
    import numpy as np
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.preprocess import design_sos, emg_envelope_batched, sosfilt_batched

    h = _lib.get_handle(0)
    out = ["# DISPATCH — which kernel serves which call by default", "",
           "Generated by `tools/dispatch_table.py` on an MI355X (gfx950, %d CUs) from real calls with default tuning: every name is what"
           % torch.cuda.get_device_properties(0).multi_processor_count,
           "`hipnmf_last_kernel` reported.  Regenerate after any change to a dispatcher; `tests/test_dispatch_doc.py` (CPU) checks this file",
           "against the `__global__` functions of `muscle_synergies_amd/csrc`.  `[sliced]`: one launch per phase and iteration over row",
           "slices of the matrices, replayed as a graph of kernel nodes; otherwise one launch per fit.  X is handed over row-major (C order);",
           "shapes whose instance streams the other order convert once per fit (`x_to_*_kernel`).", ""]
    shapes = [(4, 2), (6, 3), (8, 4), (8, 8), (12, 4), (16, 5), (16, 8), (24, 6), (32, 8), (32, 12), (48, 6), (64, 8), (64, 16), (96, 24),
              (128, 8), (128, 32), (100, 40), (200, 12), (256, 16), (300, 20), (512, 32), (512, 64)]
    calls = [(200, 1), (200, 2048), (2000, 320), (10000, 1), (10000, 320), (200000, 1)]
    out += ["## Solver: `hipnmf_fit_batched_*` (`fit_batched`, `HipNMF`, `find_synergies`)", ""]
    for dtype in (torch.float32, torch.float64):
        for loss in ("frobenius", "kullback-leibler"):
            out += [f"### {str(dtype).split('.')[1]}, beta_loss = '{loss}'", "",
                    "| channels x components | " + " | ".join(f"T = {T}, B = {B}" for T, B in calls) + " |",
                    "|---|" + "---|" * len(calls)]
            for m, k in shapes:
                cells = []
                for T, B in calls:
                    Bq = B
                    while Bq > 1 and Bq * T * m * 8 > (1 << 30):
                        Bq //= 2
                    if k > min(T, m):
                        cells.append("-")
                        continue
                    X = torch.rand((Bq, T, m), device="cuda", dtype=dtype) + 0.01
                    W0 = torch.rand((Bq, T, k), device="cuda", dtype=dtype) + 0.1
                    H0 = torch.rand((Bq, k, m), device="cuda", dtype=dtype) + 0.1
                    ms.fit_batched(X, W0, H0, max_iter=2, tol=0.0, beta_loss=loss)
                    name = h.last_kernel() + (f" (B = {Bq})" if Bq != B else "")
                    cells.append("`" + name + "`")
                    del X, W0, H0
                out.append(f"| {m} x {k} | " + " | ".join(cells) + " |")
            out.append("")
    out += ["## Other solver entry points", "",
            "| entry point | what runs |", "|---|---|",
            "| `hipnmf_fit_ragged_*` (`fit_ragged`, `find_synergies_batched`, `fit_restarts`) | the instance of the table above for the shape, one workgroup per matrix whatever the lengths; "
            "general shapes: one row-sliced fit per trial (`big1_pass_kernel` / `big_pass_w_kernel`) |",
            "| `hipnmf_rank_sweep_*`, `hipnmf_rank_sweep_stop_*` | `random_init_kernel`, then per rank the instance of the table above |",
            "| `hipnmf_shard_pass / _hupdate / _residual`, `hipnmf_fit_tsharded_*` | up to 32 channels x 8 components: `slice_pass_kernel` (fp32, 9..16 channels: `slice_pass_rowlane_kernel`) + `reduce_slices_kernel`, "
            "`hupdate_kernel`, `slice_resid_kernel` + `colsum_kernel`; beyond, and Kullback-Leibler: the general-shape kernels (`big1_pass_kernel` fp32 Frobenius, `big_pass_w_kernel` + `big_records_kernel` otherwise) |",
            "| `hipnmf_gram_*`, `hipnmf_nndsvd_stats_*`, `hipnmf_nndsvd_write_*`, `hipnmf_random_init_*` | `gram_kernel`, `nndsvd_stats_kernel`, `nndsvd_write_kernel`, `random_init_kernel` |", ""]
    out += ["## Envelope: `hipnmf_emg_envelope_*` (`emg_envelope_batched`, `rms`, `time_normalize`, `normalize`, `zero_center`)", "",
            "| dtype | samples | window | n_out | kernel |", "|---|---|---|---|---|"]
    for dtype in (torch.float32, torch.float64):
        for T in (200, 1000, 4000, 9000, 20000, 20480, 60000, 400000):
            for window in (0, 101, 255, 5001, 20001):
                for n_out in (0, 200):
                    if window >= T:
                        continue
                    raw = torch.randn((4, T, 8), device="cuda", dtype=dtype)
                    emg_envelope_batched(raw, window, reduce_to=n_out or None)
                    out.append(f"| {str(dtype).split('.')[1]} | {T} | {window} | {n_out or 'T'} | `{h.last_kernel()}` |")
    out += ["", "## Spline kinds of `time_normalize`: `hipnmf_resample_weights_*` (`time_normalize(kind='quadratic' | 'cubic')`, `time_normalize_batched`)", "",
            "| dtype | samples | n_out | kind | kernel |", "|---|---|---|---|---|"]
    from muscle_synergies_amd.preprocess import time_normalize_batched
    for dtype in (torch.float32, torch.float64):
        for T, n_out, kind in ((1000, 200, "cubic"), (9000, 200, "quadratic"), (20000, 200, "cubic")):
            raw = torch.rand((4, T, 8), device="cuda", dtype=dtype)
            time_normalize_batched(raw, n_out, kind=kind)
            out.append(f"| {str(dtype).split('.')[1]} | {T} | {n_out} | {kind} | `{h.last_kernel()}` |")
    out += ["", "## IIR filter: `hipnmf_sosfilt_*` (`sosfilt_batched`, `digital_filter`, `linear_envelope`)", "",
            "| dtype | samples | sections | zero_lag | mode | kernel |", "|---|---|---|---|---|---|"]
    for dtype in (torch.float32, torch.float64):
        for T in (300, 2000, 9000, 20000, 60000, 400000):
            for order in (2, 4, 12):
                sos = design_sos("butter", order, 2000, 6)
                for zero_lag in (True, False):
                    for mode in ("exact", "scan"):
                        raw = torch.randn((4, T, 8), device="cuda", dtype=dtype)
                        sosfilt_batched(raw, sos, zero_lag=zero_lag, mode=mode)
                        out.append(f"| {str(dtype).split('.')[1]} | {T} | {len(sos)} | {zero_lag} | {mode} | `{h.last_kernel()}` |")
    open(DOC, "w").write("\n".join(out) + "\n")
    write_tail()


TAIL_MARK = "## Helper kernels (launched beside a named one)"


def env_switches():
    names = set()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hpp", ".hip")):
            names |= set(re.findall(r'getenv\("(HIPNMF_[A-Z0-9_]+)"\)', open(os.path.join(CSRC, f)).read()))
    return sorted(names)


def write_tail():
    """(Re)writes the sections behind the tables: helpers, kernels no default route reaches, environment switches (CPU only)."""
    text = open(DOC).read()
    if TAIL_MARK in text:
        text = text[: text.index(TAIL_MARK)].rstrip("\n") + "\n"
    open(DOC, "w").write(text)
    named, _ = parse_doc()
    src = source_kernels()
    unreached = sorted(src - named - set(HELPERS))
    tail = ["", TAIL_MARK, ""]
    tail += [f"- `{k}` — {v}" for k, v in sorted(HELPERS.items()) if k in src]
    tail += ["", "## Compiled but not reached by any default route of this grid", ""]
    tail += [f"- `{k}`" for k in unreached] if unreached else ["(none)"]
    tail += ["", "## Environment switches", "",
             "Development A/B switches read by the dispatchers (each one's comment in the source holds the measurement that set the default).",
             "None of them is needed in production: the tables above are what runs with none set.", "",
             ", ".join(f"`{n}`" for n in env_switches())]
    open(DOC, "a").write("\n".join(tail) + "\n")
    print(f"wrote {DOC}: {len(named)} kernel families named, {len(unreached)} not reached: {unreached}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--refresh-tail", action="store_true", help="CPU: rewrite the sections behind the tables of an existing DISPATCH.md")
    a = ap.parse_args()
    if a.refresh_tail:
        write_tail()
        sys.exit(0)
    if a.check:
        errs = check()
        print("\n".join(errs) if errs else "DISPATCH.md is in step with the sources")
        sys.exit(1 if errs else 0)
    generate()

#!/usr/bin/env python3
"""Re-derive the narrow-shape routing crossovers of ``hipnmf_route_table`` on the box at hand.

The dispatchers send chip-filling batches of narrow shapes (<= 32 channels, <= 8 components) to the 4x4 matrix-pipe kernels up to a
number of rows per matrix and to the lane mappings beyond it; the thresholds were fitted on one box type and live in ONE table
(``csrc/hipnmf_internal.hpp``, ``struct hipnmf_route_table``; in force: ``hipnmf_routes_describe()`` / ``_lib.routes()``).
This tool measures both kernel families over a grid of row counts -- one child process per routing -- the library's own choice and the two families pinned -- (the pins are read once
per process: HIPNMF_FORCE_WIDE = -1 lanes / 1 matrix pipe), every grid point in it -- and prints, per rule, the measured crossover
next to the value in force, and a ``HIPNMF_ROUTES=...`` string that applies the measured values without a rebuild.

    python3 tools/calibrate_routes.py [--quick] [--out profiles/r06_calibrate_routes.log]

A rule is only re-fitted when the two families really cross inside the grid; otherwise the line says which one won everywhere.
The reference has no counterpart: it calls scikit-learn once per matrix (src/muscle_synergies/analysis.py:862-863, 907-912).
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# rule name -> (dtype, channels, components, loss): one representative shape per rule (the shapes the rules' comments cite)
RULES = {
    "f32_16ch_wide_max_rows": ("float32", 16, 5, "frobenius"),
    "f32_16ch_k7_wide_max_rows": ("float32", 16, 8, "frobenius"),
    "f64_16ch_wide_max_rows": ("float64", 16, 5, "frobenius"),
    "f32_32ch_wide_max_rows": ("float32", 32, 4, "frobenius"),
    "f32_32ch_k5_wide_max_rows": ("float32", 32, 5, "frobenius"),
    "kl_f32_32ch_short_max_rows": ("float32", 32, 5, "kullback-leibler"),
}
GRID = [150, 300, 450, 600, 900, 1200, 1800, 2400, 3600, 5000, 7500, 10000]
GRID_QUICK = [300, 600, 1200, 2400, 5000, 10000]


def child(grid, iters):
    """Time every (rule, rows) point under the route pinned by the environment; one JSON line per point."""
    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.synth import emg_batch_torch

    h = _lib.get_handle(0)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    for rule, (dt, m, k, loss) in RULES.items():
        for T in grid:
            B = max(2 * ncu, min(8192, int(6e6 // T)))  # chip-filling, roughly constant total rows
            X, W0, H0 = emg_batch_torch(B, T=T, m=m, k=k, k_true=min(5, m), device="cuda:0")
            if dt == "float64":
                X, W0, H0 = X.double(), W0.double(), H0.double()
            Xv = X.transpose(1, 2).contiguous()
            best = float("inf")
            for _ in range(3):
                r = ms.fit_batched(Xv, W0, H0, max_iter=iters, tol=0.0, beta_loss=loss)
                best = min(best, r.kernel_ms)
            print(json.dumps({"rule": rule, "T": T, "B": B, "ms": best, "mits": B * iters / best / 1e3, "kernel": h.last_kernel()}), flush=True)
            del X, W0, H0, Xv


def crossover(grid, lanes, matrix):
    """Largest row count up to which the matrix-pipe family is at least as fast (log-interpolated between grid points);
    None when one family wins over the whole grid."""
    wins = [matrix[T] <= lanes[T] for T in grid]
    if all(wins):
        return None, "matrix pipe faster over the whole grid"
    if not any(wins):
        return None, "lane mapping faster over the whole grid"
    last = max(i for i, w in enumerate(wins) if w)
    if last + 1 >= len(grid):
        return None, "matrix pipe faster at the top of the grid"
    import math

    a, b = grid[last], grid[last + 1]
    da = math.log(lanes[a] / matrix[a])       # > 0: matrix ahead
    db = math.log(matrix[b] / lanes[b])       # > 0: lanes ahead
    t = math.exp(math.log(a) + (math.log(b) - math.log(a)) * da / max(da + db, 1e-12))
    note = "" if all(wins[: last + 1]) else " (not monotone: the families cross more than once)"
    return t, "crossover between %d and %d rows%s" % (a, b, note)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--quick", action="store_true", help="six row counts instead of twelve")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    grid = GRID_QUICK if a.quick else GRID
    if a.child:
        return child(grid, a.iters)
    data = {}
    for tag, pin in (("default", "0"), ("lanes", "-1"), ("matrix", "1")):  # the library's own choice, and the two families pinned
        env = dict(os.environ, HIPNMF_FORCE_WIDE=pin)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--iters", str(a.iters)] + (["--quick"] if a.quick else []),
                           capture_output=True, text=True, env=env)
        if r.returncode != 0:
            sys.exit("calibrate_routes: the %s run failed:\n%s" % (tag, r.stderr[-3000:]))
        data[tag] = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    from muscle_synergies_amd import _lib

    in_force = _lib.routes()
    lines, suggest = [], []
    for rule, (dt, m, k, loss) in RULES.items():
        lanes = {d["T"]: d["ms"] for d in data["lanes"] if d["rule"] == rule}
        matrix = {d["T"]: d["ms"] for d in data["matrix"] if d["rule"] == rule}
        dflt = {d["T"]: d["ms"] for d in data["default"] if d["rule"] == rule}
        kl = {d["T"]: d["kernel"] for d in data["lanes"] if d["rule"] == rule}
        km = {d["T"]: d["kernel"] for d in data["matrix"] if d["rule"] == rule}
        kd = {d["T"]: d["kernel"] for d in data["default"] if d["rule"] == rule}
        lines.append("%s  (%s, %d channels, k = %d, %s)  in force: %g rows" % (rule, dt, m, k, loss, in_force[rule]))
        lines.append("    rows       " + " ".join("%8d" % T for T in grid))
        lines.append("    default ms " + " ".join("%8.3f" % dflt[T] for T in grid) + "   " + " | ".join(sorted({v.split("<")[0] for v in kd.values()})))
        lines.append("    lanes   ms " + " ".join("%8.3f" % lanes[T] for T in grid) + "   " + " | ".join(sorted({v.split("<")[0] for v in kl.values()})))
        lines.append("    matrix  ms " + " ".join("%8.3f" % matrix[T] for T in grid) + "   " + " | ".join(sorted({v.split("<")[0] for v in km.values()})))
        t, why = crossover(grid, lanes, matrix)
        # what the library's own choice costs: its time over the better pinned family, at every grid point (other rules -- the
        # one-wave kernel for short matrices in big batches -- act before this threshold does: hence the default's own run)
        worst_T = max(grid, key=lambda T: dflt[T] / min(lanes[T], matrix[T]))
        worst = dflt[worst_T] / min(lanes[worst_T], matrix[worst_T])
        lines.append("    measured: %s%s; the library's own choice is at most %.0f %% behind the better pinned family (at %d rows)" %
                     (why, "" if t is None else " -> %.0f rows" % t, max(0.0, worst - 1.0) * 100, worst_T))
        if t is not None:
            suggest.append("%s=%.0f" % (rule, t))
    lines.append("")
    lines.append("HIPNMF_ROUTES=" + ",".join(suggest) if suggest else "(no rule crosses inside the grid: nothing to re-fit)")
    text = "\n".join(lines)
    print(text)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Benchmarks of the NMF multiplicative-update hot path on MI355X (BASELINE.json metric and configs).

    python bench.py --gpus 1 --steps 3 --warmup 1                       # headline: config #3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config 3|4|5|2]

--config 3 (default, BASELINE.json configs[2], the configuration the metric is quoted on): one *step* = one batched
    fit (``hipnmf_fit_batched_f32`` through the Python host) of 4096 synthetic EMG matrices 16 ch x 10 000 samples
    IN TOTAL, scattered over the GPUs in contiguous runs (BASELINE.json: "batch 4096 ... scattered across 1 -> 8 MI355X":
    strong scaling; at N = 1 all 4096 on the one GPU), k = 5, fp32, ``--iters`` (500) Lee-Seung iterations (tol = 0, so
    exactly that many, as sklearn does) from a fixed ``init='custom'`` W0/H0, inputs resident in HBM.  The factorisations
    are independent: ranks share nothing, no data-path collective.  ``--batch-per-gpu N``: the weak-scaling form.
    ``--dtype f64`` (configs 2/3/4): the same workload in float64, the dtype the reference itself hands to scikit-learn;
    priced against the 8 TB/s HBM line (256 resident matrices do not fit the Infinity Cache), parity gated at 1e-9.
--config 4 (configs[3]): per-trial rank sweep k = 2..8 (500 iterations each, random init drawn on the device, VAF >=
    0.90 selection) over 1024 trials IN TOTAL, scattered by trial over the ranks (strong scaling, no collective).
--config 5 (configs[4]): ONE matrix 16 x 2e8 (``--T5``), rows sharded over the ranks (strong scaling), generated shard
    by shard on the device from counter seeds; one step = ``--iters5`` (20) iterations of the time-sharded solver:
    ``hipnmf_shard_pass`` on every sub-shard of the rank, ONE packed all-reduce of k*m + k*k = 105 floats over RCCL,
    ``hipnmf_shard_hupdate``.  A unit is one iteration of the whole matrix (20.8 GB of algorithmic traffic).
--config 2 (configs[1]): one 16 x 10 000 matrix, 500 iterations per step (cooperative kernel); N > 1 = replicas.

Rank 0 prints ONE JSON line (fields: see the task contract) including
  roofline     -- config 3/4/2: fp32 issue roof (f32 MFMA = f32 VALU = 157.3 TFLOP/s): algorithmic FLOPs per launch /
                  HIP-event duration of the solver kernel, plus ``memory``: the byte model (algorithmic bytes, bytes the
                  design really moves, the measured Infinity-Cache stream rate that binds); config 5: HBM.
  cpu_baseline -- scikit-learn's NMF(solver='mu') (the reference's arithmetic) on this host's cores, bounded sample,
                  N = 1 only; run BEFORE the GPU is initialised (worker processes).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md:36); 6290 GB/s measured copy
FP32_PEAK_TFLOPS = 157.3   # fp32 vector = fp32 matrix peak (MI355X_MICROARCH.md:41-42)
STREAM_PEAK_GBS = 7800.0   # fallback only: the ceiling is MEASURED in the run (hipnmf_diag_stream_gbs: one workgroup per CU re-reading
                           # "its" 640 KB region, no arithmetic; 7.2 - 8.1 TB/s in rounds 2-3, tools/ubench/mall_stream.hip)
WIDE_STREAM_GBS = 5880.0   # measured: the wide-shape kernel's traffic (256 B of X non-temporal + 32 B of W read + 32 B of W written
                           # per row, 1024 matrices of 64 x 10 000) with no arithmetic: 5.85 - 5.89 TB/s; 4.97 with the default
                           # cache policy on X; X alone 6.3 (7.0 non-temporal) (tools/ubench/wide_stream.hip, profiles/r03_ubench_wide_stream.log)
SHARD_STREAM_GBS = 5100.0  # measured: the shard pass's traffic (64 B X + 20 B W read, 20 B W written per row, channel-
                           # major, 2.5e7 rows) with no arithmetic: 5.0 - 5.2 TB/s (tools/ubench/shard_stream.hip)


def parse_args():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5], help="BASELINE.json configuration (1-based)")
    ap.add_argument("--batch", type=int, default=None, help="config 3: matrices per GPU (4096); config 4: trials in total (1024)")
    ap.add_argument("--iters", type=int, default=500, help="mu iterations per fit (configs 2-4)")
    ap.add_argument("--T", type=int, default=10_000)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32",
                    help="arithmetic type of configs 2/3/4: f32 = BASELINE.json's metric; f64 = what the reference itself hands to "
                         "scikit-learn (the Vicon loader builds float64 frames: vicon_data/user_data.py:391-396, analysis.py:862-863)")
    ap.add_argument("--batch-per-gpu", type=int, default=None,
                    help="config 3: the weak-scaling form (this many matrices on EVERY GPU); default: --batch (4096) matrices "
                         "IN TOTAL scattered over the GPUs, BASELINE.json's wording (strong scaling)")
    ap.add_argument("--force-nccl", action="store_true",
                    help="config 5: open the RCCL ('nccl') process group even with one rank, so the per-iteration all-reduce "
                         "really executes on the GPU (world size 1)")
    ap.add_argument("--T5", type=int, default=200_000_000, help="config 5: rows of the single long matrix")
    ap.add_argument("--iters5", type=int, default=20, help="config 5: iterations per step")
    ap.add_argument("--subshard", type=int, default=25_000_000, help="config 5: rows per sub-shard (< 2 GiB of X each)")
    ap.add_argument("--threads", type=int, default=0, help="workgroup size override (0 = library default)")
    ap.add_argument("--x-layout", choices=["row", "channel"], default="row",
                    help="config 3: memory order of the X batch handed to the engine: [B][T][m] (C order) or [B][m][T]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-orchestration", action="store_true",
                    help="no GPU work: the rank plumbing only (spawn / rendezvous / barrier / max over ranks / one JSON line) "
                         "over gloo on the CPU; the line is marked as such and is not a measurement")
    ap.add_argument("--cpu-sample", type=int, default=0, help="matrices in the CPU baseline sample (0 = auto)")
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)  # tests: this rank dies before the rendezvous
    ap.add_argument("--no-host-resident", action="store_true", help="config 3, N = 1: skip the host-resident end-to-end measurement")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity check of the last timed step's output")
    ap.add_argument("--launch-timeout", type=float, default=3600.0, help="--gpus N launcher: seconds before the ranks are ended")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# CPU baseline: sklearn (the dependency that holds the reference's NMF arithmetic).  Runs before anything touches
# the GPU.  (i) one matrix, 1 BLAS thread and all cores, best of 3; (ii) one worker process per core x 1 BLAS thread.
def _cpu_worker(job):
    seeds, T, m, k, iters, dt_name = job
    import warnings

    import numpy as np
    from threadpoolctl import threadpool_limits

    from muscle_synergies_amd.synth import emg_matrix, random_init

    warnings.simplefilter("ignore")
    from sklearn.decomposition import NMF

    data = []
    for s in seeds:
        X = emg_matrix(s, T=T, m=m, dtype=np.dtype(dt_name))
        W0, H0 = random_init(X, k, s)
        data.append((X, W0, H0))
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        for X, W0, H0 in data:
            NMF(k, solver="mu", init="custom", tol=0, max_iter=iters).fit_transform(X, W=W0, H=H0)
        dt = time.perf_counter() - t0
    return len(seeds), dt


def _cpu_single(job):
    T, m, k, iters, nthreads, dt_name = job
    import warnings

    import numpy as np
    from threadpoolctl import threadpool_limits

    from muscle_synergies_amd.synth import emg_matrix, random_init

    warnings.simplefilter("ignore")
    from sklearn.decomposition import NMF

    X = emg_matrix(0, T=T, m=m, dtype=np.dtype(dt_name))
    W0, H0 = random_init(X, k, 0)
    best = float("inf")
    with threadpool_limits(limits=nthreads):
        for _ in range(3):
            t0 = time.perf_counter()
            NMF(k, solver="mu", init="custom", tol=0, max_iter=iters).fit_transform(X, W=W0.copy(), H=H0.copy())
            best = min(best, time.perf_counter() - t0)
    return iters / best


def cpu_baseline(a):
    import multiprocessing as mp

    try:
        import sklearn
    except Exception as e:  # noqa: BLE001
        return {"value": None, "unit": "matrix-iterations/s", "cores": 0, "kind": "reference",
                "sample": f"sklearn not importable: {e}"}
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:  # noqa: BLE001
        pass
    n = a.cpu_sample or max(64, cores)
    n = (n + cores - 1) // cores * cores
    per = n // cores
    dt_name = "float64" if a.dtype == "f64" else "float32"
    jobs = [(list(range(1000 + w * per, 1000 + (w + 1) * per)), a.T, a.m, a.k, a.iters, dt_name) for w in range(cores)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:  # single-matrix numbers in a fresh process each
        one = pool.apply(_cpu_single, ((a.T, a.m, a.k, a.iters, 1, dt_name),))
        allc = pool.apply(_cpu_single, ((a.T, a.m, a.k, a.iters, cores, dt_name),))
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        out = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    slowest = max(dt for _, dt in out)
    total = sum(c for c, _ in out)
    return {
        "value": total * a.iters / slowest,
        "unit": "matrix-iterations/s",
        "cores": cores,
        "kind": "reference",
        "single_matrix_1_thread": one,
        "single_matrix_all_cores": allc,
        "sample": (f"scikit-learn {sklearn.__version__} NMF(solver='mu', init='custom', tol=0, max_iter={a.iters}) "
                   f"on {total} of the synthetic {a.m}x{a.T} k={a.k} {dt_name} matrices, {cores} worker processes x 1 BLAS "
                   f"thread; fit time of the slowest worker {slowest:.2f} s (pool wall {wall:.1f} s incl. data "
                   f"generation).  One matrix alone, best of 3: {one:.0f} it/s with 1 BLAS thread, {allc:.0f} it/s "
                   f"with {cores} BLAS threads"),
    }


# ------------------------------------------------------------------------------------------------
# In-run parity: a few matrices of the LAST timed step's output against scikit-learn (the dependency that holds the
# reference's arithmetic, sklearn/decomposition/_nmf.py:731-893) from the same W0/H0, in CPU worker processes that
# were started before this process touched the GPU.  Checker leg only: nothing here is timed or shipped.
PARITY_TOL_F64 = 1e-9  # float64 calls (the reference's own dtype): DESIGN.md section 6
PARITY_TOL = 1e-5  # BASELINE.json north_star: 1e-5 relative Frobenius error; SURVEY.md 8c: |d(WH)|_F / |X|_F and |d err| / |X|_F


def _parity_noop(_):
    try:
        import sklearn.decomposition  # noqa: F401 -- page the import in while the GPU works
    except Exception:  # noqa: BLE001
        pass
    return 0


def _parity_worker(job):
    X, W0, H0, iters, loss = job
    import warnings

    import numpy as np

    warnings.simplefilter("ignore")
    try:
        from sklearn.decomposition import NMF

        mdl = NMF(H0.shape[0], solver="mu", init="custom", tol=0, max_iter=iters, beta_loss=loss)
        W = mdl.fit_transform(X, W=W0.copy(), H=H0.copy())
        if X.dtype == np.float64:
            return "scikit-learn", W, mdl.components_, float(mdl.reconstruction_err_), None
        # the same fit in float64: how far scikit-learn's own float32 run is from it (the rounding noise of this case)
        m64 = NMF(H0.shape[0], solver="mu", init="custom", tol=0, max_iter=iters, beta_loss=loss)
        W64 = m64.fit_transform(X.astype(np.float64), W=W0.astype(np.float64), H=H0.astype(np.float64))
        return "scikit-learn", W, mdl.components_, float(mdl.reconstruction_err_), W64 @ m64.components_
    except ImportError:
        from oracle import nmf_mu_oracle as orc  # the checker may stand in for the absent dependency

        r = orc.nmf_mu_fit(X, W0, H0, max_iter=iters, tol=0.0)
        return "oracle", np.asarray(r["W"]), np.asarray(r["H"]), float(r["reconstruction_err"]), None


class ParityChecker:
    def __init__(self, enabled):
        self.pool = None
        if enabled:
            import multiprocessing as mp

            self.pool = mp.get_context("spawn").Pool(4)
            self._warm = self.pool.map_async(_parity_noop, range(4))

    def check(self, jobs, ours, names=None):
        """jobs: [(X [T, m], W0, H0, iters, loss)] NumPy; ours: [(W or None, H, err)].  Returns the ``parity`` object.

        The gate (BASELINE.json north_star: "within 1e-5 relative Frobenius error" of sklearn at the same iteration count) is the
        distance to scikit-learn's fit IN THE DTYPE OF THE CALL: float32 inputs against its float32 fit at 1e-5, float64 inputs
        against its float64 fit at 1e-9.  For float32 the distance to scikit-learn's float64 fit of the same case is reported
        beside it.  A float32 case beyond 1e-5 of the float32 fit is an EXCEPTION, listed by name with all three distances, and
        only tolerated when scikit-learn's own float32 fit is demonstrably not a 1e-5 yardstick for it (its float32 and float64
        fits of that case differ by more than the tolerance) while this engine is within the tolerance of the float64 fit."""
        import numpy as np

        if self.pool is None:
            return None
        ref = self.pool.map(_parity_worker, jobs)
        f64 = all(j[0].dtype == np.float64 for j in jobs)
        tol = PARITY_TOL_F64 if f64 else PARITY_TOL
        d_wh, d_err, d_h, checker = 0.0, 0.0, 0.0, None
        per_case, noise, ours64, exceptions = [], 0.0, 0.0, []
        for n, ((X, _w0, _h0, _it, _loss), (W, H, err), (who, Wr, Hr, err_r, WH64)) in enumerate(zip(jobs, ours, ref)):
            checker = who
            xn = float(np.linalg.norm(X.astype(np.float64)))
            Hd = H.astype(np.float64)
            WHr = Wr.astype(np.float64) @ Hr.astype(np.float64)
            case = {"name": names[n] if names else f"case {n}"}
            gate_ok = True
            if W is not None:
                WHo = W.astype(np.float64) @ Hd
                case["rel_dWH"] = float(np.linalg.norm(WHo - WHr)) / xn
                d_wh = max(d_wh, case["rel_dWH"])
                gate_ok = case["rel_dWH"] <= tol
                if WH64 is not None and not f64:
                    case["rel_dWH_vs_float64_fit"] = float(np.linalg.norm(WHo - WH64)) / xn
                    case["checker_float32_vs_its_float64_fit"] = float(np.linalg.norm(WHr - WH64)) / xn
                    ours64, noise = max(ours64, case["rel_dWH_vs_float64_fit"]), max(noise, case["checker_float32_vs_its_float64_fit"])
            d_h = max(d_h, float(np.linalg.norm(Hd - Hr) / max(np.linalg.norm(Hr), 1e-300)))
            case["rel_derr"] = abs(float(err) - err_r) / xn
            d_err = max(d_err, case["rel_derr"])
            case["within_tol_of_checker_fit_in_call_dtype"] = bool(gate_ok)
            ok = gate_ok
            if not gate_ok and "rel_dWH_vs_float64_fit" in case:
                tolerated = (case["checker_float32_vs_its_float64_fit"] > tol and case["rel_dWH_vs_float64_fit"] <= tol)
                exceptions.append({"name": case["name"], "rel_dWH_vs_float32_fit": case["rel_dWH"],
                                   "rel_dWH_vs_float64_fit": case["rel_dWH_vs_float64_fit"],
                                   "checker_float32_vs_its_float64_fit": case["checker_float32_vs_its_float64_fit"],
                                   "tolerated": bool(tolerated),
                                   "why": "scikit-learn's float32 fit of this case is itself further than the tolerance from its float64 "
                                          "fit, and this engine is within the tolerance of the float64 fit" if tolerated else
                                          "beyond the tolerance of the float32 fit with no such excuse"})
                ok = tolerated
            case["ok"] = bool(ok and case["rel_derr"] <= tol)
            per_case.append(case)
        ok = bool(all(c["ok"] for c in per_case) and np.isfinite(d_wh) and np.isfinite(d_err))
        out = {"n_checked": len(jobs), "dtype": "float64" if f64 else "float32", "max_rel_dWH": d_wh, "max_rel_derr": d_err,
               "max_rel_dH": d_h, "tol": tol, "ok": ok,
               "all_within_tol_of_checker_fit_in_call_dtype": bool(all(c["within_tol_of_checker_fit_in_call_dtype"] for c in per_case)),
               "exceptions": exceptions, "checker": checker, "per_case": per_case,
               "what": "matrices of the last timed step's output vs the checker's fit from the same W0/H0 at the same iteration "
                       "count and IN THE SAME DTYPE: |W H - W_ref H_ref|_F / |X|_F and |err - err_ref| / |X|_F per case, gated at tol; "
                       "`exceptions` names every case that is not within tol of that fit (max_rel_dH is informational: the "
                       "factors themselves drift ~1e-3 at 500 fp32 iterations under ANY change of summation order, SURVEY.md 8c)"}
        if not f64:
            out["float32_rounding_noise"] = {
                "checker_float32_vs_its_own_float64_fit_max_rel_dWH": noise, "ours_vs_the_float64_fit_max_rel_dWH": ours64,
                "note": "how far scikit-learn's float32 run is from scikit-learn's float64 run of the same case, and how far "
                        "this engine is from that float64 run: the scale of float32 rounding at this iteration count"}
        return out

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool.join()
            self.pool = None


# ------------------------------------------------------------------------------------------------
class Ctx:
    """Rank / device / process-group plumbing shared by the configurations."""

    def __init__(self, a):
        self.a = a
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.force_nccl = bool(a.force_nccl and a.config == 5 and not a.dry_orchestration)
        self.distributed = self.world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or self.force_nccl  # under torchrun: always a group
        self.dry = bool(a.dry_orchestration)

    def init_gpu(self):
        import torch

        self.torch = torch
        self.backend = None
        if self.dry:  # rank plumbing only: gloo on the CPU, nothing touches a GPU
            self.dev = torch.device("cpu")
        else:
            # every rank checks ITS device (the launcher parent stays free of GPU libraries) and exits non-zero without one
            n = torch.cuda.device_count()
            if n <= self.local_rank or not torch.cuda.is_available():
                raise SystemExit(f"bench.py rank {self.rank}: needs ROCm GPU index {self.local_rank}, {n} visible (the engine has "
                                 f"no CPU fallback; a {self.world}-GPU number is never reported from fewer devices)")
            torch.cuda.set_device(self.local_rank)
            self.dev = torch.device("cuda", self.local_rank)
        if self.distributed:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # RCCL only where the data path has a collective (config 5: one all-reduce per iteration).  Configs 2/3/4 share
            # nothing between ranks: their timing barrier and the max over ranks of one float64 go over gloo on the CPU,
            # so the scaling record of the embarrassingly parallel configurations does not depend on RCCL coming up.
            self.backend = "nccl" if (self.a.config == 5 and not self.dry) else "gloo"
            if self.backend == "nccl" and "RANK" not in os.environ:  # --force-nccl without a launcher: a world of one
                dist.init_process_group(backend="nccl", device_id=self.dev, rank=0, world_size=1,
                                        init_method=f"tcp://127.0.0.1:{_free_port()}")
            elif self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=self.dev)
            else:
                dist.init_process_group(backend="gloo")
            self.dist = dist

    def sync(self):
        if not self.dry:
            self.torch.cuda.synchronize(self.dev)

    def _coll_dev(self):
        return self.dev if self.backend == "nccl" else self.torch.device("cpu")

    def barrier(self):
        """Device work of every rank finished, then all ranks met, then (nccl: the barrier's own kernel) drained."""
        self.sync()
        if self.distributed:
            self.dist.barrier()
        self.sync()

    def ranks_seen(self):
        """Number of ranks that took part (sum of ones over the group): in the JSON line as evidence of the launch."""
        if not self.distributed:
            return 1
        t = self.torch.ones(1, dtype=self.torch.int64, device=self._coll_dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def timed(self, step):
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize, max over ranks."""
        a = self.a
        for _ in range(a.warmup):
            step()
        self.barrier()
        t0 = time.perf_counter()
        outs = [step() for _ in range(a.steps)]
        self.sync()
        elapsed = time.perf_counter() - t0
        self.barrier()
        if self.distributed:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self._coll_dev())
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, outs

    def finish(self):
        if self.distributed:
            self.dist.destroy_process_group()


def _traffic(kernel, **key):
    """Bytes per launch measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for exactly this kernel instance and
    workload (tools/measure_traffic.sh writes profiles/traffic.json); None when no matching measurement is committed."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(tpath) as f:
            entries = json.load(f)
        for e in entries if isinstance(entries, list) else [entries]:
            if e.get("kernel") == kernel and all(e.get(k) == v for k, v in key.items()):  # (the instance AND the workload)
                _traffic.source = {"file": "profiles/traffic.json", "measured_round": e.get("measured_round"),
                                   "source_commit": e.get("source_commit"), "method": e.get("method")}
                return e.get("l2_fabric_bytes_per_launch")
    except Exception:  # noqa: BLE001
        pass
    return None


_traffic.source = None


def flops_per_unit(T, m, k):
    """One mu iteration of one matrix: 4 T k (m + k) + 2 T k + 4 k^2 m + 2 k m."""
    return 4 * T * k * (m + k) + 2 * T * k + 4 * k * k * m + 2 * k * m


def _matrix_pipe(kernel, tf, m, k):
    """Useful vs issued rate of the matrix pipe for the wide-shape kernels: issued = useful x the padding of the tile shape."""
    if kernel.startswith("fit_wide4_kernel"):  # nmf_wide4.hpp: v_mfma_f32_4x4x1, components padded to a multiple of 4
        kp = (k + 3) // 4 * 4
        mp = 16 if m <= 16 else 32 if m <= 32 else 48 if m <= 48 else 64 if m <= 64 else 96 if m <= 96 else 128
        return {"achieved_tflops_useful": tf, "peak_tflops": FP32_PEAK_TFLOPS, "frac_useful": tf / FP32_PEAK_TFLOPS,
                "issued_tflops": tf * (kp / k) * (mp + kp) / (m + k),
                "note": "all four contractions on v_mfma_f32_4x4x1_16b_f32, components padded to a multiple of 4, channels to "
                        "16 / 32 / 48 / 64 / 96 / 128 (issued = useful x padding); profiles/r03_pmc_fit_wide4_64_8.txt: matrix pipe busy 31 %"}
    return {"achieved_tflops_useful": tf, "peak_tflops": FP32_PEAK_TFLOPS, "frac_useful": tf / FP32_PEAK_TFLOPS,
            "issued_tflops": tf * (16.0 / k) * ((m + 15) // 16 * 16 + 16.0) / (m + k),
            "note": "all four contractions on v_mfma_f32_16x16x4_f32 with components padded to 16 and "
                    "channels to a multiple of 16 (issued = useful x padding)"}


def _bound_detail(kernel):
    """Which pipe does the fp32 work of this kernel family (f32 VALU peak = f32 MFMA peak = 157.3 TFLOP/s on gfx950)."""
    if kernel.startswith("fit_rowlane") or "rowlane" in kernel:
        return ("fp32 issue: nmf_rowlane.hpp puts the two T-long contractions on v_mfma_f32_4x4x1_16b_f32 and keeps the row-local "
                "products on the VALU; both pipes peak at 157.3 TFLOP/s fp32")
    if kernel.startswith(("fit_persistent", "fit_small", "fit_coop", "slice_pass_kernel")):
        return ("fp32 issue, VALU only: nmf_kernels.hpp / nmf_small.hpp use v_fma_f32 / v_pk_fma_f32 with DPP reductions and no "
                "MFMA (k <= 5 x 16 channels lost the round-2 shoot-out on the matrix pipe); f32 VALU peak = f32 MFMA peak = "
                "157.3 TFLOP/s on gfx950")
    return "fp32 issue (f32 MFMA = f32 VALU = 157.3 TFLOP/s on gfx950)"


FP64_PEAK_TFLOPS = 78.6    # fp64 vector = fp64 matrix rate of gfx950 (half the packed-fp32 rate; AMD's MI355X data sheet)


def roofline_f64_narrow(kernel, kernel_ms, units_per_launch, T, m, k, traffic, stream_gbs):
    """float64 on the narrow (<= 32 channel) shapes: HBM-bound.  256 resident matrices x 8 T m bytes (328 MB at 16 x 10 000) do
    not fit the 256 MiB Infinity Cache, so -- unlike the fp32 headline -- X comes from HBM every iteration and the 8 TB/s
    line is the roof.  `achieved` = ALGORITHMIC bytes (X once, W read + written: 8 T (m + 2k) per unit) / kernel time."""
    fl = flops_per_unit(T, m, k)
    by = 8 * T * (m + 2 * k)
    sec = kernel_ms * 1e-3
    gbs = by * units_per_launch / sec / 1e9
    tf = fl * units_per_launch / sec / 1e12
    mem = {"algorithmic_bytes_per_unit": by, "hbm_peak_gbs": HBM_PEAK_GBS, "l2_fabric_bytes_per_launch": traffic,
           "note": ("rows of W that fit the workgroup's LDS (40 bytes per row at k = 5) never leave it, so the bytes really "
                    "moved are below the algorithmic count: l2_fabric_bytes_per_launch (PMC) when a measurement is committed")}
    if traffic:
        mem["l2_fabric_source"] = _traffic.source
        mem["l2_fabric_gbs"] = traffic / sec / 1e9
        mem["l2_fabric_over_algorithmic"] = traffic / (by * units_per_launch)
    if stream_gbs:
        mem["stream_peak_gbs_measured_in_this_run"] = stream_gbs
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": kernel, "kernel_ms_avg": kernel_ms, "algorithmic_bytes_per_unit": by, "flops_per_unit": fl,
            "units_per_launch": units_per_launch,
            "fp64_issue": {"achieved_tflops": tf, "peak_tflops": FP64_PEAK_TFLOPS, "frac": tf / FP64_PEAK_TFLOPS,
                           "note": "v_fma_f64 on the VALU (lane mapping G = 4, CH = 4: DPP reduce-scatter inside quads)"},
            "memory": mem}


def compute_roofline(kernel, kernel_ms, units_per_launch, T, m, k, traffic=None, moved_bytes_per_unit=None, stream_gbs=None, esize=4):
    """``stream_gbs``: the memory system's rate for the solver's access pattern measured in this run (None: the committed constant)."""
    if esize == 8 and not (m > 32 or k > 8):
        return roofline_f64_narrow(kernel, kernel_ms, units_per_launch, T, m, k, traffic, stream_gbs)
    fl = flops_per_unit(T, m, k)
    peak_stream = stream_gbs if stream_gbs else STREAM_PEAK_GBS
    by = esize * T * (m + 2 * k)  # read X once, read + write W once
    sec = kernel_ms * 1e-3
    tf = fl * units_per_launch / sec / 1e12
    mem = {
        "algorithmic_bytes_per_unit": by,
        "algorithmic_gbs": by * units_per_launch / sec / 1e9,
        "hbm_peak_gbs": HBM_PEAK_GBS,
        "l2_fabric_bytes_per_launch": traffic,
        "note": ("the 256 matrices being worked on (~190 MB) sit in the 256 MiB Infinity Cache and most rows of W in "
                 "LDS, so algorithmic bytes / time may exceed the HBM line; what binds is the rate at which the XCDs "
                 "can re-read their matrices: stream_peak_gbs, measured by tools/ubench/mall_stream.hip"),
        "stream_peak_gbs": peak_stream,
    }
    if "stream_peak_gbs" in mem:
        mem["stream_peak_source"] = ("measured in this run after the timed region: hipnmf_diag_stream_gbs (one workgroup per CU reading its "
                                     "own region of one matrix's size, the batch's number of regions, 16-byte loads, no arithmetic)"
                                     if stream_gbs else
                                     "tools/ubench/mall_stream.hip, profiles/r02_ubench_mall_stream.log (round 2, commit 6cd19eb); a constant, not re-measured in this run")
    if traffic:
        mem["l2_fabric_source"] = _traffic.source  # a PMC measurement committed under profiles/, not taken in this run
        mem["l2_fabric_gbs"] = traffic / sec / 1e9
        mem["frac_of_stream_peak"] = traffic / sec / 1e9 / peak_stream
    elif moved_bytes_per_unit:
        mem["design_bytes_per_unit"] = moved_bytes_per_unit
        mem["design_gbs"] = moved_bytes_per_unit * units_per_launch / sec / 1e9
        mem["frac_of_stream_peak"] = mem["design_gbs"] / peak_stream
    if m > 32 or k > 8:
        # wide shapes (nmf_wide.hpp): the batch does not fit the Infinity Cache (2.56 MB of X per 64-channel matrix, one
        # matrix per workgroup, several hundred in flight) and W streams too: HBM-bound, reported against the 8 TB/s line
        mem.pop("note"), mem.pop("stream_peak_gbs"), mem.pop("stream_peak_source", None)
        gbs = by * units_per_launch / sec / 1e9
        mem["measured_stream_ceiling_gbs"] = WIDE_STREAM_GBS
        mem["frac_of_measured_ceiling"] = gbs / WIDE_STREAM_GBS
        mem["measured_stream_ceiling_source"] = ("tools/ubench/wide_stream.hip, profiles/r03_ubench_wide_stream.log: the same bytes "
                                                 "per row (X read non-temporal, W read and written back) with no arithmetic, at the "
                                                 "64-channel k = 8 mix; a constant, not re-measured in this run")
        if fl / (FP32_PEAK_TFLOPS * 1e12) > by / (HBM_PEAK_GBS * 1e9):
            # beyond the ridge (157.3 TFLOP/s / 8 TB/s = 19.7 flop per algorithmic byte; e.g. 512 channels x 32 components: 30):
            # the dense fp32 matrix peak is the roof, the stream is reported beside it
            mem["frac_of_hbm_peak"] = gbs / HBM_PEAK_GBS
            return {"bound": "mfma", "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS,
                    "traffic": traffic, "kernel": kernel, "kernel_ms_avg": kernel_ms, "flops_per_unit": fl,
                    "algorithmic_bytes_per_unit": by, "units_per_launch": units_per_launch,
                    "matrix_pipe": _matrix_pipe(kernel, tf, m, k), "memory": mem}
        return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "traffic": traffic, "kernel": kernel, "kernel_ms_avg": kernel_ms, "algorithmic_bytes_per_unit": by,
                "units_per_launch": units_per_launch,
                "matrix_pipe": _matrix_pipe(kernel, tf, m, k),
                "memory": mem}
    return {
        "bound": "fp32_issue",
        "bound_detail": _bound_detail(kernel),
        "achieved": tf,
        "peak": FP32_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": tf / FP32_PEAK_TFLOPS,
        "traffic": traffic,
        "kernel": kernel,
        "kernel_ms_avg": kernel_ms,
        "flops_per_unit": fl,
        "units_per_launch": units_per_launch,
        "memory": mem,
    }


def lds_rows_of_w(k, threads=512):
    """Rows of W the one-workgroup-per-matrix kernels keep in LDS (hipnmf_api.hip): the design's own byte count."""
    nw = threads // 64
    nacc = 16 * k + k * (k + 1) // 2
    base = 4 * (2 * k * 16 + 2 * k * k + nw * max(nacc, 33) + 8)
    base = (base + 15) // 16 * 16
    return (160 * 1024 - base) // (4 * k) // 64 * 64  # whole 64-row wave tiles (round 2: whole workgroup-steps)


# ------------------------------------------------------------------------------------------------ config 3 / 2
def run_batch(cx, single):
    a, torch = cx.a, cx.torch
    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.synth import emg_batch_torch

    from muscle_synergies_amd.engine import partition

    wide = a.m > 32 or a.k > 8
    f64 = a.dtype == "f64"
    # config 3: BASELINE.json's wording is "batch 4096 ... scattered across 1 -> 8 MI355X": --batch matrices IN TOTAL, contiguous
    # runs of them per rank (strong scaling); --batch-per-gpu N keeps N matrices on every GPU instead (weak scaling)
    weak = bool(a.batch_per_gpu) and not single
    total = a.batch or (4096 if not wide else max(256, 4096 * 16 // max(a.m, 16)))
    if single:
        B = 1
    elif weak:
        B = a.batch_per_gpu
    else:
        lo_b, hi_b = partition(total, cx.world)[cx.rank]
        B = hi_b - lo_b
    # synthetic workload, generated on the device (seeded per rank).  X is handed over as [B, T, m] in C order
    # (row-major, what sklearn itself takes): the layout the fp32 16-channel kernels stream in place.  The
    # [B, m, T] storage (a DataFrame's F order) is timed too: the engine then converts it once per fit.
    X, W0, H0 = emg_batch_torch(B, T=a.T, m=a.m, k=a.k, device=cx.dev, seed=cx.rank)
    if f64:  # the same values as float64: what DataFrame.to_numpy() hands to the estimator in the reference
        X, W0, H0 = X.double(), W0.double(), H0.double()
    Xc = X.transpose(1, 2)  # logical [B, T, m] view of channel-major storage
    Xr = Xc.contiguous()
    handle = _lib.get_handle(cx.local_rank)
    if a.threads:
        handle.set_tuning(a.threads, 0, 0)
    Xv = Xr if a.x_layout == "row" else Xc
    kernel_ms = []

    def step():
        r = ms.fit_batched(Xv, W0, H0, max_iter=a.iters, tol=0.0, device=cx.dev, handle=handle)
        kernel_ms.append(r.kernel_ms)
        return r

    elapsed, outs = cx.timed(step)
    kernel = handle.last_kernel()
    kernel_ms = kernel_ms[a.warmup:]
    other = None
    if cx.rank == 0 and not single:  # the other input layout, one untimed + one timed fit, reported beside `value`
        Xo = Xc if a.x_layout == "row" else Xr
        ms.fit_batched(Xo, W0, H0, max_iter=a.iters, tol=0.0, device=cx.dev, handle=handle)
        torch.cuda.synchronize(cx.dev)
        t0 = time.perf_counter()
        ms.fit_batched(Xo, W0, H0, max_iter=a.iters, tol=0.0, device=cx.dev, handle=handle)
        torch.cuda.synchronize(cx.dev)
        other = B * a.iters / (time.perf_counter() - t0)
    if cx.rank != 0:
        return None
    r = outs[-1]
    host_res = None
    if cx.world == 1 and not single and not a.no_host_resident:
        # the reference's input is host memory (a DataFrame, analysis.py:739-746): the same batch from NumPy arrays to NumPy
        # results -- upload, fit and download through the chunked transfer pipeline of fit_batched; one untimed, one timed call
        import numpy as np

        Xh, Wh, Hh = Xr.cpu().numpy(), W0.cpu().numpy(), H0.cpu().numpy()
        # (a) the reusable form: the caller's arrays page-locked once (ms.HostBatch), first call untimed, SECOND call timed --
        # what a rank range / restarts / any repeated fit of the same host batch gets on every call after the first
        t0 = time.perf_counter()
        hb = ms.HostBatch(Xh, Wh, Hh, device=cx.dev, reuse_outputs=True)
        t_reg = time.perf_counter() - t0
        hb.fit(max_iter=a.iters, tol=0.0, handle=handle)
        torch.cuda.synchronize(cx.dev)
        t0 = time.perf_counter()
        rh = hb.fit(max_iter=a.iters, tol=0.0, handle=handle)
        dt_h = time.perf_counter() - t0
        Wb, Hb = rh.W.copy(), rh.H.copy()
        reg_s = hb.register_seconds
        hb.close()
        # (b) one-shot calls on the same arrays (nothing kept between calls): pipelined, and one upload / fit / download
        ms.fit_batched(Xh, Wh, Hh, max_iter=a.iters, tol=0.0, device=cx.dev, handle=handle)
        torch.cuda.synchronize(cx.dev)
        t0 = time.perf_counter()
        r1 = ms.fit_batched(Xh, Wh, Hh, max_iter=a.iters, tol=0.0, device=cx.dev, handle=handle)
        dt_1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        ru = ms.fit_batched(Xh, Wh, Hh, max_iter=a.iters, tol=0.0, device=cx.dev, handle=handle, host_chunk=0)
        dt_u = time.perf_counter() - t0
        host_res = {"matrix_iterations_per_s": B * a.iters / dt_h, "seconds": dt_h,
                    "how": "ms.HostBatch(X, W0, H0, reuse_outputs=True): arrays page-locked once (hipHostRegister in place), the SECOND "
                           "fit of the same host batch timed, NumPy in, NumPy out",
                    "register_seconds_once": reg_s, "constructor_seconds_once": t_reg,
                    "one_shot_matrix_iterations_per_s": B * a.iters / dt_1,
                    "unpipelined_matrix_iterations_per_s": B * a.iters / dt_u,
                    "bytes_up": int(Xh.nbytes + Wh.nbytes + Hh.nbytes), "bytes_down": int(rh.W.nbytes + rh.H.nbytes),
                    "bitwise_equal_to_device_resident_fit": bool(np.array_equal(Wb, r.W.cpu().numpy()) and np.array_equal(Hb, r.H.cpu().numpy())),
                    "bitwise_equal_to_unpipelined": bool(np.array_equal(Wb, ru.W) and np.array_equal(Hb, ru.H)),
                    "one_shot_bitwise_equal_to_unpipelined": bool(np.array_equal(r1.W, ru.W) and np.array_equal(r1.H, ru.H)),
                    "note": "NumPy in, NumPy out on one GPU: chunks of the batch uploaded / fitted / downloaded concurrently "
                            "(engine._fit_batched_pipelined); `value` keeps the inputs resident in HBM as the contract says"}
        del r1, Wb, Hb
        del Xh, Wh, Hh, rh, ru
    units_per_launch = B * a.iters
    avg_ms = sum(kernel_ms) / len(kernel_ms)
    layout = "row-major [B][T][m] (C order)" if a.x_layout == "row" else "channel-major [B][m][T] (F order)"
    traffic = _traffic(kernel, batch=B, iters=a.iters, T=a.T, m=a.m, k=a.k, x_layout=a.x_layout)
    if f64:
        cfg_dtype_note = ("float64 is the reference's own dtype (vicon_data/user_data.py:391-396 builds dtype=float frames, "
                          "analysis.py:862-863 hands them to scikit-learn unchanged); BASELINE.json's metric is quoted on fp32")
    moved = 4 * (a.T * 16 + 2 * a.k * max(0, a.T - lds_rows_of_w(a.k))) if (a.m > 8 and not single and not wide and not f64) else None
    dt_words = "float64" if f64 else "fp32"
    global_batch = 1 * cx.world if single else (B * cx.world if weak else total)
    cfg = {
        "workload": (f"one synthetic EMG matrix {a.m} ch x {a.T} samples per GPU, k={a.k}, {dt_words}, {a.iters} mu iterations "
                     f"per fit (BASELINE.json configs[1]; N > 1 = independent replicas)") if single else
                    (f"batch of {global_batch} synthetic EMG matrices {a.m} ch x {a.T} samples"
                     + (f" ({B} on every GPU: weak-scaling form)" if weak else f" scattered over {cx.world} GPU(s) ({B} on rank 0)")
                     + f", k={a.k}, {dt_words}, {a.iters} mu iterations per fit, tol=0, init='custom' (BASELINE.json configs[2])"),
        "batch_rank0": B, "global_batch": global_batch, "n_samples": a.T, "n_features": a.m, "n_components": a.k,
        "iters_per_step": a.iters, "x_layout": layout,
        "parallelism": f"independent factorisations scattered over {cx.world} GPU(s), no collective",
        "all_fits_ran_full_iters": bool((r.n_iter == a.iters).all().item()),
        "all_residuals_finite": bool(torch.isfinite(r.reconstruction_err).all().item()),
    }
    if f64:
        cfg["dtype_note"] = cfg_dtype_note
    if other is not None:
        cfg["value_other_x_layout"] = {"x_layout": "channel-major [B][m][T] (F order, DataFrame.to_numpy())"
                                       if a.x_layout == "row" else "row-major [B][T][m]",
                                       "matrix_iterations_per_s_1gpu": other,
                                       "note": "one conversion kernel per fit inside the timed call"}
    stream = None
    if not single and not wide:  # the ceiling the headline kernel is priced against, measured now (after the timed region)
        region = ((8 if f64 else 4) * a.T * a.m + 1023) // 1024 * 1024
        try:
            stream = handle.stream_gbs(region, B, 20)
        except Exception as e:  # noqa: BLE001 -- a diagnostic must not cost the benchmark line
            cfg["stream_peak_error"] = str(e)
    # (last: the checker's CPU fits leave the GPU idle for about a second, and the clocks follow -- the stream ceiling above and
    #  the host-resident rate are measured before, back to back with the timed region)
    parity = None
    if cx.parity is not None:  # matrices spread over the batch, from the LAST timed step's output
        idx = sorted({int(round(i * (B - 1) / 3)) for i in range(4)})
        host = lambda t: t.detach().cpu().numpy()  # noqa: E731
        jobs = [(host(Xr[i]), host(W0[i]), host(H0[i]), a.iters, "frobenius") for i in idx]
        ours = [(host(r.W[i]), host(r.H[i]), float(r.reconstruction_err[i])) for i in idx]
        parity = cx.parity.check(jobs, ours, names=[f"matrix {i} of rank 0's batch" for i in idx])
        parity["matrices"] = idx
    return {"units": global_batch * a.iters * a.steps, "elapsed": elapsed, "scaling": "weak" if (weak or single) else "strong",
            "config": cfg, "parity": parity, "host_resident": host_res,
            "roofline": compute_roofline(kernel, avg_ms, units_per_launch, a.T, a.m, a.k, traffic, moved, stream, esize=8 if f64 else 4)}


# ------------------------------------------------------------------------------------------------ config 4
def run_rank_sweep(cx):
    a, torch = cx.a, cx.torch
    import muscle_synergies_amd as ms
    from muscle_synergies_amd.engine import partition
    from muscle_synergies_amd.engine import rank_sweep_native
    from muscle_synergies_amd.synth import emg_rank_trials_torch

    total = a.batch or 1024
    lo, hi = partition(total, cx.world)[cx.rank]  # scattered by trial: contiguous slices, no collective
    B = hi - lo
    # trials whose smallest sufficient rank differs (bursting synergies, k_true = 2..6: synth.emg_rank_trials_torch); round
    # 2's smoothed-noise batch selected k = 2 for every trial and could not tell a real stop from a post-hoc one
    X, k_true = emg_rank_trials_torch(B, T=a.T, m=a.m, device=cx.dev, seed=1000 + cx.rank)
    f64 = a.dtype == "f64"
    if f64:
        X = X.double()
    Xv = X.transpose(1, 2).contiguous()
    kmin, kmax = 2, 8
    kms = []

    def step():
        r = ms.rank_sweep_batched(Xv, kmin, kmax, vaf_threshold=0.90, max_iter=a.iters, tol=0.0, seed=1, device=cx.dev)
        kms.append(r.kernel_ms)
        return r

    elapsed, outs = cx.timed(step)
    # the same sweep as one library call, computing every rank (the reference's behaviour, analysis.py:907-912) and with
    # the real stop (hipnmf_rank_sweep_stop_*): untimed warm-up + one timed call each, on every rank; reported by rank 0
    native = {}
    for stop in (False, True):
        for rep in range(2):
            torch.cuda.synchronize(cx.dev)
            t0 = time.perf_counter()
            rn = rank_sweep_native(Xv, kmin, kmax, vaf_threshold=0.90, max_iter=a.iters, tol=0.0, seed=1, device=cx.dev,
                                   stop_at_threshold=stop)
            torch.cuda.synchronize(cx.dev)
            native[stop] = (rn, time.perf_counter() - t0)
    cx.barrier()
    if cx.rank != 0:
        return None
    r = outs[-1]
    parity = None
    if cx.parity is not None:
        # the timed sweep draws W0/H0 with random_init_batched(X, k, seed + k): the same call gives the same factors again
        from muscle_synergies_amd.engine import random_init_batched

        rw = ms.rank_sweep_batched(Xv, kmin, kmax, vaf_threshold=0.90, max_iter=a.iters, tol=0.0, seed=1, device=cx.dev, keep_W=True)
        host = lambda t: t.detach().cpu().numpy()  # noqa: E731
        jobs, ours, what = [], [], []
        for k in sorted({kmin, (kmin + kmax) // 2, kmax}):
            W0k, H0k = random_init_batched(Xv, k, seed=1 + k)
            for i in sorted({0, B - 1}):
                jobs.append((host(Xv[i]), host(W0k[i]), host(H0k[i]), a.iters, "frobenius"))
                ours.append((host(rw.W[k][i]), host(rw.components[k][i]), float(rw.reconstruction_err[k][i])))
                what.append([i, k])
            del W0k, H0k
        same = all(bool(torch.equal(rw.reconstruction_err[k], r.reconstruction_err[k])) for k in r.ranks)
        del rw
        parity = cx.parity.check(jobs, ours, names=[f"trial {i}, k={k}" for i, k in what])
        parity["trial_rank_pairs"] = what
        parity["untimed_rerun_bitwise_equal_to_last_timed_step"] = same
        parity["ok"] = bool(parity["ok"] and same)
    nk = kmax - kmin + 1
    r_all, t_all = native[False]
    r_stop, t_stop = native[True]
    fits_run = sum(int((r_stop.n_iter[k] > 0).sum().item()) for k in r_stop.ranks)
    kms = kms[a.warmup:]
    avg_ms = sum(kms) / len(kms)
    fl = sum(flops_per_unit(a.T, a.m, k) for k in range(kmin, kmax + 1)) * B * a.iters
    tf = fl / (avg_ms * 1e-3) / 1e12
    hist = torch.bincount(r.selected.clamp(min=0), minlength=kmax + 1).tolist()
    return {"units": total * nk * a.iters * a.steps, "elapsed": elapsed, "scaling": "strong", "parity": parity,
            "config": {"workload": (f"rank sweep k={kmin}..{kmax} ({a.iters} mu iterations each, random init drawn on the "
                                    f"device, smallest k with VAF >= 0.90 selected) over {total} synthetic EMG trials "
                                    f"{a.m} ch x {a.T} samples in total, {'float64' if f64 else 'fp32'} (BASELINE.json configs[3])"),
                       "trials_total": total, "trials_rank0": B, "n_samples": a.T, "n_features": a.m,
                       "ranks": [kmin, kmax], "iters_per_fit": a.iters,
                       "parallelism": f"trials scattered over {cx.world} GPU(s), no collective",
                       "selected_rank_histogram_rank0": hist,
                       "true_rank_histogram_rank0": torch.bincount(k_true, minlength=kmax + 1).tolist(),
                       "native_sweep_rank0": {
                           "compute_all_ms": t_all * 1e3, "stop_at_threshold_ms": t_stop * 1e3,
                           "fits_run_with_stop": fits_run, "fits_total": B * nk,
                           "selected_identical": bool(torch.equal(r_all.selected, r_stop.selected)),
                           "matrix_iterations_per_s_compute_all": B * nk * a.iters / t_all,
                           "note": "hipnmf_rank_sweep_* vs hipnmf_rank_sweep_stop_* (one library call each, random init "
                                   "drawn inside); `value` above is the compute-all sweep through the Python host"}},
            "roofline": _sweep_roofline(a, f64, tf, fl, avg_ms, B, nk, kmin, kmax)}


def _sweep_roofline(a, f64, tf, fl, avg_ms, B, nk, kmin, kmax):
    if f64:  # seven kernels, one per rank; HBM-bound like the float64 batch (roofline_f64_narrow)
        by = sum(8 * a.T * (a.m + 2 * k) for k in range(kmin, kmax + 1)) * B * a.iters
        gbs = by / (avg_ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                "kernel": "fit_persistent_kernel<double,4,4,k,0> (k <= 6), fit_wide4d_kernel<16,2,8,..> (k >= 7): DISPATCH.md",
                "kernel_ms_avg": avg_ms, "units_per_launch": B * nk * a.iters,
                "fp64_issue": {"achieved_tflops": tf, "peak_tflops": FP64_PEAK_TFLOPS, "frac": tf / FP64_PEAK_TFLOPS}}
    return {"bound": "fp32_issue", "bound_detail": "fp32 issue (f32 MFMA = f32 VALU = 157.3 TFLOP/s); seven kernels, one per rank",
            "achieved": tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_PEAK_TFLOPS,
            "traffic": None, "kernel": "fit_persistent_kernel<float,1,16,k,0> (k <= 5), fit_rowlane_kernel<k,...> (k >= 6)",
            "kernel_ms_avg": avg_ms, "units_per_launch": B * nk * a.iters}


# ------------------------------------------------------------------------------------------------ config 5
def run_tsharded(cx):
    a, torch = cx.a, cx.torch
    from muscle_synergies_amd.synth import emg_shard_torch
    from muscle_synergies_amd.tsharded import HipShardOps, MultiShardOps, fit_tsharded, plan_subshards, shard_bounds

    T, m, k = a.T5, a.m, a.k
    lo, hi = shard_bounds(T, cx.world)[cx.rank]
    # this rank's rows as sub-shards of at most --subshard rows (tsharded.plan_subshards: the same recording for every N)
    H = None
    shards = []
    for _t0, n, sid in plan_subshards(T, cx.world, cx.rank, a.subshard):
        Xs, Ws, H0 = emg_shard_torch(5, sid, n, m=m, k=k, device=cx.dev)
        if H is None:
            H = H0.clone()
        shards.append(HipShardOps.from_native(Xs, Ws, H))
    ops = MultiShardOps(shards)
    torch.cuda.synchronize(cx.dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    pass_ms = []
    coll = {"calls": 0, "elements": 0, "max_elements": 0}

    def counted_all_reduce(t):
        coll["calls"] += 1
        coll["elements"] += t.numel()
        coll["max_elements"] = max(coll["max_elements"], t.numel())
        if cx.distributed:
            cx.dist.all_reduce(t, op=cx.dist.ReduceOp.SUM)
        return t

    def step():
        # the solver runs on torch's current stream (handles are bound to it), so torch events see its kernels
        ev[0].record()
        r = fit_tsharded(ops, max_iter=a.iters5, tol=0.0, all_reduce=counted_all_reduce)
        ev[1].record()
        torch.cuda.synchronize(cx.dev)
        pass_ms.append(ev[0].elapsed_time(ev[1]))
        return r

    elapsed, outs = cx.timed(step)
    if cx.rank != 0:
        return None
    r = outs[-1]
    pass_ms = pass_ms[a.warmup:]
    by = 4 * T * (m + 2 * k)  # per iteration of the whole matrix
    rows_rank0 = hi - lo
    by_rank = 4 * rows_rank0 * (m + 2 * k)
    # the dominant kernel (the shard pass over this rank's sub-shards: slice_pass_rowlane_kernel + reduce_slices) timed on
    # its own with events on the solver's stream, after the timed region: three untimed + ten timed passes (W keeps being
    # updated in place, H is left alone, which does not change the traffic).  The whole-iteration time (pass + all-reduce +
    # H update, final residual spread over the iterations) is reported beside it.
    for _ in range(3):
        ops.shard_pass()
    ev[0].record()
    for _ in range(10):
        ops.shard_pass()
    ev[1].record()
    torch.cuda.synchronize(cx.dev)
    it_ms = ev[0].elapsed_time(ev[1]) / 10
    step_it_ms = (sum(pass_ms) / len(pass_ms)) / a.iters5  # includes 1 / iters5 of the final residual pass
    # the exchange step on its own: 200 packed all-reduces of k*m + k*k values back to back on the solver's stream
    ar_us = None
    if cx.distributed:
        buf = torch.zeros(k * m + k * k, dtype=torch.float32, device=cx.dev)
        for _ in range(20):
            cx.dist.all_reduce(buf, op=cx.dist.ReduceOp.SUM)
        ev[0].record()
        for _ in range(200):
            cx.dist.all_reduce(buf, op=cx.dist.ReduceOp.SUM)
        ev[1].record()
        torch.cuda.synchronize(cx.dev)
        ar_us = ev[0].elapsed_time(ev[1]) / 200 * 1e3
    achieved = by_rank / (it_ms * 1e-3) / 1e9
    parity = None
    if cx.parity is not None:
        # a 2e8-row matrix cannot go through scikit-learn: the same entry points, kernels and loop on a 200 000-row
        # replica of the recording (sub-shard 0's generator), rank 0 alone, against the checker at the same iteration count
        Tp = 200_000
        Xp, Wp, Hp = emg_shard_torch(5, 0, Tp, m=m, k=k, device=cx.dev)
        host = lambda t: t.detach().cpu().numpy()  # noqa: E731
        job = (host(Xp[0].t().contiguous()), host(Wp[0].t().contiguous()), host(Hp[0]), a.iters5, "frobenius")
        rp = fit_tsharded(HipShardOps.from_native(Xp, Wp, Hp.clone()), max_iter=a.iters5, tol=0.0, all_reduce=lambda t: t)
        parity = cx.parity.check([job], [(host(rp.W_local[0]), host(rp.H[0]), float(rp.reconstruction_err[0]))])
        parity["what"] = (f"a {Tp}-row replica (sub-shard 0's generator) through the same shard entry points on rank 0 alone, "
                          f"{a.iters5} iterations, " + parity["what"])
        del Xp, Wp, Hp, rp
    n_fit = a.steps + a.warmup
    return {"units": a.iters5 * a.steps, "elapsed": elapsed, "scaling": "strong", "parity": parity,
            "collective": {"backend": (cx.backend + (" (RCCL over xGMI)" if cx.backend == "nccl" else "")) if cx.distributed else "none (single rank)",
                           "world_size": cx.world,
                           "all_reduce_us_each_back_to_back": ar_us,
                           "all_reduce_share_of_iteration": (ar_us * 1e-3 / step_it_ms) if ar_us else None,
                           "all_reduce_calls_per_fit": coll["calls"] // n_fit,
                           "all_reduce_elements_per_iteration": k * m + k * k,
                           "all_reduce_elements_issued_per_fit": coll["elements"] // n_fit,
                           "largest_all_reduce_elements": coll["max_elements"],
                           "note": f"per fit: {a.iters5} x ({k * m} + {k * k}) floats for the H update + one 2 x {m} residual/VAF reduction at the end"},
            "config": {"workload": (f"ONE synthetic EMG matrix {m} ch x {T} samples, k={k}, fp32, rows sharded over "
                                    f"{cx.world} GPU(s) ({rows_rank0} rows on rank 0 in {len(shards)} sub-shard(s)), "
                                    f"{a.iters5} mu iterations per step, tol=0 (BASELINE.json configs[4])"),
                       "n_samples": T, "n_features": m, "n_components": k, "iters_per_step": a.iters5,
                       "unit_of_work": "one mu iteration of the whole matrix",
                       "parallelism": (f"time-sharded over {cx.world} GPU(s): one all-reduce of {k * m + k * k} floats per "
                                       f"iteration (RCCL over xGMI), H replicated"),
                       "reconstruction_err": float(r.reconstruction_err[0]), "vaf_all": float(r.vaf[0, 0])},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "slice_pass_rowlane_kernel<5> (+ reduce_slices, hupdate, all-reduce per iteration)",
                         "measured_stream_ceiling_gbs": SHARD_STREAM_GBS,
                         "frac_of_measured_ceiling": achieved / SHARD_STREAM_GBS,
                         "kernel_ms_avg": it_ms, "iteration_ms_in_timed_steps": step_it_ms, "algorithmic_bytes_per_unit": by,
                         "algorithmic_bytes_per_iteration_rank0": by_rank, "units_per_launch": 1,
                         "note": "per-GPU rate of rank 0: its rows x 4 (m + 2k) bytes per pass / duration of one shard pass over "
                                 "its sub-shards (events on the solver's stream around ten passes, measured live after the "
                                 "timed region); iteration_ms_in_timed_steps = timed step / iters5 (adds the all-reduce, the H "
                                 "update and 1 / iters5 of the final residual pass)"}}


# ------------------------------------------------------------------------------------------------
def run_dry(cx):
    """--dry-orchestration: the units of config 3 with a sleep for a step -- checks the launcher, not the engine."""
    a = cx.a
    B = a.batch or 4096

    def step():
        time.sleep(0.01 * (1 + cx.rank))  # ranks differ: the reported time must be the slowest rank's
        return None

    elapsed, _ = cx.timed(step)
    if cx.rank != 0:
        return None
    weak = bool(a.batch_per_gpu)
    total = a.batch_per_gpu * cx.world if weak else B  # the units of config 3: --batch in total, or --batch-per-gpu on every rank
    return {"units": total * a.iters * a.steps, "elapsed": elapsed, "scaling": "weak" if weak else "strong",
            "config": {"workload": "DRY ORCHESTRATION: no GPU work, a sleep per step (rank plumbing check only)",
                       "global_batch": total}, "roofline": None}


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a):
    """``python bench.py --gpus N`` without a torchrun environment: start N fresh rank processes (one per GPU, rendezvous on
    127.0.0.1) and relay rank 0's JSON line.  The parent imports no GPU library at all (every rank checks its own device and
    exits non-zero without one) and nothing is exec'ed from a process that has touched a GPU: the ranks are children, their
    exit codes are ours.  All ranks are polled: the first non-zero exit (or the deadline) ends the others."""
    import subprocess
    import tempfile

    port = _free_port()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HIPNMF_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.monotonic() + a.launch_timeout
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if bad or time.monotonic() > deadline:
            failed = (f"rank {bad[0]} exited with code {codes[bad[0]]}" if bad else f"no result within --launch-timeout {a.launch_timeout} s")
            for p in procs:  # exactly the processes started above
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            break
        time.sleep(0.05)
    codes = [p.returncode for p in procs]
    out0.seek(0)
    text = out0.read()
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    if failed or any(codes) or len(lines) != 1:
        sys.stdout.write(text)
        raise SystemExit(f"bench.py --gpus {a.gpus}: {failed or 'failed'}; rank exit codes {codes}, {len(lines)} JSON line(s) from rank 0")
    print(lines[0], flush=True)


def main():
    a = parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a)  # invoked directly: be the launcher (under torchrun the environment is there already)
    # ONE JSON line on stdout, whatever the libraries under us print: file descriptor 1 is pointed at stderr for the life of the
    # process (RCCL writes a version banner to it when the first communicator comes up; C stdio flushes it at exit, after the
    # line) and the line itself goes to a saved copy of the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    cx = Ctx(a)
    if a.dry_orchestration and a.dry_fail_rank == cx.rank:
        raise SystemExit(7)
    if cx.world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={cx.world}: start one rank per GPU (python bench.py --gpus N does "
                         f"that by itself when no launcher environment is present)")
    cpu = None
    if cx.rank == 0 and cx.world == 1 and not a.no_cpu_baseline and not a.dry_orchestration:
        cpu = cpu_baseline(a)  # before the GPU is initialised (spawns worker processes)
    # in-run parity checker (rank 0): its CPU workers start before this process touches the GPU
    cx.parity = ParityChecker(cx.rank == 0 and not a.no_parity and not a.dry_orchestration)
    if cx.parity.pool is None:
        cx.parity = None
    cx.init_gpu()
    if a.dry_orchestration:
        res = run_dry(cx)
    elif a.config == 3:
        res = run_batch(cx, single=False)
    elif a.config == 2:
        res = run_batch(cx, single=True)
    elif a.config == 4:
        res = run_rank_sweep(cx)
    else:
        res = run_tsharded(cx)
    seen = cx.ranks_seen()
    if cx.rank == 0:
        parity = res.get("parity")
        out = {
            "metric": "NMF mu-iters/sec",
            "value": res["units"] / res["elapsed"],
            "unit": "matrix-iterations/s",
            "n_gpus": cx.world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": res["elapsed"] / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": res["scaling"],
            "vs_baseline": None,
            "dtype": a.dtype if a.config in (2, 3, 4) else "f32",
            "data": "none (dry orchestration)" if a.dry_orchestration else "synthetic",
            "config": res["config"],
            "roofline": res["roofline"],
            "cpu_baseline": cpu,
            "parity": parity,
            "ranks_seen": seen,
            "process_group_backend": cx.backend,
        }
        if "collective" in res:
            out["collective"] = res["collective"]
        if res.get("host_resident"):
            out["value_host_resident"] = res["host_resident"]["matrix_iterations_per_s"]
            out["host_resident"] = res["host_resident"]
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if cx.parity is not None:
        cx.parity.close()
    cx.finish()
    if cx.rank == 0 and res.get("parity") is not None and not res["parity"]["ok"]:
        raise SystemExit("bench.py: the in-run parity check FAILED (see \"parity\" in the JSON line): the number above is not valid")


if __name__ == "__main__":
    main()

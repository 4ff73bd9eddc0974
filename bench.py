#!/usr/bin/env python3
"""Headline benchmark: NMF mu-update iterations/sec on a batch of synthetic EMG matrices.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One *step* = one batched fit (``hipnmf_fit_batched_f32`` through the Python host): ``--iters`` (500)
Lee-Seung multiplicative-update iterations (tol = 0, so exactly that many, as sklearn does) on every
matrix of the rank's batch, from a fixed ``init='custom'`` W0/H0, inputs already resident in HBM.
Workload = BASELINE.json configs[2]: 4096 synthetic EMG matrices 16 ch x 10 000 samples, k = 5, fp32,
per GPU.  The factorisations are independent, so ranks share nothing: no data-path collective, weak
scaling (the batch per GPU is fixed); the only collectives are the timing barrier and the max over ranks.

Rank 0 prints ONE JSON line (fields: see the task contract) including
  roofline     -- algorithmic bytes per launch / HIP-event duration of the solver kernel vs HBM peak
  cpu_baseline -- scikit-learn's NMF(solver='mu') (the reference's arithmetic) on this host's cores,
                  bounded sample, N = 1 only; run BEFORE the GPU is initialised (worker processes).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md:36); 6290 GB/s measured copy


def parse_args():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="matrices per GPU")
    ap.add_argument("--iters", type=int, default=500, help="mu iterations per fit (= per step)")
    ap.add_argument("--T", type=int, default=10_000)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--threads", type=int, default=0, help="workgroup size override (0 = library default)")
    ap.add_argument("--x-layout", choices=["row", "channel"], default="row",
                    help="memory order of the X batch handed to the engine: [B][T][m] (C order) or [B][m][T]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="matrices in the CPU baseline sample (0 = auto)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# CPU baseline: sklearn (the dependency that holds the reference's NMF arithmetic), one BLAS thread per
# worker process, all host cores.  Runs before anything touches the GPU.
def _cpu_worker(job):
    seeds, T, m, k, iters = job
    import warnings

    import numpy as np
    from threadpoolctl import threadpool_limits

    from muscle_synergies_amd.synth import emg_matrix, random_init

    warnings.simplefilter("ignore")
    from sklearn.decomposition import NMF

    data = []
    for s in seeds:
        X = emg_matrix(s, T=T, m=m, dtype=np.float32)
        W0, H0 = random_init(X, k, s)
        data.append((X, W0, H0))
    with threadpool_limits(limits=1):
        t0 = time.perf_counter()
        for X, W0, H0 in data:
            NMF(k, solver="mu", init="custom", tol=0, max_iter=iters).fit_transform(X, W=W0, H=H0)
        dt = time.perf_counter() - t0
    return len(seeds), dt


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
    jobs = [(list(range(1000 + w * per, 1000 + (w + 1) * per)), a.T, a.m, a.k, a.iters) for w in range(cores)]
    ctx = mp.get_context("spawn")
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
        "sample": (f"scikit-learn {sklearn.__version__} NMF(solver='mu', init='custom', tol=0, max_iter={a.iters}) "
                   f"on {total} of the synthetic {a.m}x{a.T} k={a.k} fp32 matrices, {cores} worker processes x 1 BLAS "
                   f"thread; fit time of the slowest worker {slowest:.2f} s (pool wall {wall:.1f} s incl. data generation)"),
    }


# ------------------------------------------------------------------------------------------------
def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # under torchrun always use the process group

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a)  # before the GPU is initialised (spawns worker processes)

    import torch

    import muscle_synergies_amd as ms
    from muscle_synergies_amd import _lib
    from muscle_synergies_amd.synth import emg_batch_torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no ROCm GPU visible (the engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    # synthetic workload, generated on the device (seeded per rank).  X is handed over as [B, T, m] in C order
    # (row-major, what sklearn itself takes): the layout the fp32 16-channel kernel streams in place.
    # --x-layout channel passes the [B, m, T] storage (a DataFrame's F order) instead; the engine then converts it
    # once per fit inside the timed step.
    X, W0, H0 = emg_batch_torch(a.batch, T=a.T, m=a.m, k=a.k, device=dev, seed=rank)
    Xv = X.transpose(1, 2)  # logical [B, T, m] (sklearn orientation), zero-copy view of channel-major storage
    if a.x_layout == "row":
        Xv = Xv.contiguous()
        del X
    handle = _lib.get_handle(local_rank)
    if a.threads:
        handle.set_tuning(a.threads, 0, 0)

    def step():
        return ms.fit_batched(Xv, W0, H0, max_iter=a.iters, tol=0.0, device=dev, handle=handle)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(a.steps):
        r = step()
        kernel_ms.append(r.kernel_ms)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    barrier()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_iter_ok = bool((r.n_iter == a.iters).all().item())
        finite = bool(torch.isfinite(r.reconstruction_err).all().item())
        total_units = world * a.batch * a.iters * a.steps
        value = total_units / elapsed
        bytes_per_unit = 4 * a.T * (a.m + 2 * a.k)  # read X once, read + write W once
        bytes_per_launch = a.batch * a.iters * bytes_per_unit
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("batch") == a.batch and tj.get("iters") == a.iters and tj.get("T") == a.T:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "NMF mu-iters/sec",
            "value": value,
            "unit": "matrix-iterations/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (f"batch of {a.batch} synthetic EMG matrices {a.m} ch x {a.T} samples per GPU, k={a.k}, fp32, "
                             f"{a.iters} mu iterations per fit, tol=0, init='custom' (BASELINE.json configs[2])"),
                "batch_per_gpu": a.batch,
                "global_batch": a.batch * world,
                "n_samples": a.T,
                "n_features": a.m,
                "n_components": a.k,
                "iters_per_step": a.iters,
                "x_layout": "row-major [B][T][m] (C order)" if a.x_layout == "row" else "channel-major [B][m][T] (F order)",
                "parallelism": f"independent factorisations scattered over {world} GPU(s), no collective",
                "all_fits_ran_full_iters": n_iter_ok,
                "all_residuals_finite": finite,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "fit_persistent_kernel<float,1,16,5,0>" if (8 < a.m <= 16 and a.k <= 5) else "fit_persistent_kernel",
                "kernel_ms_avg": avg_ms,
                "algorithmic_bytes_per_unit": bytes_per_unit,
                "units_per_launch": a.batch * a.iters,
            },
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

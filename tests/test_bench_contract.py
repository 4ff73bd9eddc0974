"""bench.py's bookkeeping (no GPU): the per-unit FLOP / byte model of SURVEY.md section 8(d), the LDS-capacity
formula the byte model relies on, the roofline object's fields, and the command-line contract."""
import importlib.util
import os
import subprocess
import sys

from conftest import ROOT

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_unit_of_work_model_matches_the_survey():
    assert 4 * 10_000 * (16 + 2 * 5) == 1_040_000                       # algorithmic bytes per matrix-iteration
    assert bench.flops_per_unit(10_000, 16, 5) == 4 * 10_000 * 5 * 21 + 2 * 10_000 * 5 + 4 * 25 * 16 + 2 * 5 * 16
    assert abs(bench.flops_per_unit(10_000, 16, 5) - 4.30e6) < 0.01e6
    assert bench.lds_rows_of_w(5) == 7936                                # rows of W resident in LDS at k = 5 (64-row granule)
    assert bench.lds_rows_of_w(2) >= 10_240 and bench.lds_rows_of_w(4) >= 10_000
    assert 4 * (10_000 * 16 + 2 * 5 * (10_000 - bench.lds_rows_of_w(5))) == 722_560   # design bytes per unit


def test_roofline_object_is_against_the_binding_roofs():
    r = bench.compute_roofline("fit_persistent_kernel<float,1,16,5,0>", 200.0, 4096 * 500, 10_000, 16, 5,
                               traffic=1.5185e12)
    assert r["bound"] == "fp32_issue" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert 0 < r["frac"] < 1 and abs(r["achieved"] - 4301760 * 2048000 / 0.2 / 1e12) < 1e-6
    mem = r["memory"]
    assert mem["algorithmic_bytes_per_unit"] == 1_040_000 and mem["l2_fabric_bytes_per_launch"] == 1.5185e12
    assert 0.9 < mem["frac_of_stream_peak"] < 1.05
    r2 = bench.compute_roofline("k", 200.0, 4096 * 500, 10_000, 16, 5, traffic=None, moved_bytes_per_unit=722_560)
    assert r2["traffic"] is None and "design_gbs" in r2["memory"]


def test_wide_shapes_report_against_the_hbm_line():
    r = bench.compute_roofline("fit_wide_kernel<float,64,16,4>", 26.6, 1024 * 40, 10_000, 64, 8)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["algorithmic_bytes_per_unit"] == 4 * 10_000 * (64 + 16)
    assert abs(r["achieved"] - 3_200_000 * 40960 / 26.6e-3 / 1e9) < 1e-6 and 0 < r["frac"] < 1
    assert r["matrix_pipe"]["issued_tflops"] > r["matrix_pipe"]["achieved_tflops_useful"]
    r4 = bench.compute_roofline("fit_wide4_kernel<64,2,12,2>", 26.6, 1024 * 40, 10_000, 64, 8)  # 4x4x1 tiles: no padding at 64 x 8
    assert r4["bound"] == "hbm" and abs(r4["matrix_pipe"]["issued_tflops"] - r4["matrix_pipe"]["achieved_tflops_useful"]) < 1e-9
    r5 = bench.compute_roofline("fit_wide4_kernel<48,2,12,2>", 26.6, 1024 * 40, 10_000, 40, 6)
    assert r5["matrix_pipe"]["issued_tflops"] > r5["matrix_pipe"]["achieved_tflops_useful"]


def test_committed_traffic_measurements_name_their_kernel_and_source():
    """profiles/traffic.json feeds roofline.traffic: every entry names the kernel instance and workload it was measured
    on and the commit it was measured at (the -m gpu suite checks that the library still launches that instance)."""
    import json

    entries = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert entries
    for e in entries:
        for key in ("kernel", "batch", "iters", "T", "m", "k", "x_layout", "l2_fabric_bytes_per_launch", "source_commit", "method"):
            assert key in e, key
        assert e["kernel"].startswith(("fit_persistent_kernel<", "fit_rowlane_kernel<", "fit_wide_kernel<", "fit_wide4_kernel<", "big1_pass_kernel<"))
    assert bench._traffic("no-such-kernel", batch=1) is None


def test_command_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--config"):
        assert flag in out.stdout


def _json_lines(text):
    import json

    return [json.loads(ln) for ln in text.splitlines() if ln.startswith("{")]


def test_gpus_n_invoked_directly_starts_n_ranks_over_gloo():
    """`python bench.py --gpus 2` with no launcher environment must run TWO ranks (round 3: it silently ran one and printed
    n_gpus 1).  --dry-orchestration swaps the GPU work for a sleep and RCCL for gloo; everything else is the real plumbing."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-orchestration", "--steps", "3",
                          "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = _json_lines(out.stdout)
    assert len(lines) == 1
    r = lines[0]
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["steps"] == 3 and r["warmup"] == 1
    assert r["process_group_backend"] == "gloo"
    assert r["ms_per_step"] >= 19.0          # rank 1 sleeps 20 ms per step, rank 0 10 ms: the slowest rank's time is reported
    assert r["data"].startswith("none") and r["cpu_baseline"] is None


def test_gpus_n_under_torchrun_uses_the_launcher_environment():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--dry-orchestration", "--steps", "2", "--warmup", "0"], capture_output=True, text=True, env=env,
                         timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = _json_lines(out.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_gpus_n_refuses_to_run_on_fewer_devices():
    """Without a GPU (this container) --gpus 2 must fail loudly, never print a 1-GPU line."""
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("two GPUs visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and not _json_lines(out.stdout)
    assert "needs ROCm GPU index" in (out.stderr + out.stdout)  # said by the rank itself: the launcher parent loads no GPU library
    # a launcher environment that disagrees with --gpus is an error too
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-orchestration"],
                         capture_output=True, text=True, env=env2, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)


def test_launcher_ends_the_other_ranks_when_one_dies():
    """A rank that dies before the rendezvous must not leave rank 0 waiting for the collective timeout (round-4 advisor
    finding): the launcher polls every rank, ends the others and exits non-zero within seconds."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-orchestration", "--dry-fail-rank", "1"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and not _json_lines(out.stdout)
    assert "rank 1 exited with code 7" in (out.stderr + out.stdout)
    assert time.monotonic() - t0 < 60


def test_only_config5_uses_rccl():
    """Configs 2-4 have no data-path collective: their barrier / max-over-ranks run over gloo (bench.py Ctx.init_gpu)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'self.backend = "nccl" if (self.a.config == 5 and not self.dry) else "gloo"' in src
    assert src.count('backend="nccl"') == 2  # under a launcher, and --force-nccl's world of one (config 5 only: Ctx.force_nccl)
    assert 'self.force_nccl = bool(a.force_nccl and a.config == 5' in src


def test_float64_batches_report_against_the_hbm_line():
    """--dtype f64 (the reference's own dtype): 256 resident matrices of 16 x 10 000 doubles exceed the Infinity Cache, so the
    roofline object is priced against the 8 TB/s HBM line with 8-byte elements."""
    import bench

    r = bench.compute_roofline("fit_persistent_kernel<double,4,4,5,0>", 600.0, 4096 * 500, 10_000, 16, 5, esize=8)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["algorithmic_bytes_per_unit"] == 8 * 10_000 * 26
    assert abs(r["achieved"] - 8 * 10_000 * 26 * 4096 * 500 / 0.6 / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12


def test_parity_gate_is_the_call_dtype_fit_and_names_exceptions():
    """the gate is the distance to scikit-learn's fit in the dtype of the call; a float32 case beyond it is
    listed by name and tolerated only when sklearn's own float32 fit is further than the tolerance from its float64 fit."""
    import numpy as np

    import bench

    class FakePool:
        def __init__(self, refs):
            self.refs = refs

        def map(self, _f, _jobs):
            return self.refs

    rng = np.random.default_rng(0)
    X = np.abs(rng.standard_normal((50, 4))).astype(np.float32)
    W = np.abs(rng.standard_normal((50, 2))).astype(np.float32)
    H = np.abs(rng.standard_normal((2, 4))).astype(np.float32)
    xn = float(np.linalg.norm(X))
    WH = W.astype(np.float64) @ H.astype(np.float64)

    def shifted(rel):  # a reconstruction at relative distance `rel` from ours
        D = rng.standard_normal(WH.shape)
        return WH + D * (rel * xn / np.linalg.norm(D))

    pc = bench.ParityChecker(False)
    job = (X, W, H, 5, "frobenius")
    # (a) within 1e-5 of the float32 fit: plain pass, no exception
    pc.pool = FakePool([("scikit-learn", W, H, 1.0, shifted(3e-6))])
    out = pc.check([job], [(W, H, 1.0)], names=["a"])
    assert out["ok"] and out["exceptions"] == [] and out["all_within_tol_of_checker_fit_in_call_dtype"]
    # (b) 3e-5 from the float32 fit, 0 from the float64 fit, sklearn's two fits 3e-5 apart: a NAMED, tolerated exception
    W32 = W.copy()
    W32[0, 0] += np.float32(3e-5 * xn / np.linalg.norm(H[0]))
    pc.pool = FakePool([("scikit-learn", W32, H, 1.0, WH)])
    out = pc.check([job], [(W, H, 1.0)], names=["b"])
    assert out["ok"] and not out["all_within_tol_of_checker_fit_in_call_dtype"]
    assert [e["name"] for e in out["exceptions"]] == ["b"] and out["exceptions"][0]["tolerated"]
    # (c) the same distance from the float32 fit while sklearn's own fits agree: fails
    pc.pool = FakePool([("scikit-learn", W32, H, 1.0, W32.astype(np.float64) @ H.astype(np.float64))])
    out = pc.check([job], [(W, H, 1.0)], names=["c"])
    assert not out["ok"] and not out["exceptions"][0]["tolerated"]
    # (d) float64 calls are gated at 1e-9 against the float64 fit, no second clause
    X64, W64, H64 = X.astype(np.float64), W.astype(np.float64), H.astype(np.float64)
    pc.pool = FakePool([("scikit-learn", W64 * (1 + 1e-7), H64, 1.0, None)])
    out = pc.check([(X64, W64, H64, 5, "frobenius")], [(W64, H64, 1.0)])
    assert out["tol"] == 1e-9 and not out["ok"]
    pc.pool = None


def test_config3_is_strong_scaling_by_default():
    """BASELINE.json: 'batch 4096 ... scattered across 1 -> 8 MI355X'.  --batch is the total; --batch-per-gpu the weak form."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'lo_b, hi_b = partition(total, cx.world)[cx.rank]' in src and '"--batch-per-gpu"' in src
    from muscle_synergies_amd.engine import partition

    for n in (1, 2, 4, 8):
        parts = partition(4096, n)
        assert sum(hi - lo for lo, hi in parts) == 4096 and len(parts) == n

"""Host side of libhip_nmf.so under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on
the CPU build only -- the device code is NOT instrumented: -fno-gpu-sanitize).  The four translation units that hold
host logic (argument validation, workspace carve-up, launch orchestration) are rebuilt with -fsanitize=address,undefined
and driven through every entry point's error paths by tools/abi_host_drive.py in a child process that preloads the
sanitizer runtime."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

HOST_TUS = ("hipnmf_api", "hipnmf_envelope", "hipnmf_init", "hipnmf_sosfilt")


def _asan_runtime():
    hits = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return hits[0] if hits else None


def test_host_side_is_clean_under_asan_and_ubsan():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("clang's ASan runtime not found under /opt/rocm")
    from muscle_synergies_amd.build import build

    lib = build(variant="asan", only=HOST_TUS,
                extra_flags=("-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g"))
    assert os.path.exists(lib)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", HIPNMF_LIBRARY=lib,
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "abi_host_drive.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "abi-host-drive: ok" in r.stdout
    for marker in ("ERROR: AddressSanitizer", "runtime error:", "SUMMARY: UndefinedBehaviorSanitizer"):
        assert marker not in r.stderr, r.stderr[-4000:]

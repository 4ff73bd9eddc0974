"""Randomised engine-vs-oracle cross-check (tests/fuzz_gpu.py with a fixed seed): shapes from 1 x 1 to 12 345 x 32,
both dtypes and memory orders, every solver path, both losses, stop rule, regularisation, transform."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_fuzz_against_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_gpu.py"), "--cases", "120", "--seed", "7"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 problems" in r.stdout

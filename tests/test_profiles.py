"""Committed measurements must not outlive the code they describe (CPU; needs the git history, skipped without it)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = "muscle_synergies_amd/csrc/"
# kernel family -> the headers that hold its device code
HEADERS = {
    "fit_persistent_kernel": ["nmf_kernels.hpp"],
    "fit_rowlane_kernel": ["nmf_rowlane.hpp", "nmf_kernels.hpp"],
    "fit_wide4_kernel": ["nmf_wide4.hpp"],
    "fit_wide_kernel": ["nmf_wide.hpp"],
    "big1_pass_kernel": ["nmf_big1.hpp", "nmf_big.hpp"],
}


def _git(*args):
    return subprocess.run(["git", "-C", ROOT, *args], capture_output=True, text=True)


def test_traffic_entries_are_not_older_than_their_kernels():
    """profiles/traffic.json feeds bench.py's roofline.traffic: an entry whose source_commit predates the last commit that touched
    the kernel's header describes code that no longer ships (round-4 finding: a round-3 PMC number on a round-4 kernel)."""
    if _git("rev-parse", "--git-dir").returncode != 0:
        pytest.skip("no git history here")
    entries = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert entries
    for e in entries:
        fam = next((f for f in HEADERS if e["kernel"].startswith(f)), None)
        assert fam, f"no header mapping for {e['kernel']}: add it to tests/test_profiles.py"
        paths = [CSRC + h for h in HEADERS[fam]]
        last = _git("log", "-1", "--format=%H", "--", *paths).stdout.strip()
        assert last, paths
        src = e.get("source_commit", "")
        if _git("cat-file", "-e", src + "^{commit}").returncode != 0:
            pytest.fail(f"{e['kernel']}: source_commit {src!r} is not a commit of this repository")
        # the kernel's last change must be the measured commit or one of its ancestors
        ok = _git("merge-base", "--is-ancestor", last, src).returncode == 0
        assert ok, (f"{e['kernel']}: measured at {src} ({e.get('measured_round')}), but {paths} changed afterwards in {last[:7]}: "
                    f"re-run tools/measure_traffic.sh (HIPNMF_SOURCE_COMMIT=$(git rev-parse --short HEAD)) for this workload")

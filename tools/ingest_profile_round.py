#!/usr/bin/env python3
"""Copy what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into profiles/ (the tracked evidence), and check every bench line
against the rocprofv3 kernel statistics of the same command: the dominant kernel's average duration must agree with the
HIP-event duration bench.py measured itself.

    python3 tools/ingest_profile_round.py r06"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
copied = []
for f in sorted(glob.glob(os.path.join(src, f"{tag}_*"))):
    if os.path.isfile(f) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
        copied.append(os.path.basename(f))
for name, to in (("traffic.json", os.path.join(dst, "traffic.json")), ("DISPATCH.md", os.path.join(ROOT, "DISPATCH.md"))):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, to)
        copied.append(name)
print(f"copied {len(copied)} files into profiles/")
rows = []
for j in sorted(glob.glob(os.path.join(dst, f"{tag}_bench*.json"))):
    try:
        d = json.loads([l for l in open(j).read().splitlines() if l.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001
        print("unreadable:", j, e)
        continue
    name = os.path.basename(j)[len(tag) + 1:-5]
    stats = os.path.join(dst, f"{tag}_kernel_stats_{'bench_steps2' if name == 'bench' else name}.csv")
    roof = d.get("roofline") or {}
    kms, kern = roof.get("kernel_ms_avg"), roof.get("kernel", "")
    best = None
    if os.path.exists(stats):
        fam = kern.split("<")[0].split(" ")[0]
        for r in csv.DictReader(open(stats)):
            if fam and fam in r["Name"]:
                avg = float(r["AverageNs"]) / 1e6
                if best is None or float(r["TotalDurationNs"]) > best[2]:
                    best = (r["Name"][:70], avg, float(r["TotalDurationNs"]), int(r["Calls"]))
    rows.append((name, d["value"], d["dtype"], roof.get("bound"), roof.get("frac"), kms, best, (d.get("parity") or {}).get("ok")))
print("%-24s %14s %5s %10s %7s %12s  %s" % ("bench line", "value", "dtype", "bound", "frac", "kernel ms", "rocprofv3: avg ms x calls of the same family"))
for name, v, dt, b, fr, kms, best, ok in rows:
    print("%-24s %14.1f %5s %10s %7.3f %12s  %s  parity=%s" % (name, v, dt, b, fr or 0, "%.3f" % kms if kms else "-",
          "%.3f ms x %d (%s)" % (best[1], best[3], best[0]) if best else "-", ok))

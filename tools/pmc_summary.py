#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel: python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, collections, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "fit_persistent"
for f in sorted(glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f, {k: f"{v:.4g} (n={n[k]})" for k, v in agg.items()})
for f in sorted(glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            print("  ", r["Kernel_Name"][:60], "dur_ms", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                  "vgpr", r["VGPR_Count"], "sgpr", r["SGPR_Count"], "lds", r["LDS_Block_Size"], "scratch", r["Scratch_Size"], "grid", r["Grid_Size_X"], "wg", r["Workgroup_Size_X"])

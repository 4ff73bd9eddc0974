#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin) as one line per kernel."""
import re, sys, subprocess
txt = sys.stdin.read()
cur = None
rows = []
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except Exception:
            pass
        name = re.sub(r"\(hipnmf::SolveArgs<\w+>\)", "", name).replace("hipnmf::", "").replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
        continue
    for key in ("TotalSGPRs", "VGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None and "remark:     " + key in line:
            cur[key] = int(m.group(1))
print(f"{'kernel':60s} {'VGPR':>5} {'SGPR':>5} {'scr':>5} {'occ':>4} {'vsp':>4} {'ssp':>4}")
for r in rows:
    print(f"{r['name'][:60]:60s} {r.get('VGPRs',0):5d} {r.get('TotalSGPRs',0):5d} {r.get('ScratchSize [bytes/lane]',0):5d} {r.get('Occupancy [waves/SIMD]',0):4d} {r.get('VGPRs Spill',0):4d} {r.get('SGPRs Spill',0):4d}")

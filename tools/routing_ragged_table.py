#!/usr/bin/env python3
"""Join the three logs of tools/routing_ragged_audit.sh: per call the wall time under the library's own routing, with the lane
mappings pinned and with the matrix-pipe kernels pinned; LOSING marks a default more than 10 % slower than the better pinned route."""
import re
import sys


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r"(\S+ \S+ \S+ m=\d+ k=\S+ trials=\d+ rows=\S+ iters=\d+): wall ([\d.]+) ms, device (\S+) ms, .*?(\[.*\])\s*$", line)
        if m:
            out[m.group(1)] = (float(m.group(2)), m.group(4))
    return out


d, lanes, mat = (parse(p) for p in sys.argv[1:4])
losing = 0
print("%-86s %9s %9s %9s  %s" % ("call", "default", "lanes", "matrix", "kernel(s) of the default route"))
for key, (t, kern) in d.items():
    tl, tm = lanes.get(key, (float("nan"), ""))[0], mat.get(key, (float("nan"), ""))[0]
    best = min(x for x in (tl, tm) if x == x) if any(x == x for x in (tl, tm)) else float("nan")
    flag = ""
    if best == best and t > 1.10 * best:
        flag = "  LOSING (%.2fx)" % (t / best)
        losing += 1
    print("%-86s %9.2f %9.2f %9.2f  %s%s" % (key, t, tl, tm, kern, flag))
print("ROUTING-RAGGED cases=%d losing_defaults=%d  (wall ms per call, best of 3; lanes = HIPNMF_FORCE_WIDE=-1, matrix = HIPNMF_FORCE_WIDE=1)" % (len(d), losing))

#!/usr/bin/env python3
"""ISA of a kernel's hottest loop (the INNERMOST s_cbranch back-edge block with the most VALU instructions) with a count per
instruction category.  Used for profiles/r02_isa_*.txt:

    python tools/isa_tile_loop.py inst_f32_g1c16 'fit_persistent_kernelIfLi1ELi16ELi5ELi0E' > profiles/r02_isa_fit_persistent_k5_tile_loop.txt
    python tools/isa_tile_loop.py inst_f32_rowlane 'fit_rowlane_kernelILi5E' > profiles/r02_isa_fit_rowlane_k5_tile_loop.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tu, pat = sys.argv[1], sys.argv[2]
must = sys.argv[3] if len(sys.argv) > 3 else "v_rcp_f32"  # the loop must contain this mnemonic (the W-update quotient)
merge = True  # two back edges that overlap without nesting (fast path / exact-division path of one tile loop) count as one loop
src = os.path.join(ROOT, "muscle_synergies_amd", "csrc", tu + ".hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-I", os.path.dirname(src),
                    "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)s_endpgm" % re.escape(pat), text, flags=re.S | re.M)
if not m:
    sys.exit(f"kernel matching {pat} not found")
name, body = m.group(1), m.group(2).split("\n")
labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
loops = []
for i, l in enumerate(body):
    b = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if b and b.group(1) in labels and labels[b.group(1)] < i:  # back edge
        loops.append((labels[b.group(1)], i))
if merge:
    changed = True
    while changed:
        changed = False
        for a in list(loops):
            for b in list(loops):
                if a != b and a[0] < b[0] <= a[1] < b[1]:
                    loops.remove(a); loops.remove(b); loops.append((a[0], b[1])); changed = True
                    break
            if changed:
                break
best = None
for lo, hi in loops:
    if any(lo <= a and b <= hi and (a, b) != (lo, hi) for a, b in loops):
        continue  # not innermost
    if not any(must in x for x in body[lo:hi + 1]):
        continue
    n_valu = sum(1 for x in body[lo:hi + 1] if re.match(r"\s+v_", x))
    if best is None or n_valu > best[0]:
        best = (n_valu, lo, hi)


def cat(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith(("v_fmac_f32", "v_fma_f32", "v_pk_fma")): return "VALU fma"
    if op.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32")): return "VALU mul/add"
    if op.startswith("v_rcp"): return "VALU rcp"
    if op.startswith(("v_cmp", "v_cndmask", "v_min", "v_max")): return "VALU compare/select"
    if op.startswith(("v_mov", "v_accvgpr")): return "VALU move"
    if op.startswith(("v_add_u32", "v_lshl", "v_and", "v_or", "v_mad_u", "v_add_co", "v_sub_u", "v_mul_lo", "v_mul_hi", "v_ashr", "v_lshr", "v_readfirstlane", "v_readlane", "v_writelane")): return "VALU integer/address"
    if op.startswith("v_"): return "VALU other"
    if op.startswith(("buffer_", "global_", "scratch_", "flat_")): return "VMEM " + ("store" if "store" in op else "load")
    if op.startswith("ds_"): return "LDS"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "SALU / branch"
    return "other"


n_valu, lo, hi = best
blk = [l for l in body[lo:hi + 1] if l.strip() and not l.strip().startswith(";")]
counts = collections.Counter()
for l in blk:
    t = l.split()
    if t and not t[0].endswith(":"):
        counts[cat(t[0])] += 1
print(f"kernel {name}")
print(f"hottest loop: lines {lo}..{hi} of the kernel body, {sum(counts.values())} instructions, {n_valu} VALU + MFMA")
for k, v in counts.most_common():
    print(f"  {k:24s} {v}")
print()
print("\n".join(l.split(";")[0].rstrip() for l in blk))

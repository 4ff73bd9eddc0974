"""ISA lint for the hipcc (ROCm 7.2, LLVM AMDGPU) register-allocator defect that made fit_small_kernel<double,16,5> return wrong
factors in round 3 (profiles/r04_small_f64_miscompile.md): a 64-bit value is spilled HALF to scratch ("4-byte Folded Spill") and
HALF to an AGPR ("Reload Reuse"); at one reload site both halves come back, at another only the scratch half is reloaded and the
register that should receive the AGPR half is read uninitialised.

For every kernel of every device assembly file given: find each (scratch offset X, AGPR aN) split pair -- a `Reload Reuse`
v_accvgpr_write of v(M+1) [or vM] next to a 4-byte folded spill of vM [or v(M+1)] -- and require that EVERY 4-byte folded reload
of offset X has a read of aN next to it.  Prints the offending kernels; exit code 1 if any.

    python3 tools/isa_split_spill_lint.py muscle_synergies_amd/csrc/_build/*.s
"""
import re
import subprocess
import sys

NEAR_SPILL, NEAR_RELOAD = 4, 16
RE_FUNC = re.compile(r"^(_Z\w+):\s*(;.*)?$")
RE_REUSE_W = re.compile(r"v_accvgpr_write_b32 a(\d+), v(\d+)\s*;\s*Reload Reuse")
RE_SPILL4 = re.compile(r"scratch_store_dword off, v(\d+), \w+(?: offset:(\d+))?\s*; 4-byte Folded Spill")
RE_RELOAD4 = re.compile(r"scratch_load_dword [av](\d+), off, \w+(?: offset:(\d+))?\s*; 4-byte Folded Reload")


def kernels(path):
    name, body = None, []
    for line in open(path, errors="replace"):
        m = RE_FUNC.match(line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif line.startswith(".Lfunc_end"):
            if name:
                yield name, body
            name, body = None, []
        elif name is not None:
            body.append(line)
    if name:
        yield name, body


def lint_kernel(body):
    reuse_w = [(i, int(m.group(1)), int(m.group(2))) for i, ln in enumerate(body) for m in [RE_REUSE_W.search(ln)] if m]
    if not reuse_w:
        return [], 0
    spills = [(i, int(m.group(1)), int(m.group(2) or 0)) for i, ln in enumerate(body) for m in [RE_SPILL4.search(ln)] if m]
    reloads = [(i, int(m.group(2) or 0)) for i, ln in enumerate(body) for m in [RE_RELOAD4.search(ln)] if m]
    pairs = set()
    for i, areg, vreg in reuse_w:
        for j, sreg, off in spills:
            if abs(i - j) <= NEAR_SPILL and abs(sreg - vreg) == 1:
                pairs.add((off, areg))
    problems = []
    for off, areg in sorted(pairs):
        pat = re.compile(r"v_accvgpr_(?:read|mov)_b32 \w+, a%d\b" % areg)
        for j, roff in reloads:
            if roff != off:
                continue
            lo, hi = max(0, j - NEAR_RELOAD), min(len(body), j + NEAR_RELOAD + 1)
            if not any(pat.search(body[q]) for q in range(lo, hi)):
                problems.append("scratch offset %d is half of a value whose other half lives in a%d, but the reload at kernel line %d "
                                "restores only the scratch half: %s" % (off, areg, j + 1, body[j].strip()))
    return problems, len(pairs)


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        return name


def lint_files(paths):
    bad, n_kernels, n_split = [], 0, 0
    for path in paths:
        for name, body in kernels(path):
            n_kernels += 1
            problems, pairs = lint_kernel(body)
            n_split += 1 if pairs else 0
            for p in problems:
                bad.append((path, demangle(name), p))
    return bad, n_kernels, n_split


def main(argv):
    bad, n_kernels, n_split = lint_files(argv)
    for path, name, p in bad:
        print("SPLIT-SPILL DEFECT %s\n  %s\n  %s" % (path, name, p))
    print("isa_split_spill_lint: %d kernel(s) in %d file(s), %d with a value split between scratch and an AGPR, %d defective reload(s)"
          % (n_kernels, len(argv), n_split, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

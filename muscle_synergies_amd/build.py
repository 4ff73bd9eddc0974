"""Build libhip_nmf.so (hand-written HIP for gfx950) in-tree.

    python -m muscle_synergies_amd.build [--force] [--jobs N]

One `hipcc -c` per translation unit (run in parallel), then one link step.  The shared library lands in
``muscle_synergies_amd/lib/libhip_nmf.so`` so that it travels with a repo snapshot to the GPU box.
hipcc cross-compiles gfx950 code objects without a GPU.
"""
from __future__ import annotations

import argparse
import os
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libhip_nmf.so")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")

ARCH = "gfx950"
# -fno-slp-vectorize: hipcc otherwise packs the fp32 FMAs into v_pk_fma_f32 plus v_mov shuffles, which costs
# ~40 VGPRs and measured 14 % slower on MI355X (profiles/README.md, round 1).
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-slp-vectorize", "-Wall",
            "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the ROCm toolchain is required to build libhip_nmf.so")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(INCLUDE, "hip_nmf.h"))
    return hs


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _obj_stale(obj: str, src: str, hdrs) -> bool:
    """An object is stale when its source or one of the project headers IT INCLUDES is newer (hipcc -MD writes the list next to
    the object); without a dependency file every project header counts."""
    dep = obj[:-2] + ".d"
    if not os.path.exists(obj) or not os.path.exists(dep):
        return _stale(obj, [src, *hdrs])
    try:
        words = open(dep).read().replace("\\\n", " ").split()
    except OSError:
        return True
    root = os.path.dirname(PKG)
    mine = [w for w in words[1:] if os.path.abspath(w).startswith(root) and os.path.exists(w)]
    return _stale(obj, [src, *mine])


def _device_asm(src: str, objdir: str) -> str:
    return os.path.join(objdir, os.path.basename(src)[:-4] + f"-hip-amdgcn-amd-amdhsa-{ARCH}.s")


def _compile(src: str, extra, objdir: str = OBJ) -> str:
    """One translation unit.  -save-temps=obj keeps the device assembly next to the object (same code generation, the
    assembly is what the ISA lint below reads); the other intermediates are deleted again."""
    base = os.path.basename(src)[:-4]
    obj = os.path.join(objdir, base + ".o")
    cmd = [_hipcc(), *CXXFLAGS, *extra, "-I", CSRC, "-save-temps=obj", "-MD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    keep = {obj, _device_asm(src, objdir)}
    for f in os.listdir(objdir):
        full = os.path.join(objdir, f)
        if full not in keep and (f.startswith(base + "-hip-") or f.startswith(base + "-host-") or f.startswith(base + ".hip-")):
            os.remove(full)
    return obj


def lint_isa(srcs, objdir: str = OBJ, verbose: bool = False) -> None:
    """Refuse a build whose device code shows the split-spill defect of hipcc that made one fp64 instance return wrong factors in
    round 3 (tools/isa_split_spill_lint.py, profiles/r04_small_f64_miscompile.md).  Reads the assembly -save-temps left."""
    tools = os.path.join(os.path.dirname(PKG), "tools")
    if not os.path.exists(os.path.join(tools, "isa_split_spill_lint.py")):
        return  # installed without the repo's tools/: nothing to run
    import importlib.util

    spec = importlib.util.spec_from_file_location("isa_split_spill_lint", os.path.join(tools, "isa_split_spill_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    asm = [a for a in (_device_asm(s, objdir) for s in srcs) if os.path.exists(a)]
    bad, n_kernels, n_split = lint.lint_files(asm)
    if verbose:
        print(f"[build] ISA lint: {n_kernels} kernels in {len(asm)} assembly file(s), {n_split} with a scratch/AGPR split value, "
              f"{len(bad)} defective reload(s)", flush=True)
    if bad:
        raise RuntimeError("hipcc emitted the split-spill defect (a 64-bit value reloaded by halves from scratch and an AGPR, one half "
                           "missing): refusing the library.  (The check is a pattern match on the assembly's spill comments; if it "
                           "misfires on another hipcc, HIPNMF_SKIP_ISA_LINT=1 builds without it.)\n"
                           + "\n".join(f"{a_}: {n}: {p}" for a_, n, p in bad))


def build(force: bool = False, jobs: int | None = None, extra_flags=(), verbose: bool = False,
          variant: str | None = None, only=(), record_profile: bool = False) -> str:
    """Build the library.  ``variant`` (development aid) builds ``libhip_nmf_<variant>.so`` with
    ``extra_flags`` in its own object directory; select it at run time with ``HIPNMF_LIBRARY``.
    ``only``: with a variant, the translation units (base names without ``.hip``) the flags apply to -- every
    other object is taken from the default build (which is brought up to date first)."""
    objdir = OBJ if not variant else os.path.join(CSRC, "_build_" + variant)
    lib = LIB if not variant else os.path.join(LIBDIR, f"libhip_nmf_{variant}.so")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = headers()
    srcs = sources()
    if variant and only:
        build(jobs=jobs, verbose=verbose)  # default objects for the shared translation units
        mine = [s for s in srcs if os.path.basename(s)[:-4] in only]
        stamp = os.path.join(objdir, "flags.txt")  # same flags + objects newer than the sources: nothing to do
        same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(extra_flags)
        todo = [s for s in mine if force or not same_flags
                or _obj_stale(os.path.join(objdir, os.path.basename(s)[:-4] + ".o"), s, hdrs)]
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(lambda s: _compile(s, list(extra_flags), objdir), todo))
        with open(stamp, "w") as f:
            f.write(" ".join(extra_flags))
        objs = [os.path.join(objdir if s in mine else OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
        if todo or _stale(lib, objs):
            cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        return lib
    todo = [s for s in srcs
            if force or _obj_stale(os.path.join(objdir, os.path.basename(s)[:-4] + ".o"), s, hdrs)]
    skip_lint = bool(variant) or os.environ.get("HIPNMF_SKIP_ISA_LINT") == "1"
    if not skip_lint:
        # an object whose assembly is gone cannot be linted: it is compiled again rather than trusted
        todo += [s for s in srcs if s not in todo and not _lint_ok(s, objdir) and not os.path.exists(_device_asm(s, objdir))]
    jobs = jobs or min(8, os.cpu_count() or 1)
    t_compile = time.monotonic()
    if todo:
        if verbose:
            print(f"[build] compiling {len(todo)} translation unit(s) for {ARCH} with {jobs} job(s)", flush=True)
        for s_ in todo:  # a recompiled unit loses its stamp first: if this call dies half-way the next one lints it again
            _drop(_lint_stamp(s_, objdir))
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(lambda s: _compile(s, list(extra_flags), objdir), todo))
    t_compile = time.monotonic() - t_compile
    if not skip_lint:
        # BEFORE linking, and over every unit that has no 'lint passed' stamp newer than its object -- not only over the units
        # this call compiled (round-5 advisor finding: a refused library used to be accepted by the next call, which found nothing
        # to compile and skipped the lint).  A refused unit loses its object, assembly and stamp, and the library is removed:
        # re-running the build can only reproduce the refusal or succeed on different code.
        pending = [s_ for s_ in srcs if not _lint_ok(s_, objdir)]
        try:
            lint_isa(pending, objdir, verbose)
        except RuntimeError as e:
            bad_asm = {line.split(":", 1)[0] for line in str(e).splitlines() if ARCH + ".s" in line}
            for s_ in pending:
                if _device_asm(s_, objdir) in bad_asm or not bad_asm:
                    base = os.path.join(objdir, os.path.basename(s_)[:-4])
                    for f in (base + ".o", base + ".d", _device_asm(s_, objdir), _lint_stamp(s_, objdir)):
                        _drop(f)
            _drop(lib)
            raise
        for s_ in pending:
            with open(_lint_stamp(s_, objdir), "w") as f:
                f.write("isa_split_spill_lint passed\n")
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    linked = False
    if todo or _stale(lib, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        linked = True
        if verbose:
            print(f"[build] linked {lib}", flush=True)
    if not variant:
        _record(lib, srcs, todo, linked, verbose, record_profile, compile_seconds=t_compile, jobs=jobs)
    return lib


def _drop(path: str) -> None:
    try:
        os.remove(path)
    except OSError:
        pass


def _lint_stamp(src: str, objdir: str) -> str:
    return os.path.join(objdir, os.path.basename(src)[:-4] + ".lint_ok")


def _lint_ok(src: str, objdir: str) -> bool:
    """The unit's object passed the ISA lint: a stamp written after the lint, not older than the object."""
    stamp, obj = _lint_stamp(src, objdir), os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
    return os.path.exists(stamp) and os.path.exists(obj) and os.path.getmtime(stamp) >= os.path.getmtime(obj)


def _record(lib: str, srcs, todo, linked: bool, verbose: bool, record_profile: bool = False, compile_seconds: float = 0.0,
            jobs: int = 0) -> None:
    """What this call did -- compiled from source or found up to date -- next to the library (lib/build_info.json, untracked).
    Only ``record_profile`` (``--record-profile``, used by tools/profile_round.sh) also updates the tracked
    profiles/build_info.json: an ordinary rebuild must not dirty the work tree."""
    import hashlib
    import json
    import time

    with open(lib, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    info = {
        "library": os.path.relpath(lib, os.path.dirname(PKG)),
        "sha256": sha,
        "arch": ARCH,
        "translation_units": len(srcs),
        "compiled_in_this_call": sorted(os.path.basename(s) for s in todo),
        "linked_in_this_call": linked,
        "build_mode": ("compiled all from source" if len(todo) == len(srcs) else
                       "compiled %d of %d translation units" % (len(todo), len(srcs)) if todo else "up to date: reused"),
        "flags": CXXFLAGS,
        "compile_seconds": round(compile_seconds, 1), "jobs": jobs,
        "time_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
    }
    paths = [os.path.join(LIBDIR, "build_info.json")]
    if record_profile and os.path.isdir(os.path.join(os.path.dirname(PKG), "profiles")):
        paths.append(os.path.join(os.path.dirname(PKG), "profiles", "build_info.json"))
    lib_record = os.path.join(LIBDIR, "build_info.json")
    for path in paths:
        try:
            old = json.load(open(path)) if os.path.exists(path) else {}
            if "profiles" in path and not todo and not linked and os.path.exists(lib_record):
                # --record-profile on an up-to-date library: the tracked record takes the record of the build that PRODUCED this
                # library (kept beside it), not "reused"
                made = json.load(open(lib_record))
                if made.get("sha256") == sha:
                    with open(path, "w") as f:
                        json.dump({k: v for k, v in made.items() if k not in ("last_checked_utc", "last_check")}, f, indent=1)
                    continue
            if not todo and not linked and old.get("sha256") == sha:
                if "profiles" in path:
                    continue  # the tracked record keeps the build that produced this library; reuse is logged beside the .so
                info_w = dict(old, last_checked_utc=info["time_utc"], last_check="up to date: reused")
            else:
                info_w = info
            with open(path, "w") as f:
                json.dump(info_w, f, indent=1)
        except OSError:
            pass
    if verbose:
        print(f"[build] {info['build_mode']}; sha256 {sha[:16]}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--flag", action="append", default=[], help="extra hipcc flag (repeatable), e.g. --flag=-DHIPNMF_PF=3")
    ap.add_argument("--variant", default=None, help="build lib/libhip_nmf_<variant>.so instead of the default library")
    ap.add_argument("--only", action="append", default=[],
                    help="with --variant: translation unit (base name) the flags apply to (repeatable); the rest is shared")
    ap.add_argument("--record-profile", action="store_true", help="also update the tracked profiles/build_info.json")
    a = ap.parse_args()
    print(build(force=a.force, jobs=a.jobs, extra_flags=a.flag, verbose=True, variant=a.variant, only=a.only,
                record_profile=a.record_profile))

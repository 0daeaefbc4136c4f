r"""Builds libzuko_amd.so (gfx950 only) with hipcc — no torch extension machinery involved.

    python zuko_amd/_build.py          # incremental (run as a script: it must not import the package)
    python zuko_amd/_build.py --force

Objects and the shared library land in zuko_amd/lib/ (git-ignored, but shipped to the GPU box
with the repo snapshot).
"""

from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "libzuko_amd.so"
ARCH = "gfx950"
MAX_JOBS = int(os.environ.get("ZUKO_AMD_BUILD_JOBS", "12"))  # concurrent hipcc processes (the heavy units take 2-3 GB each)

# translation unit -> extra flags.  The univariate math is built without FMA contraction so that
# its expression trees round like the reference's op-by-op PyTorch evaluation.
SOURCES = {
    "elementwise.hip": ["-ffp-contract=off"],
    "linear.hip": [],
    "linear_bf16.hip": ["-ffp-contract=off"],  # its spline epilogue must round like elementwise.hip's
    "linear_bf16_lanes.hip": ["-ffp-contract=off"],
    "fused_ar.hip": ["-ffp-contract=off"],
    "fused_ar_gsplit.hip": ["-ffp-contract=off"],
    "backward.hip": [],
    "train.hip": [],
    "gemm_half.hip": [],
    # (pragma-unroll-threshold: the 8 x 32 tile loop of a 512-wide layer exceeds the default cap of forced unrolling; a
    #  partially unrolled loop indexes the activation arrays dynamically, which puts them in scratch memory: 4x slower)
    "fused_coupling.hip": ["-ffp-contract=off", "-mllvm", "-pragma-unroll-threshold=1000000"],
    "inc_inverse.hip": ["-ffp-contract=off"],
    "backward_poly.hip": ["-ffp-contract=off"],
    "backward_bern.hip": ["-ffp-contract=off"],
    "backward_bern_u.hip": ["-ffp-contract=off"],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"] + (["-DZK_INC_FAST_BUILD"] if os.environ.get("ZUKO_AMD_FAST_BUILD") == "1" else [])


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _torch_lib_dir() -> str:
    import importlib.util

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        raise RuntimeError("zuko_amd build: PyTorch-ROCm is required (its libamdhip64.so is the HIP runtime we link against)")
    d = os.path.join(os.path.dirname(spec.origin), "lib")
    if not os.path.exists(os.path.join(d, "libamdhip64.so")):
        raise RuntimeError(f"zuko_amd build: {d}/libamdhip64.so not found (need a ROCm build of PyTorch)")
    return d


def _includes(path: str, seen: set) -> None:
    """Transitive closure of the `#include "x.h"` lines of a translation unit, within csrc/."""
    import re

    with open(path, "r", errors="replace") as f:
        text = f.read()
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        full = os.path.normpath(os.path.join(os.path.dirname(path), name))  # (also ../../include/zuko_amd.h)
        if full not in seen and os.path.exists(full):
            seen.add(full)
            _includes(full, seen)


def _digest(path: str, flags: list[str]) -> str:
    """Flags + the unit + the headers it actually includes (a header only the generated static-shape kernels use — zuko_amd/static_ar.py
    keeps its own digest — does not rebuild the library)."""
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    with open(path, "rb") as f:
        h.update(f.read())
    seen: set = set()
    _includes(path, seen)
    for full in sorted(seen):
        h.update(os.path.basename(full).encode())
        with open(full, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        flags = COMMON + extra
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        stamp = obj + ".sha"
        dig = _digest(path, flags)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((path, obj, stamp, dig, flags))

    def compile_one(job):
        import time

        path, obj, stamp, dig, flags = job
        cmd = [hipcc, *flags, "-c", path, "-o", obj]
        if verbose:
            print("[zuko_amd build]", " ".join(cmd), flush=True)
        t0 = time.time()
        subprocess.run(cmd, check=True)
        if verbose:
            print(f"[zuko_amd build] {os.path.basename(path)}: {time.time() - t0:.0f} s", flush=True)
        with open(stamp, "w") as f:
            f.write(dig)

    if jobs:
        # every translation unit at once (nine, the longest ~4 min on its own): the wall time is the slowest unit's; the long ones first
        jobs.sort(key=lambda j: -os.path.getsize(j[0]))
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(jobs), MAX_JOBS)) as ex:
            list(ex.map(compile_one, jobs))
    out = lib_path()
    if jobs or not os.path.exists(out):
        # Link against the SAME HIP runtime PyTorch uses (its bundled libamdhip64.so, soname
        # "libamdhip64.so"), never the system one (soname "libamdhip64.so.7"): two runtimes in one
        # process would give our kernels their own, unordered null stream.  No rpath is recorded;
        # zuko_amd._C imports torch first so the soname is already resident when we are dlopen'ed.
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-no-hip-rt", *objs, f"-L{_torch_lib_dir()}", "-l:libamdhip64.so", "-o", out]
        if verbose:
            print("[zuko_amd build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    prebuild_static(verbose)
    return out


def prebuild_static(verbose: bool = True) -> None:
    """Static-shape instantiations of the fused autoregressive kernel for the BASELINE.json conditioners and a few common
    shapes (zuko_amd/static_ar.py: PREBUILT): generated + compiled into zuko_amd/lib/ars/, a no-op when they are current.
    Runs in a child process because it imports the package (which needs the library that was just linked)."""
    code = "import sys; sys.path.insert(0, %r); import zuko_amd.static_ar as s; s.prebuild(verbose=%r, jobs=%d)" % (os.path.dirname(HERE), bool(verbose), min(MAX_JOBS, max(4, os.cpu_count() or 4)))
    r = subprocess.run([sys.executable, "-c", code])
    if r.returncode != 0:
        # best effort: the library itself is built; a conditioner without a prebuilt kernel compiles one on first use or runs the generic kernel
        sys.stderr.write("[zuko_amd build] WARNING: prebuilding the static-shape kernels failed (exit code %d); the library is usable without them\n" % r.returncode)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

"""Build libtstar_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m tstar_amd.build [--force]

Per-source object files are cached under tstar_amd/csrc/build/ and rebuilt when
the source or any header is newer.  The .so is git-ignored but travels to the
GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libtstar_hip.so")
FIT_LIB = os.path.join(HERE, "libtstar_fitpack.so")     # host-only (no HIP runtime): the spline workers load it in a fraction of a second
FIT_SRC = os.path.join(CSRC, "fitpack.cpp")
# -ffp-contract=off: the FITPACK restatement is bit-identical to scipy's only without fused multiply-adds;
# -fno-math-errno lets sqrt vectorise and changes no result; never -ffast-math
FIT_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-math-errno", "-Wall"]
ARCH = "gfx950"

# -ffp-contract=off on the searcher file keeps its float64 arithmetic bit-identical
# to the numpy statement of the reference (no FMA contraction).
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-function"]
PER_FILE = {"searcher.hip": ["-ffp-contract=off"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "tstar_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdr = _newest_header()
    jobs = []
    objs = []
    for src in sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(op)
        stale = force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr)
        if stale:
            jobs.append([hipcc] + FLAGS + PER_FILE.get(src, []) + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    need_link = force or jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if need_link:
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    # the FITPACK restatement is OPTIONAL at run time (spline_worker falls back to scipy's own fit, same bits): a host without g++,
    # or one where it does not compile, gets a warning, not a failed build.  Rebuilt when the source or the flags change.
    stamp = FIT_LIB + ".flags"
    flags_now = " ".join(FIT_FLAGS)
    flags_built = None
    if os.path.exists(stamp):
        with open(stamp) as f:
            flags_built = f.read()
    stale_fit = (force or not os.path.exists(FIT_LIB) or os.path.getmtime(FIT_LIB) < os.path.getmtime(FIT_SRC) or flags_built != flags_now)
    if stale_fit:
        gxx = os.environ.get("CXX") or shutil.which("g++")
        try:
            if not gxx:
                raise RuntimeError("g++ not found (set CXX)")
            run([gxx] + FIT_FLAGS + [FIT_SRC, "-o", FIT_LIB + ".tmp"])
            os.replace(FIT_LIB + ".tmp", FIT_LIB)
            with open(stamp, "w") as f:
                f.write(flags_now)
        except Exception as e:                         # noqa: BLE001 -- any failure here only costs the fast fit
            # a library OLDER than its source (or built with other flags) must not be loaded in its place: new Python against an old
            # binary may lack entry points (tstar_curfit_pairing) or differ in bits.  Without the file the workers take scipy's own fit.
            for stale in (FIT_LIB, FIT_LIB + ".tmp", stamp):
                try:
                    os.remove(stale)
                except OSError:
                    pass
            print(f"[build] WARNING: libtstar_fitpack.so not built ({str(e).splitlines()[0]}); the stale library was removed, "
                  "the spline fit falls back to scipy", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

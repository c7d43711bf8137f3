"""Build libaircraft_hip.so in-tree with hipcc for gfx950 (no cmake/ninja, no JIT cache).
One object per translation unit, compiled in parallel, then linked."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ = os.path.join(_HERE, "csrc", "_obj")
OUT = os.path.join(_HERE, "libaircraft_hip.so")
HEADERS = glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(_HERE, "..", "include", "aircraft_hip.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-pass-failed", "-ffp-contract=on"]
# Per-unit flags of the headline kernels (k_nn_step_sens<8,true> and its wave-pair twin own the whole 512-register file):
# two output tiles per accumulator chunk instead of four and SLP packing only where it is clearly profitable bring the
# register-spill scratch from 228 to 36 B/lane (HBM traffic 1.9x -> 1.1x of the algorithmic bytes) at +0.6 % speed
# (same-box A/Bs).  Results are bit-identical: neither changes the order of any accumulation.
UNIT_FLAGS = {
    # (-mllvm -amdgpu-sched-strategy=max-ilp was measured on these three units: it takes the instructions that consume the result
    # of the one or two before them from 158 of 689 to 12 in the default model's stage body, and the kernels run 2-6 % SLOWER:
    # they are bound by vector-ALU throughput, not by dependent issue — DESIGN.md §6.2)
    "an_inst_sens_poly": ["-fno-slp-vectorize"],
    "nn_inst_wt2_mfma_sens_w2": ["-fno-slp-vectorize"],  # (the small-net kernel at two waves per SIMD)
    "nn_inst_wt8_mfma_sens": ["-DAC_CH=2", "-mllvm", "-slp-threshold=6"],
    "nn_inst_wt8_mfma_pair": ["-DAC_CH=2", "-mllvm", "-slp-threshold=6"],
}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def source_sha() -> str:
    """Identity of the library BUILD INPUTS (kernel sources, headers, compile flags): stable across rebuilds of the same
    tree, different as soon as a kernel or a flag changes.  Measurements that are committed and quoted later
    (profiles/*_pmc_traffic.json) carry it, and bench.py quotes them only for the build they were taken on."""
    import hashlib

    hh = hashlib.sha256()
    for f in sources() + sorted(HEADERS):
        hh.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            hh.update(fh.read())
    hh.update(repr((CFLAGS, sorted(UNIT_FLAGS.items()))).encode())
    return hh.hexdigest()[:16]


def _deps(obj):
    """Headers a translation unit really includes (the compiler's own -MMD list from the last build of this object), so a
    change to one engine header rebuilds only the units built on it; all headers when no list exists yet."""
    d = obj + ".d"
    if not os.path.exists(d):
        return HEADERS
    txt = open(d).read().replace("\\\n", " ")
    files = [f for f in txt.split(":", 1)[-1].split() if f.endswith((".hpp", ".h")) and "/opt/rocm" not in f and not f.startswith("/usr")]
    return [f for f in files if os.path.exists(f)] or HEADERS


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, jobs: int | None = None, diag: bool = False) -> str:
    """diag=True builds libaircraft_hip_diag.so with -DAC_STAMPS (in-kernel phase stamps, tools/diag_stamps.py);
    it is a measurement aid, never loaded by the product path unless AIRCRAFT_HIP_LIB points at it."""
    global OBJ, OUT
    obj_dir, out = (OBJ + "_diag", OUT.replace(".so", "_diag.so")) if diag else (OBJ, OUT)
    cflags = CFLAGS + (["-DAC_STAMPS"] if diag else [])
    return _build(force, verbose, jobs, obj_dir, out, cflags)


def _build(force, verbose, jobs, OBJ, OUT, CFLAGS) -> str:
    os.makedirs(OBJ, exist_ok=True)
    todo = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + _deps(obj) + [os.path.abspath(__file__)]):
            todo.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = ["hipcc", *CFLAGS, *UNIT_FLAGS.get(os.path.basename(src)[:-4], []), "-MMD", "-MF", obj + ".d", "-c", src, "-o", obj]
        if verbose:
            print("[aircraft_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(cc, todo))
    if todo or force or _stale(OUT, objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", OUT, *objs]
        if verbose:
            print("[aircraft_amd.build]", " ".join(cmd[:6]), f"... ({len(objs)} objects)", flush=True)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, diag="--diag" in sys.argv)

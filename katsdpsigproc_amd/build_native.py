"""Build ``_native/libkatsdpsigproc_hip.so`` from ``csrc/*.hip`` with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the shared
library then travels to the GPU box in-tree. Usage: ``python -m katsdpsigproc_amd.build_native``.
"""

import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_native")
OUT = os.path.join(OUT_DIR, "libkatsdpsigproc_hip.so")

# -ffp-contract=off: the kernels reproduce numpy/pandas float arithmetic bit for bit,
# so a*b+c must never be fused behind our back (explicit fmaf where numpy fuses).
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


# The three-run long-band kernel repeats the median and threshold phases once per run of a
# lane; at the default limit the compiler gives up unrolling those loops, the run number
# becomes a run-time index and the 192 deviations of a lane move to scratch memory.
PER_SOURCE_FLAGS = {
    "flagger_fused_long3.hip": ["-mllvm", "-pragma-unroll-threshold=1000000"],
}


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source and link the shared library; returns its path."""
    hipcc = _hipcc()
    extra = os.environ.get("KSP_EXTRA_HIPCC_FLAGS", "").split()  # experiments only
    os.makedirs(OUT_DIR, exist_ok=True)
    # objects built with other flags (an experiment's -D...) must not be reused
    stamp = os.path.join(OUT_DIR, ".flags")
    flags_now = " ".join(FLAGS + extra)
    try:
        with open(stamp) as f:
            flags_before = f.read()
    except OSError:
        flags_before = None
    if flags_before != flags_now:
        force = True
    sources = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(HERE, "..", "include", "katsdpsigproc_hip.h")
    ]
    objects = []
    jobs = []
    for src in sources:
        obj = os.path.join(OUT_DIR, os.path.basename(src)[:-4] + ".o")
        objects.append(obj)
        # (every source is a dependency of every object: one translation unit includes
        # another's .hip file)
        if force or _stale(obj, sources + headers):
            own = PER_SOURCE_FLAGS.get(os.path.basename(src), [])
            jobs.append([hipcc] + FLAGS + own + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)}\n{proc.stdout}\n{proc.stderr}")
        return proc

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as pool:
        list(pool.map(run, jobs))
    if jobs or force or _stale(OUT, objects):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objects)
    with open(stamp, "w") as f:
        f.write(flags_now)
    return OUT


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print("built", path)

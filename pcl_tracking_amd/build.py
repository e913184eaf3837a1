"""Build recipe for the HIP extension (gfx950 only), in-tree so the .so travels with the repo snapshot.

  python -m pcl_tracking_amd.build            # -> pcl_tracking_amd/_build/libpft_hip.so
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT_DIR = os.path.join(_HERE, "_build")
LIB = os.path.join(OUT_DIR, "libpft_hip.so")
SOURCES = ["pft_kernels.hip", "pft_octree.hip", "pft_likelihood.hip", "pft_population.hip", "pft_api.hip"]
HEADERS = ["pft_internal.h", "pft_device_utils.h", os.path.join("..", "..", "include", "pft.h")]

# -ffp-contract=off: PCL's float arithmetic on x86-64 has no FMA contraction; the greedy octree descent
# compares float sums, so contraction would flip near-ties (DESIGN.md "numerics").
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
    "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [hipcc()] + HIPCC_FLAGS + list(extra_flags) + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed")
    if verbose and (r.stdout or r.stderr):
        print(r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

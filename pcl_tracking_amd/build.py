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
SOURCES = ["pft_kernels.hip", "pft_octree.hip", "pft_octree_sorted.hip", "pft_likelihood.hip", "pft_population.hip", "pft_kld.hip", "pft_exact_nn.hip", "pft_hull.hip", "pft_api.hip",
           "pft_filters.hip"]
HEADERS = ["pft_internal.h", "pft_device_utils.h", os.path.join("..", "..", "include", "pft.h"),
           os.path.join("..", "..", "include", "pft_filters.h")]

# -ffp-contract=off: PCL's float arithmetic on x86-64 has no FMA contraction; the greedy octree descent
# compares float sums, so contraction would flip near-ties (DESIGN.md "numerics").
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
    # the SLP vectoriser packs adjacent f32 adds / muls into v_pk_add_f32 / v_pk_mul_f32, which issue at 1.1x the
    # scalar rate on gfx950 and cost v_mov shuffles on top: k_likelihood 215 -> 205 us without it
    "-fno-slp-vectorize",
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
    extra_flags = list(extra_flags) + os.environ.get("PFT_EXTRA_HIPCC_FLAGS", "").split()
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


EXAMPLES_DIR = os.path.join(_HERE, "examples")
EXAMPLE_BIN = os.path.join(OUT_DIR, "auto_tracking_amd")
DIST_EXAMPLE_BIN = os.path.join(OUT_DIR, "dist_tracking_amd")
_EXAMPLE_DEPS = [os.path.join(EXAMPLES_DIR, "tracking_app.hpp")] + [
    os.path.join(_HERE, "include", "pft", h) for h in ("particle_filter_tracker.hpp", "filters.hpp", "pcd_io.hpp", "common.hpp", "id_exchange.hpp")]


def _build_host_program(src, out, extra, force, verbose):
    lib = build()
    deps = [src, lib] + _EXAMPLE_DEPS
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    root = os.path.dirname(_HERE)
    cmd = [hipcc(), "-std=c++17", "-O2", "-Wall", "-ffp-contract=off", "-I", os.path.join(root, "include"),
           "-I", os.path.join(_HERE, "include"), src, "-o", out, "-L", OUT_DIR, "-lpft_hip", "-Wl,-rpath,$ORIGIN"] + extra
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("example build failed: " + os.path.basename(src))
    return out


def build_example(force=False, verbose=False):
    """the ROS-free C++ driver (host side in the reference's language) over the header-only mirror of the
    PCL classes and the C-ABI library: any number of objects, one tracker each"""
    return _build_host_program(os.path.join(EXAMPLES_DIR, "auto_tracking_amd.cpp"), EXAMPLE_BIN, [], force, verbose)


def build_dist_example(force=False, verbose=False):
    """the C++ multi-GPU host: one process per GPU, pft_dist_* phases with ncclAllReduce / ncclAllGather (RCCL) in
    between on the handle's stream"""
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    extra = ["-I", os.path.join(rocm, "include"), "-L", os.path.join(rocm, "lib"), "-lrccl", "-lpthread",
             "-Wl,-rpath," + os.path.join(rocm, "lib")]
    return _build_host_program(os.path.join(EXAMPLES_DIR, "dist_tracking_amd.cpp"), DIST_EXAMPLE_BIN, extra, force, verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_example(force="--force" in sys.argv, verbose=True))
    print(build_dist_example(force="--force" in sys.argv, verbose=True))

"""Host logic of A3's input selection (pcl_tracking_amd/csrc/pft_hull.hip), checked on the CPU: the bounding box of a rigidly
transformed reference cloud, evaluated in float32 exactly as the device and PCL's transformPointCloud + getMinMax3D do
(((r0 x + r1 y) + r2 z), then + t), must be the same over the support subset as over all points -- for thousands of random
orientations, exact quarter turns and clouds of different character; degenerate clouds keep every point.
No GPU is needed: the function is host code of the library, reached through a debug entry point of the C ABI."""
import ctypes as C

import numpy as np
import pytest

from pcl_tracking_amd import _lib, scene


def support_subset(pts):
    L = _lib.load()
    pts = np.ascontiguousarray(pts)
    keep = np.zeros(len(pts), np.uint32)
    n_keep = C.c_size_t(0)
    st = L.pft_debug_aabb_support_subset(pts.ctypes.data_as(C.c_void_p), len(pts), keep.ctypes.data_as(C.c_void_p), C.byref(n_keep))
    assert st == 0
    return keep[: n_keep.value].copy()


def rotations(rng, n):
    """n rotation matrices (float32, rows = the directions of the box's axes): uniformly random + the quarter turns"""
    q = rng.normal(0, 1, (n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                  2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                  2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)
    quarter = []
    for perm in ((0, 1, 2), (1, 2, 0), (2, 0, 1), (0, 2, 1), (2, 1, 0), (1, 0, 2)):
        for signs in ((1, 1, 1), (-1, 1, 1), (1, -1, 1), (1, 1, -1)):
            m = np.zeros((3, 3))
            for r in range(3):
                m[r, perm[r]] = signs[r]
            quarter.append(m)
    return np.concatenate([R, np.array(quarter)]).astype(np.float32)


def boxes(xyz, R):
    """min / max over the points of fl(fl(fl(r0 x) + fl(r1 y)) + fl(r2 z)) per rotation row, in float32 (no FMA)"""
    x, y, z = (xyz[:, k].astype(np.float32) for k in range(3))
    lo, hi = [], []
    for r in R.reshape(-1, 3):
        v = (r[0] * x + r[1] * y) + r[2] * z
        lo.append(v.min())
        hi.append(v.max())
    return np.array(lo, np.float32), np.array(hi, np.float32)


def cloud(xyz):
    m = np.zeros(len(xyz), scene.POINT_DTYPE)
    m["w"] = 1.0
    xyz = np.asarray(xyz, np.float32)
    m["x"], m["y"], m["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    return m


CASES = ["scan", "ball", "shell", "noisy_planes", "clusters_with_duplicates", "slab", "tiny", "huge_offset", "heavy_tail"]


@pytest.mark.parametrize("kind", CASES)
def test_box_over_the_support_subset_equals_box_over_all_points(kind):
    rng = np.random.default_rng(CASES.index(kind))
    n = 2048
    if kind == "scan":
        m = scene.make_model(n)
        xyz = np.stack([m["x"], m["y"], m["z"]], 1)
    elif kind == "ball":
        xyz = rng.uniform(-0.2, 0.2, (n, 3))
    elif kind == "shell":
        v = rng.normal(0, 1, (1500, 3))
        xyz = v / np.linalg.norm(v, axis=1, keepdims=True) * (0.3 + rng.normal(0, 3e-4, (1500, 1)))
    elif kind == "noisy_planes":
        xyz = rng.uniform(-0.2, 0.2, (n, 3))
        xyz[: n // 2, 2] = 0.1 + rng.normal(0, 1e-4, n // 2)
        xyz[n // 2:, 0] = -0.15 + rng.normal(0, 1e-6, n - n // 2)
    elif kind == "clusters_with_duplicates":
        c = rng.uniform(-1, 1, (12, 3))
        xyz = c[rng.integers(0, 12, n)] + rng.normal(0, 0.05, (n, 3)) * (rng.uniform(0, 1, (n, 1)) > 0.3)
    elif kind == "slab":
        xyz = rng.uniform(-1, 1, (n, 3)) * np.array([1.0, 0.7, 2e-3])
    elif kind == "tiny":
        xyz = rng.uniform(-1, 1, (n, 3)) * 1e-3
    elif kind == "huge_offset":
        xyz = rng.uniform(-0.2, 0.2, (n, 3)) + np.array([31.0, -17.0, 55.0])
    else:
        xyz = rng.standard_t(2.5, (n, 3)) * 0.1
    pts = cloud(xyz)
    xyz = np.stack([pts["x"], pts["y"], pts["z"]], 1)
    keep = support_subset(pts)
    assert len(keep) >= 4 and (np.diff(keep.astype(np.int64)) > 0).all() and keep[-1] < len(pts)
    R = rotations(rng, 3000)
    lo_all, hi_all = boxes(xyz, R)
    lo_sub, hi_sub = boxes(xyz[keep], R)
    np.testing.assert_array_equal(lo_sub, lo_all)
    np.testing.assert_array_equal(hi_sub, hi_all)
    if kind in ("scan", "ball", "noisy_planes", "huge_offset", "tiny"):
        assert len(keep) < len(pts) // 2, len(keep)


@pytest.mark.parametrize("kind", ["planar", "collinear", "few", "coincident", "nan"])
def test_degenerate_clouds_keep_every_point(kind):
    rng = np.random.default_rng(5)
    n = 500
    xyz = rng.uniform(-1, 1, (n, 3))
    if kind == "planar":
        xyz[:, 1] = 0.25
    elif kind == "collinear":
        xyz = np.outer(rng.uniform(-1, 1, n), [0.3, -0.2, 0.9])
    elif kind == "few":
        xyz = xyz[:50]
    elif kind == "coincident":
        xyz[:] = [0.1, 0.2, 0.3]
    else:
        xyz[17, 2] = np.nan
    keep = support_subset(cloud(xyz))
    np.testing.assert_array_equal(keep, np.arange(len(xyz)))


def test_the_subset_of_a_lattice_cloud_is_still_exact():
    """exactly coplanar faces and collinear edges (a lattice): whatever the hull code makes of the ties, the box is the box"""
    rng = np.random.default_rng(9)
    xyz = np.round(rng.uniform(-0.15, 0.15, (2048, 3)) / 0.05) * 0.05
    pts = cloud(xyz)
    xyz = np.stack([pts["x"], pts["y"], pts["z"]], 1)
    keep = support_subset(pts)
    R = rotations(rng, 3000)
    a, b = boxes(xyz, R), boxes(xyz[keep], R)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.parametrize("kind", ["sphere", "thick_shell", "ball"])
def test_largest_reference_cloud_is_bounded_in_time_and_exact(kind):
    """8 192 points (the size limit of the subset search) with MANY hull vertices -- a sphere bails out (more than
    half of the points are vertices: all are kept), a thick shell and a ball do not -- must not stall pft_set_reference:
    dead facets are compacted away, so a point is tested against the live facets only.  The box stays exact."""
    import time

    rng = np.random.default_rng(11)
    n = 8192
    v = rng.normal(0, 1, (n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    if kind == "sphere":
        xyz = v * 0.3
    elif kind == "thick_shell":
        xyz = v * rng.uniform(0.2, 0.3, (n, 1))
    else:
        xyz = v * (rng.uniform(0, 1, (n, 1)) ** (1 / 3)) * 0.3
    t0 = time.perf_counter()
    keep = support_subset(cloud(xyz))
    dt = time.perf_counter() - t0
    assert dt < 5.0, dt  # (0.1 - 0.6 s here; the un-compacted facet list took an order of magnitude longer on the sphere)
    if kind == "sphere":
        assert len(keep) == n
    else:
        assert len(keep) < n // 2
    R = rotations(rng, 300)
    lo_all, hi_all = boxes(xyz, R)
    lo_sub, hi_sub = boxes(xyz[keep], R)
    np.testing.assert_array_equal(lo_all, lo_sub)
    np.testing.assert_array_equal(hi_all, hi_sub)
    print(kind, "kept", len(keep), "of", n, "in %.2f s" % dt)

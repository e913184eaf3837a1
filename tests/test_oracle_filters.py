"""Hand-derived known answers for the oracle's input filters (oracle/pft_oracle_filters.c): PassThrough,
ApproximateVoxelGrid (history table + flush order), VoxelGrid.  CPU only.
PARITY UNPINNED: these pin the restatement of PCL 1.8.0, not PCL (oracle/pft_oracle.h)."""
import numpy as np

from pcl_tracking_amd import scene


def pts(rows):
    """rows of (x, y, z, r, g, b[, a])"""
    p = np.zeros(len(rows), scene.POINT_DTYPE)
    for i, r in enumerate(rows):
        p["x"][i], p["y"][i], p["z"][i] = r[0], r[1], r[2]
        a = r[6] if len(r) > 6 else 255
        p["rgba"][i] = (a << 24) | (r[3] << 16) | (r[4] << 8) | r[5]
    p["w"] = 1.0
    return p


def test_pass_through_limits_are_inclusive_and_nonfinite_is_dropped(orc):
    nan, inf = float("nan"), float("inf")
    c = pts([(0, 0, -0.1, 1, 1, 1), (0, 0, 0.0, 1, 1, 1), (0, 0, 5.0, 1, 1, 1), (0, 0, 10.0, 1, 1, 1),
             (0, 0, 10.001, 1, 1, 1), (0, 0, nan, 1, 1, 1), (nan, 0, 5.0, 1, 1, 1), (0, inf, 5.0, 1, 1, 1),
             (0, 0, inf, 1, 1, 1)])
    assert orc.pass_through(c, "z", 0.0, 10.0).tolist() == [1, 2, 3]
    # negative: keeps what lies strictly outside the limits; non-finite points are still dropped
    assert orc.pass_through(c, "z", 0.0, 10.0, negative=True).tolist() == [0, 4]
    assert orc.pass_through(c, "x", -1.0, 1.0).tolist() == [0, 1, 2, 3, 4]
    assert len(orc.pass_through(c[:0])) == 0


def test_approx_voxel_grid_flush_order_by_hand(orc):
    # leaf 0.01: cell = floor(coord * 100); entry = (ix*7171 + iy*3079 + iz*4231) & 511
    # A, B in cell (0,0,0) -> entry 0;  C in cell (512,0,0): 512*7171 & 511 = 0 -> collides, flushes {A,B};
    # E in cell (1,0,0): 7171 & 511 = 3;  D back in cell (0,0,0): flushes C;  F in cell (-1,0,0): (-7171) & 511 = 509
    A = (0.001, 0.001, 0.001, 10, 20, 30)
    B = (0.003, 0.005, 0.007, 20, 30, 41)
    Cc = (5.125, 0.002, 0.002, 200, 100, 50)
    E = (0.015, 0.001, 0.001, 1, 2, 3)
    D = (0.004, 0.004, 0.004, 7, 8, 9)
    F = (-0.001, 0.001, 0.001, 90, 80, 70)
    cloud = pts([A, B, Cc, E, D, F])
    out = orc.approx_voxel_grid(cloud, 0.01, 512)
    assert len(out) == 5
    f32 = np.float32
    # slot 0: centroid of A, B (float sums in arrival order, / 2.0f); colour truncates 35.5 -> 35; alpha byte 0
    assert out["x"][0] == (f32(A[0]) + f32(B[0])) / f32(2) and out["z"][0] == (f32(A[2]) + f32(B[2])) / f32(2)
    assert out["rgba"][0] == (15 << 16) | (25 << 8) | 35
    # slot 1: C flushed when D arrived;  then the open entries in table order: 0 (D), 3 (E), 509 (F)
    assert out["x"][1] == f32(Cc[0]) and out["rgba"][1] == (200 << 16) | (100 << 8) | 50
    assert [float(v) for v in out["x"][2:]] == [float(f32(D[0])), float(f32(E[0])), float(f32(F[0]))]
    assert (out["w"] == 1.0).all() and (out["pad"] == 0).all()
    # a larger table separates cell (512,0,0) (entry 512*7171 & 2047 = 512) from cell (0,0,0): A, B, D merge
    out2 = orc.approx_voxel_grid(cloud, 0.01, 2048)
    assert len(out2) == 4
    assert out2["x"][0] == ((f32(A[0]) + f32(B[0])) + f32(D[0])) / f32(3)
    assert len(orc.approx_voxel_grid(cloud[:0])) == 0


def test_approx_voxel_grid_same_voxel_revisited_after_flush_is_a_new_point(orc):
    # A (cell 0) | C (cell 512, same entry) | A' (cell 0 again): three outputs, the two visits of cell 0 are not merged
    cloud = pts([(0.001, 0, 0, 1, 1, 1), (5.125, 0, 0, 2, 2, 2), (0.002, 0, 0, 3, 3, 3)])
    out = orc.approx_voxel_grid(cloud, 0.01, 512)
    assert [int(v) & 255 for v in out["rgba"]] == [1, 2, 3]


def test_voxel_grid_by_hand(orc):
    # two voxels along x; output ordered by voxel index (x fastest), centroid incl. alpha
    cloud = pts([(0.021, 0.001, 0.001, 100, 0, 0, 255), (0.001, 0.001, 0.001, 10, 20, 30, 200),
                 (0.002, 0.003, 0.004, 20, 40, 61, 100), (float("nan"), 0, 0, 9, 9, 9)])
    out = orc.voxel_grid(cloud, 0.01)
    assert len(out) == 2
    f32 = np.float32
    assert out["x"][0] == (f32(0.001) + f32(0.002)) / f32(2) and out["z"][0] == (f32(0.001) + f32(0.004)) / f32(2)
    assert out["rgba"][0] == (150 << 24) | (15 << 16) | (30 << 8) | 45
    assert out["x"][1] == f32(0.021) and out["rgba"][1] == (255 << 24) | (100 << 16)
    # leaf far too small for the extent: PCL refuses (index overflow)
    wide = pts([(0, 0, 0, 1, 1, 1), (1000, 1000, 1000, 1, 1, 1)])
    assert orc.voxel_grid(wide, 0.0001) is None


def test_filters_on_a_sensor_frame(orc):
    """shape checks on the synthetic qhd frame: PassThrough drops NaN and z > 10, the grids reduce it"""
    c = scene.make_depth_frame(320, 180)
    idx = orc.pass_through(c)
    z = c["z"]
    want = np.flatnonzero(np.isfinite(c["x"]) & np.isfinite(c["y"]) & np.isfinite(z) & (z >= 0) & (z <= 10))
    np.testing.assert_array_equal(idx, want)
    assert 0 < len(idx) < len(c)
    a = orc.approx_voxel_grid(c[idx])
    v = orc.voxel_grid(c[idx])
    assert len(v) <= len(a) < len(idx)  # the approximate grid emits a voxel once per visit
    # every exact-grid output lies in a distinct voxel
    cells = np.floor(np.stack([v["x"], v["y"], v["z"]], 1) * np.float32(100)).astype(np.int64)
    assert len(np.unique(cells, axis=0)) == len(v)

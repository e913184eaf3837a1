"""Known answers for the oracle's restatement of the host-side steps around the trackers (oracle/pft_oracle_app.c,
SURVEY.md 8f row 3): removeZeroPoints (/root/reference/src/auto_tracking.cpp:577-595), pcl::compute3DCentroid, the
re-centring of :663-668 and the result consumer of :309-316 / :432-433.  Hand-derived, CPU only."""
import numpy as np

from pcl_tracking_amd import scene


def pts(xyz):
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    return scene.make_points(xyz, np.tile(np.array([[10, 20, 30]]), (len(xyz), 1)))


def test_remove_zero_points(orc):
    c = pts([[0.001, 0.002, 0.003],      # inside the 1 cm cube around the sensor origin: dropped
             [0.0099, -0.0099, 0.0],     # still inside (all three |.| < 0.01)
             [0.0101, 0.0, 0.0],         # one axis outside: kept
             [np.nan, 1.0, 1.0], [1.0, np.nan, 1.0], [1.0, 1.0, np.nan],  # NaN on any axis: dropped
             [np.inf, 1.0, 1.0],         # infinity is not NaN: kept, as the reference's isnan test keeps it
             [0.5, -0.25, 2.0]])
    out = orc.remove_zero_points(c)
    assert len(out) == 3
    np.testing.assert_array_equal(out["x"], np.array([0.0101, np.inf, 0.5], np.float32))
    assert (out["rgba"] == c["rgba"][0]).all()  # whole points are copied
    assert len(orc.remove_zero_points(pts(np.zeros((0, 3))))) == 0


def test_compute_3d_centroid_is_a_sequential_float_sum(orc):
    # 1e8 + 1 - 1e8 in float: the 1 is absorbed when it is added to 1e8 first
    c, n = orc.compute_3d_centroid(pts([[1e8, 0, 0], [1.0, 3.0, 0], [-1e8, 3.0, 6.0]]))
    assert n == 3 and c[3] == 1.0
    assert c[0] == 0.0 and c[1] == np.float32(2.0) and c[2] == np.float32(2.0)
    c, n = orc.compute_3d_centroid(pts([[1.0, 3.0, 0], [1e8, 0, 0], [-1e8, 3.0, 6.0]]))
    assert c[0] == 0.0  # (1 + 1e8) rounds to 1e8 as well
    c, n = orc.compute_3d_centroid(pts([[1e8, 0, 0], [-1e8, 3.0, 6.0], [1.0, 3.0, 0]]))
    assert c[0] == np.float32(1.0) / np.float32(3.0)  # order matters: the restatement adds in index order
    # not dense: non-finite points are skipped and not counted
    c, n = orc.compute_3d_centroid(pts([[1, 2, 3], [np.nan, 0, 0], [3, 4, 5]]), is_dense=False)
    assert n == 2 and list(c[:3]) == [2.0, 3.0, 4.0]
    c0 = np.array([7, 7, 7, 7], np.float32)
    assert orc.compute_3d_centroid(pts(np.zeros((0, 3))))[1] == 0


def test_recentre_model(orc):
    c = pts([[1.0, 2.0, 3.0], [3.0, 2.0, 1.0]])
    cen, n = orc.compute_3d_centroid(c)
    out, trans = orc.recentre_model(c, cen)
    np.testing.assert_array_equal(trans, np.array([[1, 0, 0, 2], [0, 1, 0, 2], [0, 0, 1, 2], [0, 0, 0, 1]], np.float32))
    np.testing.assert_array_equal(out["x"], [-1.0, 1.0])
    np.testing.assert_array_equal(out["y"], [0.0, 0.0])
    np.testing.assert_array_equal(out["z"], [1.0, -1.0])
    assert (out["rgba"] == c["rgba"]).all() and (out["w"] == 1.0).all()


def test_object_position(orc):
    ref = pts([[0.1, 0.0, 0.0], [-0.1, 0.0, 0.0], [0.0, 0.2, 0.0], [0.0, -0.2, 0.0]])
    r = np.zeros(1, scene.PARTICLE_DTYPE)
    r["x"], r["y"], r["z"], r["yaw"] = 0.5, -0.25, 1.0, np.float32(np.pi / 2)
    moved, c = orc.object_position(ref, r)
    # yaw = pi/2: (x, y) -> (-y, x); then the translation, z 5 mm towards the camera (auto_tracking.cpp:313)
    np.testing.assert_allclose(moved["x"], [0.5, 0.5, 0.3, 0.7], atol=1e-6)
    np.testing.assert_allclose(moved["y"], [-0.15, -0.35, -0.25, -0.25], atol=1e-6)
    np.testing.assert_array_equal(moved["z"], np.full(4, np.float32(1.0) + np.float32(-0.005)))
    np.testing.assert_allclose(c[:3], [0.5, -0.25, 0.995], atol=1e-6)
    assert c[3] == 1.0
    # identity pose: the centroid of a centred model is the z shift alone
    r2 = np.zeros(1, scene.PARTICLE_DTYPE)
    _, c2 = orc.object_position(ref, r2)
    assert c2[0] == 0.0 and c2[1] == 0.0 and c2[2] == np.float32(-0.005)

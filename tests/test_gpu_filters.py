"""GPU parity tests of the input filters (include/pft_filters.h) against the CPU oracle, through the C ABI:
PassThrough, ApproximateVoxelGrid (history-table flush order included) and VoxelGrid outputs are compared
BIT FOR BIT (x, y, z, data[3], rgba, padding; point order; counts) -- the device pipeline sums every voxel
in arrival order like the sequential algorithm, so there is no tolerance anywhere in this file.
PARITY UNPINNED: the oracle restates PCL 1.8.0, which is not available here (oracle/pft_oracle.h)."""
import numpy as np
import pytest

from pcl_tracking_amd import scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    from pcl_tracking_amd import filters

    return filters


@pytest.fixture(scope="module")
def frame():
    return scene.make_depth_frame(960, 540)  # Kinect2 qhd, auto_tracking.cpp:775


def same_cloud(got, want):
    assert len(got) == len(want)
    assert got.tobytes() == want.tobytes()


def random_cloud(n, seed, span=0.3, nan_frac=0.0):
    rng = np.random.default_rng(seed)
    c = np.zeros(n, scene.POINT_DTYPE)
    for k in ("x", "y", "z"):
        c[k] = rng.uniform(-span, span, n).astype(np.float32)
    c["z"] += np.float32(1.0)
    c["w"] = 1.0
    c["rgba"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    if nan_frac:
        bad = rng.random(n) < nan_frac
        c["x"][bad] = np.nan
        c["y"][bad] = np.nan
        c["z"][bad] = np.nan
    return c


@pytest.mark.parametrize("hist", [512, 2048, 64])
def test_reference_front_end_on_a_qhd_frame(F, orc, frame, hist):
    """filterPassThrough + gridSampleApprox fused (auto_tracking.cpp:637, 683)"""
    f = F.make_reference_input_filter()
    f.setHistorySize(hist)
    f.setInputCloud(frame)
    got = f.filter()
    idx = orc.pass_through(frame, "z", 0.0, 10.0)
    want = orc.approx_voxel_grid(frame[idx], 0.01, hist)
    assert f.counts() == (len(idx), len(want))
    np.testing.assert_array_equal(f.passIndices(), idx)
    same_cloud(got, want)
    assert 0 < f.lastMilliseconds() < 100


def test_pass_through_alone(F, orc, frame):
    p = F.PassThrough()
    p.setFilterFieldName("z")
    p.setFilterLimits(0, 10)
    p.setKeepOrganized(False)
    p.setInputCloud(frame)
    same_cloud(p.filter(), frame[orc.pass_through(frame, "z", 0.0, 10.0)])
    p.setFilterLimitsNegative(True)
    p.setInputCloud(frame)
    same_cloud(p.filter(), frame[orc.pass_through(frame, "z", 0.0, 10.0, negative=True)])
    p.setFilterLimitsNegative(False)
    p.setFilterFieldName("x")
    p.setFilterLimits(-0.5, 0.25)
    p.setInputCloud(frame)
    same_cloud(p.filter(), frame[orc.pass_through(frame, "x", -0.5, 0.25)])


@pytest.mark.parametrize("n", [1, 2, 63, 1023, 1024, 1025, 5000, 70001])
def test_approx_voxel_grid_ragged_sizes(F, orc, n):
    c = random_cloud(n, n)
    g = F.ApproximateVoxelGrid()
    g.setLeafSize(0.01, 0.01, 0.01)
    g.setInputCloud(c)
    same_cloud(g.filter(), orc.approx_voxel_grid(c, 0.01, 512))
    g.setLeafSize(0.05, 0.02, 0.1)
    g.setInputCloud(c)
    same_cloud(g.filter(), orc.approx_voxel_grid(c, (0.05, 0.02, 0.1), 512))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_approx_voxel_grid_long_runs_across_tiles(F, orc, seed):
    """runs of very different lengths (1 ... 2500 consecutive points of one voxel) that start and end at arbitrary
    offsets of the 1024-point device tiles, over few voxels so that table entries collide and flush each other"""
    rng = np.random.default_rng(seed)
    lens = rng.choice([1, 2, 3, 7, 60, 255, 256, 257, 700, 1024, 2500], 120)
    vox = rng.integers(0, 40, (len(lens), 3))  # 40^3 voxels over 512 table entries: collisions are common
    cell = np.repeat(vox, lens, axis=0)
    n = len(cell)
    c = np.zeros(n, scene.POINT_DTYPE)
    jit = rng.uniform(0.05, 0.95, (n, 3))
    for a, k in enumerate(("x", "y", "z")):
        c[k] = ((cell[:, a] + jit[:, a]) * 0.01).astype(np.float32)
    c["w"] = 1.0
    c["rgba"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    for hist in (512, 64, 1024):
        g = F.ApproximateVoxelGrid()
        g.setLeafSize(0.01)
        g.setHistorySize(hist)
        g.setInputCloud(c)
        same_cloud(g.filter(), orc.approx_voxel_grid(c, 0.01, hist))


def test_approx_voxel_grid_edge_cases(F, orc):
    g = F.ApproximateVoxelGrid()
    g.setLeafSize(0.01)
    # empty cloud
    g.setInputCloud(np.zeros(0, scene.POINT_DTYPE))
    assert len(g.filter()) == 0 and g.counts() == (0, 0)
    # every point in one voxel: one run of 20 000 points summed in order
    c = random_cloud(20000, 5, span=0.004)
    c["x"] = np.abs(c["x"]); c["y"] = np.abs(c["y"]); c["z"] = np.float32(1.001) + np.abs(c["z"] - 1) * np.float32(0.5)
    g.setInputCloud(c)
    want = orc.approx_voxel_grid(c, 0.01, 512)
    assert len(want) == 1
    same_cloud(g.filter(), want)
    # two voxels that share a table entry, alternating: every point flushes the other voxel
    c = random_cloud(4000, 6, span=0.001)
    c["x"] = np.where(np.arange(4000) % 2 == 0, np.float32(0.005), np.float32(5.125))
    c["y"] = np.float32(0.005); c["z"] = np.float32(0.005)
    g.setInputCloud(c)
    want = orc.approx_voxel_grid(c, 0.01, 512)
    assert len(want) == 4000
    same_cloud(g.filter(), want)
    # NaN points without a PassThrough in front: PCL feeds them through (cell = INT_MIN), centroids turn NaN
    c = random_cloud(3000, 7, nan_frac=0.1)
    g.setInputCloud(c)
    got, want = g.filter(), orc.approx_voxel_grid(c, 0.01, 512)
    assert len(got) == len(want)
    for k in ("x", "y", "z"):
        np.testing.assert_array_equal(np.isnan(got[k]), np.isnan(want[k]))
        np.testing.assert_array_equal(got[k][~np.isnan(want[k])], want[k][~np.isnan(want[k])])
    np.testing.assert_array_equal(got["rgba"], want["rgba"])
    # all points invalid behind a PassThrough: empty output
    f = F.make_reference_input_filter()
    c = random_cloud(2000, 8, nan_frac=1.0)
    f.setInputCloud(c)
    assert len(f.filter()) == 0 and f.counts() == (0, 0)


@pytest.mark.parametrize("n", [1, 3, 1024, 4097, 60000])
def test_voxel_grid(F, orc, n):
    """gridSample (auto_tracking.cpp:549-561): model preparation and the waiting frames"""
    c = random_cloud(n, 100 + n, nan_frac=0.05 if n > 3 else 0.0)
    g = F.VoxelGrid()
    g.setLeafSize(0.01, 0.01, 0.01)
    g.setInputCloud(c)
    same_cloud(g.filter(), orc.voxel_grid(c, 0.01))
    g.setLeafSize(0.03, 0.01, 0.02)
    g.setInputCloud(c)
    same_cloud(g.filter(), orc.voxel_grid(c, (0.03, 0.01, 0.02)))


def test_voxel_grid_on_the_frame_and_leaf_too_small(F, orc, frame):
    g = F.VoxelGrid()
    g.setLeafSize(0.01)
    idx = orc.pass_through(frame)
    g.setInputCloud(frame[idx])
    same_cloud(g.filter(), orc.voxel_grid(frame[idx], 0.01))
    # PCL: "Leaf size is too small for the input dataset" -> the input cloud is handed through unchanged
    wide = random_cloud(500, 3, span=40.0)  # 8e5 cells per axis: the int64 product stays defined
    assert orc.voxel_grid(wide, 0.0001) is None
    g.setLeafSize(0.0001)
    g.setInputCloud(wide)
    same_cloud(g.filter(), wide)


def test_filter_output_feeds_the_tracker_in_hbm(F, orc, frame):
    """front end -> tracker without touching the host: same poses as the tracker fed the oracle-filtered cloud"""
    import torch

    from pcl_tracking_amd import tracker

    f = F.make_reference_input_filter()
    dev_frame = torch.from_numpy(frame.view(np.uint8).reshape(-1).copy()).cuda()
    f.setInputCloudDevice(dev_frame.data_ptr(), len(frame), keepalive=dev_frame)
    ptr, n = f.filterDevice()
    want_cloud = orc.approx_voxel_grid(frame[orc.pass_through(frame)], 0.01, 512)
    assert n == len(want_cloud)
    model = scene.make_model(1024)
    res = []
    for mode in ("device", "host"):
        t = tracker.make_reference_tracker(particle_num=400, seed=5)
        t.setReferenceCloud(model)
        t.setTrans(scene.initial_trans())
        if mode == "device":
            t.setInputCloudDevice(ptr, n, keepalive=f)
        else:
            t.setInputCloud(want_cloud)
        t.compute()
        t.compute()
        res.append(t.getResult().tobytes())
    assert res[0] == res[1]

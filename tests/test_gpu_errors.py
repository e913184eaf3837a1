"""Device-side failures must reach the caller (run with -m gpu on an MI355X).

PftHeader::error carries bit0 (octree node capacity), bit1 (depth / bounding-box growth steps) and bit2 (the one-pass
crop's bounded wait).  The likelihood launch of such an iteration runs without a target (all weights zero), which is
the worst failure mode for a tracker if nobody is told: every host synchronisation point of include/pft.h
(pft_get_result, pft_get_particles, pft_get_fit_ratio, pft_synchronize, pft_eval_weights) returns PFT_ERR_CAPACITY
/ PFT_ERR_HIP with pft_last_error_string naming the flag.  The flags are per iteration: the next frame is clean.
The reference's caller looks for a failure signal around compute() (auto_tracking.cpp:692-696).

Also here: the sorted builder's radix-pass guess (taken from the previous iteration's depth, read without
synchronising) being too small is NOT an error -- the rescue launch behind it rebuilds the tree.
"""
import numpy as np
import pytest

from pcl_tracking_amd import scene

pytestmark = pytest.mark.gpu

KEYS = ("x", "y", "z", "roll", "pitch", "yaw")


@pytest.fixture(scope="module")
def gpu():
    from pcl_tracking_amd import tracker

    return tracker


@pytest.fixture(scope="module")
def data():
    return dict(model=scene.make_model(1024), scene=scene.make_scene(50000))


def fresh(gpu, data, P=256, seed=2):
    t = gpu.make_reference_tracker(particle_num=P, seed=seed)
    t.setReferenceCloud(data["model"])
    t.setTrans(scene.initial_trans())
    t.setInputCloud(data["scene"])
    return t


@pytest.mark.parametrize("builder", ["single", "sorted"])
def test_octree_capacity_overflow_is_reported_and_not_sticky(gpu, data, builder, monkeypatch):
    from pcl_tracking_amd._lib import PftError

    monkeypatch.setenv("PFT_FORCE_BUILDER", builder)
    t = fresh(gpu, data)
    t.compute()
    good = t.getResult()
    assert (t.debugHostStat()[2:] == 0).all()
    t.debugSetLimits(max_words=300)  # a few thousand cropped points need a few thousand node words
    t.compute()
    with pytest.raises(PftError) as e:
        t.getResult()
    assert e.value.status == 6 and "bit0" in str(e.value), str(e.value)
    hs = t.debugHostStat()
    assert hs[2] & 1 and hs[3] == 0  # reported once, then cleared
    # the failed iterations ran without a target: weights uniform, the pose is the unweighted mean (finite)
    p = t.getParticles()
    assert np.isfinite(p["x"]).all() and np.allclose(p["weight"], 1.0 / len(p))
    # capacity back: the very next frame is clean
    t.debugSetLimits(max_words=len(data["scene"]) * 8 + 64)
    t.compute()
    r = t.getResult()
    assert all(np.isfinite(float(r[k])) for k in KEYS)
    assert abs(float(r["x"]) - float(good["x"])) < 0.05
    p = t.getParticles()
    assert p["weight"].max() > 2.0 / len(p)  # the likelihood is back
    t.synchronize()


def test_too_deep_tree_is_reported(gpu):
    """two clusters 4e7 m apart at 1 cm resolution need depth 32 > PFT_MAX_DEPTH: bit1, PFT_ERR_CAPACITY"""
    from pcl_tracking_amd._lib import PftError

    rng = np.random.default_rng(5)
    xyz = rng.uniform(-0.2, 0.2, (64, 3)).astype(np.float32)
    xyz[32:, 0] += 4.0e7
    pts = scene.make_points(xyz, rng.integers(0, 255, (64, 3)))
    t = gpu.make_reference_tracker(particle_num=16, seed=1)
    t.setReferenceCloud(pts)
    t.setTrans(np.eye(4, dtype=np.float32))
    t.setInputCloud(pts)
    # identity poses: the crop box is the model's own AABB and keeps both clusters (a rotation of 1e-3 rad would swing
    # the far cluster by kilometres and leave it outside)
    p = np.zeros(16, scene.PARTICLE_DTYPE)
    p["w"], p["weight"] = 1.0, 1.0 / 16
    t.setParticles(p)
    t.compute()
    with pytest.raises(PftError) as e:
        t.synchronize()
    assert e.value.status == 6 and "bit1" in str(e.value), str(e.value)
    # an ordinary cloud afterwards: clean
    t.setInputCloud(pts[:32])
    t.compute()
    t.synchronize()


@pytest.mark.parametrize("call", ["getResult", "getParticles", "getFitRatio", "synchronize"])
def test_crop_timeout_flag_is_reported_by_every_sync_point(gpu, data, call):
    """bit2 cannot be provoked on a healthy GPU (a workgroup would have to wait 2^24 polls for a predecessor): the flag is
    injected right behind the crop launch, which is all the rest of the path sees of such a failure"""
    from pcl_tracking_amd._lib import PftError

    t = fresh(gpu, data)
    t.compute()
    t.synchronize()
    t.debugInjectError(4)
    t.compute()
    with pytest.raises(PftError) as e:
        getattr(t, call)()
    assert e.value.status == 5 and "bit2" in str(e.value), str(e.value)
    getattr(t, call)()  # reported once
    t.compute()  # per iteration, not sticky: the next frame builds its tree again
    t.synchronize()
    p = t.getParticles()
    assert p["weight"].max() > 2.0 / len(p)


def test_population_barrier_timeout_writes_nothing_and_is_named(gpu, data):
    """bit4: a device-scope barrier of the population kernel that times out must not leave weights, mean and
    alias table computed from values that never arrived.  The flag cannot be provoked on an idle GPU (the workgroups are
    co-resident at once), so it is injected behind the crop launch: every workgroup of the population launch then sees it
    at its first barrier, as it would after a real time-out elsewhere in the grid."""
    from pcl_tracking_amd._lib import PftError

    t = fresh(gpu, data, P=2048)  # 8 workgroups: the barriers are real
    t.compute()
    t.synchronize()
    before = t.getResult()
    w_before = t.getParticles()["weight"].copy()
    assert w_before.max() > 2.0 / 2048
    t.debugInjectError(16)
    t.compute()
    with pytest.raises(PftError) as e:
        t.getResult()
    msg = str(e.value)
    assert e.value.status == 5 and "bit4" in msg and "NOT written" in msg, msg
    assert "without a target cloud" not in msg  # that sentence belongs to bits 0-2
    # the first iteration of that frame wrote nothing; its second iteration was clean, so the handle is consistent again
    after = t.getResult()
    for k in KEYS:
        assert np.isfinite(after[k])
    t.compute()
    t.synchronize()
    p = t.getParticles()
    assert abs(float(p["weight"].sum()) - 1.0) < 1e-3 and p["weight"].max() > 2.0 / len(p)
    assert abs(t.getResult()["x"] - before["x"]) < 0.05


def test_eval_weights_reports_the_flag(gpu, data):
    from pcl_tracking_amd._lib import PftError

    t = fresh(gpu, data)
    p = np.zeros(8, scene.PARTICLE_DTYPE)
    gt = scene.model_gt_pose()
    for k, name in enumerate(KEYS):
        p[name] = gt[k]
    t.debugInjectError(4)  # (bits 0 / 1 belong to the builder, which starts from the crop's bit 2 alone)
    with pytest.raises(PftError) as e:
        t.evalWeights(p)
    assert e.value.status == 5
    G = t.evalWeights(p)
    assert (G["raw"] < 0).all()


@pytest.mark.parametrize("npass", [1, 2, 3])
def test_sorted_builder_with_too_few_radix_passes_is_rescued(gpu, data, npass, monkeypatch):
    """the host sizes the radix passes from the previous depth (stale, unsynchronised): when the tree turns out deeper
    the rescue launch rebuilds it -- same tree as the single-workgroup builder, no error, bit for bit"""
    gt = scene.model_gt_pose()
    rng = np.random.default_rng(7)
    p = np.zeros(96, scene.PARTICLE_DTYPE)
    for k, name in enumerate(KEYS):
        p[name] = gt[k] + rng.normal(0, 0.02 if k < 3 else 0.1, len(p))
    monkeypatch.setenv("PFT_FORCE_BUILDER", "single")
    ref = fresh(gpu, data).evalWeights(p, want_nn=True)
    assert ref["octree_depth"] >= 7  # 21+ bits of Morton code: more than 1-2 passes of 8 bits
    monkeypatch.setenv("PFT_FORCE_BUILDER", "sorted")
    t = fresh(gpu, data)
    t.debugSetLimits(sorted_npass=npass)
    G = t.evalWeights(p, want_nn=True)
    assert (t.debugHostStat()[2:] == 0).all()
    assert G["octree_depth"] == ref["octree_depth"] and G["n_words"] == ref["n_words"] and G["n_leaves"] == ref["n_leaves"]
    np.testing.assert_array_equal(G["point_keys"], ref["point_keys"])
    np.testing.assert_array_equal(G["nn_idx"], ref["nn_idx"])
    np.testing.assert_array_equal(G["nn_d2"], ref["nn_d2"])
    np.testing.assert_array_equal(G["raw"], ref["raw"])


def test_depth_jump_between_computes_with_the_sorted_builder(gpu, data, monkeypatch):
    """the depth grows by >= 3 levels between two evaluations on one handle (tight particle set, then one
    spread over metres) while the sorted builder still holds the pass count of the shallow tree"""
    gt = scene.model_gt_pose()
    rng = np.random.default_rng(9)

    def particles(sig_t, sig_r, n=64):
        p = np.zeros(n, scene.PARTICLE_DTYPE)
        for k, name in enumerate(KEYS):
            p[name] = gt[k] + rng.normal(0, sig_t if k < 3 else sig_r, n)
        return p

    tiny = scene.make_model(64, seed=3)
    cloud = data["scene"]
    # first evaluation: only the two dozen input points nearest to the object centre exist (a crop a few centimetres
    # wide, depth <= 4); second: the whole scene under particles spread over metres (depth 9)
    dist2 = (cloud["x"] - gt[0]) ** 2 + (cloud["y"] - gt[1]) ** 2 + (cloud["z"] - gt[2]) ** 2
    near = cloud[np.sort(np.argsort(dist2)[:24])]
    tight, wide = particles(0.001, 0.01), particles(1.5, 1.0)
    out = {}
    for builder in ("single", "sorted"):
        monkeypatch.setenv("PFT_FORCE_BUILDER", builder)
        t = gpu.make_reference_tracker(particle_num=64, seed=1)
        t.setReferenceCloud(tiny)
        t.setTrans(scene.initial_trans())
        t.setInputCloud(near)
        a = t.evalWeights(tight, want_nn=True)
        t.setInputCloud(cloud)
        b = t.evalWeights(wide, want_nn=True)
        assert (t.debugHostStat()[2:] == 0).all()
        out[builder] = (a, b)
    (a1, b1), (a2, b2) = out["single"], out["sorted"]
    assert b1["octree_depth"] >= a1["octree_depth"] + 3, (a1["octree_depth"], b1["octree_depth"])
    for x, y in ((a1, a2), (b1, b2)):
        assert x["octree_depth"] == y["octree_depth"] and x["n_words"] == y["n_words"]
        np.testing.assert_array_equal(x["nn_idx"], y["nn_idx"])
        np.testing.assert_array_equal(x["raw"], y["raw"])

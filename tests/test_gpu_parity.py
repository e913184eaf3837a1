"""GPU parity tests (run with -m gpu on an MI355X): every stage of the HIP path, called through the
C ABI (include/pft.h), against the CPU oracle on the same seeded inputs.

Bars: integer / index results (crop set, octree keys and depth, approximate-NN index) bit-exact; float
results that involve no transcendental (AABB, NN squared distance) bit-exact; sums and transcendentals
within the tolerance written at each assert; weighted-mean pose within 1e-4 (BASELINE.json north_star).
PARITY UNPINNED: the oracle restates PCL 1.8.0, which is not available here (oracle/pft_oracle.h).
"""
import numpy as np
import pytest

from pcl_tracking_amd import scene

pytestmark = pytest.mark.gpu


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def particles_around(pose, n, seed, sig_t=0.015, sig_r=0.09):
    rng = np.random.default_rng(seed)
    p = np.zeros(n, scene.PARTICLE_DTYPE)
    for k, name in enumerate(("x", "y", "z")):
        p[name] = pose[k] + rng.normal(0, sig_t, n)
    for k, name in enumerate(("roll", "pitch", "yaw")):
        p[name] = pose[3 + k] + rng.normal(0, sig_r, n)
    p["w"] = 1.0
    p["weight"] = 1.0 / n
    return p


@pytest.fixture(scope="module")
def gpu():
    from pcl_tracking_amd import tracker

    return tracker


@pytest.fixture(scope="module")
def data():
    return dict(model=scene.make_model(2048), scene=scene.make_scene(50000), gt=scene.model_gt_pose())


def make_pair(gpu, orc, model, cloud, P, seed=1, **cfg):
    g = gpu.make_reference_tracker(particle_num=P, seed=seed)
    o = orc.Tracker(orc.default_config(particle_num=P, seed=seed, threads=0, emulate_pcl_alloc=0, **cfg))
    for t, ref, tr, inp in ((g, g.setReferenceCloud, g.setTrans, g.setInputCloud),
                            (o, o.set_reference, o.set_trans, o.set_input)):
        ref(model)
        tr(scene.initial_trans())
        inp(cloud)
    return g, o


# ---- A1 -------------------------------------------------------------------------------------------
def test_pose_to_matrix(gpu, orc, data):
    g = gpu.make_reference_tracker(particle_num=64)
    p = particles_around(data["gt"], 4096, 3, 0.5, 2.0)
    got = g.debugPoseToMatrix(p)
    want = np.stack([orc.get_transformation(*[q[k] for k in ("x", "y", "z", "roll", "pitch", "yaw")])[:3] for q in p])
    # sin/cos: double->float on the GPU vs glibc cosf/sinf on the CPU; products of two such factors
    # (entries like A*DF - B*E cancel, so the bound is absolute: a few float ulps of 1.0)
    np.testing.assert_allclose(got, want, atol=3e-7, rtol=0)
    assert (ulp_diff(got, want) == 0).mean() > 0.9


# ---- A0 / A11 RNG -----------------------------------------------------------------------------------
def test_init_particles(gpu, orc):
    g = gpu.make_reference_tracker(particle_num=1000, seed=77)
    cfg = orc.default_config(particle_num=1000, seed=77)
    rep = np.zeros(1, scene.PARTICLE_DTYPE)
    rep["x"], rep["y"], rep["z"], rep["yaw"], rep["w"], rep["weight"] = 0.3, -0.2, 1.1, 0.7, 1.0, 1e-3
    want = orc.init_particles(cfg, rep, 0, 1000)
    got = g.debugInitParticles(rep, 0, 1000)
    for k in ("x", "y", "z", "roll", "pitch", "yaw"):
        # double log/sin/cos differ by <= 1 ulp(double) between glibc and ocml: invisible after the float cast
        assert ulp_diff(got[k], want[k]).max() <= 1
        assert (got[k] == want[k]).mean() > 0.999
    np.testing.assert_array_equal(got["weight"], want["weight"])
    np.testing.assert_array_equal(g.debugInitParticles(rep, 600, 400)["x"], got["x"][600:])


def test_resample(gpu, orc, data):
    P = 2048
    g = gpu.make_reference_tracker(particle_num=P, seed=5)
    cfg = orc.default_config(particle_num=P, seed=5)
    old = particles_around(data["gt"], P, 11)
    rng = np.random.default_rng(0)
    w = rng.random(P).astype(np.float32) ** 4
    w[rng.random(P) < 0.1] = 0
    w /= w.sum()
    old["weight"] = w
    a, q = orc.gen_alias_table(w)
    rep = old[:1].copy()
    rep["x"] += 0.5
    for epoch in (0, 3):
        want = orc.resample(cfg, old, a, q, rep, epoch)
        got = g.debugResample(old, a, q, rep, epoch)
        assert got[0].tobytes() == rep[0].tobytes()
        for k in ("x", "y", "z", "roll", "pitch", "yaw"):
            assert ulp_diff(got[k], want[k]).max() <= 1
            assert (got[k] == want[k]).mean() > 0.999
        np.testing.assert_array_equal(got["weight"], want["weight"])
    np.testing.assert_array_equal(g.debugResample(old, a, q, rep, 3, 512, 256)["yaw"], got["yaw"][512:768])


# ---- A2-A7 ------------------------------------------------------------------------------------------
def check_eval(gpu, orc, model, cloud, P, pose, seed, expect_empty=False):
    g, o = make_pair(gpu, orc, model, cloud, P)
    p = particles_around(pose, P, seed)
    mats = g.debugPoseToMatrix(p)
    G = g.evalWeights(p, want_nn=True)
    O = o.eval_weights(p, want_nn=True, mats=mats)  # same matrices: everything downstream must be bit-exact
    # A3: AABB (float min/max, no rounding freedom)
    np.testing.assert_array_equal(G["bbox"], O["bbox"].astype(np.float32))
    # A4: crop set and order
    np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
    if expect_empty is None:
        expect_empty = len(O["crop_idx"]) == 0
    if expect_empty:
        assert len(G["crop_idx"]) == 0
        assert (G["raw"] == 0).all() and (O["raw"] == 0).all() and (G["nn_idx"] == -1).all()
        return G, O
    assert len(G["crop_idx"]) > 0
    # A5: box replay (doubles) and keys
    assert G["octree_depth"] == O["octree_depth"]
    np.testing.assert_array_equal(G["octree_min"], O["octree_min"])
    np.testing.assert_array_equal(G["octree_max"], O["octree_max"])
    ot = orc.Octree(np.ascontiguousarray(cloud)[O["crop_idx"]])
    np.testing.assert_array_equal(G["point_keys"], ot.point_keys())
    assert G["n_leaves"] == ot.info()["leaves"]
    # A6: approximate nearest neighbour: index and squared distance, every pair
    np.testing.assert_array_equal(G["nn_idx"], O["nn_idx"])
    np.testing.assert_array_equal(G["nn_d2"], O["nn_d2"])
    assert G["scan_queries"] == O["scan_queries"] and G["scan_points"] == O["scan_points"]
    # A7: per-particle sum of ~M doubles, reduced in a different order, then cast to float
    d = ulp_diff(G["raw"], O["raw"])
    assert d.max() <= 1, d.max()
    assert (d == 0).mean() > 0.99
    return G, O


def test_eval_weights_scene(gpu, orc, data):
    G, O = check_eval(gpu, orc, data["model"], data["scene"], 256, data["gt"], 21)
    assert G["octree_depth"] >= 6 and len(G["crop_idx"]) > 1000
    assert (O["raw"] < -100).all()


def test_eval_weights_real_trig(gpu, orc, data):
    """same chain with each side computing its own matrices (cosf/sinf vs double->float): 1-ulp
    differences in a matrix entry may flip a few neighbours; weights stay within 1e-3 absolute"""
    g, o = make_pair(gpu, orc, data["model"], data["scene"], 256)
    p = particles_around(data["gt"], 256, 5)
    G = g.evalWeights(p, want_nn=True)
    O = o.eval_weights(p, want_nn=True)
    np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
    assert (G["nn_idx"] == O["nn_idx"]).mean() > 0.9999
    np.testing.assert_allclose(G["raw"], O["raw"], atol=1e-3, rtol=0)


@pytest.mark.parametrize("M,N,P", [(1, 1, 1), (63, 500, 7), (65, 1025, 33), (513, 3000, 129), (2048, 307200, 64)])
def test_eval_weights_ragged_sizes(gpu, orc, M, N, P):
    model = scene.make_model(M, seed=M)
    if N == 307200:
        cloud = scene.make_scene(N, mode="organized")  # BASELINE config 3: no downsample, many points per leaf
    elif N == 1:
        cloud = np.zeros(1, scene.POINT_DTYPE)
        cloud["x"], cloud["y"], cloud["z"], cloud["w"] = scene.model_gt_pose()[:3] + (1.0,)
    else:
        cloud = scene.make_scene(50000)[:N]
    # (a one-point model has a degenerate AABB: the crop is empty unless the input point sits exactly on it)
    G, O = check_eval(gpu, orc, model, cloud, P, scene.model_gt_pose(), 100 + M, expect_empty=None if N == 1 else False)
    if N == 307200:
        assert G["scan_points"] > 2 * G["scan_queries"]


@pytest.mark.parametrize("M,N,P", [(65, 1025, 33), (2048, 50000, 128), (2048, 307200, 64)])
def test_eval_weights_sorted_builder(gpu, orc, monkeypatch, M, N, P):
    """the many-workgroup builder (Morton keys -> stable radix sort -> level construction), forced on at sizes
    where the single-workgroup builder would normally run: same tree, same neighbours, bit for bit"""
    monkeypatch.setenv("PFT_FORCE_BUILDER", "sorted")
    model = scene.make_model(M, seed=M)
    cloud = scene.make_scene(N, mode="organized") if N == 307200 else scene.make_scene(50000)[:N]
    G, O = check_eval(gpu, orc, model, cloud, P, scene.model_gt_pose(), 300 + M)
    monkeypatch.setenv("PFT_FORCE_BUILDER", "single")
    G2, _ = check_eval(gpu, orc, model, cloud, P, scene.model_gt_pose(), 300 + M)
    np.testing.assert_array_equal(G["nn_idx"], G2["nn_idx"])
    assert G["n_words"] == G2["n_words"] and G["n_leaves"] == G2["n_leaves"]


def test_compute_with_sorted_builder(gpu, orc, data, monkeypatch):
    monkeypatch.setenv("PFT_FORCE_BUILDER", "sorted")
    g, o = make_pair(gpu, orc, data["model"], data["scene"], 400, seed=3)
    for f in range(3):
        g.compute()
        o.compute()
        rg, ro = g.getResult(), o.get_result()
        for k in ("x", "y", "z", "roll", "pitch", "yaw"):
            assert abs(float(rg[k]) - float(ro[k])) < 1e-4


def test_eval_weights_empty_crop(gpu, orc, data):
    far = (5.0, 5.0, 5.0, 0, 0, 0)
    check_eval(gpu, orc, data["model"], data["scene"], 64, far, 9, expect_empty=True)


def test_eval_weights_nonfinite_input_points(gpu, orc, data):
    cloud = data["scene"].copy()
    cloud["x"][::97] = np.nan
    cloud["z"][5::101] = np.inf
    check_eval(gpu, orc, data["model"], cloud, 64, data["gt"], 2)


def test_eval_weights_wide_spread_deep_tree(gpu, orc, data):
    """particles spread over metres: big crop, deep octree, many box-growth steps"""
    g, o = make_pair(gpu, orc, data["model"], data["scene"], 128)
    p = particles_around(data["gt"], 128, 4, sig_t=0.6, sig_r=1.0)
    mats = g.debugPoseToMatrix(p)
    G = g.evalWeights(p, want_nn=True)
    O = o.eval_weights(p, want_nn=True, mats=mats)
    assert G["octree_depth"] == O["octree_depth"] >= 8
    np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
    np.testing.assert_array_equal(G["nn_idx"], O["nn_idx"])
    np.testing.assert_array_equal(G["nn_d2"], O["nn_d2"])
    assert ulp_diff(G["raw"], O["raw"]).max() <= 1


# ---- A8, A9, A10 --------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 400, 1000, 8192, 16384, 16385, 40000, 65536, 300000])
def test_normalize(gpu, orc, n):
    g = gpu.make_reference_tracker(particle_num=64)
    rng = np.random.default_rng(n)
    raw = (-rng.random(n) * 2000).astype(np.float32)
    raw[rng.random(n) < 0.05] = 0.0
    want, fw = orc.normalize_weights(raw)
    got, fg = g.debugNormalize(raw)
    assert fw == fg
    assert ulp_diff(got, want).max() <= 1  # exp() in double differs by <= 1 ulp(double) glibc vs ocml
    assert (got == want).mean() > 0.99
    for special in (np.zeros(n, np.float32), np.full(n, -3.0, np.float32)):
        np.testing.assert_array_equal(g.debugNormalize(special)[0], orc.normalize_weights(special)[0])


@pytest.mark.parametrize("n", [1, 2, 3, 400, 1000, 8192, 16384, 16385, 40000, 65536, 300000])
def test_alias_table(gpu, orc, n):
    g = gpu.make_reference_tracker(particle_num=64)
    rng = np.random.default_rng(n + 1)
    cases = []
    w = rng.random(n).astype(np.float32) ** 6
    w[rng.random(n) < 0.15] = 0
    cases.append(w / max(w.sum(), 1e-30))
    cases.append(np.full(n, np.float32(1) / np.float32(n), np.float32))  # uniform (after an all-equal frame)
    w = np.zeros(n, np.float32)
    w[n // 2] = 1.0
    cases.append(w)  # all mass on one particle
    w = np.full(n, np.float32(1) / np.float32(n), np.float32)
    w[: n // 2] *= np.float32(0.5)
    cases.append(w)  # exact ties in q
    for w in cases:
        a_w, q_w = orc.gen_alias_table(w)
        a_g, q_g = g.debugAlias(w)
        # same Walker table: identical aliases, probabilities equal up to the rounding of prefix sums vs
        # sequential updates
        np.testing.assert_array_equal(a_g, a_w)
        np.testing.assert_allclose(q_g, q_w, atol=1e-9 * max(1, n), rtol=0)


def test_weighted_mean(gpu, orc, data):
    g = gpu.make_reference_tracker(particle_num=64)
    for n in (1, 400, 8192, 40000, 65536):
        p = particles_around(data["gt"], n, n)
        w = np.random.default_rng(n).random(n).astype(np.float32)
        p["weight"] = w / w.sum()
        want = orc.weighted_mean(p)
        got = g.debugWeightedMean(p)
        for k in ("x", "y", "z", "roll", "pitch", "yaw"):
            # PCL accumulates sequentially in float, the kernel tree-reduces in double
            assert abs(float(got[k]) - float(want[k])) < 2e-6 * max(1.0, abs(float(want[k])))
        assert got["weight"] == want["weight"]


@pytest.mark.parametrize("n", [1, 2, 3, 159, 400, 500, 1000, 4097, 8192, 16385, 40000, 65536, 300000])
def test_population_sums_follow_the_specified_tree(gpu, orc, n):
    """The weight sum and the weighted mean are ADJACENT-PAIR TREES in double over the index range padded to a power of
    two (pft_population.hip): whatever the number of workgroups and particles per thread the launcher picks for n, the
    results equal the oracle's restatement of that order bit for bit -- and PCL's sequential order to
    the tolerances of test_normalize / test_weighted_mean above."""
    g = gpu.make_reference_tracker(particle_num=64)
    rng = np.random.default_rng(n)
    p = particles_around(scene.model_gt_pose(), n, n + 1)
    w = rng.random(n).astype(np.float32)
    p["weight"] = w / w.sum()
    got, want = g.debugWeightedMean(p), orc.weighted_mean_tree(p)
    assert got.tobytes() == want.tobytes(), (got, want)
    raw = (-rng.random(n) * 80 - 900).astype(np.float32)
    raw[rng.random(n) < 0.05] = 0.0
    gw, gf = g.debugNormalize(raw)
    ow, of = orc.normalize_weights_tree(raw)
    assert gf == of
    # exp() in double on both sides (ocml / glibc): the cast to float agrees except in rare last-bit cases
    d = ulp_diff(gw, ow)
    assert d.max() <= 1 and (d == 0).mean() > 0.999, (int(d.max()), float((d == 0).mean()))


# ---- A12: the whole tracker -------------------------------------------------------------------------------
@pytest.mark.parametrize("P", [400, 8192])
def test_compute_tracks_like_oracle(gpu, orc, data, P):
    """identical frames + RNG seed: weighted-mean pose within 1e-4 of the CPU tracker (north_star)"""
    g, o = make_pair(gpu, orc, data["model"], data["scene"], P, seed=3)
    frames = 4 if P == 400 else 2
    for f in range(frames):
        g.compute()
        assert o.compute() == 0
        rg, ro = g.getResult(), o.get_result()
        for k in ("x", "y", "z", "roll", "pitch", "yaw"):
            assert abs(float(rg[k]) - float(ro[k])) < 1e-4, (f, k, rg, ro)
        assert rg["weight"] == ro["weight"]
        assert abs(g.getFitRatio() - o.fit_ratio()) < 1e-2
    pg, po = g.getParticles(), o.get_particles()
    assert len(pg) == len(po) == P
    same = np.ones(P, bool)
    for k in ("x", "y", "z", "roll", "pitch", "yaw"):
        same &= np.abs(pg[k] - po[k]) < 1e-5
    assert same.mean() > 0.995  # a weight-rounding flip may redirect a handful of alias draws
    np.testing.assert_allclose(pg["weight"][same], po["weight"][same], rtol=1e-3, atol=1e-9)


def test_compute_moving_sequence(gpu, orc, data):
    g, o = make_pair(gpu, orc, data["model"], data["scene"], 400, seed=8)
    for f in range(3):
        cloud = scene.make_scene(20000, obj_pose=scene.advance_pose(scene.GT_POSE, 4 * f))
        g.setInputCloud(cloud)
        o.set_input(cloud)
        g.compute()
        o.compute()
        rg, ro = g.getResult(), o.get_result()
        for k in ("x", "y", "z", "roll", "pitch", "yaw"):
            assert abs(float(rg[k]) - float(ro[k])) < 1e-4


def test_frame_graph_mode_is_bit_identical(gpu, data, monkeypatch):
    """PFT_GRAPH=1 (opt-in): the launches of a steady-state frame are captured and replayed as one hipGraph, updated in
    place from frame to frame.  Same kernels, same arguments: the results must not change by a bit.  (Measured on the
    headline workload it is no faster -- 0.546 against 0.539 ms per frame -- so it is not the default: DESIGN.md 5.)"""
    def run(graph):
        if graph:
            monkeypatch.setenv("PFT_GRAPH", "1")
        else:
            monkeypatch.delenv("PFT_GRAPH", raising=False)
        t = gpu.make_reference_tracker(particle_num=1024, seed=12)
        t.setReferenceCloud(data["model"])
        t.setTrans(scene.initial_trans())
        out = []
        for f in range(8):  # the first frames run directly, then the graph takes over; the crop size changes on the way
            t.setInputCloud(data["scene"][:20000 + 4000 * f])
            t.compute()
            out.append(t.getResult().tobytes())
        out.append(t.getParticles().tobytes())
        return out

    assert run(True) == run(False)


def test_error_behaviour(gpu):
    from pcl_tracking_amd._lib import PftError

    t = gpu.make_reference_tracker(particle_num=16)
    t.setReferenceCloud(scene.make_model(64))
    with pytest.raises(PftError) as e:
        t.compute()  # no input cloud: PCL prints PCL_ERROR and returns; the ABI reports PFT_ERR_NO_INPUT
    assert e.value.status == 2
    t2 = gpu.make_reference_tracker(particle_num=16)
    with pytest.raises(PftError) as e:
        t2.setInputCloud(scene.make_scene(50000)[:100])
        t2.compute()
    assert e.value.status == 3


def test_native_library_is_loaded(gpu):
    """the GPU tests must run on the HIP extension, not on a fallback"""
    import os

    from pcl_tracking_amd import _lib

    _lib.load()
    with open("/proc/%d/maps" % os.getpid()) as f:
        assert "libpft_hip.so" in f.read()


def test_handle_survives_changing_inputs(gpu, orc, data):
    """input clouds of very different sizes, a new reference cloud mid-run, empty input: the handle regrows its buffers
    and keeps matching the oracle (auto_tracking.cpp feeds a different cloud every frame)"""
    g, o = make_pair(gpu, orc, data["model"], data["scene"], 256, seed=9)
    clouds = [data["scene"], scene.make_scene(307200, mode="organized"), data["scene"][:900], data["scene"][:20000]]
    for c in clouds:
        g.setInputCloud(c)
        o.set_input(c)
        g.compute()
        assert o.compute() == 0
        rg, ro = g.getResult(), o.get_result()
        for k in ("x", "y", "z", "roll", "pitch", "yaw"):
            assert abs(float(rg[k]) - float(ro[k])) < 1e-4, (len(c), k)
    # a new (larger) reference cloud: the reference's "set object to track" step can be repeated
    m2 = scene.make_model(4000, seed=5)
    g.setReferenceCloud(m2)
    o.set_reference(m2)
    p = particles_around(data["gt"], 256, 2)
    G = g.evalWeights(p)
    O = o.eval_weights(p, mats=g.debugPoseToMatrix(p))
    np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
    assert ulp_diff(G["raw"], O["raw"]).max() <= 1
    # many short-lived handles (create / destroy cycles)
    for k in range(20):
        t = gpu.make_reference_tracker(particle_num=64, seed=k)
        t.setReferenceCloud(data["model"][:200])
        t.setTrans(scene.initial_trans())
        t.setInputCloud(data["scene"][:3000])
        t.compute()
        t.close()


# ---- non-default parameters: everything a caller of the PCL classes can set ---------------------------------
@pytest.mark.parametrize("params", [
    dict(octree_resolution=0.02, max_distance=0.05),
    dict(octree_resolution=0.005, max_distance=0.2, distance_weight=4.0),
    dict(hsv_weight=1.5, h_weight=0.5, s_weight=2.0, v_weight=1.0),
    dict(hsv_pcl180_argorder=0, alpha=5.0),
    dict(octree_resolution=0.037, max_distance=0.01, hsv_weight=0.0),
])
def test_eval_weights_non_default_parameters(gpu, orc, data, params):
    P = 96
    o = orc.Tracker(orc.default_config(particle_num=P, threads=0, emulate_pcl_alloc=0, **params))
    g = gpu.ParticleFilterTracker(seed=1)
    g.setParticleNum(P)
    coh = gpu.ApproxNearestPairPointCloudCoherence()
    dc, hc = gpu.DistanceCoherence(), gpu.HSVColorCoherence()
    dc.setWeight(params.get("distance_weight", 1.0))
    hc.setWeight(params.get("hsv_weight", 0.1))
    hc.setHWeight(params.get("h_weight", 1.0))
    hc.setSWeight(params.get("s_weight", 1.0))
    hc.setVWeight(params.get("v_weight", 0.0))
    coh.addPointCoherence(dc)
    coh.addPointCoherence(hc)
    coh.setSearchMethod(gpu.OctreeSearch(params.get("octree_resolution", 0.01)))
    coh.setMaximumDistance(params.get("max_distance", 0.1))
    g.setCloudCoherence(coh)
    g.setAlpha(params.get("alpha", 15.0))
    g._cfg.hsv_pcl180_argorder = params.get("hsv_pcl180_argorder", 1)
    for ref, tr, inp in ((g.setReferenceCloud, g.setTrans, g.setInputCloud), (o.set_reference, o.set_trans, o.set_input)):
        ref(data["model"])
        tr(scene.initial_trans())
        inp(data["scene"])
    p = particles_around(data["gt"], P, 17)
    G = g.evalWeights(p, want_nn=True)
    O = o.eval_weights(p, want_nn=True, mats=g.debugPoseToMatrix(p))
    np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
    assert G["octree_depth"] == O["octree_depth"]
    np.testing.assert_array_equal(G["nn_idx"], O["nn_idx"])
    np.testing.assert_array_equal(G["nn_d2"], O["nn_d2"])
    d = ulp_diff(G["raw"], O["raw"])
    assert d.max() <= 1, d.max()
    # A8 with the configured alpha
    w_o, fit_o = orc.normalize_weights(O["raw"], alpha=params.get("alpha", 15.0))
    w_g, fit_g = g.debugNormalize(O["raw"])
    assert fit_g == fit_o and ulp_diff(w_g, w_o).max() <= 1


@pytest.mark.parametrize("iters", [1, 3])
def test_compute_other_iteration_counts(gpu, orc, data, iters):
    g = gpu.make_reference_tracker(particle_num=300, seed=12)
    g.setIterationNum(iters)
    o = orc.Tracker(orc.default_config(particle_num=300, seed=12, threads=0, emulate_pcl_alloc=0, iteration_num=iters))
    for ref, tr, inp in ((g.setReferenceCloud, g.setTrans, g.setInputCloud), (o.set_reference, o.set_trans, o.set_input)):
        ref(data["model"])
        tr(scene.initial_trans())
        inp(data["scene"])
    for f in range(3):
        g.compute()
        assert o.compute() == 0
        rg, ro = g.getResult(), o.get_result()
        for k in ("x", "y", "z", "roll", "pitch", "yaw"):
            assert abs(float(rg[k]) - float(ro[k])) < 1e-4, (f, k)


def test_eval_weights_deep_tree_branch_levels_only_in_lds(gpu, orc, data):
    """a depth-10 tree (5 mm leaves) whose centre tables, jump table and branch levels fit the likelihood kernel's 80 KiB
    of LDS but whose leaf starts do not: the kernel then reads the leaf starts from L2 (third LDS layout of
    pft_likelihood.hip); the layout is asserted from the tree's sizes so that this case keeps covering it"""
    P, res = 64, 0.005
    model, cloud = scene.make_model(2048), scene.make_scene(50000)
    g = gpu.make_reference_tracker(particle_num=P, seed=1)
    coh = gpu.ApproxNearestPairPointCloudCoherence()
    coh.addPointCoherence(gpu.DistanceCoherence())
    hc = gpu.HSVColorCoherence()
    hc.setWeight(0.1)
    coh.addPointCoherence(hc)
    coh.setSearchMethod(gpu.OctreeSearch(res))
    coh.setMaximumDistance(0.1)
    g.setCloudCoherence(coh)
    o = orc.Tracker(orc.default_config(particle_num=P, seed=1, threads=0, emulate_pcl_alloc=0, octree_resolution=res))
    for ref, tr, inp in ((g.setReferenceCloud, g.setTrans, g.setInputCloud), (o.set_reference, o.set_trans, o.set_input)):
        ref(model)
        tr(scene.initial_trans())
        inp(cloud)
    rng = np.random.default_rng(3)
    gt = scene.model_gt_pose()
    p = np.zeros(P, scene.PARTICLE_DTYPE)
    for k, name in enumerate(("x", "y", "z")):
        p[name] = gt[k] + rng.normal(0, 0.2, P)
    for k, name in enumerate(("roll", "pitch", "yaw")):
        p[name] = gt[3 + k] + rng.normal(0, 0.2, P)
    p["w"] = 1.0
    p["weight"] = 1.0 / P
    G = g.evalWeights(p, want_nn=True)
    O = o.eval_weights(p, want_nn=True, mats=g.debugPoseToMatrix(p))
    D, nl, nw = G["octree_depth"], G["n_leaves"], G["n_words"]
    leaf_start = nw - nl - 1
    base = 2048 + 3 * (2 << D) * 4 + (2 << 12)
    assert D == O["octree_depth"] == 10
    assert base + leaf_start * 4 <= 80 * 1024 < base + leaf_start * 4 + (nl + 1) * 2, (D, leaf_start, nl)
    np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
    np.testing.assert_array_equal(G["nn_idx"], O["nn_idx"])
    np.testing.assert_array_equal(G["nn_d2"], O["nn_d2"])
    assert ulp_diff(G["raw"], O["raw"]).max() <= 1


def test_crop_one_pass_and_two_pass_agree(gpu, orc, data, monkeypatch):
    """the order-preserving crop in one launch (ticketed workgroups) against the count + scatter pair and the oracle"""
    p = particles_around(data["gt"], 96, 17, sig_t=0.2, sig_r=0.5)  # a wide box: the crop keeps most of the cloud
    res = []
    for two in (False, True):
        if two:
            monkeypatch.setenv("PFT_CROP_TWO_PASS", "1")
        g, o = make_pair(gpu, orc, data["model"], data["scene"], 96)
        G = g.evalWeights(p, want_nn=True)
        res.append(G)
        if not two:
            O = o.eval_weights(p, want_nn=True, mats=g.debugPoseToMatrix(p))
            np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
            assert len(G["crop_idx"]) > 5000
    np.testing.assert_array_equal(res[0]["crop_idx"], res[1]["crop_idx"])
    np.testing.assert_array_equal(res[0]["nn_idx"], res[1]["nn_idx"])
    np.testing.assert_array_equal(res[0]["raw"], res[1]["raw"])


# ---- NearestPairPointCloudCoherence (true nearest neighbour) -----------------------------------
EXACT_PATH_ENV = {"sorted": None, "per_query": "PFT_EXACT_PER_QUERY", "shells": "PFT_EXACT_SHELLS_ONLY"}


@pytest.mark.parametrize("M,N,P,maxd,path", [(256, 6000, 48, 0.1, "sorted"), (513, 20000, 33, 0.1, "sorted"), (64, 900, 20, 0.03, "sorted"),
                                             (300, 50000, 16, 0.25, "sorted"), (256, 6000, 48, 0.1, "per_query"),
                                             (513, 20000, 33, 0.1, "per_query"), (300, 50000, 16, 0.25, "per_query"),
                                             (256, 6000, 48, 0.1, "shells"), (300, 50000, 16, 0.25, "shells")])
def test_exact_nearest_pair_coherence(gpu, orc, data, M, N, P, maxd, path, monkeypatch):
    """the three search paths of pft_exact_nn.hip: per-cell candidate lists walked by waves of cell-sorted queries
    (default), the same lists walked per query, and the per-query shell search alone"""
    if EXACT_PATH_ENV[path]:
        monkeypatch.setenv(EXACT_PATH_ENV[path], "1")
    model = scene.make_model(M, seed=900 + M)
    cloud = data["scene"][:N]
    o = orc.Tracker(orc.default_config(particle_num=P, threads=0, emulate_pcl_alloc=0, exact_nearest=1, max_distance=maxd))
    g = gpu.ParticleFilterTracker(seed=1)
    g.setParticleNum(P)
    coh = gpu.NearestPairPointCloudCoherence()
    coh.addPointCoherence(gpu.DistanceCoherence())
    hc = gpu.HSVColorCoherence()
    hc.setWeight(0.1)
    coh.addPointCoherence(hc)
    coh.setSearchMethod(gpu.OctreeSearch(0.01))
    coh.setMaximumDistance(maxd)
    g.setCloudCoherence(coh)
    for ref, tr, inp in ((g.setReferenceCloud, g.setTrans, g.setInputCloud), (o.set_reference, o.set_trans, o.set_input)):
        ref(model)
        tr(scene.initial_trans())
        inp(cloud)
    p = particles_around(data["gt"], P, 31 + M)
    G = g.evalWeights(p, want_nn=True)
    O = o.eval_weights(p, want_nn=True, mats=g.debugPoseToMatrix(p))
    np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
    # inside the gate: the same neighbour (index and float squared distance); outside: the GPU does not search
    gate = O["nn_d2"].astype(np.float64) < maxd * maxd
    assert gate.any()
    np.testing.assert_array_equal(G["nn_idx"][gate], O["nn_idx"][gate])
    np.testing.assert_array_equal(G["nn_d2"][gate], O["nn_d2"][gate])
    assert (G["nn_idx"][~gate] == -1).all()
    d = ulp_diff(G["raw"], O["raw"])
    assert d.max() <= 1, d.max()
    # and a short tracking run in this mode
    g2 = gpu.ParticleFilterTracker(seed=5)
    g2.setParticleNum(200)
    g2.setCloudCoherence(coh)
    o2 = orc.Tracker(orc.default_config(particle_num=200, seed=5, threads=0, emulate_pcl_alloc=0, exact_nearest=1,
                                        max_distance=maxd))
    for ref, tr, inp in ((g2.setReferenceCloud, g2.setTrans, g2.setInputCloud), (o2.set_reference, o2.set_trans, o2.set_input)):
        ref(model)
        tr(scene.initial_trans())
        inp(cloud)
    for f in range(2):
        g2.compute()
        assert o2.compute() == 0
        rg, ro = g2.getResult(), o2.get_result()
        for k in ("x", "y", "z", "roll", "pitch", "yaw"):
            assert abs(float(rg[k]) - float(ro[k])) < 1e-4, (f, k)


def test_exact_nearest_sorted_and_per_query_paths_agree_bit_for_bit(gpu, data, monkeypatch):
    """the cell-sorted search (one candidate list per wave) and the per-query list walk are the same arithmetic in the same
    summation order: a tracking run at 1 024 particles ends in identical populations"""
    out = []
    for path in ("sorted", "per_query"):
        if EXACT_PATH_ENV[path]:
            monkeypatch.setenv(EXACT_PATH_ENV[path], "1")
        g = gpu.ParticleFilterTracker(seed=3)
        g.setParticleNum(1024)
        coh = gpu.NearestPairPointCloudCoherence()
        coh.addPointCoherence(gpu.DistanceCoherence())
        hc = gpu.HSVColorCoherence()
        hc.setWeight(0.1)
        coh.addPointCoherence(hc)
        coh.setMaximumDistance(0.1)
        g.setCloudCoherence(coh)
        g.setReferenceCloud(data["model"])
        g.setTrans(scene.initial_trans())
        g.setInputCloud(data["scene"])
        for _ in range(4):
            g.compute()
        out.append((g.getParticles(), g.getResult()))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1]
    assert out[0][0]["weight"].max() > 1.2 / 1024


def test_exact_nearest_wide_particle_cloud_and_tiny_sizes(gpu, orc, data):
    """(a) particles spread over half a metre: a tile of 32 particles then touches far more grid cells than the 8 192 entries
    of the workgroup's LDS count table -- the cells that find their probe window full are counted and placed with global
    atomics -- and most queries have no neighbour inside the gate; (b) one particle, one to three reference points"""
    def pair(model, P):
        o = orc.Tracker(orc.default_config(particle_num=P, threads=0, emulate_pcl_alloc=0, exact_nearest=1, max_distance=0.1))
        g = gpu.ParticleFilterTracker(seed=1)
        g.setParticleNum(P)
        coh = gpu.NearestPairPointCloudCoherence()
        coh.addPointCoherence(gpu.DistanceCoherence())
        hc = gpu.HSVColorCoherence()
        hc.setWeight(0.1)
        coh.addPointCoherence(hc)
        coh.setMaximumDistance(0.1)
        g.setCloudCoherence(coh)
        for ref, tr, inp in ((g.setReferenceCloud, g.setTrans, g.setInputCloud), (o.set_reference, o.set_trans, o.set_input)):
            ref(model)
            tr(scene.initial_trans())
            inp(data["scene"])
        return g, o

    def check(g, o, p):
        G = g.evalWeights(p, want_nn=True)
        O = o.eval_weights(p, want_nn=True, mats=g.debugPoseToMatrix(p))
        gate = O["nn_d2"].astype(np.float64) < 0.01
        np.testing.assert_array_equal(G["nn_idx"][gate], O["nn_idx"][gate])
        np.testing.assert_array_equal(G["nn_d2"][gate], O["nn_d2"][gate])
        assert (G["nn_idx"][~gate] == -1).all()
        assert ulp_diff(G["raw"], O["raw"]).max() <= 1
        return gate

    g, o = pair(data["model"], 96)
    gate = check(g, o, particles_around(data["gt"], 96, 77, sig_t=0.25, sig_r=0.8))
    assert 0.02 < gate.mean() < 0.9
    for M in (1, 2, 3):
        g, o = pair(data["model"][:M], 1)
        check(g, o, particles_around(data["gt"], 1, 5, sig_t=0.0, sig_r=0.0))


def test_exact_nearest_edge_cases(gpu, orc, data):
    """empty crop (no input point inside the particles' box) and a gate wider than the crop box"""
    def pair(maxd, cloud, P=8):
        o = orc.Tracker(orc.default_config(particle_num=P, threads=0, emulate_pcl_alloc=0, exact_nearest=1, max_distance=maxd))
        g = gpu.ParticleFilterTracker(seed=1)
        g.setParticleNum(P)
        coh = gpu.NearestPairPointCloudCoherence()
        coh.addPointCoherence(gpu.DistanceCoherence())
        hc = gpu.HSVColorCoherence()
        hc.setWeight(0.1)  # auto_tracking.cpp:246 (the class default is 1.0, the oracle's config default 0.1)
        coh.addPointCoherence(hc)
        coh.setMaximumDistance(maxd)
        g.setCloudCoherence(coh)
        for ref, tr, inp in ((g.setReferenceCloud, g.setTrans, g.setInputCloud), (o.set_reference, o.set_trans, o.set_input)):
            ref(data["model"][:128])
            tr(scene.initial_trans())
            inp(cloud)
        return g, o

    far = data["scene"][:2000].copy()
    far["z"] += 50.0  # nothing inside the crop box
    g, o = pair(0.1, far)
    p = particles_around(data["gt"], 8, 3)
    G, O = g.evalWeights(p, want_nn=True), o.eval_weights(p, want_nn=True, mats=g.debugPoseToMatrix(p))
    assert len(G["crop_idx"]) == 0 == len(O["crop_idx"])
    assert (G["raw"] == 0).all() and (O["raw"] == 0).all() and (G["nn_idx"] == -1).all()
    g, o = pair(2.0, data["scene"][:5000])  # every cropped point is inside the gate of every query
    G, O = g.evalWeights(p, want_nn=True), o.eval_weights(p, want_nn=True, mats=g.debugPoseToMatrix(p))
    np.testing.assert_array_equal(G["nn_idx"], O["nn_idx"])
    np.testing.assert_array_equal(G["nn_d2"], O["nn_d2"])
    assert ulp_diff(G["raw"], O["raw"]).max() <= 1


@pytest.mark.parametrize("kind", ["lattice", "duplicates", "line", "growth_all_directions", "far_from_origin"])
def test_eval_weights_adversarial_clouds(gpu, orc, data, kind, monkeypatch):
    """input clouds built to sit on the octree's decision boundaries: points exactly on cell faces (coordinates that
    are multiples of the resolution), many coincident points, a degenerate line, a box that has to grow in every
    direction, and a scene 60 m from the origin (large float ulps); both builders"""
    rng = np.random.default_rng(hash(kind) % 1000)
    gt = np.array(data["gt"][:3], np.float32)
    n = 6000
    c = np.zeros(n, scene.POINT_DTYPE)
    c["w"] = 1.0
    c["rgba"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    if kind == "lattice":
        xyz = np.round(rng.uniform(-0.3, 0.3, (n, 3)) / 0.01) * 0.01 + np.round(gt / 0.01) * 0.01
    elif kind == "duplicates":
        base = rng.uniform(-0.25, 0.25, (40, 3)) + gt
        xyz = base[rng.integers(0, 40, n)]
    elif kind == "line":
        s = rng.uniform(-0.4, 0.4, n)
        xyz = gt + np.stack([s, 0.3 * s, -0.2 * s], 1)
    elif kind == "growth_all_directions":
        r = np.linspace(0.001, 0.45, n)  # an outward spiral: every few points leave the box on another side
        a = np.linspace(0, 60 * np.pi, n)
        xyz = gt + np.stack([r * np.cos(a), r * np.sin(a), r * np.sin(2.3 * a)], 1)
    else:
        xyz = rng.uniform(-0.3, 0.3, (n, 3)) + gt
    off = np.array([60.0, -35.0, 20.0], np.float32) if kind == "far_from_origin" else np.zeros(3, np.float32)
    xyz = xyz.astype(np.float32) + off
    c["x"], c["y"], c["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    P = 48
    p = particles_around(tuple(np.array(data["gt"][:3]) + off) + tuple(data["gt"][3:]), P, 9, sig_t=0.03, sig_r=0.2)
    trans = scene.initial_trans().copy()
    trans[:3, 3] += off
    for builder in ("single", "sorted"):
        monkeypatch.setenv("PFT_FORCE_BUILDER", builder)
        g = gpu.make_reference_tracker(particle_num=P, seed=1)
        o = orc.Tracker(orc.default_config(particle_num=P, seed=1, threads=0, emulate_pcl_alloc=0))
        for ref, tr, inp in ((g.setReferenceCloud, g.setTrans, g.setInputCloud), (o.set_reference, o.set_trans, o.set_input)):
            ref(data["model"][:600])
            tr(trans)
            inp(c)
        G = g.evalWeights(p, want_nn=True)
        O = o.eval_weights(p, want_nn=True, mats=g.debugPoseToMatrix(p))
        np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
        assert len(G["crop_idx"]) > 0 and G["octree_depth"] == O["octree_depth"]
        np.testing.assert_array_equal(G["octree_min"], O["octree_min"])
        ot = orc.Octree(np.ascontiguousarray(c)[O["crop_idx"]])
        np.testing.assert_array_equal(G["point_keys"], ot.point_keys())
        np.testing.assert_array_equal(G["nn_idx"], O["nn_idx"])
        np.testing.assert_array_equal(G["nn_d2"], O["nn_d2"])
        assert ulp_diff(G["raw"], O["raw"]).max() <= 1


@pytest.mark.parametrize("kind", ["lattice", "duplicates"])
@pytest.mark.parametrize("path", ["sorted", "per_query"])
def test_exact_nearest_ties_take_the_lowest_index(gpu, orc, data, kind, path, monkeypatch):
    """clouds with many equal distances (points on a 1 cm lattice, 40 positions repeated 150 times each): the true nearest
    neighbour is the lowest cloud index among the closest points, whatever order the grid cells hold them in"""
    if EXACT_PATH_ENV[path]:
        monkeypatch.setenv(EXACT_PATH_ENV[path], "1")
    rng = np.random.default_rng(11)
    gt = np.array(data["gt"][:3], np.float32)
    n = 6000
    c = np.zeros(n, scene.POINT_DTYPE)
    c["w"] = 1.0
    c["rgba"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    if kind == "lattice":
        xyz = np.round(rng.uniform(-0.12, 0.12, (n, 3)) / 0.01) * 0.01 + np.round(gt / 0.01) * 0.01
    else:
        xyz = (rng.uniform(-0.2, 0.2, (40, 3)) + gt)[rng.integers(0, 40, n)]
    xyz = xyz.astype(np.float32)
    c["x"], c["y"], c["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    P = 24
    o = orc.Tracker(orc.default_config(particle_num=P, threads=0, emulate_pcl_alloc=0, exact_nearest=1, max_distance=0.1))
    g = gpu.ParticleFilterTracker(seed=1)
    g.setParticleNum(P)
    coh = gpu.NearestPairPointCloudCoherence()
    coh.addPointCoherence(gpu.DistanceCoherence())
    hc = gpu.HSVColorCoherence()
    hc.setWeight(0.1)
    coh.addPointCoherence(hc)
    coh.setMaximumDistance(0.1)
    g.setCloudCoherence(coh)
    # model points on the same lattice: queries at identity rotation are equidistant from several cloud points
    m = data["model"][:400].copy()
    for k in ("x", "y", "z"):
        m[k] = (np.round(m[k] / 0.005) * 0.005).astype(np.float32)
    for ref, tr, inp in ((g.setReferenceCloud, g.setTrans, g.setInputCloud), (o.set_reference, o.set_trans, o.set_input)):
        ref(m)
        tr(scene.initial_trans())
        inp(c)
    p = particles_around(data["gt"], P, 5, sig_t=0.02, sig_r=0.0)
    for k in ("roll", "pitch", "yaw"):
        p[k] = 0.0
    for k, v in zip(("x", "y", "z"), np.round(gt / 0.005) * 0.005):
        p[k][: P // 2] = np.float32(v)  # half of the particles exactly on the lattice
    G = g.evalWeights(p, want_nn=True)
    O = o.eval_weights(p, want_nn=True, mats=g.debugPoseToMatrix(p))
    gate = O["nn_d2"].astype(np.float64) < 0.01
    assert gate.any()
    np.testing.assert_array_equal(G["nn_idx"][gate], O["nn_idx"][gate])
    np.testing.assert_array_equal(G["nn_d2"][gate], O["nn_d2"][gate])
    assert ulp_diff(G["raw"], O["raw"]).max() <= 1


# ---- A3 on the support subset of the reference cloud (pft_hull.hip) ---------------------------------------------
def _box_points(t):
    import ctypes as C

    dbg = np.zeros(32, np.uint64)
    t._check(t._L.pft_debug_get_descent_stats(t._h, dbg.ctypes.data_as(C.c_void_p)))
    return int(dbg[30]), int(dbg[31])


@pytest.mark.parametrize("kind", ["scan", "ball", "noisy_planes", "lattice", "planar", "duplicates", "far_from_origin"])
def test_bounding_box_over_the_hull_shell_equals_the_box_over_all_points(gpu, data, kind, monkeypatch):
    """the box of the particles' transformed reference clouds is taken over the reference points that can be extreme in some
    rigidly transformed coordinate (convex hull + the shell the float evaluation can reach); it must be the box over ALL
    points bit for bit, for any rotation -- here 4 096 poses with uniformly random orientations and a few exact
    quarter turns, on clouds that are generic, degenerate (the subset then is the whole cloud) or far from the origin"""
    rng = np.random.default_rng(abs(hash(kind)) % 1000)
    n = 2048
    if kind == "scan":
        m = data["model"]
    else:
        m = np.zeros(n, scene.POINT_DTYPE)
        m["w"] = 1.0
        m["rgba"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        if kind == "ball":
            xyz = rng.normal(0, 1, (n, 3))
            xyz *= (rng.uniform(0, 1, (n, 1)) ** (1 / 3)) * 0.2 / np.linalg.norm(xyz, axis=1, keepdims=True)
        elif kind == "noisy_planes":
            xyz = rng.uniform(-0.2, 0.2, (n, 3))
            xyz[: n // 2, 2] = 0.1 + rng.normal(0, 1e-4, n // 2)
            xyz[n // 2:, 0] = -0.15 + rng.normal(0, 1e-6, n - n // 2)
        elif kind == "lattice":
            xyz = np.round(rng.uniform(-0.15, 0.15, (n, 3)) / 0.05) * 0.05
        elif kind == "planar":
            xyz = rng.uniform(-0.2, 0.2, (n, 3))
            xyz[:, 1] = 0.03125
        elif kind == "duplicates":
            xyz = rng.uniform(-0.2, 0.2, (40, 3))[rng.integers(0, 40, n)]
        else:
            xyz = rng.uniform(-0.2, 0.2, (n, 3)) + np.array([31.0, -17.0, 55.0])
        xyz = xyz.astype(np.float32)
        m["x"], m["y"], m["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    P = 4096
    p = np.zeros(P, scene.PARTICLE_DTYPE)
    p["w"], p["weight"] = 1.0, 1.0 / P
    for k, name in enumerate(("x", "y", "z")):
        p[name] = data["gt"][k] + rng.normal(0, 0.3, P)
    p["roll"], p["yaw"] = rng.uniform(-np.pi, np.pi, P), rng.uniform(-np.pi, np.pi, P)
    p["pitch"] = np.arcsin(rng.uniform(-1, 1, P))
    for i, (r, pt, y) in enumerate([(0, 0, 0), (np.pi / 2, 0, 0), (0, np.pi / 2, 0), (0, 0, np.pi / 2), (np.pi, 0, -np.pi / 2)]):
        p["roll"][i], p["pitch"][i], p["yaw"][i] = r, pt, y
    out = []
    for full in (False, True):
        if full:
            monkeypatch.setenv("PFT_AABB_FULL", "1")
        t = gpu.make_reference_tracker(particle_num=P, seed=1)
        t.setReferenceCloud(m)
        t.setTrans(scene.initial_trans())
        t.setInputCloud(data["scene"][:4000])
        boxes = [t.evalWeights(p[a:a + 512])["bbox"] for a in range(0, P, 512)]  # eight different boxes per cloud
        out.append((np.stack(boxes), _box_points(t)))
    (b_sub, (m_sub, m_all)), (b_full, (m_full, _)) = out
    assert m_full == m_all == n
    np.testing.assert_array_equal(b_sub, b_full)
    if kind in ("scan", "ball", "noisy_planes", "far_from_origin"):
        assert m_sub < n // 2, m_sub  # the subset is what makes the kernel cheap
    if kind == "planar":
        assert m_sub == n  # a degenerate cloud keeps every point
    print(kind, "box over", m_sub, "of", n, "points")


def test_bounding_box_hull_shell_random_clouds(gpu, data, monkeypatch):
    """two dozen random reference clouds of different character and scale (thin shells, clusters with exact duplicates, almost
    flat slabs, sizes from a millimetre to a hundred metres, off-centre): subset box == full box, bit for bit"""
    rng = np.random.default_rng(2024)
    P = 1024
    for case in range(24):
        n = int(rng.integers(130, 3000))
        kind = case % 6
        if kind == 0:  # thin spherical shell
            v = rng.normal(0, 1, (n, 3))
            xyz = v / np.linalg.norm(v, axis=1, keepdims=True) * (1.0 + rng.normal(0, 1e-3, (n, 1)))
        elif kind == 1:  # clusters with exact duplicates
            c = rng.uniform(-1, 1, (12, 3))
            xyz = c[rng.integers(0, 12, n)] + rng.normal(0, 0.05, (n, 3)) * (rng.uniform(0, 1, (n, 1)) > 0.3)
        elif kind == 2:  # almost flat slab
            xyz = rng.uniform(-1, 1, (n, 3)) * np.array([1.0, 0.7, 2e-3])
        elif kind == 3:  # a box surface
            xyz = rng.uniform(-1, 1, (n, 3))
            ax = rng.integers(0, 3, n)
            xyz[np.arange(n), ax] = np.sign(xyz[np.arange(n), ax])
        elif kind == 4:  # heavy-tailed
            xyz = rng.standard_t(2.5, (n, 3)) * 0.1
        else:  # uniform
            xyz = rng.uniform(-1, 1, (n, 3))
        scale = 10.0 ** rng.uniform(-3, 2)
        xyz = xyz * scale + rng.uniform(-1, 1, 3) * scale * rng.choice([0.0, 1.0, 30.0])
        m = np.zeros(n, scene.POINT_DTYPE)
        m["w"] = 1.0
        xyz = xyz.astype(np.float32)
        m["x"], m["y"], m["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        p = np.zeros(P, scene.PARTICLE_DTYPE)
        p["w"], p["weight"] = 1.0, 1.0 / P
        for k, name in enumerate(("x", "y", "z")):
            p[name] = rng.normal(0, 1.0, P) * scale
        p["roll"], p["yaw"] = rng.uniform(-np.pi, np.pi, P), rng.uniform(-np.pi, np.pi, P)
        p["pitch"] = np.arcsin(rng.uniform(-1, 1, P))
        boxes = []
        for full in (False, True):
            if full:
                monkeypatch.setenv("PFT_AABB_FULL", "1")
            else:
                monkeypatch.delenv("PFT_AABB_FULL", raising=False)
            t = gpu.make_reference_tracker(particle_num=P, seed=1)
            t.setReferenceCloud(m)
            t.setTrans(scene.initial_trans())
            t.setInputCloud(data["scene"][:1000])
            boxes.append(np.stack([t.evalWeights(p[a:a + 256])["bbox"] for a in range(0, P, 256)]))
        np.testing.assert_array_equal(boxes[0], boxes[1], err_msg="case %d kind %d n %d scale %g" % (case, kind, n, scale))


@pytest.mark.parametrize("kind", ["scan", "planar"])
def test_fused_resample_box_equals_the_separate_launches(gpu, data, kind, monkeypatch):
    """pft_compute's default steady-state iteration is ONE launch for resample + pose -> matrix + box partials
    (k_resample4<BOX>); PFT_SPLIT_RESAMPLE=1 takes k_resample4 + k_aabb and PFT_RESAMPLE_ONE_LANE=1 the one-lane resample
    kernel.  All three must leave the same bounding box, the same crop and the same particles, bit for bit -- also for a
    reference whose box support set is too large for the fused form ("planar": a degenerate cloud keeps all 2 048 points,
    more than PFT_BOX_FUSED_MAX), which then falls back to the separate launches by itself."""
    import ctypes as C

    if kind == "scan":
        m = data["model"]
    else:
        rng = np.random.default_rng(5)
        m = data["model"].copy()
        m["y"] = np.float32(0.03125)
        m["x"] = rng.uniform(-0.2, 0.2, len(m)).astype(np.float32)
        m["z"] = rng.uniform(-0.15, 0.15, len(m)).astype(np.float32)
    runs = []
    for env in (None, "PFT_SPLIT_RESAMPLE", "PFT_RESAMPLE_ONE_LANE"):
        for k in ("PFT_SPLIT_RESAMPLE", "PFT_RESAMPLE_ONE_LANE"):
            monkeypatch.delenv(k, raising=False)
        if env:
            monkeypatch.setenv(env, "1")
        t = gpu.make_reference_tracker(particle_num=1024, seed=3)
        t.setReferenceCloud(m)
        t.setTrans(scene.initial_trans())
        t.setInputCloud(data["scene"])
        frames = []
        for _ in range(3):
            t.compute()
            t.synchronize()
            bbox = np.zeros(6, np.float32)
            t._check(t._L.pft_debug_get_bbox(t._h, bbox.ctypes.data_as(C.c_void_p)))
            n = C.c_size_t()
            t._check(t._L.pft_debug_get_crop(t._h, None, 0, C.byref(n)))
            idx = np.zeros(n.value, np.int32)
            t._check(t._L.pft_debug_get_crop(t._h, idx.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
            frames.append((bbox, idx, t.getParticles().copy(), t.getResult().copy()))
        runs.append(frames)
        if env is None:
            m_sub, m_all = _box_points(t)
            assert (m_sub <= 256) == (kind == "scan"), (kind, m_sub)
    for other in runs[1:]:
        for (b0, i0, p0, r0), (b1, i1, p1, r1) in zip(runs[0], other):
            np.testing.assert_array_equal(b0, b1)
            np.testing.assert_array_equal(i0, i1)
            assert p0.tobytes() == p1.tobytes()
            assert r0.tobytes() == r1.tobytes()
            assert len(i0) > 100


@pytest.mark.parametrize("P", [512, 8192])
def test_leaf_records_copied_or_followed_give_the_same_bits(gpu, data, P, monkeypatch):
    """Round 3: for small launches the single-workgroup builder leaves the point records where the crop put them and the
    likelihood kernel reads crop_pts[leaf_order[pos]] (PftHeader::leaf_indirect; default below ~8 million queries);
    PFT_LEAF_INDIRECT=0 / 1 force the copied / the followed form.  Same neighbours, same distances, same frames."""
    runs = []
    for env in ("0", "1"):
        monkeypatch.setenv("PFT_LEAF_INDIRECT", env)
        t = gpu.make_reference_tracker(particle_num=P, seed=5)
        t.setReferenceCloud(data["model"])
        t.setTrans(scene.initial_trans())
        t.setInputCloud(data["scene"])
        frames = []
        for _ in range(3):
            t.compute()
            t.synchronize()
            frames.append((t.getParticles().copy(), t.getResult().copy()))
        ev = t.evalWeights(t.getParticles()[:64], want_nn=True)
        runs.append((frames, ev))
    monkeypatch.delenv("PFT_LEAF_INDIRECT")
    (f0, e0), (f1, e1) = runs
    for (p0, r0), (p1, r1) in zip(f0, f1):
        assert p0.tobytes() == p1.tobytes() and r0.tobytes() == r1.tobytes()
    np.testing.assert_array_equal(e0["nn_idx"], e1["nn_idx"])
    np.testing.assert_array_equal(e0["nn_d2"].view(np.uint32), e1["nn_d2"].view(np.uint32))
    np.testing.assert_array_equal(e0["raw"].view(np.uint32), e1["raw"].view(np.uint32))
    assert (e0["nn_idx"] >= 0).all()


def test_deep_tree_without_centre_tables_in_the_branch_only_layout(gpu, orc):
    """Found by tools/fuzz_parity.py (round 3): wild particles (0.5 m / 1 rad spread) give a crop box of tens of metres, the
    octree gets 12 levels (> PFT_TABLE_MAX_DEPTH: no centre tables, all-generic descent), and with ~20 000 cropped points
    the likelihood kernel keeps the branch levels in LDS and the leaf starts in L2 -- a layout whose dispatch used to take
    the table-reading descent regardless, sending every query to one leaf."""
    model = scene.make_model(65, seed=65)
    cloud = scene.make_scene(20000)
    g, o = make_pair(gpu, orc, model, cloud, 700)
    p = particles_around(scene.model_gt_pose(), 700, 12345, 0.5, 1.0)
    mats = g.debugPoseToMatrix(p)
    G = g.evalWeights(p, want_nn=True)
    O = o.eval_weights(p, want_nn=True, mats=mats)
    assert O["octree_depth"] >= 11 and len(O["crop_idx"]) > 15000, (O["octree_depth"], len(O["crop_idx"]))
    assert G["octree_depth"] == O["octree_depth"]
    np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
    np.testing.assert_array_equal(G["nn_idx"], O["nn_idx"])
    np.testing.assert_array_equal(G["nn_d2"].view(np.uint32), O["nn_d2"].view(np.uint32))
    assert ulp_diff(G["raw"], O["raw"]).max() <= 1
    assert len(np.unique(G["nn_idx"])) > 1000

"""Known-answer tests for the CPU oracle (SURVEY.md section 8c list).

PARITY UNPINNED: PCL 1.8.0 is not available and the reference ships no vectors for this path, so the
oracle is pinned by hand-derivable answers only.  Each case cites the PCL 1.8.0 routine restated.
"""
import math

import numpy as np
import pytest

from pcl_tracking_amd import scene


def P(x, y, z, rgba=0xFF000000):
    p = np.zeros(1, scene.POINT_DTYPE)
    p["x"], p["y"], p["z"], p["w"], p["rgba"] = x, y, z, 1.0, rgba
    return p


# (1) pcl::getTransformation -------------------------------------------------------------------
def test_get_transformation_yaw_quarter_turn(orc):
    m = orc.get_transformation(1, 2, 3, 0, 0, math.pi / 2)
    np.testing.assert_allclose(m[:3, :3], [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-7)
    np.testing.assert_array_equal(m[:3, 3], [1, 2, 3])
    np.testing.assert_array_equal(m[3], [0, 0, 0, 1])


def test_get_transformation_is_rz_ry_rx(orc):
    rng = np.random.default_rng(0)
    for _ in range(50):
        x, y, z = rng.uniform(-2, 2, 3)
        r, p, w = rng.uniform(-3, 3, 3)
        got = orc.get_transformation(x, y, z, r, p, w).astype(np.float64)
        np.testing.assert_allclose(got, scene.pose_matrix(np.float32(x), np.float32(y), np.float32(z), np.float32(r),
                                                          np.float32(p), np.float32(w)), atol=4e-7)


# (2) toState(getTransformation(s)) == s for |pitch| < pi/2 --------------------------------------
def test_to_state_round_trip(orc):
    rng = np.random.default_rng(1)
    for _ in range(50):
        s = np.concatenate([rng.uniform(-2, 2, 3), rng.uniform(-1.5, 1.5, 3)]).astype(np.float32)
        st = orc.to_state(orc.get_transformation(*s))
        got = np.array([st["x"], st["y"], st["z"], st["roll"], st["pitch"], st["yaw"]])
        np.testing.assert_allclose(got, s, atol=2e-6)


def test_transform_cloud_order_and_copy(orc):
    pts = np.zeros(3, scene.POINT_DTYPE)
    pts["x"], pts["y"], pts["z"], pts["rgba"] = [1, 0, 0.5], [0, 1, 0.25], [0, 0, 2], [1, 2, 3]
    m = orc.get_transformation(0.1, 0.2, 0.3, 0.3, -0.2, 0.7)
    out = orc.transform_cloud(pts, m)
    f = np.float32
    for i in range(3):
        x, y, z = f(pts["x"][i]), f(pts["y"][i]), f(pts["z"][i])
        for r, name in enumerate("xyz"):
            want = f(f(f(m[r, 0] * x) + f(m[r, 1] * y)) + f(m[r, 2] * z)) + m[r, 3]
            assert out[name][i] == f(want)
    np.testing.assert_array_equal(out["rgba"], pts["rgba"])


# (3) RGB2HSV ---------------------------------------------------------------------------------
def test_div_table_spot_values(orc):
    L = orc.lib()
    assert [L.orc_div_table(i) for i in (0, 1, 2, 3, 7, 9, 11, 13, 14, 254, 255)] == [
        0, 1044480, 522240, 348160, 149211, 116053, 94953, 80345, 74606, 4112, 4096]
    for i in range(1, 256):  # no rounding ties: 2*1044480/i is never an odd integer
        assert (2 * 1044480) % i != 0 or ((2 * 1044480) // i) % 2 == 0


def test_rgb2hsv_integer_cases(orc):
    assert orc.rgb2hsv_int(77, 77, 77) == (0, 0, 77)  # grey
    assert orc.rgb2hsv_int(255, 0, 0) == (0, 255, 255)  # red
    assert orc.rgb2hsv_int(0, 255, 0) == (60, 255, 255)  # green -> 120 deg / 2
    assert orc.rgb2hsv_int(0, 0, 255) == (120, 255, 255)  # blue -> 240 deg / 2
    assert orc.rgb2hsv_int(255, 0, 255)[0] == 150  # magenta 300 deg / 2
    assert orc.rgb2hsv_int(255, 0, 1)[0] == 0  # (-1*4096*15 + 2^18) >> 19 = 0
    assert orc.rgb2hsv_int(255, 0, 17)[0] == 178  # (-17*61440 + 2^18) >> 19 = -2 -> +180
    assert orc.rgb2hsv_int(255, 17, 0)[0] == 2
    h, s, v = orc.rgb2hsv(0, 255, 0)
    assert h == np.float32(60) / np.float32(180) and s == 1.0 and v == 1.0


# (4) DistanceCoherence ------------------------------------------------------------------------
def test_distance_coherence(orc):
    cfg = orc.default_config()
    assert orc.distance_coherence(cfg, P(0, 0, 0), P(0, 0, 0)) == 1.0
    assert orc.distance_coherence(cfg, P(0, 0, 0), P(1, 0, 0)) == 0.5
    d = orc.distance_coherence(cfg, P(0, 0, 0), P(0.03, 0.04, 0))
    assert abs(d - 1 / (1 + 0.0025)) < 1e-8


# (5) HSVColorCoherence ------------------------------------------------------------------------
def test_hsv_coherence(orc):
    cfg = orc.default_config()
    red = int(scene.pack_rgba(255, 0, 0))
    assert orc.hsv_coherence(cfg, red, red) == 1.0
    # as written upstream RGB2HSV(Red, Blue, Green): pure green is read as (r=0,g=0,b=255) -> h=120
    green = int(scene.pack_rgba(0, 255, 0))
    hs, ht = 0.0, np.float32(120) / np.float32(180)
    hd = min(abs(hs - ht), abs(1 + hs - ht))
    want = 1.0 / (1.0 + 0.1 * float(np.float32(hd) * np.float32(hd)))
    assert abs(orc.hsv_coherence(cfg, red, green) - want) < 1e-7
    cfg2 = orc.default_config(hsv_pcl180_argorder=0)  # textbook order: green h=60
    ht = np.float32(60) / np.float32(180)
    want = 1.0 / (1.0 + 0.1 * float(ht * ht))
    assert abs(orc.hsv_coherence(cfg2, red, green) - want) < 1e-7
    # hue wraps: h=2 vs h=178 are 4/180 apart
    a, b = int(scene.pack_rgba(255, 17, 0)), int(scene.pack_rgba(255, 0, 17))
    if cfg2.hsv_pcl180_argorder == 0:
        ha, hb = orc.rgb2hsv(255, 17, 0)[0], orc.rgb2hsv(255, 0, 17)[0]
        assert ha < 0.05 and hb > 0.95
        c = orc.hsv_coherence(cfg2, a, b)
        assert c > 1.0 / (1.0 + 0.1 * 0.01)


# (7) octree box replay + greedy descent --------------------------------------------------------
def test_octree_first_point_box(orc):
    t = orc.Octree(P(0.5, -0.25, 1.0))
    i = t.info()
    assert i["depth"] == 1 and i["leaves"] == 1
    eps = float(np.finfo(np.float32).eps)
    p0 = np.array([np.float32(0.5), np.float32(-0.25), np.float32(1.0)], np.float64)
    lo, hi = p0 - 0.01 / 2, p0 + 0.01 / 2
    over = ((2 * 0.01 - eps) - (hi - lo)) / 2.0
    np.testing.assert_array_equal(i["min"], lo - over)
    np.testing.assert_array_equal(i["max"], hi + over)


def test_octree_growth_replay_by_hand(orc):
    pts = np.concatenate([P(0, 0, 0), P(0.05, 0, 0)])
    t = orc.Octree(pts)
    i = t.info()
    eps = float(np.finfo(np.float32).eps)
    # first point: box = +-res (minus eps/2), depth 1.  second point violates only the upper x bound:
    # two growth steps, each lowering min_y and min_z (axes without an upper violation extend downwards)
    mn = np.array([-0.005, -0.005, -0.005]) - ((0.02 - eps) - (0.005 - -0.005)) / 2
    mn[1] -= 0.02
    mn[2] -= 0.02
    mn[1] -= 0.04
    mn[2] -= 0.04
    assert i["depth"] == 3
    np.testing.assert_allclose(i["min"], mn, atol=1e-15)
    np.testing.assert_allclose(i["max"], mn + (0.08 - eps), atol=1e-15)
    keys = t.point_keys()
    # point 0 was keyed (1,1,1) at depth 1, then shifted by 2 and 4 cells in y and z
    np.testing.assert_array_equal(keys[0], [0, 0 + 2 + 4, 0 + 2 + 4])
    np.testing.assert_array_equal(keys[1], [5, 6, 6])
    idx, d2 = t.approx_nearest(P(0.049, 0, 0))
    assert idx[0] == 1 and abs(d2[0] - 1e-6) < 1e-9


def test_octree_greedy_is_not_true_nn_but_consistent(orc):
    rng = np.random.default_rng(7)
    n = 3000
    xyz = rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32) * np.array([1, 1, 0.05], np.float32)
    pts = scene.make_points(xyz, np.zeros((n, 3)))
    t = orc.Octree(pts)
    qxyz = rng.uniform(-0.35, 0.35, (2000, 3)).astype(np.float32) * np.array([1, 1, 0.3], np.float32)
    q = scene.make_points(qxyz, np.zeros((len(qxyz), 3)))
    idx, d2 = t.approx_nearest(q)
    assert (idx >= 0).all()
    d = ((qxyz[:, None, :].astype(np.float64) - xyz[None].astype(np.float64)) ** 2).sum(-1)
    true = d.argmin(1)
    got = d[np.arange(len(q)), idx]
    np.testing.assert_allclose(got, d2, rtol=1e-5, atol=1e-9)
    assert (got >= d.min(1) - 1e-12).all()
    assert (idx != true).sum() > 0  # greedy descent is approximate by construction
    assert (idx == true).mean() > 0.3
    # leaf membership: every point's final-frame key addresses a 1 cm cell that contains it
    i = t.info()
    keys = t.point_keys()
    cell = np.floor((xyz.astype(np.float64) - i["min"]) / 0.01)
    assert np.abs(cell - keys).max() <= 1  # equal up to double rounding at cell faces
    assert (cell == keys).mean() > 0.999


def test_octree_ties_keep_first_inserted(orc):
    pts = np.concatenate([P(0.001, 0, 0), P(-0.001, 0, 0), P(0.001, 0, 0)])
    t = orc.Octree(pts)
    idx, d2 = t.approx_nearest(P(0.0, 0.0, 0.0))
    assert idx[0] == 0


# (6) coherence gate at maximum distance ---------------------------------------------------------
def test_weight_one_point_target_inside_and_outside(orc):
    cfg = orc.default_config(particle_num=1, threads=1)
    tr = orc.Tracker(cfg)
    tr.set_reference(P(0, 0, 0, int(scene.pack_rgba(10, 200, 30))))
    ident = np.zeros(1, scene.PARTICLE_DTYPE)
    # the crop keeps only input points inside the AABB of the transformed reference: a single
    # reference point gives a degenerate box, so put the target exactly on it, and a second one away
    tr.set_input(np.concatenate([P(0, 0, 0, int(scene.pack_rgba(10, 200, 30)))]))
    r = tr.eval_weights(ident, want_nn=True)
    assert r["raw"][0] == -1.0 and r["nn_idx"][0, 0] == 0
    # two reference points span a box; target 0.05 from ref 0 (inside 0.1) -> -c
    ref = np.concatenate([P(0, 0, 0), P(0.3, 0, 0)])
    tr.set_reference(ref)
    tr.set_input(P(0.05, 0, 0))
    r = tr.eval_weights(ident, want_nn=True)
    c0 = 1 / (1 + float(np.float32(0.05)) ** 2)
    assert abs(r["nn_d2"][0, 0] - 0.0025) < 1e-8
    assert r["nn_d2"][0, 1] > 0.01  # 0.25 away: outside the gate, contributes nothing
    assert abs(-r["raw"][0] - c0) < 1e-6
    tr.set_input(P(0.15, 0, 0))  # 0.15 from both: gated out for both -> weight 0
    r = tr.eval_weights(ident)
    assert r["raw"][0] == 0.0


def test_weight_empty_crop_defined(orc):
    cfg = orc.default_config(particle_num=2, threads=1)
    tr = orc.Tracker(cfg)
    tr.set_reference(np.concatenate([P(0, 0, 0), P(0.1, 0.1, 0.1)]))
    tr.set_input(P(5, 5, 5))
    r = tr.eval_weights(np.zeros(2, scene.PARTICLE_DTYPE), want_nn=True)
    assert len(r["crop_idx"]) == 0 and (r["raw"] == 0).all() and (r["nn_idx"] == -1).all()


# (8) normalizeWeight ---------------------------------------------------------------------------
def test_normalize_weights(orc):
    w, fit = orc.normalize_weights([-10, -5, 0, -1])
    e = np.array([math.exp(1.0), math.exp(1 - 15 * 5 / 9), 0.0, math.exp(1 - 15.0)], np.float32)
    want = e / np.float32(e.astype(np.float64).sum())
    np.testing.assert_array_equal(w, want)
    assert fit == -10 and w[2] == 0 and abs(w.sum() - 1) < 1e-6
    w, _ = orc.normalize_weights([-3, -3, -3])
    np.testing.assert_array_equal(w, np.float32(1) / np.float32(3))
    w, _ = orc.normalize_weights([0, 0, 0, 0])
    np.testing.assert_array_equal(w, np.float32(0.25))


# (9) Walker alias table ------------------------------------------------------------------------
def alias_probabilities(a, q):
    n = len(a)
    p = np.zeros(n)
    for k in range(n):
        p[k] += q[k] / n
        p[a[k]] += (1 - q[k]) / n
    return p


def test_alias_table_reconstructs_weights(orc):
    w = np.array([0.5, 0.25, 0.125, 0.125], np.float32)
    a, q = orc.gen_alias_table(w)
    np.testing.assert_allclose(alias_probabilities(a, q), w, atol=1e-12)
    rng = np.random.default_rng(3)
    w = rng.random(1000).astype(np.float32)
    w[rng.random(1000) < 0.2] = 0
    w /= w.sum()
    a, q = orc.gen_alias_table(w)
    np.testing.assert_allclose(alias_probabilities(a, np.minimum(q, 1.0)), w, atol=2e-6)


# (10) update() -----------------------------------------------------------------------------------
def test_weighted_mean_two_particles(orc):
    p = np.zeros(2, scene.PARTICLE_DTYPE)
    p["x"], p["yaw"], p["weight"] = [1, 3], [0.2, -0.2], [0.25, 0.75]
    r = orc.weighted_mean(p)
    assert r["x"] == 2.5 and abs(r["yaw"] - (-0.1)) < 1e-7 and r["weight"] == 0.5


# RNG spec: Philox4x32-10 known-answer vectors (Random123 kat_vectors) -----------------------------
def test_philox_known_answers(orc):
    f = 0xFFFFFFFF
    np.testing.assert_array_equal(orc.philox4x32([0, 0, 0, 0], [0, 0]),
                                  [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8])
    np.testing.assert_array_equal(orc.philox4x32([f, f, f, f], [f, f]),
                                  [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD])
    np.testing.assert_array_equal(orc.philox4x32([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344],
                                                 [0xA4093822, 0x299F31D0]),
                                  [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1])


def test_rng_moments(orc):
    z = np.array([orc.rng_normal_pair(5, i, 1, 0, 1) for i in range(20000)]).ravel()
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    u = np.array([orc.rng_uniform(5, i, 0, 0, 1) for i in range(20000)])
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01


def test_init_and_resample_semantics(orc):
    cfg = orc.default_config(particle_num=64, seed=9)
    rep = np.zeros(1, scene.PARTICLE_DTYPE)
    rep["x"], rep["yaw"], rep["weight"] = 1.0, 0.5, 1 / 64
    p = orc.init_particles(cfg, rep, 0, 64)
    assert (p["weight"] == np.float32(1) / np.float32(64)).all()
    assert abs(p["x"].mean() - 1.0) < 0.002 and 0.001 < p["x"].std() < 0.006
    # sharding by global id: the second half computed alone equals the second half of the whole
    np.testing.assert_array_equal(orc.init_particles(cfg, rep, 32, 32), p[32:])
    w = np.zeros(64, np.float32)
    w[5] = 1.0  # all mass on particle 5
    a, q = orc.gen_alias_table(w)
    p["weight"] = w
    out = orc.resample(cfg, p, a, q, rep, 0)
    assert out[0].tobytes() == rep[0].tobytes()  # slot 0 = representative state, verbatim
    assert np.abs(out["x"][1:] - p["x"][5]).max() < 0.1 and out["x"][1:].std() > 0.005
    np.testing.assert_array_equal(orc.resample(cfg, p, a, q, rep, 0, 16, 16), out[16:32])
    assert (orc.resample(cfg, p, a, q, rep, 1)["x"][1:] != out["x"][1:]).all()  # epoch changes draws


# ---- KLD-adaptive variant ---------------------------------------------------------------
def test_kld_normal_quantile_is_the_algorithm_209_cdf(orc):
    """PCL's normalQuantile is CACM Algorithm 209 (a normal CDF polynomial, not a quantile)"""
    from math import erf, sqrt

    assert orc.kld_normal_quantile(0.0) == 0.5
    for u in (-3.0, -1.0, -0.3, 0.5, 0.99, 1.7, 2.5, 5.0):
        want = 0.5 * (1.0 + erf(u / sqrt(2.0)))
        assert abs(orc.kld_normal_quantile(u) - want) < 2e-7, u
    assert orc.kld_normal_quantile(13.0) == 1.0 and orc.kld_normal_quantile(-13.0) == 0.0


def test_kld_bound_by_hand(orc):
    # calcKLBound(k) = (k-1)/(2 eps) * (1 - 2/(9(k-1)) + sqrt(2/(9(k-1))) z)^3, z = normalQuantile(0.99)
    from math import sqrt

    z = orc.kld_normal_quantile(0.99)
    for k in (2, 3, 17, 120):
        chi = 1.0 - 2.0 / (9.0 * (k - 1)) + sqrt(2.0 / (9.0 * (k - 1))) * z
        assert orc.kld_bound(k) == ((k - 1.0) / 2.0 / 0.2) * chi * chi * chi
    assert 4.0 < orc.kld_bound(2) < 4.1 and 276.0 < orc.kld_bound(100) < 277.0


def test_kld_resample_stops_at_the_bound(orc):
    """one bin only -> the loop runs to the maximum (k < 2 never ends it); many bins -> it stops at the first n
    with n >= calcKLBound(k), and the particle count / bins / k are those of the sequential loop"""
    import numpy as np

    from pcl_tracking_amd import scene

    cfg = orc.default_config(kld_adaptive=1, seed=9)
    old = np.zeros(50, scene.PARTICLE_DTYPE)
    old["w"] = 1.0
    old["weight"] = 1.0 / 50
    old["x"] = 0.05
    old["y"] = 0.05
    old["z"] = 0.05
    old["roll"] = 0.05
    old["pitch"] = 0.05
    old["yaw"] = 0.05
    a, q = orc.gen_alias_table(old["weight"])
    motion = np.zeros(1, scene.PARTICLE_DTYPE)
    # tiny step noise: every sample stays in bin (0,...,0) -> k == 1 -> runs to maximum_particle_number_
    tiny = orc.default_config(kld_adaptive=1, seed=9, step_cov=[1e-12] * 6)
    p, bins, k = orc.kld_resample(tiny, old, a, q, motion, 0)
    assert len(p) == 500 and k == 1 and (bins == 0).all()
    # reference noise: bins spread out; replay the stopping rule from the returned bins
    p, bins, k = orc.kld_resample(cfg, old, a, q, motion, 0)
    seen, kk, stop = set(), 0, None
    for n, b in enumerate(map(tuple, bins), 1):
        if b not in seen:
            seen.add(b)
            kk += 1
        if not (n < 500 and (kk < 2 or n < orc.kld_bound(kk))):
            stop = n
            break
    assert stop == len(p) and kk == k
    # bins are truncations toward zero of value / 0.1f
    v = np.stack([p[c] for c in ("x", "y", "z", "roll", "pitch", "yaw")], 1)
    np.testing.assert_array_equal(bins, (v / np.float32(0.1)).astype(np.int32))


# ---- the oracle's test-only summation order (sum mode 1): the tree the product specifies for its reductions ----
def test_tree_order_weighted_mean_known_answer(orc):
    """Adjacent-pair tree in double over the index range padded to a power of two: ((x0 + x1) + (x2 + x3)) + ...  With
    x = {2^60, 1, -2^60, 1} (weights 1) PCL's sequential float sum gives 1 (the first 1 is absorbed, the second
    survives), the tree gives 0 (both are absorbed inside their pairs): the two orders are told apart."""
    import numpy as np

    p = np.zeros(4, orc.PARTICLE_DTYPE)
    p["x"] = [2.0 ** 60, 1.0, -(2.0 ** 60), 1.0]
    p["y"] = [1.0, 2.0, 3.0, 4.0]
    p["weight"] = 1.0
    seq, tree = orc.weighted_mean(p), orc.weighted_mean_tree(p)
    assert float(seq["x"]) == 1.0 and float(tree["x"]) == 0.0
    assert float(seq["y"]) == 10.0 and float(tree["y"]) == 10.0
    assert float(tree["weight"]) == 0.25 and float(tree["w"]) == 1.0
    # five elements: padded to eight with +0.0; ((a+b)+(c+d)) + ((e+0)+(0+0))
    q = np.zeros(5, orc.PARTICLE_DTYPE)
    q["z"] = [1.0, 2.0 ** -30, 2.0 ** -30, 1.0, 2.0 ** -60]
    q["weight"] = 1.0
    want = ((1.0 + 2.0 ** -30) + (2.0 ** -30 + 1.0)) + ((2.0 ** -60 + 0.0) + 0.0)
    assert float(orc.weighted_mean_tree(q)["z"]) == np.float32(want)


def test_tree_order_normalize_matches_the_sequential_one_up_to_the_sum(orc):
    import numpy as np

    rng = np.random.default_rng(3)
    raw = -rng.random(1000).astype(np.float32) * 50 - 1000
    raw[::17] = 0.0  # particles without a correspondence keep weight zero
    a, fa = orc.normalize_weights(raw)
    b, fb = orc.normalize_weights_tree(raw)
    assert fa == fb == float(raw.min())
    assert (a[::17] == 0).all() and (b[::17] == 0).all()
    d = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    assert d.max() <= 1  # only (float) sum can differ, by one ulp at most
    assert abs(float(b.astype(np.float64).sum()) - 1.0) < 1e-5

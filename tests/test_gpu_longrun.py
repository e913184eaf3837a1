"""Long tracking runs on the GPU against the CPU oracle (run with -m gpu on an MI355X), SURVEY 8a rows A1 + A12.

>= 30 frames of a MOVING object (scene.advance_pose) with input clouds whose cropped size crosses
PFT_SORTED_BUILD_MIN in both directions, so the run passes through everything a short run never sees: the
builder switch driven by the previous iteration's crop size (single-workgroup <-> sorted many-workgroup
builder), its radix-pass guess from the previous depth and the rescue launch behind it, the double-buffered
particle arrays, the resample epochs, the one-pass crop's launch tags and (KLD variant) a particle count that
changes every resample.

Two comparisons per configuration:
  * SAME ARITHMETIC -- the oracle in its two test-only modes: pose -> matrix with double sin / cos rounded to float
    (set_trig_mode(1); PCL's cosf / sinf stays the default) and the weight sum / weighted mean in the adjacent-pair
    tree order the product specifies for its parallel reductions (set_sum_mode(1); PCL's sequential sums stay the
    default).  Both sides then see identical matrices, identical Philox draws and identical sums: every frame must be
    BIT-IDENTICAL -- result pose, every particle, every weight, the KLD particle count.  (With PCL's sums the runs
    part after a few frames: the mean re-enters the population through slot 0 of the resample, and PCL's Walker alias
    table is a discontinuous function of the weights -- DESIGN.md section 4.)
  * OWN TRIG   -- each side with its own sin / cos (glibc cosf / sinf against the device's double -> float): the
    north_star bar of 1e-4 on the weighted-mean pose; the first frame at which 1e-4 is exceeded is reported
    (DESIGN.md section 4 records it).
PARITY UNPINNED: the oracle restates PCL 1.8.0, which is not available here (oracle/pft_oracle.h).
"""
import numpy as np
import pytest

from pcl_tracking_amd import scene

pytestmark = pytest.mark.gpu

KEYS = ("x", "y", "z", "roll", "pitch", "yaw")
FRAMES = 32


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


@pytest.fixture(scope="module")
def gpu():
    from pcl_tracking_amd import tracker

    return tracker


_clouds = {}


def frame_cloud(f):
    """the object advances 1 mm / 0.5 deg per frame; the sensor alternates between a sparse organised
    image (160x120: crops of a few thousand points), a dense one (320x240: crops above PFT_SORTED_BUILD_MIN) and,
    every eighth frame, the voxel-downsampled 50 000-point cloud of the headline benchmark"""
    if f not in _clouds:
        pose = scene.advance_pose(scene.GT_POSE, f)
        if f % 8 == 5:
            _clouds[f] = scene.make_scene(50000, obj_pose=pose)
        elif (f // 3) % 2 == 1:
            _clouds[f] = scene.make_scene(320 * 240, obj_pose=pose, mode="organized")
        else:
            _clouds[f] = scene.make_scene(160 * 120, obj_pose=pose, mode="organized")
    return _clouds[f]


def make_pair(gpu, orc, P, seed, kld, trig_mode, sum_mode=0):
    model = scene.make_model(2048)
    g = gpu.make_reference_tracker(particle_num=P, seed=seed, kld=kld)
    o = orc.Tracker(orc.default_config(particle_num=P, seed=seed, threads=0, emulate_pcl_alloc=0,
                                       kld_adaptive=1 if kld else 0))
    o.set_trig_mode(trig_mode)
    o.set_sum_mode(sum_mode)
    for ref, tr in ((g.setReferenceCloud, g.setTrans), (o.set_reference, o.set_trans)):
        ref(model)
        tr(scene.initial_trans())
    return g, o


@pytest.mark.parametrize("P,kld", [(400, False), (8192, False), (400, True)])
def test_long_run_same_trig_is_bit_stable(gpu, orc, P, kld):
    g, o = make_pair(gpu, orc, P, seed=11, kld=kld, trig_mode=1, sum_mode=1)
    crops, depths, worst_pose, worst_frame, flipped = [], [], 0.0, -1, 0
    for f in range(FRAMES):
        cloud = frame_cloud(f)
        g.setInputCloud(cloud)
        o.set_input(cloud)
        g.compute()
        assert o.compute() == 0
        rg, ro = g.getResult(), o.get_result()  # getResult also reports device-side failures of the frame
        hs = g.debugHostStat()
        crops.append(int(hs[0]))
        depths.append(int(hs[1]))
        assert hs[2] == 0 and hs[3] == 0, (f, hs)
        pg, po = g.getParticles(), o.get_particles()
        # KLD variant: the particle count of every frame is the oracle's
        assert len(pg) == len(po), (f, len(pg), len(po))
        a = max(abs(float(rg[k]) - float(ro[k])) for k in KEYS)
        # identical matrices, identical Philox draws, identical summation order: nothing is left to differ
        assert rg.tobytes() == ro.tobytes(), (f, a, rg, ro)
        if a > worst_pose:
            worst_pose, worst_frame = a, f
        assert rg["weight"] == ro["weight"]
        # particle set: the same draws from the same alias entries, the same Box-Muller noise (double on both sides:
        # ocml and glibc could differ in a last bit that survives the cast to float -- it has not happened in these
        # runs); the per-particle likelihood sums are re-associated, but agree after the cast to float
        dk = np.zeros(len(pg), np.int64)
        for k in KEYS:
            dk = np.maximum(dk, ulp_diff(pg[k], po[k]))
        n_off = int((dk > 0).sum())
        flipped += n_off
        assert n_off == 0, (f, n_off, int(dk.max()))
        np.testing.assert_array_equal(pg["weight"].view(np.uint32), po["weight"].view(np.uint32), err_msg="frame %d" % f)
    # the run really did cross the builder threshold in both directions, and the depth changed on the way
    big = [c > 18000 for c in crops]
    assert any(big) and not all(big), crops
    assert any(big[i] != big[i + 1] for i in range(len(big) - 1)), crops
    assert len(set(depths)) >= 2, depths
    if kld:
        assert len(set(len(frame_cloud(f)) for f in range(FRAMES))) >= 2
    print("long run P=%d kld=%s: %d frames, crops %d..%d, depths %s, worst pose difference %.3g at frame %d, "
          "particles that took another alias entry over the run: %d"
          % (P, kld, FRAMES, min(crops), max(crops), sorted(set(depths)), worst_pose, worst_frame, flipped))


@pytest.mark.parametrize("P,kld", [(400, False), (8192, False), (400, True)])
def test_long_run_own_trig_reports_where_1e_4_ends(gpu, orc, P, kld, record_property):
    """PCL's cosf / sinf on the oracle side (with PCL's sequential sums), the device's own trig and tree sums on the
    other.  A matrix entry differs by 1 ulp now and then, a neighbour flips, a raw weight changes in its last digits --
    and PCL's Walker alias table (genAliasTable) is a DISCONTINUOUS function of the weights: its sequential pairing of
    small and large entries restructures wholesale when one running excess crosses 1 at a different step, so from
    that resample on the two runs hold different (equally distributed) particle sets.  Until then the weighted-mean
    poses agree to ~1e-6; the north_star's 1e-4 therefore holds up to a seed-dependent frame, which this test
    reports (DESIGN.md section 4 records it) and requires to lie beyond the short runs of test_gpu_parity.py.  The
    same-trig test above is the one that pins the schedule bit for bit over the whole run."""
    g, o = make_pair(gpu, orc, P, seed=11, kld=kld, trig_mode=0)
    first_bad, worst, worst_before = None, 0.0, 0.0
    for f in range(FRAMES):
        cloud = frame_cloud(f)
        g.setInputCloud(cloud)
        o.set_input(cloud)
        g.compute()
        assert o.compute() == 0
        rg, ro = g.getResult(), o.get_result()
        assert all(np.isfinite(float(rg[k])) for k in KEYS)
        a = max(abs(float(rg[k]) - float(ro[k])) for k in KEYS)
        worst = max(worst, a)
        if a >= 1e-4 and first_bad is None:
            first_bad = f
        if first_bad is None:
            worst_before = max(worst_before, a)
        if not kld:
            assert len(g.getParticles()) == P
    record_property("first_frame_over_1e-4", first_bad)
    print("own-trig run P=%d kld=%s: first frame over 1e-4: %s (worst before it %.3g, worst over %d frames %.3g)"
          % (P, kld, first_bad, worst_before, FRAMES, worst))
    # (the KLD variant's stopping rule and bin counts are discrete on top of that: one different draw changes the
    # particle COUNT of the next resample, so its runs part earlier)
    assert first_bad is None or first_bad >= (1 if kld else 5), (first_bad, worst)
    assert worst_before < 1e-4
    # (afterwards the two runs are two different samples of the posterior; with 400-500 particles on these sparse frames
    # their means can be far apart, Euler angles included -- nothing to assert beyond finiteness, checked above)

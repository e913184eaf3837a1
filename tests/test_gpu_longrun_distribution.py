"""Distribution-level long-run parity (VERDICT r2 weak #1 / missing #2; run with -m gpu on an MI355X).

The oracle here runs in its DEFAULT modes -- PCL's own arithmetic: pose -> matrix with cosf / sinf, the weight sum added
sequentially, the weighted mean sequentially in float (tracking/impl/particle_filter.hpp).  The device forms the matrix
from double sin / cos and sums in its specified trees, so a matrix entry differs by an ulp now and then,
and once a raw weight differs in a last digit PCL's Walker alias table -- a discontinuous function of the weights -- pairs
its entries differently: from that resample on the two runs hold different particle sets (DESIGN.md section 4;
tests/test_gpu_longrun.py pins the schedule bit for bit with the oracle's device-arithmetic modes).  What must still
hold after the runs have parted is that they are the SAME FILTER: two samples of one distribution.  Checked here, for
the fixed tracker at 8 192 and 400 particles and the KLD-adaptive tracker, over 32 moving frames and 8 seeds per side:

  (a) pointwise: until a seed's first frame with a pose difference >= 1e-4 the weighted-mean poses agree to < 1e-4
      (north_star bar 1), and frame 0 always agrees;
  (b) after it: the per-frame tracking error against the ground truth (scene.model_gt_pose(advance_pose(...)):
      translation distance and rotation angle of the result pose) of the device runs and of the oracle runs agree in
      mean and in 90th percentile within the spread that the ORACLE ITSELF shows between seeds -- for two independent
      samples of 8 seeds with per-seed standard deviation s the difference of the means has standard deviation s / 2,
      the test allows 2 s (four of those) plus a floor of 0.5 mm / 0.1 deg;
  (c) same-seed distance: after parting, the device's pose of seed k is no farther from the oracle's pose of seed k than
      oracle runs of DIFFERENT seeds are from each other (median over frames and pairs; factor 2: the frames of one run are
      correlated, so eight seeds give a noisy median -- the oracle against its own device-arithmetic modes shows 0.7 ... 1.7);
  (d) neither side loses the object: no settled frame of any seed is farther from the truth than 1.25 x the worst frame of
      the other side + 2 cm / 5 deg, and all stay within 15 cm / 30 deg.

The sequence: the box seen corner-on (three visible faces: all six degrees of freedom are constrained; the one-face view
of the other tests leaves the in-plane motion of a planar model to chance and makes the posterior multi-modal -- seeds then
fall into one mode or the other, which no 8-seed statistic can tell from a difference between the implementations),
moving 1 mm / 0.5 deg per frame, sensor frames alternating between sparse / dense organised images and the 50 000-point
voxel cloud as in test_gpu_longrun.py; the filter starts as the reference starts it (auto_tracking.cpp:663-674: the model
centroid, identity rotation -- 63 deg off here) and pulls in during the first frames.  PCL's likelihood is close to a
count of the points inside the 10 cm gate, so its weighted mean sits centimetres / some 14 deg from the truth on BOTH
sides: the test is about the two sides being the same filter, not about the filter being accurate.
PARITY UNPINNED: the oracle restates PCL 1.8.0, which is not available here (oracle/pft_oracle.h).
"""
import numpy as np
import pytest

from pcl_tracking_amd import scene

from test_gpu_longrun import FRAMES, KEYS

pytestmark = pytest.mark.gpu

SEEDS = [21, 22, 23, 24, 25, 26, 27, 28]
SETTLE = 6  # frames the filter gets to pull in from the initial pose (identity rotation, 1 cm off)
POSE3 = (0.05, -0.05, 0.9, 0.65, -0.55, 0.7)  # the box corner-on: faces -x, -y, -z visible

_cache = {}


def model3(M=1024):
    """the model (points, centroid offset); M model points: the CPU oracle is what this test waits for (8 seeds x 32 frames,
    cost proportional to particles x model points), so the 8 192-particle case takes 384 points, the others 512"""
    if ("model", M) not in _cache:
        assert len(scene._visible_faces(scene.MODEL_DIMS, POSE3)) == 3
        _cache[("model", M)] = scene.make_model(M, view_pose=POSE3, return_offset=True)
    return _cache[("model", M)]


def gt_pose(f, M=1024):
    """pose the tracker should estimate at frame f for the re-centred model"""
    pose = scene.advance_pose(POSE3, f)
    T = scene.pose_matrix(*pose)
    t = T[:3, :3] @ model3(M)[1] + T[:3, 3]
    return (t[0], t[1], t[2], pose[3], pose[4], pose[5])


def initial_trans3(M=1024):
    g = gt_pose(0, M)
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (g[0] + 0.01, g[1] + 0.01, g[2] + 0.01)
    return m


def frame_cloud(f):
    if ("cloud", f) not in _cache:
        pose = scene.advance_pose(POSE3, f)
        if f % 8 == 5:
            c = scene.make_scene(50000, obj_pose=pose)
        elif (f // 3) % 2 == 1:
            c = scene.make_scene(320 * 240, obj_pose=pose, mode="organized")
        else:
            c = scene.make_scene(160 * 120, obj_pose=pose, mode="organized")
        _cache[("cloud", f)] = c
    return _cache[("cloud", f)]


def pose_error(r, f, M=1024):
    """(translation distance [m], rotation angle [rad]) between a result pose and the frame's ground truth"""
    gt = gt_pose(f, M)
    A = scene.pose_matrix(*[float(r[k]) for k in KEYS])
    B = scene.pose_matrix(*gt)
    R = A[:3, :3].T @ B[:3, :3]
    return float(np.linalg.norm(A[:3, 3] - B[:3, 3])), float(np.arccos(np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)))


def pose_distance(a, b):
    A = scene.pose_matrix(*[float(a[k]) for k in KEYS])
    B = scene.pose_matrix(*[float(b[k]) for k in KEYS])
    R = A[:3, :3].T @ B[:3, :3]
    return float(np.linalg.norm(A[:3, 3] - B[:3, 3])), float(np.arccos(np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)))


@pytest.fixture(scope="module")
def gpu():
    from pcl_tracking_amd import tracker

    return tracker


@pytest.mark.parametrize("P,kld", [(8192, False), (400, False), (400, True)])
def test_device_and_pcl_arithmetic_are_the_same_filter_in_distribution(gpu, orc, P, kld, record_property):
    M = 384 if P >= 8192 else 512
    model = model3(M)[0]
    S = len(SEEDS)
    err_g = np.zeros((S, FRAMES, 2))
    err_o = np.zeros((S, FRAMES, 2))
    res_g, res_o = [[None] * FRAMES for _ in range(S)], [[None] * FRAMES for _ in range(S)]
    first_bad = []
    for si, seed in enumerate(SEEDS):
        g = gpu.make_reference_tracker(particle_num=P, seed=seed, kld=kld)
        o = orc.Tracker(orc.default_config(particle_num=P, seed=seed, threads=0, emulate_pcl_alloc=0,
                                           kld_adaptive=1 if kld else 0))  # default modes: PCL's cosf / sinf and sums
        for ref, tr in ((g.setReferenceCloud, g.setTrans), (o.set_reference, o.set_trans)):
            ref(model)
            tr(initial_trans3(M))
        bad, worst_before = None, 0.0
        for f in range(FRAMES):
            cloud = frame_cloud(f)
            g.setInputCloud(cloud)
            o.set_input(cloud)
            g.compute()
            assert o.compute() == 0
            rg, ro = g.getResult(), o.get_result()
            assert all(np.isfinite(float(rg[k])) for k in KEYS)
            res_g[si][f], res_o[si][f] = rg.copy(), ro.copy()
            err_g[si, f] = pose_error(rg, f, M)
            err_o[si, f] = pose_error(ro, f, M)
            a = max(abs(float(rg[k]) - float(ro[k])) for k in KEYS)
            if a >= 1e-4 and bad is None:
                bad = f
            if bad is None:
                worst_before = max(worst_before, a)
        # (a) pointwise agreement until the runs part; the first frame always agrees
        assert worst_before < 1e-4
        assert bad is None or bad >= 1, (seed, bad)
        first_bad.append(bad)
    record_property("first_frame_over_1e-4_per_seed", first_bad)
    parted = [FRAMES if b is None else b for b in first_bad]

    # (b) tracking error against the ground truth: mean and 90th percentile per seed, over the settled frames
    sl = slice(SETTLE, FRAMES)
    names, floors = ("translation [m]", "rotation [rad]"), (0.5e-3, np.deg2rad(0.1))
    report = []
    for c in range(2):
        mg, mo = err_g[:, sl, c].mean(1), err_o[:, sl, c].mean(1)
        pg, po = np.percentile(err_g[:, sl, c], 90, axis=1), np.percentile(err_o[:, sl, c], 90, axis=1)
        tol_mean = 2.0 * mo.std(ddof=1) + floors[c]
        tol_p90 = 2.0 * po.std(ddof=1) + floors[c]
        report.append("%s: mean device %.4g oracle %.4g (allowed difference %.3g), p90 device %.4g oracle %.4g (allowed %.3g)"
                      % (names[c], mg.mean(), mo.mean(), tol_mean, pg.mean(), po.mean(), tol_p90))
        assert abs(mg.mean() - mo.mean()) <= tol_mean, report[-1]
        assert abs(pg.mean() - po.mean()) <= tol_p90, report[-1]

    # (c) same-seed distance after parting against the oracle's own seed-to-seed distance
    same, cross = [[], []], [[], []]
    for si in range(S):
        for f in range(max(parted[si], SETTLE), FRAMES):
            d = pose_distance(res_g[si][f], res_o[si][f])
            same[0].append(d[0])
            same[1].append(d[1])
    for si in range(S):
        for sj in range(si + 1, S):
            for f in range(SETTLE, FRAMES):
                d = pose_distance(res_o[si][f], res_o[sj][f])
                cross[0].append(d[0])
                cross[1].append(d[1])
    if same[0]:
        for c in range(2):
            ms, mc = float(np.median(same[c])), float(np.median(cross[c]))
            report.append("%s: median distance device-oracle (same seed, after parting) %.4g, oracle-oracle (different seeds) %.4g"
                          % (names[c], ms, mc))
            assert ms <= 2.0 * mc + floors[c], report[-1]

    # (d) nobody loses the object (settled frames; the first ones start 63 deg off)
    lim = (0.15, np.deg2rad(30.0))  # well inside the gate + half the object's smallest extent
    slack = (0.02, np.deg2rad(5.0))
    for c in range(2):
        wg, wo = err_g[:, sl, c].max(), err_o[:, sl, c].max()
        report.append("%s: worst frame device %.4g oracle %.4g" % (names[c], wg, wo))
        assert wg <= 1.25 * wo + slack[c] and wo <= 1.25 * wg + slack[c], report[-1]
        assert wg < lim[c] and wo < lim[c], report[-1]
    print("P=%d kld=%s, %d seeds x %d frames; first frame over 1e-4 per seed: %s" % (P, kld, S, FRAMES, first_bad))
    for r in report:
        print("   ", r)

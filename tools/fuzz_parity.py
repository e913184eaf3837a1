"""Randomised parity campaign on the GPU box: the HIP path through the C ABI against the CPU oracle on generated cases,
for a time budget.  Five kinds of case:

  eval   one weight evaluation (crop -> octree -> approximate nearest neighbour -> coherence) of random particles on a
         random (model, cloud) pair: bounding box, crop list, octree depth / box / per-point keys / leaf count, the
         neighbour index and squared distance of EVERY (particle, reference point) pair must be bit-equal, the raw
         weights within 1 ulp (tests/test_gpu_parity.py::check_eval); builder (single workgroup / sorted), leaf-record
         form (copied / followed) and descent (fast / all-generic) are drawn at random per case
  filter the input front end (PassThrough + ApproximateVoxelGrid fused, VoxelGrid) on random clouds, leaf sizes, history
         sizes and limits: output bytes, counts and pass indices are the oracle's
  exact  NearestPairPointCloudCoherence mode against the oracle's exhaustive search, all three search paths
  shard  the particle-sharded phases with 2 / 3 / 4 / 8 ranks as handles of one process, exchange steps done by hand:
         every rank reproduces the single handle bit for bit
  track  2 - 6 frames of a whole tracker (fixed or KLD-adaptive, 1 - 3 iterations per frame) against the oracle in its
         device-arithmetic modes: result pose, every particle, every weight bit-identical (tests/test_gpu_longrun.py)

Clouds: the ray-cast scene (voxel / organised, several sizes), uniform cubes, clusters with duplicates, planes, clouds
far from the origin, tiny extents, NaN / Inf contamination; models of 1 ... 3 000 points; particle sets from a single
particle to a few thousand, tight and wild, some far outside the cloud (empty crops).

    python tools/fuzz_parity.py [minutes] [seed]        (GPU box; prints a line per 25 cases, a summary at the end)
A failing case prints its seed and parameters and the campaign goes on; exit code 1 if any case failed."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc  # noqa: E402
from pcl_tracking_amd import filters, scene, tracker  # noqa: E402
import test_gpu_parity as TP  # noqa: E402

KEYS = ("x", "y", "z", "roll", "pitch", "yaw")
_scene_cache = {}
LAST = {}  # description of the case being run (printed when it fails)


def cached_scene(kind, n):
    if (kind, n) not in _scene_cache:
        _scene_cache[(kind, n)] = scene.make_scene(n, mode="organized" if kind == "organized" else "voxel")
    return _scene_cache[(kind, n)]


def random_cloud(rng):
    kind = rng.choice(["voxel", "voxel", "organized", "cube", "clusters", "plane", "far", "tiny", "nan"])
    gt = np.array(scene.model_gt_pose()[:3])
    if kind == "voxel":
        c = cached_scene("voxel", int(rng.choice([3000, 20000, 50000])))
    elif kind == "organized":
        w = int(rng.choice([80, 160, 320, 320, 640]))  # 640 x 480: BASELINE configs[2]'s cloud, crops of 100 000 points and more
        c = cached_scene("organized", w * (w * 3 // 4))
    else:
        n = int(rng.integers(1, 30000))
        c = np.zeros(n, scene.POINT_DTYPE)
        c["w"] = 1.0
        c["rgba"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        if kind == "cube":
            xyz = gt + rng.uniform(-0.4, 0.4, (n, 3))
        elif kind == "clusters":
            cen = gt + rng.uniform(-0.3, 0.3, (max(1, n // 50), 3))
            xyz = cen[rng.integers(0, len(cen), n)] + rng.normal(0, 0.004, (n, 3)) * (rng.random((n, 1)) < 0.7)
        elif kind == "plane":
            xyz = gt + rng.uniform(-0.3, 0.3, (n, 3))
            xyz[:, int(rng.integers(0, 3))] = gt[0] + 0.05
        elif kind == "far":
            off = rng.uniform(-40, 40, 3)
            xyz = gt + off + rng.uniform(-0.3, 0.3, (n, 3))
        elif kind == "tiny":
            xyz = gt + rng.uniform(-0.004, 0.004, (n, 3))
        else:  # nan: a cube with non-finite coordinates mixed in (PCL's PassThrough drops them)
            xyz = gt + rng.uniform(-0.3, 0.3, (n, 3))
            bad = rng.random(n) < 0.05
            xyz[bad, int(rng.integers(0, 3))] = rng.choice([np.nan, np.inf, -np.inf])
        xyz = xyz.astype(np.float32)
        c["x"], c["y"], c["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    return kind, c


def random_model(rng):
    M = int(rng.choice([1, 7, 63, 64, 65, 200, 513, 1024, 2048, 3000]))
    return scene.make_model(M, seed=int(rng.integers(1, 1000)))


def cloud_centre(cloud, kind):
    if kind == "far":
        ok = np.isfinite(cloud["x"]) & np.isfinite(cloud["y"]) & np.isfinite(cloud["z"])
        return np.array([cloud["x"][ok].mean(), cloud["y"][ok].mean(), cloud["z"][ok].mean()])
    return np.array(scene.model_gt_pose()[:3])


def eval_case(rng, env):
    kind, cloud = random_cloud(rng)
    model = random_model(rng)
    P = int(rng.choice([1, 2, 33, 64, 100, 257, 700]))
    while P * len(model) > 600000:
        P = max(1, P // 2)
    c = cloud_centre(cloud, kind)
    pose = (c[0], c[1], c[2]) + tuple(rng.uniform(-3.1, 3.1, 3) if rng.random() < 0.3 else scene.model_gt_pose()[3:])
    sig_t = float(rng.choice([0.0, 0.005, 0.015, 0.1, 0.5]))
    sig_r = float(rng.choice([0.0, 0.05, 0.09, 1.0]))
    if rng.random() < 0.05:  # far outside the cloud: empty crop
        pose = (pose[0] + 30.0,) + pose[1:]
    params = {}
    if rng.random() < 0.4:  # everything a caller of the PCL classes can set (tests/test_gpu_parity.py: non-default parameters)
        params = dict(octree_resolution=float(rng.choice([0.003, 0.005, 0.01, 0.02, 0.037, 0.1])),
                      max_distance=float(rng.choice([0.01, 0.05, 0.1, 0.3])),
                      distance_weight=float(rng.choice([1.0, 4.0])), hsv_weight=float(rng.choice([0.0, 0.1, 1.5])),
                      h_weight=float(rng.choice([1.0, 0.5])), s_weight=float(rng.choice([1.0, 2.0])),
                      v_weight=float(rng.choice([0.0, 1.0])), hsv_pcl180_argorder=int(rng.integers(0, 2)))
    desc = dict(kind=kind, N=len(cloud), M=len(model), P=P, sig_t=sig_t, sig_r=sig_r, pose=[round(float(v), 4) for v in pose], env=dict(env),
                params=params)
    LAST.clear()
    LAST.update(desc)
    if params:
        o = orc.Tracker(orc.default_config(particle_num=P, threads=0, emulate_pcl_alloc=0, **params))
        g = tracker.ParticleFilterTracker(seed=1)
        g.setParticleNum(P)
        coh = tracker.ApproxNearestPairPointCloudCoherence()
        dc, hc = tracker.DistanceCoherence(), tracker.HSVColorCoherence()
        dc.setWeight(params["distance_weight"])
        hc.setWeight(params["hsv_weight"])
        hc.setHWeight(params["h_weight"])
        hc.setSWeight(params["s_weight"])
        hc.setVWeight(params["v_weight"])
        coh.addPointCoherence(dc)
        coh.addPointCoherence(hc)
        coh.setSearchMethod(tracker.OctreeSearch(params["octree_resolution"]))
        coh.setMaximumDistance(params["max_distance"])
        g.setCloudCoherence(coh)
        g._cfg.hsv_pcl180_argorder = params["hsv_pcl180_argorder"]
        for ref, tr, inp in ((g.setReferenceCloud, g.setTrans, g.setInputCloud), (o.set_reference, o.set_trans, o.set_input)):
            ref(model)
            tr(scene.initial_trans())
            inp(cloud)
    else:
        g, o = TP.make_pair(tracker, orc, model, cloud, P)
    p = TP.particles_around(pose, P, int(rng.integers(1, 1 << 30)), sig_t, sig_r)
    mats = g.debugPoseToMatrix(p)
    G = g.evalWeights(p, want_nn=True)
    O = o.eval_weights(p, want_nn=True, mats=mats)
    LAST.update(crop=len(O["crop_idx"]), depth_dev=int(G["octree_depth"]), depth_orc=int(O["octree_depth"]), leaves_dev=int(G["n_leaves"]),
                omin=[float(v) for v in O["octree_min"]], omax=[float(v) for v in O["octree_max"]])
    np.testing.assert_array_equal(G["bbox"], O["bbox"].astype(np.float32))
    np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
    if len(O["crop_idx"]) == 0:
        assert (G["raw"] == 0).all() and (O["raw"] == 0).all() and (G["nn_idx"] == -1).all()
        return desc, "empty"
    assert G["octree_depth"] == O["octree_depth"], (G["octree_depth"], O["octree_depth"])
    np.testing.assert_array_equal(G["octree_min"], O["octree_min"])
    np.testing.assert_array_equal(G["octree_max"], O["octree_max"])
    ot = orc.Octree(np.ascontiguousarray(cloud)[O["crop_idx"]], resolution=params.get("octree_resolution", 0.01))
    if len(O["crop_idx"]) <= 20000:  # (orc_octree_point_key is a Python-side loop)
        np.testing.assert_array_equal(G["point_keys"], ot.point_keys())
    assert G["n_leaves"] == ot.info()["leaves"]
    np.testing.assert_array_equal(G["nn_idx"], O["nn_idx"])
    np.testing.assert_array_equal(G["nn_d2"].view(np.uint32), O["nn_d2"].view(np.uint32))
    assert G["scan_queries"] == O["scan_queries"] and G["scan_points"] == O["scan_points"]
    d = TP.ulp_diff(G["raw"], O["raw"])
    assert d.max() <= 1, d.max()
    return desc, "depth %d crop %d" % (G["octree_depth"], len(G["crop_idx"]))


def track_case(rng, env):
    kld = bool(rng.random() < 0.35)
    P = int(rng.choice([300, 400, 500])) if kld else int(rng.choice([1, 64, 400, 1000, 3000]))
    iters = int(rng.integers(1, 4))
    frames = int(rng.integers(2, 7))
    seed = int(rng.integers(1, 1 << 30))
    M = int(rng.choice([200, 1024, 2048]))
    model = scene.make_model(M)
    desc = dict(kld=kld, P=P, iters=iters, frames=frames, seed=seed, M=M, env=dict(env))
    LAST.clear()
    LAST.update(desc)
    g = tracker.make_reference_tracker(particle_num=P, seed=seed, kld=kld)
    g.setIterationNum(iters)
    o = orc.Tracker(orc.default_config(particle_num=P, seed=seed, threads=0, emulate_pcl_alloc=0, iteration_num=iters,
                                       kld_adaptive=1 if kld else 0))
    o.set_trig_mode(1)
    o.set_sum_mode(1)
    for ref, tr in ((g.setReferenceCloud, g.setTrans), (o.set_reference, o.set_trans)):
        ref(model)
        tr(scene.initial_trans())
    for f in range(frames):
        kind = rng.choice(["voxel", "organized"])
        if kind == "voxel":
            cloud = cached_scene("voxel", int(rng.choice([3000, 20000, 50000])))
        else:
            w = int(rng.choice([80, 160, 320]))
            cloud = cached_scene("organized", w * (w * 3 // 4))
        g.setInputCloud(cloud)
        o.set_input(cloud)
        if f > 0 and rng.random() < 0.25:
            # a weight evaluation between two frames (device only) must leave the running filter alone: it builds its own
            # crop and octree with its own particles, through the same buffers and builder hints
            pe = TP.particles_around(scene.model_gt_pose(), min(P, int(rng.choice([1, 33, 300]))), int(rng.integers(1, 1 << 30)),
                                     float(rng.choice([0.015, 0.3])), float(rng.choice([0.09, 1.0])))
            g.evalWeights(pe, want_nn=bool(rng.random() < 0.5))
            LAST["eval_between"] = LAST.get("eval_between", 0) + 1
        g.compute()
        assert o.compute() == 0
        rg, ro = g.getResult(), o.get_result()
        assert rg.tobytes() == ro.tobytes(), (f, rg, ro)
        pg, po = g.getParticles(), o.get_particles()
        assert len(pg) == len(po), (f, len(pg), len(po))
        for k in KEYS + ("weight",):
            np.testing.assert_array_equal(pg[k].view(np.uint32), po[k].view(np.uint32), err_msg="frame %d field %s" % (f, k))
    return desc, "ok"


def filter_case(rng, env):
    """the input front end (include/pft_filters.h): PassThrough + ApproximateVoxelGrid fused, ApproximateVoxelGrid and
    VoxelGrid alone, on a random cloud with random leaf sizes / history sizes / limits: the output bytes are the oracle's"""
    n = int(rng.choice([0, 1, 2, 63, 1023, 1024, 1025, 4097, 20000, 70001, 150000]))
    kind = rng.choice(["uniform", "runs", "frame", "nan", "negative"])
    c = np.zeros(n, scene.POINT_DTYPE)
    c["w"] = 1.0
    c["rgba"] = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    span = float(rng.choice([0.004, 0.05, 0.3, 2.0]))
    if kind == "frame" and n > 0:
        w = int(rng.choice([120, 240, 480]))
        c = scene.make_depth_frame(w, w * 9 // 16, seed=int(rng.integers(1, 1000)))
        n = len(c)
    elif kind == "runs" and n > 0:  # long runs of one voxel, few voxels: table entries collide and flush each other
        lens = rng.choice([1, 2, 3, 7, 60, 255, 256, 257, 700, 1024, 2500], max(1, n // 200))
        vox = rng.integers(0, 40, (len(lens), 3))
        cell = np.repeat(vox, lens, axis=0)[:n]
        c = c[: len(cell)]
        n = len(c)
        jit = rng.uniform(0.05, 0.95, (n, 3))
        for a, k in enumerate(("x", "y", "z")):
            c[k] = ((cell[:, a] + jit[:, a]) * 0.01).astype(np.float32)
    else:
        for k in ("x", "y", "z"):
            c[k] = rng.uniform(-span, span, n).astype(np.float32)
        if kind != "negative":
            c["z"] += np.float32(1.0)
        if kind == "nan" and n > 0:
            bad = rng.random(n) < 0.08
            for k in ("x", "y", "z"):
                c[k][bad] = np.nan
    leaf = tuple(float(v) for v in rng.choice([0.005, 0.01, 0.02, 0.05, 0.1], 3)) if rng.random() < 0.3 else (0.01, 0.01, 0.01)
    hist = int(rng.choice([64, 512, 512, 1024, 2048]))
    LAST.clear()
    LAST.update(dict(filter=str(kind), n=n, leaf=leaf, hist=hist, span=span))
    # the reference's fused front end
    lo, hi = (0.0, 10.0) if rng.random() < 0.6 else (float(rng.uniform(-1, 1)), float(rng.uniform(1, 3)))
    f = filters.make_reference_input_filter()
    f.setPassThrough("z", lo, hi)
    f.setLeafSize(*leaf)
    f.setHistorySize(hist)
    f.setInputCloud(c)
    got = f.filter()
    idx = orc.pass_through(c, "z", lo, hi)
    want = orc.approx_voxel_grid(c[idx], leaf, hist)
    assert f.counts() == (len(idx), len(want)), (f.counts(), len(idx), len(want))
    np.testing.assert_array_equal(f.passIndices(), idx)
    assert got.tobytes() == want.tobytes()
    # VoxelGrid (exact) on the finite points
    if n <= 80000:
        g = filters.VoxelGrid()
        g.setLeafSize(*leaf)
        fin = c[idx] if kind in ("nan", "frame") else c
        g.setInputCloud(fin)
        gv, wv = g.filter(), orc.voxel_grid(fin, leaf)
        if wv is None:  # PCL: "Leaf size is too small for the input dataset" -> the input is handed through unchanged
            wv = fin
        assert gv.tobytes() == wv.tobytes(), (len(gv), len(wv))
    return dict(LAST), "filter"


def shard_case(rng, env):
    """the particle-sharded phases (pft_dist_*): W ranks as W handles of this one process, the two exchange steps done by
    hand (element-wise max of the bbox6 buffers, concatenation of the shards = what all-reduce(MAX) and all-gather
    deliver): every rank reproduces the single handle bit for bit, whatever W"""
    import torch

    from pcl_tracking_amd.dist import HipPhases

    world = int(rng.choice([2, 3, 4, 8]))
    P = world * int(rng.choice([1, 50, 128, 500, 1024]))
    frames = int(rng.integers(1, 4))
    iters = int(rng.integers(1, 4))
    seed = int(rng.integers(1, 1 << 30))
    M = int(rng.choice([200, 1024, 2048]))
    LAST.clear()
    LAST.update(dict(shard=True, world=world, P=P, frames=frames, iters=iters, seed=seed, M=M, env=dict(env)))
    model = scene.make_model(M)
    dev = torch.device("cuda", 0)
    single = tracker.make_reference_tracker(particle_num=P, seed=seed)
    single.setIterationNum(iters)
    single.setReferenceCloud(model)
    single.setTrans(scene.initial_trans())
    phs = [HipPhases(P, r, world, dev, seed=seed, iteration_num=iters) for r in range(world)]
    for ph in phs:
        ph.set_reference(model)
        ph.set_trans(scene.initial_trans())
    for f in range(frames):
        cloud = cached_scene("voxel", int(rng.choice([3000, 20000, 50000]))) if rng.random() < 0.6 else cached_scene("organized", 160 * 120)
        single.setInputCloud(cloud)
        single.compute()
        want = single.getResult().tobytes()
        for ph in phs:
            ph.set_input(cloud)
            ph.begin_frame()
        for it in range(iters):
            for ph in phs:
                ph.phase_a(it)
            bb = torch.stack([ph.bbox6 for ph in phs]).max(0).values
            for ph in phs:
                ph.bbox6.copy_(bb)
                ph.phase_b()
            g = torch.cat([ph.shard for ph in phs])
            for ph in phs:
                ph.gathered.copy_(g)
                ph.phase_c()
        for r, ph in enumerate(phs):
            assert ph.get_result().tobytes() == want, (f, r)
    want_p = single.getParticles().view(np.float32).reshape(-1, 8)
    for r, ph in enumerate(phs):
        np.testing.assert_array_equal(ph.get_particles().view(np.float32).reshape(-1, 8).view(np.uint32), want_p.view(np.uint32), err_msg="rank %d" % r)
    return dict(LAST), "shard"


def exact_case(rng, env):
    """NearestPairPointCloudCoherence (pft_config.exact_nearest): the true nearest neighbour inside the gate -- index and
    float squared distance of every in-gate pair equal the oracle's exhaustive search, outside the gate no neighbour;
    search path (cell-sorted lists / per-query lists / shells only) drawn at random"""
    path = rng.choice(["", "PFT_EXACT_PER_QUERY", "PFT_EXACT_SHELLS_ONLY"])
    for k in ("PFT_EXACT_PER_QUERY", "PFT_EXACT_SHELLS_ONLY"):
        os.environ.pop(k, None)
    if path:
        os.environ[str(path)] = "1"
    kind, cloud = random_cloud(rng)
    if len(cloud) > 20000:
        cloud = cloud[:20000]
    M = int(rng.choice([1, 7, 64, 200, 513]))
    model = scene.make_model(M, seed=int(rng.integers(1, 1000)))
    P = int(rng.choice([1, 16, 48]))
    maxd = float(rng.choice([0.05, 0.1, 0.25]))
    c = cloud_centre(cloud, kind)
    pose = (c[0], c[1], c[2]) + tuple(scene.model_gt_pose()[3:])
    sig_t, sig_r = float(rng.choice([0.0, 0.015, 0.1])), float(rng.choice([0.0, 0.09, 1.0]))
    LAST.clear()
    LAST.update(dict(exact=str(path), kind=str(kind), N=len(cloud), M=M, P=P, maxd=maxd, sig_t=sig_t, sig_r=sig_r))
    try:
        o = orc.Tracker(orc.default_config(particle_num=P, threads=0, emulate_pcl_alloc=0, exact_nearest=1, max_distance=maxd))
        g = tracker.ParticleFilterTracker(seed=1)
        g.setParticleNum(P)
        coh = tracker.NearestPairPointCloudCoherence()
        coh.addPointCoherence(tracker.DistanceCoherence())
        hc = tracker.HSVColorCoherence()
        hc.setWeight(0.1)
        coh.addPointCoherence(hc)
        coh.setSearchMethod(tracker.OctreeSearch(0.01))
        coh.setMaximumDistance(maxd)
        g.setCloudCoherence(coh)
        for ref, tr, inp in ((g.setReferenceCloud, g.setTrans, g.setInputCloud), (o.set_reference, o.set_trans, o.set_input)):
            ref(model)
            tr(scene.initial_trans())
            inp(cloud)
        p = TP.particles_around(pose, P, int(rng.integers(1, 1 << 30)), sig_t, sig_r)
        G = g.evalWeights(p, want_nn=True)
        O = o.eval_weights(p, want_nn=True, mats=g.debugPoseToMatrix(p))
        np.testing.assert_array_equal(G["crop_idx"], O["crop_idx"])
        if len(O["crop_idx"]) == 0:
            assert (G["raw"] == 0).all()
            return dict(LAST), "exact-empty"
        gate = O["nn_d2"].astype(np.float64) < maxd * maxd
        np.testing.assert_array_equal(G["nn_idx"][gate], O["nn_idx"][gate])
        np.testing.assert_array_equal(G["nn_d2"][gate].view(np.uint32), O["nn_d2"][gate].view(np.uint32))
        assert (G["nn_idx"][~gate] == -1).all()
        assert TP.ulp_diff(G["raw"], O["raw"]).max() <= 1
    finally:
        for k in ("PFT_EXACT_PER_QUERY", "PFT_EXACT_SHELLS_ONLY"):
            os.environ.pop(k, None)
    return dict(LAST), "exact"


def run_case(cseed):
    crng = np.random.default_rng(cseed)
    env = {}
    b = crng.choice(["", "single", "sorted"])
    if b:
        env["PFT_FORCE_BUILDER"] = str(b)
    li = crng.choice(["", "0", "1"])
    if li:
        env["PFT_LEAF_INDIRECT"] = str(li)
    if crng.random() < 0.15:
        env["PFT_GENERIC_DESCENT"] = "1"
    for k in ("PFT_FORCE_BUILDER", "PFT_LEAF_INDIRECT", "PFT_GENERIC_DESCENT"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for k, v in os.environ.items():  # (repro runs: overrides given on the command line as FUZZ_<NAME>=value)
        if k.startswith("FUZZ_"):
            os.environ[k[5:]] = v
            env[k[5:]] = v
    u = crng.random()
    kind = "track" if u < 0.20 else ("filter" if u < 0.32 else ("shard" if u < 0.40 else ("exact" if u < 0.48 else "eval")))
    return kind, env, {"track": track_case, "filter": filter_case, "eval": eval_case, "shard": shard_case, "exact": exact_case}[kind](crng, env)


def campaign(minutes, seed, max_cases=None, verbose=True):
    """runs cases until the time budget or the case count is used up; returns (counts per kind, failures)"""
    rng = np.random.default_rng(seed)
    t_end = time.time() + minutes * 60.0
    n = {"eval": 0, "track": 0, "filter": 0, "shard": 0, "exact": 0}
    failed = []
    t0 = time.time()
    case = 0
    while time.time() < t_end and (max_cases is None or case < max_cases):
        case += 1
        cseed = int(rng.integers(1, 1 << 62))
        kind, env = "?", {}
        try:
            kind, env, (desc, note) = run_case(cseed)
            n[kind] += 1
        except Exception as e:  # noqa: BLE001  (the campaign goes on; the case is reported with its seed)
            failed.append((kind, cseed, repr(e)[:300], dict(LAST)))
            if verbose:
                print("FAILED case seed %d env %s: %s\n   case: %s" % (cseed, env, repr(e)[:500], LAST), flush=True)
        if verbose and case % 25 == 0:
            print("%5d cases in %.0f s (%s), failures %d" % (case, time.time() - t0, n, len(failed)), flush=True)
    for k in ("PFT_FORCE_BUILDER", "PFT_LEAF_INDIRECT", "PFT_GENERIC_DESCENT"):
        os.environ.pop(k, None)
    return n, failed


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--case":  # python tools/fuzz_parity.py --case SEED: one case, verbosely
        try:
            print(run_case(int(sys.argv[2])))
        finally:
            print("case:", LAST)
        return 0
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    n, failed = campaign(minutes, seed0)
    print("campaign seed %d: %s cases in %.1f min; FAILURES: %d" % (seed0, n, (time.time() - t0) / 60.0, len(failed)))
    for f in failed:
        print("   ", f)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
